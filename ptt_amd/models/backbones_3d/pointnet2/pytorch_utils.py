"""Mirror of the layer builders of ptt/models/backbones_3d/pointnet2/pytorch_utils.py that the
hot path uses: SharedMLP (:12-36) made of Conv2d(1x1) + BatchNorm2d + ReLU units (:39-189),
plus the Conv1d/FC/Seq builders the heads assemble their small stacks from (:124-155, :263-427).

Child-module NAMES are part of the checkpoint contract (tracker3d_template.py:96-124 matches
state_dict entries by key and shape): a unit is an nn.Sequential with children `conv`,
`normlayer` (itself a Sequential holding `bn`) and `activation`; SharedMLP names its units
`layer0`, `layer1`, ... Parameters are initialised as the reference does (kaiming-normal conv
weights, BN weight 1 / bias 0, no conv bias when BN follows).
"""
import torch
import torch.nn as nn


class _Norm(nn.Sequential):
    def __init__(self, channels, norm_cls, name=""):
        super().__init__()
        self.add_module(name + "bn", norm_cls(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0.0)


class BatchNorm1d(_Norm):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm1d, name)


class BatchNorm2d(_Norm):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm2d, name)


class BatchNorm3d(_Norm):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm3d, name)


class _ConvUnit(nn.Sequential):
    """[norm ->][act ->] conv [-> norm][-> act]; preact puts norm/act first (reference :39-91)."""

    def __init__(self, conv_cls, norm_wrapper, in_size, out_size, kernel_size, stride, padding, dilation,
                 activation, bn, init, bias, preact, name):
        super().__init__()
        use_bias = bias and not bn
        conv = conv_cls(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                        dilation=dilation, bias=use_bias)
        init(conv.weight)
        if use_bias:
            nn.init.constant_(conv.bias, 0)

        def add_norm_act():
            if bn:
                self.add_module(name + "normlayer", norm_wrapper(in_size if preact else out_size))
            if activation is not None:
                self.add_module(name + "activation", activation)

        if preact:
            add_norm_act()
        self.add_module(name + "conv", conv)
        if not preact:
            add_norm_act()


class Conv1d(_ConvUnit):
    def __init__(self, in_size, out_size, kernel_size=1, stride=1, padding=0, dilation=1,
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name="", norm_layer=BatchNorm1d):
        super().__init__(nn.Conv1d, norm_layer, in_size, out_size, kernel_size, stride, padding, dilation,
                         activation, bn, init, bias, preact, name)


class Conv2d(_ConvUnit):
    def __init__(self, in_size, out_size, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), dilation=(1, 1),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name="", norm_layer=BatchNorm2d):
        super().__init__(nn.Conv2d, norm_layer, in_size, out_size, kernel_size, stride, padding, dilation,
                         activation, bn, init, bias, preact, name)


class Conv3d(_ConvUnit):
    def __init__(self, in_size, out_size, kernel_size=(1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0),
                 dilation=(1, 1, 1), activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_,
                 bias=True, preact=False, name="", norm_layer=BatchNorm3d):
        super().__init__(nn.Conv3d, norm_layer, in_size, out_size, kernel_size, stride, padding, dilation,
                         activation, bn, init, bias, preact, name)


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False, init=None, preact=False,
                 name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        if preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(in_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
        self.add_module(name + "fc", fc)
        if not preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(out_size))
            if activation is not None:
                self.add_module(name + "activation", activation)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d units over a (B,C,M,nsample) grouped tensor (reference :12-36)."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0      # the very first pre-activated unit gets no norm/act
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=(not plain) and bn, activation=None if plain else activation,
                       preact=preact))


class Seq(nn.Sequential):
    """Fluent builder the heads use: Seq(c).conv1d(c1, bn=True).conv1d(c2, activation=None) (reference :263-427).
    Children are named by their position, which is what the reference's checkpoints contain."""

    def __init__(self, input_channels):
        super().__init__()
        self.count = 0
        self.current_channels = input_channels

    def _push(self, module, out_size):
        self.add_module(str(self.count), module)
        self.count += 1
        self.current_channels = out_size
        return self

    def conv1d(self, out_size, kernel_size=1, stride=1, padding=0, dilation=1, activation=nn.ReLU(inplace=True),
               bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False, name="", norm_layer=BatchNorm1d):
        return self._push(Conv1d(self.current_channels, out_size, kernel_size, stride, padding, dilation, activation,
                                 bn, init, bias, preact, name, norm_layer), out_size)

    def conv2d(self, out_size, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), dilation=(1, 1),
               activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False,
               name="", norm_layer=BatchNorm2d):
        return self._push(Conv2d(self.current_channels, out_size, kernel_size, stride, padding, dilation, activation,
                                 bn, init, bias, preact, name, norm_layer), out_size)

    def fc(self, out_size, activation=nn.ReLU(inplace=True), bn=False, init=None, preact=False, name=""):
        return self._push(FC(self.current_channels, out_size, activation=activation, bn=bn, init=init, preact=preact,
                             name=name), out_size)

    def dropout(self, p=0.5):
        self.add_module(str(self.count), nn.Dropout(p=p))
        self.count += 1
        return self


# ---------------------------------------------------------------------------------------------------------------
# Eval-mode execution of a Conv1d(k=1)(+BatchNorm)(+ReLU) stack on point-major rows (no reference counterpart: the
# reference runs these as stock cuDNN convs on (B,C,N) tensors). One ptt_linear_f32 launch per layer with the
# BatchNorm folded from its running statistics, instead of conv + batch_norm + relu (+ layout copies) per layer.
# ---------------------------------------------------------------------------------------------------------------
def rows_fusable(seq, rows):
    from .... import ops
    if seq.training or not rows.is_cuda:
        return False
    name = 'Conv1d stack %s' % [u.conv.weight.shape[0] for u in seq if hasattr(u, 'conv')]
    if ops.autograd_recording(seq, rows):
        return ops.note_unfused(name, 'autograd is recording (wrap inference in torch.no_grad())')
    if rows.dtype != torch.float32:
        return ops.note_unfused(name, 'inputs must be float32')
    for unit in seq:
        conv = getattr(unit, 'conv', None)
        if not isinstance(conv, nn.Conv1d) or conv.kernel_size != (1,) or conv.stride != (1,) or conv.padding != (0,):
            return ops.note_unfused(name, 'not a stack of 1x1 Conv1d units')
        if list(unit._modules.keys())[0] != 'conv':                       # pre-activation units are not folded
            return ops.note_unfused(name, 'pre-activation units')
        if hasattr(unit, 'activation') and not isinstance(unit.activation, nn.ReLU):
            return ops.note_unfused(name, 'activation other than ReLU')
        if hasattr(unit, 'normlayer') and not isinstance(getattr(unit.normlayer, 'bn', None), nn.BatchNorm1d):
            return ops.note_unfused(name, 'normalisation other than BatchNorm1d')
    return len(seq) > 0


def _rows_params(seq):
    from .... import ops
    tensors = []
    for unit in seq:
        tensors.append(unit.conv.weight)
        if unit.conv.bias is not None:
            tensors.append(unit.conv.bias)
        if hasattr(unit, 'normlayer'):
            bn = unit.normlayer.bn
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
            if bn.num_batches_tracked is not None:
                tensors.append(bn.num_batches_tracked)         # bumped by every train-mode forward
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    cache = getattr(seq, '_rows_cache', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    layers = []
    with torch.no_grad():
        for unit in seq:
            w = unit.conv.weight
            scale = shift = None
            if hasattr(unit, 'normlayer'):
                bn = unit.normlayer.bn
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                if unit.conv.bias is not None:
                    shift = (shift + unit.conv.bias * scale).contiguous()
            elif unit.conv.bias is not None:
                shift = unit.conv.bias.detach().float().contiguous()
            w2 = w.reshape(w.shape[0], w.shape[1])
            layers.append((ops.pack_weight(w2), w.shape[0], scale, shift, hasattr(unit, 'activation'),
                           # the one-launch form (ptt_rows_mlp_f32): BatchNorm scale folded into the packed weights
                           ops.pack_weight(w2 if scale is None else w2 * scale[:, None]), w.shape[1]))
    ops.publish_params(layers[0][0].device)
    object.__setattr__(seq, '_rows_cache', (key, layers))
    return layers


def rows_layers(seq, rot0=0):
    """The folded layers of an eval-mode Conv1d stack for ptt_row_jobs_f32: [(wpacked, cout, scale, shift, relu), ...].
    rot0: the first layer's input channels rotated left by rot0 before packing — a stack whose reference input is
    cat((xyz, feats)) (centroids_voting_head.py:86-90) then reads [feats | xyz], two tensors, no concatenation."""
    from .... import ops
    layers = [L[:5] for L in _rows_params(seq)]
    if rot0:
        cache = getattr(seq, '_rows_rot_cache', None)
        key = (id(layers[0][0]), rot0)
        if cache is None or cache[0] != key:
            w = seq[0].conv.weight
            cache = (key, ops.pack_weight(w.reshape(w.shape[0], w.shape[1]), rot0))
            ops.publish_params(w.device)
            object.__setattr__(seq, '_rows_rot_cache', cache)
        layers[0] = (cache[1],) + tuple(layers[0][1:])
    return layers


def _one_launch(layers, rows):
    """ptt_rows_mlp_f32's envelope (at most 4 layers, K <= 264, inner widths <= 256, last <= 384) — and enough rows: the
    one-launch form runs a 32-row tile's layers back to back on ONE CU (25 us for 259 -> 256 -> 256 -> 259), while
    per-layer launches spread each layer's column tiles over idle CUs. At one tracklet frame (128 rows) the per-layer form
    is faster (1.12 vs 1.14 ms per frame); at 48 frames the one-launch form (4.19 vs 4.21 ms per step)."""
    n_rows = rows.numel() // rows.shape[-1]
    return (n_rows >= 1024 and len(layers) <= 4 and rows.shape[-1] <= 264 and all(L[1] <= 256 for L in layers[:-1])
            and layers[-1][1] <= 384)


def rows_forward(seq, rows, residual=None):
    """seq(rows^T)^T for an eval-mode Conv1d(k=1) stack: rows (..., Cin) -> (..., Cout); `residual` (..., Cout) is
    added to the last layer's output. Call only when rows_fusable(seq, rows). The whole stack is one launch when it fits
    ptt_rows_mlp_f32, else one ptt_linear_f32 launch per layer."""
    from .... import ops
    layers = _rows_params(seq)
    if _one_launch(layers, rows) and rows.stride(-1) == 1:
        return ops.rows_mlp(rows, [(L[5], None, L[3], L[6], L[1], L[4]) for L in layers], residual)
    x = rows
    for i, L in enumerate(layers):
        wp, cout, scale, shift, relu = L[:5]
        x = ops.linear(x, wp, cout, scale, shift, relu, residual if i == len(layers) - 1 else None)
    return x

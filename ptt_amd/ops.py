"""Operator boundary: torch-tensor front-ends of the C ABI (include/ptt_hip.h).

The first group has the names, argument order and error behaviour of the third-party
`pointnet2_ops._ext` functions the reference calls (pointnet2_utils.py:78,112,118,237,
257,287): device tensors only (a CPU tensor raises RuntimeError, as upstream does),
fp32 / int32, contiguous. Kernels are enqueued on the current torch stream; nothing
synchronises. The second group exposes the fused fp32-MFMA kernels.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import AttnDesc, SaDesc, SaLayer, XcorrDesc


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing with HIP events recorded on the SAME stream the kernels are
# enqueued on (the current torch stream). Off by default; bench.py switches it on for the
# kernels whose roofline it reports.
_timing = None


def start_kernel_timing(names):
    global _timing
    _timing = {n: [] for n in names}


def stop_kernel_timing():
    """-> {name: [milliseconds per launch]} (synchronises)."""
    global _timing
    t, _timing = _timing, None
    torch.cuda.synchronize()
    return {n: [a.elapsed_time(b) for a, b in ev] for n, ev in (t or {}).items()}


class _timed(object):
    def __init__(self, name):
        self.ev = None
        if _timing is not None and name in _timing:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.name = name

    def __enter__(self):
        if self.ev:
            self.ev[0].record()

    def __exit__(self, *a):
        if self.ev:
            self.ev[1].record()
            _timing[self.name].append(self.ev)


# --------------------------------------------------------------------------- which path runs, and saying so
# The fused kernels have no autograd graph. A module takes them only when (a) it is in eval mode on a HIP device,
# (b) its shape is one the library instantiates and (c) nothing in the call is being recorded by autograd. When an
# eval-mode call on a HIP device has to take the stock-torch path instead, the module says so ONCE per reason
# (warnings.warn) and counts it here, so nobody benchmarks MIOpen/rocBLAS believing it is the hand-written path.
unfused_calls = {}          # (module name, reason) -> number of eval-mode HIP calls that took the stock-torch path


def autograd_recording(module, *tensors):
    """True when this call must stay differentiable: grad mode is on and an input or a parameter requires grad
    (eval-mode fine-tuning with frozen BatchNorm, saliency / adversarial gradients w.r.t. the points, ...). The
    reference stays differentiable in eval mode; inference wraps the model in torch.no_grad()
    (tools/eval_utils/eval_tracking_utils.py:33)."""
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and t.requires_grad for t in tensors):
        return True
    return module is not None and any(p.requires_grad for p in module.parameters())


def note_unfused(name, reason):
    """An eval-mode call on a HIP device is leaving the hand-written kernels: count it, warn once per (name, reason)."""
    key = (name, reason)
    n = unfused_calls.get(key, 0)
    unfused_calls[key] = n + 1
    if n == 0:
        import warnings
        warnings.warn("ptt_amd: %s runs on stock torch ops, not the fused HIP kernels: %s" % (name, reason),
                      RuntimeWarning, stacklevel=3)
    return False


def publish_params(device):
    """Called by the modules right after they (re)build a cached set of packed / folded parameters: blocks the host
    until the kernels that produced them have finished. The caches are shared by every stream that later runs the
    module (the search and template branches run on two streams and share their SA modules), and a cache entry built
    on one stream must not be read by a kernel on another stream before it is complete. Once per weight version."""
    if device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).synchronize()


def require_finite(*tensors):
    """Raises ValueError if any of the device tensors holds a NaN or an infinity (one reduction per tensor and ONE host
    synchronisation: a guard for the boundary of a pipeline, not something the hot path calls per frame). Non-finite coordinates
    are outside the contract of this library ("Non-finite input")."""
    bad = [k for k, t in enumerate(tensors) if t is not None and t.is_floating_point() and not bool(torch.isfinite(t).all())]
    if bad:
        raise ValueError("non-finite values in tensor(s) %s" % ", ".join("#%d %s" % (k, tuple(tensors[k].shape)) for k in bad))


def _chk(t, name, dtype, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a device (HIP) tensor — CPU tensors are not supported" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("%s must have %d dimensions, got %s" % (name, ndim, tuple(t.shape)))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# --------------------------------------------------------------------------- _ext-compatible ops
FPS_RESIDENT_MAX_N, FPS_RESIDENT_MAX_NPOINT = 16384, 15360         # ptt_fps_f32's register / LDS-resident reach


def furthest_point_sampling(xyz, npoint):
    """(B,N,3) f32 -> (B,npoint) i32.  Replaces _ext.furthest_point_sampling (pointnet2_utils.py:78)."""
    _chk(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    out = torch.empty((B, int(npoint)), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device), _timed('ptt_fps_f32'):
        if N <= FPS_RESIDENT_MAX_N and int(npoint) <= FPS_RESIDENT_MAX_NPOINT:
            _lib.check(_lib.lib().ptt_fps_f32(_ptr(xyz), B, N, int(npoint), _ptr(out), _stream()), "ptt_fps_f32")
        else:           # the reference's op has no size limit: min-distances in a workspace, identical picks, slower
            ws = torch.empty((B * N,), dtype=torch.float32, device=xyz.device)
            _lib.check(_lib.lib().ptt_fps_ws_f32(_ptr(xyz), B, N, int(npoint), _ptr(out), _ptr(ws), ws.numel(), _stream()),
                       "ptt_fps_ws_f32")
    return out


def gather_points(features, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M).  Replaces _ext.gather_points (pointnet2_utils.py:112)."""
    _chk(features, "features", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 2)
    B, C, N = features.shape
    M = idx.shape[1]
    out = torch.empty((B, C, M), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        _lib.check(_lib.lib().ptt_gather_f32(_ptr(features), _ptr(idx), B, C, N, M, _ptr(out), _stream()),
                   "ptt_gather_f32")
    return out


def scatter_add_det(src, idx, N):
    """Deterministic out[b,c,n] = sum_{e: idx[b,e]==n} src[b,c,e] in ascending e (ptt_scatter_add_det_f32).
    src (B,C,E) f32, idx (B,E) i32 -> (B,C,N); E <= 16384."""
    B, C, E = src.shape
    out = torch.empty((B, C, int(N)), dtype=torch.float32, device=src.device)
    nbytes = _lib.lib().ptt_scatter_add_det_workspace(B, int(N), E)
    ws = torch.empty((max(1, (nbytes + 3) // 4),), dtype=torch.int32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().ptt_scatter_add_det_f32(_ptr(src), _ptr(idx), B, C, int(N), E, _ptr(out), _ptr(ws),
                                                      ws.numel() * 4, _stream()), "ptt_scatter_add_det_f32")
    return out


_ATOMIC_GRADS = os.environ.get("PTT_ATOMIC_GRADS", "0")      # read once at import; set_atomic_grads() at run time


def set_atomic_grads(on):
    """Backward of gather / group with upstream's atomicAdd behaviour (True) or in the fixed summation order (False,
    default). Returns the previous setting."""
    global _ATOMIC_GRADS
    prev = _ATOMIC_GRADS == "1"
    _ATOMIC_GRADS = "1" if on else "0"
    return prev


def _det_grads(entries, N):
    """The backward scatter-adds run in a fixed summation order (bit-reproducible, equal to the sequential loop) unless
    PTT_ATOMIC_GRADS=1 asks for upstream's atomicAdd behaviour, a cloud has more than 16384 entries, or the sort key
    idx * 16384 + e would not fit 32 bits (N >= 262144): those cases take the atomic kernels."""
    if int(N) * 16384 > 0xffffffff:
        return False
    return entries <= 16384 and _ATOMIC_GRADS != "1"


def gather_points_grad(grad_out, idx, N):
    """(B,C,M) f32, (B,M) i32 -> (B,C,N).  Replaces _ext.gather_points_grad (pointnet2_utils.py:118)."""
    _chk(grad_out, "grad_out", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 2)
    B, C, M = grad_out.shape
    if M > 0 and _det_grads(M, N):
        return scatter_add_det(grad_out, idx, N)
    out = torch.empty((B, C, int(N)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.lib().ptt_gather_grad_f32(_ptr(grad_out), _ptr(idx), B, C, int(N), M, _ptr(out), _stream()),
                   "ptt_gather_grad_f32")
    return out


def select_centres(xyz, idx, npoint, want_idx64=True):
    """new_xyz (B,npoint,3) = xyz[b, idx[b,:]] (idx None: the first npoint points) and idx as int64, one launch.
    Replaces the gather_operation + transposes + int64 cast of pointnet2_modules.py:79-81,90."""
    _chk(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    if idx is not None:
        _chk(idx, "idx", torch.int32, 2)
    new_xyz = torch.empty((B, int(npoint), 3), dtype=torch.float32, device=xyz.device)
    idx64 = torch.empty((B, int(npoint)), dtype=torch.int64, device=xyz.device) if (want_idx64 and idx is not None) else None
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().ptt_select_centres_f32(_ptr(xyz), _ptr(idx), B, N, int(npoint), _ptr(new_xyz), _ptr(idx64),
                                                     _stream()), "ptt_select_centres_f32")
    return new_xyz, idx64


GRID_BALL_QUERY_MIN_POINTS = 4096      # clouds from this size on take the uniform-grid ball query (same results)


def ball_query(new_xyz, xyz, radius, nsample):
    """centres first: (B,M,3), (B,N,3) -> (B,M,nsample) i32.  Replaces _ext.ball_query (pointnet2_utils.py:287)."""
    _chk(new_xyz, "new_xyz", torch.float32, 3)
    _chk(xyz, "xyz", torch.float32, 3)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    out = torch.empty((B, M, int(nsample)), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device), _timed('ptt_ball_query_f32'):
        if GRID_BALL_QUERY_MIN_POINTS <= N <= 131072:          # large clouds: 27 cells of a uniform grid instead of the whole cloud
            ws = _ws(_lib.lib().ptt_ball_query_grid_workspace(B, N), xyz.device)
            _lib.check(_lib.lib().ptt_ball_query_grid_f32(_ptr(new_xyz), _ptr(xyz), B, M, N, float(radius), int(nsample), _ptr(out),
                                                          _ptr(ws), ws.numel() * 8, _stream()), "ptt_ball_query_grid_f32")
        else:
            _lib.check(_lib.lib().ptt_ball_query_f32(_ptr(new_xyz), _ptr(xyz), B, M, N, float(radius), int(nsample),
                                                     _ptr(out), _stream()), "ptt_ball_query_f32")
    return out


def centres_ball_query(xyz, sel, npoint, radius, nsample, want_idx64=True):
    """select_centres + ball_query of one SA level in one launch: -> (new_xyz (B,npoint,3), idx64 (B,npoint) | None,
    idx (B,npoint,nsample) i32). sel: (B,npoint) i32 sample indices, or None for the first npoint points."""
    _chk(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    if sel is not None:
        _chk(sel, "sel", torch.int32, 2)
    M = int(npoint)
    new_xyz = torch.empty((B, M, 3), dtype=torch.float32, device=xyz.device)
    idx64 = torch.empty((B, M), dtype=torch.int64, device=xyz.device) if (want_idx64 and sel is not None) else None
    idx = torch.empty((B, M, int(nsample)), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device), _timed('ptt_ball_query_f32'):
        if GRID_BALL_QUERY_MIN_POINTS <= N <= 131072:
            ws = _ws(_lib.lib().ptt_ball_query_grid_workspace(B, N), xyz.device)
            _lib.check(_lib.lib().ptt_centres_ball_query_grid_f32(_ptr(xyz), _ptr(sel), B, N, M, float(radius), int(nsample), _ptr(new_xyz),
                                                                  _ptr(idx64), _ptr(idx), _ptr(ws), ws.numel() * 8, _stream()),
                       "ptt_centres_ball_query_grid_f32")
        else:
            _lib.check(_lib.lib().ptt_centres_ball_query_f32(_ptr(xyz), _ptr(sel), B, N, M, float(radius), int(nsample), _ptr(new_xyz),
                                                             _ptr(idx64), _ptr(idx), _stream()), "ptt_centres_ball_query_f32")
    return new_xyz, idx64, idx


def group_points(features, idx):
    """(B,C,N) f32, (B,M,ns) i32 -> (B,C,M,ns).  Replaces _ext.group_points (pointnet2_utils.py:237)."""
    _chk(features, "features", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 3)
    B, C, N = features.shape
    _, M, ns = idx.shape
    out = torch.empty((B, C, M, ns), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        _lib.check(_lib.lib().ptt_group_f32(_ptr(features), _ptr(idx), B, C, N, M, ns, _ptr(out), _stream()),
                   "ptt_group_f32")
    return out


def group_points_grad(grad_out, idx, N):
    """(B,C,M,ns) f32 -> (B,C,N).  Replaces _ext.group_points_grad (pointnet2_utils.py:257)."""
    _chk(grad_out, "grad_out", torch.float32, 4)
    _chk(idx, "idx", torch.int32, 3)
    B, C, M, ns = grad_out.shape
    if M * ns > 0 and _det_grads(M * ns, N):
        return scatter_add_det(grad_out.view(B, C, M * ns), idx.view(B, M * ns), N)
    out = torch.empty((B, C, int(N)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.lib().ptt_group_grad_f32(_ptr(grad_out), _ptr(idx), B, C, int(N), M, ns, _ptr(out),
                                                 _stream()), "ptt_group_grad_f32")
    return out


def _unreached(name):
    def fn(*a, **k):
        raise NotImplementedError("%s is never reached by PTT (SURVEY.md §2.1) and is not provided" % name)
    fn.__name__ = name
    return fn


three_nn = _unreached("three_nn")
three_interpolate = _unreached("three_interpolate")
three_interpolate_grad = _unreached("three_interpolate_grad")
furthest_point_sampling_with_dist = _unreached("furthest_point_sampling_with_dist")


# --------------------------------------------------------------------------- fused / MFMA ops
def knn(xyz, k, want_rel=False):
    """(B,N,3) f32 -> (B,N,k) i32 ascending by (squared distance, index).
    Replaces square_distance(xyz, xyz).argsort()[:, :, :k] (transformer_block/variants.py:150-151).
    want_rel: also return rel (B,N,k,3) = xyz_i - xyz_neighbour (variants.py:158), for pt_attn_pair."""
    _chk(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    out = torch.empty((B, N, int(k)), dtype=torch.int32, device=xyz.device)
    rel = torch.empty((B, N, int(k), 3), dtype=torch.float32, device=xyz.device) if want_rel else None
    with torch.cuda.device(xyz.device), _timed('ptt_knn_f32'):
        _lib.check(_lib.lib().ptt_knn_rel_f32(_ptr(xyz), B, N, int(k), _ptr(out), _ptr(rel), _stream()), "ptt_knn_rel_f32")
    return (out, rel) if want_rel else out


def pack_weight(weight, rot=0):
    """(Cout,K[,1[,1]]) f32 -> packed MFMA B-fragment buffer (1-D f32 tensor).
    rot: rotate the input channels left (the first SharedMLP layer of sa_fused_forward needs rot=3 when
    use_xyz: the kernel keeps grouped rows as [features | xyz])."""
    w = weight.detach()
    w = w.reshape(w.shape[0], -1).contiguous().float()
    _chk(w, "weight", torch.float32, 2)
    Cout, K = w.shape
    n = _lib.lib().ptt_packed_weight_elems(Cout, K)
    out = torch.empty((n,), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().ptt_pack_weight_rot_f32(_ptr(w), Cout, K, int(rot), _ptr(out), _stream()),
                   "ptt_pack_weight_rot_f32")
    return out


ONE_FRAME_MAX_SA_ROWS = 4096   # (= 4 frames of vote_aggregation, the batch range of ONE_FRAME_MAX_POINTS) grouped rows (B * npoint * nsample) up to which a hoisted SA level runs as two row-job launches
ROW_JOB_MAX_ROWS = 1024       # ops.linear hands launches of at most this many rows (and K >= 192) to ptt_row_jobs_f32
ONE_FRAME_MAX_POINTS = int(os.environ.get("PTT_PT_PER_LAYER_MAX", "512"))   # B * N up to which the modules take the one-frame launch chain


def linear(x, wpacked, cout, scale=None, shift=None, relu=False, residual=None, out=None):
    """Row-wise y = act(x @ W^T * scale + shift) (+ residual) on fp32 MFMA.
    x: (..., K) with contiguous last dim and uniform row stride; returns (..., cout)."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("x must be a float32 device tensor")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (int(cout),), dtype=torch.float32, device=x.device)
    o2 = out.view(-1, int(cout))
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, int(cout))
        if r2.stride(1) != 1:
            r2 = r2.contiguous()
    # above 8192 rows the persistent row GEMM is the faster launch where it takes the shape (12288 x 128 -> 128: 7.3 vs
    # 9.2 us, 12288 x 256 -> 128: 11.0 vs 13.5; profiles/r03p_linear_infer_bench.log); below, the short-launch linear kernel —
    # except 256-column layers of 6144+ rows (fc2 and cov_final of 48 frames), which the row GEMM's 128-column workgroups of
    # round 5 take faster: 6144 x 512 -> 256 18.7 vs 25.2 us, 6144 x 256 -> 256 11.1 vs 14.3 (1536 or 128 columns, 3072 rows: slower)
    if (scale is None and (rows > 8192 or (rows >= 6144 and int(cout) == 256)) and x2.data_ptr() % 16 == 0 and o2.is_contiguous()
            and _lib.lib().ptt_rows_gemm_supported(rows, K, int(cout), x2.stride(0), int(cout))):
        with torch.cuda.device(x.device), _timed('ptt_linear_f32'):
            _lib.check(_lib.lib().ptt_rows_gemm_f32(_ptr(x2), rows, K, x2.stride(0), None, None, _ptr(wpacked), int(cout), _ptr(shift),
                                                    1 if relu else 0, _ptr(r2), r2.stride(0) if r2 is not None else int(cout),
                                                    _ptr(o2), int(cout), None, 0, _stream()), "ptt_rows_gemm_f32")
        return out
    # at most 1024 rows of >= 192 channels (the launches of ONE tracklet frame): K split over the waves of a workgroup
    # (ptt_row_jobs_f32) — 128 x 512 -> 512: 9.0 against 12.8 us, 1024 x 512 -> 512: 10.9 against 13.1
    # (profiles/r04a_launch_floor.log); wider or shorter-K launches stay on the kernels below
    if rows <= ROW_JOB_MAX_ROWS and 192 <= K <= 1024 and o2.stride(1) == 1:
        row_jobs([row_job(wpacked, cout, x=x2, scale=scale, shift=shift, act=1 if relu else 0, res=r2, out=o2)])
        return out
    with torch.cuda.device(x.device), _timed('ptt_linear_f32'):
        _lib.check(_lib.lib().ptt_linear_f32(
            _ptr(x2), rows, K, x2.stride(0) if rows > 1 else K, _ptr(wpacked), int(cout), _ptr(scale), _ptr(shift),
            1 if relu else 0, _ptr(r2), (r2.stride(0) if (r2 is not None and rows > 1) else int(cout)),
            _ptr(o2), o2.stride(0) if rows > 1 else int(cout), _stream()), "ptt_linear_f32")
    return out


def sa_fused_forward(xyz, new_xyz, idx, features, layers, radius, use_xyz=True, normalize_xyz=False,
                     point_major_out=True, l0=None):
    """Fused group -> normalise -> SharedMLP(eval) -> max-pool (QueryAndGroup + SharedMLP + max_pool2d,
    pointnet2_utils.py:320-380, pytorch_utils.py:12-36, pointnet2_modules.py:84-88).

    xyz (B,N,3), new_xyz (B,M,3), idx (B,M,ns) i32, features (B,C,N) in ANY strides (a transposed
    view of point-major storage gathers coalesced) or None.
    layers: list of (wpacked, scale|None, shift|None, cin, cout, relu); layers[0] packed with rot=3 if use_xyz.
    l0: optional (point_term (B,N,C0) contiguous, xyz_weight (3,C0), relu) — layer 0 hoisted to one row per point
    (include/ptt_hip.h, ptt_sa_desc.l0_*); then features must be None and `layers` are the remaining layers.
    Returns (B,Cout,M); with point_major_out it is a transposed view of (B,M,Cout) storage."""
    _chk(xyz, "xyz", torch.float32, 3)
    _chk(new_xyz, "new_xyz", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 3)
    B, N, _ = xyz.shape
    _, M, ns = idx.shape
    d = SaDesc()
    d.xyz, d.new_xyz, d.idx = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr()
    if features is not None:
        if not features.is_cuda or features.dtype != torch.float32 or features.dim() != 3:
            raise RuntimeError("features must be a (B,C,N) float32 device tensor")
        d.feat = features.data_ptr()
        d.feat_sb, d.feat_sc, d.feat_sn = features.stride()
        d.C = features.shape[1]
    else:
        d.feat, d.C = None, 0
    cout = layers[-1][4]
    if point_major_out:
        store = torch.empty((B, M, cout), dtype=torch.float32, device=xyz.device)
        out = store.transpose(1, 2)
    else:
        out = torch.empty((B, cout, M), dtype=torch.float32, device=xyz.device)
    d.out = out.data_ptr()
    d.out_sb, d.out_sc, d.out_sm = out.stride()
    d.B, d.N, d.M, d.nsample = B, N, M, ns
    d.radius, d.use_xyz, d.normalize_xyz = float(radius), int(bool(use_xyz)), int(bool(normalize_xyz))
    d.n_layers = len(layers)
    for i, (wp, sc, sh, cin, co, relu) in enumerate(layers):
        L = d.layers[i]
        L.Wpacked = wp.data_ptr()
        L.scale = sc.data_ptr() if sc is not None else None
        L.shift = sh.data_ptr() if sh is not None else None
        L.Cin, L.Cout, L.relu = int(cin), int(co), int(bool(relu))
    if l0 is not None:
        term, wx, l0_relu = l0
        if features is not None:
            raise RuntimeError("with a hoisted layer 0 the point features are already inside l0[0]")
        _chk(term, "l0 point term", torch.float32, 3)
        _chk(wx, "l0 xyz weight", torch.float32, 2)
        if term.shape[0] != B or term.shape[1] != N or wx.shape[0] != 3 or wx.shape[1] != term.shape[2]:
            raise RuntimeError("l0 point term must be (B,N,C0) and the xyz weight (3,C0)")
        d.l0_point_term, d.l0_xyz_weight = term.data_ptr(), wx.data_ptr()
        d.l0_channels, d.l0_relu = term.shape[2], int(bool(l0_relu))
    with torch.cuda.device(xyz.device), _timed('ptt_sa_fused_fwd_f32'):
        _lib.check(_lib.lib().ptt_sa_fused_fwd_f32(ctypes.byref(d), _stream()), "ptt_sa_fused_fwd_f32")
    return out


def cosine_map(search_feats, templ_feats, eps=1e-8):
    """cos_t (B,Ns,Nt) = F.cosine_similarity(templ_i, search_j) for every pair (p2b_xcoor.py:35-36); inputs (B,C,N) in
    any strides."""
    for t_, n_ in ((search_feats, "search_feats"), (templ_feats, "templ_feats")):
        if not isinstance(t_, torch.Tensor) or not t_.is_cuda or t_.dtype != torch.float32 or t_.dim() != 3:
            raise RuntimeError("%s must be a (B,C,N) float32 device tensor" % n_)
    if search_feats.shape[:2] != templ_feats.shape[:2] or search_feats.device != templ_feats.device:
        raise RuntimeError("search_feats and templ_feats must share batch size, channel count and device")
    B, C, Ns = search_feats.shape
    Nt = templ_feats.shape[2]
    out = torch.empty((B, Ns, Nt), dtype=torch.float32, device=search_feats.device)
    ssb, ssc, ssn = search_feats.stride()
    tsb, tsc, tsn = templ_feats.stride()
    with torch.cuda.device(out.device), _timed('ptt_cosine_map_f32'):
        _lib.check(_lib.lib().ptt_cosine_map_f32(search_feats.data_ptr(), ssb, ssn, ssc, templ_feats.data_ptr(), tsb, tsn,
                                                 tsc, B, Ns, Nt, C, float(eps), out.data_ptr(), _stream()),
                   "ptt_cosine_map_f32")
    return out


def rows_mlp(x, layers, residual=None):
    """A stack of 1x1 convolutions over rows in ONE launch — ptt_rows_mlp_f32. x (..., K) float32 with contiguous rows,
    layers = [(wpacked, scale | None, shift | None, cin, cout, relu)] (at most 4; inner cout <= 256, last <= 384, K <= 264),
    residual (..., cout_last) added to the result. -> (..., cout_last)."""
    if not x.is_cuda or x.dtype != torch.float32 or x.stride(-1) != 1:
        raise RuntimeError("rows_mlp: x must be a float32 device tensor with contiguous rows")
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    rows = x2.shape[0]
    cout = int(layers[-1][4])
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    res2 = None
    if residual is not None:
        res2 = residual.reshape(-1, cout)
        _chk(res2 if res2.is_contiguous() else res2.contiguous(), "residual", torch.float32, 2)
        res2 = res2 if res2.stride(1) == 1 else res2.contiguous()
    arr = (SaLayer * len(layers))()
    for i, (wp, sc, sh, cin, co, relu) in enumerate(layers):
        arr[i].Wpacked = wp.data_ptr()
        arr[i].scale = sc.data_ptr() if sc is not None else None
        arr[i].shift = sh.data_ptr() if sh is not None else None
        arr[i].Cin, arr[i].Cout, arr[i].relu = int(cin), int(co), int(bool(relu))
    with torch.cuda.device(x.device), _timed('ptt_rows_mlp_f32'):
        _lib.check(_lib.lib().ptt_rows_mlp_f32(_ptr(x2), rows, K, x2.stride(0), arr, len(layers), _ptr(res2),
                                               res2.stride(0) if res2 is not None else 0, _ptr(out), cout, _stream()),
                   "ptt_rows_mlp_f32")
    return out.view(*x.shape[:-1], cout)


def xcorr_fused(search_feats, templ_feats, P, w_sim, scale0, shift0, layers, eps=1e-8, want_sim=False, cos_t=None, split=False):
    """Fused CosineSimAug core (similarity_modules/p2b_xcoor.py:25-42): cosine map, concat, SharedMLP, max over
    the template axis. search_feats (B,C,Ns) / templ_feats (B,C,Nt) in any strides; P (B,Nt,C0) = layer-0
    pre-activation without the similarity term; layers = remaining (wpacked, scale, shift, cin, cout, relu).
    cos_t: the (B,Ns,Nt) map of cosine_map() if the caller already has it (else computed here, one launch).
    split: a handful of frames — two workgroups per search point, cosines inside the kernel (no cosine_map launch); needs
    point-major features (unit channel stride), B * Ns % 8 == 0, no want_sim; silently the plain form otherwise. The split
    form returns the two halves' maxima, out (2,B,Cout,Ns) views: the maximum over the template axis is out[0].maximum(out[1])
    (row_job(x=..., xmax=...) takes it while it stages its operand).
    Returns (out (B,Cout,Ns) as a view of point-major storage, sim (B,Nt,Ns) | None)."""
    for t_, n_ in ((search_feats, "search_feats"), (templ_feats, "templ_feats")):
        if not t_.is_cuda or t_.dtype != torch.float32 or t_.dim() != 3:
            raise RuntimeError("%s must be a (B,C,N) float32 device tensor" % n_)
    _chk(P, "P", torch.float32, 3)
    _chk(w_sim, "w_sim", torch.float32, 1)
    B, C, Ns = search_feats.shape
    Nt = templ_feats.shape[2]
    C0 = P.shape[2]
    if Nt <= 0 or Nt % 64:
        raise RuntimeError("xcorr_fused: Nt=%d (ptt_xcorr_fused_fwd_f32 walks the template seeds in chunks of 64)" % Nt)
    if P.shape[0] != B or P.shape[1] != Nt or w_sim.shape[0] != C0 or templ_feats.shape[:2] != search_feats.shape[:2]:
        raise RuntimeError("xcorr_fused: inconsistent shapes")
    cout = layers[-1][4]
    split = (split and not want_sim and search_feats.stride(1) == 1 and templ_feats.stride(1) == 1 and C % 4 == 0 and (B * Ns) % 8 == 0
             and all(st % 4 == 0 for st in (search_feats.stride(0), search_feats.stride(2), templ_feats.stride(0), templ_feats.stride(2)))
             and search_feats.data_ptr() % 16 == 0 and templ_feats.data_ptr() % 16 == 0 and bool(layers[-1][5]))
    if not split:
        if cos_t is None:
            cos_t = cosine_map(search_feats, templ_feats, eps=eps)
        _chk(cos_t, "cos_t", torch.float32, 3)
        if tuple(cos_t.shape) != (B, Ns, Nt):
            raise RuntimeError("xcorr_fused: cos_t must be (B,Ns,Nt)")
    store = torch.empty(((2, B, Ns, cout) if split else (B, Ns, cout)), dtype=torch.float32, device=P.device)
    out = store.transpose(-1, -2)
    sim = torch.empty((B, Nt, Ns), dtype=torch.float32, device=P.device) if want_sim else None
    d = XcorrDesc()
    if split:
        d.split, d.out_sh = 1, store.stride(0)
        d.search_feat, d.templ_feat = search_feats.data_ptr(), templ_feats.data_ptr()
        d.s_sb, d.s_sn, d.t_sb, d.t_sn = search_feats.stride(0), search_feats.stride(2), templ_feats.stride(0), templ_feats.stride(2)
        d.C, d.eps = int(C), float(eps)
    else:
        d.cos_t = cos_t.data_ptr()
    d.P, d.w_sim = P.data_ptr(), w_sim.data_ptr()
    d.scale0 = scale0.data_ptr() if scale0 is not None else None
    d.shift0 = shift0.data_ptr() if shift0 is not None else None
    d.out = out.data_ptr()
    d.out_sb, d.out_sc, d.out_sn = out.stride()[-3:]
    d.sim_out = sim.data_ptr() if sim is not None else None
    d.B, d.Ns, d.Nt, d.C0 = B, Ns, Nt, C0
    d.n_layers = len(layers)
    for i, (wp, sc, sh, cin, co, relu) in enumerate(layers):
        L = d.layers[i]
        L.Wpacked = wp.data_ptr()
        L.scale = sc.data_ptr() if sc is not None else None
        L.shift = sh.data_ptr() if sh is not None else None
        L.Cin, L.Cout, L.relu = int(cin), int(co), int(bool(relu))
    with torch.cuda.device(P.device), _timed('ptt_xcorr_fused_fwd_f32'):
        _lib.check(_lib.lib().ptt_xcorr_fused_fwd_f32(ctypes.byref(d), _stream()), "ptt_xcorr_fused_fwd_f32")
    return out, sim


def pack_delta0(weight, bias):
    """[fc_delta[0].weight (D,3) | fc_delta[0].bias (D)] in packed MFMA order (K = 4): ptt_attn_desc.Wd1p."""
    return pack_weight(torch.cat([weight.detach().float().reshape(weight.shape[0], 3),
                                  bias.detach().float().reshape(-1, 1)], dim=1).contiguous())


def spatial_order(xyz):
    """(B,N,3) -> (B,N) int32: every cloud's points along a Morton curve through its bounding box, as flat indices b*N + n
    (ptt_spatial_order_f32, N <= 8192) — the launch order that keeps pt_attn_pair's neighbour gathers in L2."""
    _chk(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    order = torch.empty((B, N), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().ptt_spatial_order_f32(_ptr(xyz), B, N, _ptr(order), _stream()), "ptt_spatial_order_f32")
    return order


def pt_attn_pair(xyz, knn_idx, qkv, wd1p, wd2p, bd2, wg1p, bg1, wg2p, bg2, d_model, want_attn=True, rel=None, order=None):
    """Fused per-(point,neighbour) part of TransformerBlock.forward (variants.py:158-163).
    wd1p = pack_delta0(fc_delta[0].weight, fc_delta[0].bias); the other weights from pack_weight. order: spatial_order(xyz) or
    None — which point every launch slot works on (the results do not depend on it).
    Returns (res (B,N,D), attn (B,N,k,D) | None)."""
    _chk(xyz, "xyz", torch.float32, 3)
    _chk(knn_idx, "knn_idx", torch.int32, 3)
    _chk(qkv, "qkv", torch.float32, 3)
    B, N, _ = xyz.shape
    k = knn_idx.shape[2]
    D = int(d_model)
    res = torch.empty((B, N, D), dtype=torch.float32, device=xyz.device)
    attn = torch.empty((B, N, k, D), dtype=torch.float32, device=xyz.device) if want_attn else None
    d = AttnDesc()
    d.xyz, d.knn, d.qkv = xyz.data_ptr(), knn_idx.data_ptr(), qkv.data_ptr()
    d.rel = rel.data_ptr() if rel is not None else None
    d.Wd1p, d.Wd2p, d.bd2 = wd1p.data_ptr(), wd2p.data_ptr(), bd2.data_ptr()
    d.Wg1p, d.bg1, d.Wg2p, d.bg2 = wg1p.data_ptr(), bg1.data_ptr(), wg2p.data_ptr(), bg2.data_ptr()
    d.res = res.data_ptr()
    d.attn = attn.data_ptr() if attn is not None else None
    d.B, d.N, d.k, d.D = B, N, k, D
    if order is not None:
        _chk(order, "order", torch.int32, 2)
        if tuple(order.shape) != (B, N):
            raise ValueError("order: (B,N) int32 expected")
        d.order = order.data_ptr()
    with torch.cuda.device(xyz.device), _timed('ptt_pt_attn_pair_f32'):
        _lib.check(_lib.lib().ptt_pt_attn_pair_f32(ctypes.byref(d), _stream()), "ptt_pt_attn_pair_f32")
    return res, attn


# --------------------------------------------------------------------------- N4: tracking-loop pre/post-processing
import numpy as np      # noqa: E402  (host-side job tables only)

CROP_JOB = np.dtype([('points', '<u8'), ('ld', '<i8'), ('lo1', '<f8', 3), ('hi1', '<f8', 3), ('trans', '<f8', 3),
                     ('rot', '<f8', 9), ('lo2', '<f8', 3), ('hi2', '<f8', 3), ('out', '<u8'), ('count', '<u8'),
                     ('n_points', '<i4'), ('capacity', '<i4'), ('label_out', '<u8'), ('ltrans', '<f8', 3), ('lrot', '<f8', 9),
                     ('llo', '<f8', 3), ('lhi', '<f8', 3)])                          # = ptt_crop_job, 384 bytes
REGULARIZE_JOB = np.dtype([('seg', '<u8', 4), ('seg_count', '<u8', 4), ('seg_capacity', '<i4', 4), ('out', '<u8'),
                           ('info', '<u8'), ('n_seg', '<i4'), ('input_size', '<i4')])   # = ptt_regularize_job, 104 bytes
assert CROP_JOB.itemsize == ctypes.sizeof(_lib.CropJob) and REGULARIZE_JOB.itemsize == ctypes.sizeof(_lib.RegularizeJob)

TRACK_BOX = np.dtype([('center', '<f8', 3), ('wlh', '<f8', 3), ('quat', '<f8', 4)])     # = ptt_track_box, 80 bytes


def track_crop_bounds(boxes, offset, scale, extra2, jobs, job_stride=1):
    """ptt_track_crop_bounds: the float64 crop quantities of crop_center_pc for every box (TRACK_BOX array) into the
    lo1/hi1/trans/rot/lo2/hi2 fields of jobs[i * job_stride] (a CROP_JOB array, typically a view of pinned memory)."""
    ex = None if extra2 is None else np.ascontiguousarray(extra2, np.float64)
    _lib.check(_lib.lib().ptt_track_crop_bounds(boxes.ctypes.data, len(boxes), float(offset), float(scale),
                                                ex.ctypes.data if ex is not None else None, jobs.ctypes.data,
                                                int(job_stride)), "ptt_track_crop_bounds")


def track_box_by_offset(boxes, offsets, use_z, active=None, rng_pos=None):
    """ptt_track_box_by_offset: boxes[i] <- get_box_by_offset(boxes[i], offsets[i, 0:4], use_z) in place (TRACK_BOX array,
    float32 (n, >=4) C-contiguous offsets, optional int32 active mask and int64 generator positions)."""
    assert offsets.dtype == np.float32 and offsets.flags['C_CONTIGUOUS'] and offsets.shape[1] >= 4
    _lib.check(_lib.lib().ptt_track_box_by_offset(boxes.ctypes.data, len(boxes), offsets.ctypes.data, offsets.shape[1],
                                                  int(bool(use_z)), active.ctypes.data if active is not None else None,
                                                  rng_pos.ctypes.data if rng_pos is not None else None),
               "ptt_track_box_by_offset")


def track_select_update(proposals, info, boxes, use_z, active, rng_pos, est_out):
    """ptt_track_select_update: the host side of one step's post-processing in one call — proposals: float32 host array (B,P,5) or
    (B,5); info: int32 host array (B,2,2); boxes TRACK_BOX (B,) in place; rng_pos int64 (B,) in place; est_out float32 (B,5)."""
    B = len(boxes)
    P = proposals.shape[1] if proposals.ndim == 3 else 1
    assert proposals.dtype == np.float32 and proposals.flags['C_CONTIGUOUS'] and info.dtype == np.int32 and info.flags['C_CONTIGUOUS']
    assert est_out.dtype == np.float32 and est_out.shape == (B, 5) and est_out.flags['C_CONTIGUOUS'] and rng_pos.dtype == np.int64
    rc = _lib.lib().ptt_track_select_update(proposals.ctypes.data, P, info.ctypes.data, boxes.ctypes.data, B, int(bool(use_z)),
                                            active.ctypes.data if active is not None else None, rng_pos.ctypes.data, est_out.ctypes.data)
    _lib.check(rc, "ptt_track_select_update")


_mt_tables = {}


def mt19937_draws(device, n=8192, seed=1):
    """The first n 32-bit outputs of MT19937(seed) as a device uint32 buffer (viewed int32): the index stream of
    regularize_pc's np.random.randint after set_manual_seed(1) (kitti_tracking_utils.py:349-353). Built once per
    (device, n, seed) by ptt_mt19937_fill and uploaded."""
    key = (str(device), int(n), int(seed))
    if key not in _mt_tables:
        host = np.empty(int(n), np.uint32)
        _lib.check(_lib.lib().ptt_mt19937_fill(int(seed), host.ctypes.data, int(n)), "ptt_mt19937_fill")
        _mt_tables[key] = torch.from_numpy(host.view(np.int32)).to(device)
        publish_params(torch.device(device))
    return _mt_tables[key]


def upload_jobs(jobs, out=None):
    """numpy structured job table -> device bytes (non-blocking copy from pinned memory on the current stream).
    `out`: an existing device uint8 buffer of the same size to refill (fixed address: hipGraph replays read it)."""
    host = torch.from_numpy(np.ascontiguousarray(jobs).view(np.uint8).reshape(-1)).pin_memory()
    if out is None:
        return host.to('cuda', non_blocking=True)
    out.copy_(host, non_blocking=True)
    return out


def crop_compact(jobs_dev, n_jobs):
    """ptt_crop_compact_f32 over a device-resident table of `n_jobs` ptt_crop_job records (uint8 tensor)."""
    with torch.cuda.device(jobs_dev.device), _timed('ptt_crop_compact_f32'):
        _lib.check(_lib.lib().ptt_crop_compact_f32(_ptr(jobs_dev), int(n_jobs), _stream()), "ptt_crop_compact_f32")


CROP_JOBS_BY_VALUE_MAX = 8


def crop_compact_host(jobs_np, n_jobs, device):
    """ptt_crop_compact_host_f32: the crops of a host-resident (numpy structured) job table, passed by value with the launch —
    at most CROP_JOBS_BY_VALUE_MAX jobs; the table may be rewritten as soon as this returns."""
    with torch.cuda.device(device), _timed('ptt_crop_compact_f32'):
        _lib.check(_lib.lib().ptt_crop_compact_host_f32(ctypes.c_void_p(jobs_np.ctypes.data), int(n_jobs), _stream()),
                   "ptt_crop_compact_host_f32")


def crop_compact_pinned(jobs_pinned, n_jobs, device):
    """ptt_crop_compact_f32 reading its job table straight from PINNED host memory (device-visible under unified addressing):
    the launch can then sit inside a hipGraph whose table the host rewrites between replays — no upload, no per-frame launch."""
    with torch.cuda.device(device), _timed('ptt_crop_compact_f32'):
        _lib.check(_lib.lib().ptt_crop_compact_f32(ctypes.c_void_p(jobs_pinned.data_ptr()), int(n_jobs), _stream()), "ptt_crop_compact_f32")


def crop_regularize_pinned(crop_jobs_pinned, reg_jobs_dev, n_jobs, draws):
    """ptt_crop_regularize_f32: crop job w then resampling job w per workgroup, the crop table read from pinned host memory."""
    with torch.cuda.device(reg_jobs_dev.device), _timed('ptt_crop_compact_f32'):
        _lib.check(_lib.lib().ptt_crop_regularize_f32(ctypes.c_void_p(crop_jobs_pinned.data_ptr()), _ptr(reg_jobs_dev), int(n_jobs),
                                                      _ptr(draws), draws.numel(), _stream()), "ptt_crop_regularize_f32")


def regularize(jobs_dev, n_jobs, draws):
    """ptt_regularize_f32 over a device-resident table of ptt_regularize_job records."""
    with torch.cuda.device(jobs_dev.device), _timed('ptt_regularize_f32'):
        _lib.check(_lib.lib().ptt_regularize_f32(_ptr(jobs_dev), int(n_jobs), _ptr(draws), draws.numel(), _stream()),
                   "ptt_regularize_f32")


def select_box(pred_box_data, out=None, idx_out=None):
    """(B,P,5) f32 -> (B,5): per frame the proposal row with the highest score (first among equals), replacing the
    `.cpu().numpy()` of the whole tensor + np.argmax of eval_tracking_utils.py:267-269."""
    _chk(pred_box_data, "pred_box_data", torch.float32, 3)
    B, P, five = pred_box_data.shape
    if five != 5:
        raise RuntimeError("pred_box_data must be (B,P,5)")
    if out is None:
        out = torch.empty((B, 5), dtype=torch.float32, device=pred_box_data.device)
    with torch.cuda.device(pred_box_data.device):
        _lib.check(_lib.lib().ptt_select_box_f32(_ptr(pred_box_data), B, P, _ptr(out), _ptr(idx_out), _stream()),
                   "ptt_select_box_f32")
    return out


# --------------------------------------------------------------------------- N3: training-step kernels (rows x channels)
WGRAD2 = os.environ.get("PTT_WGRAD2", "1") != "0"        # dev A/B: the round-2 128 x 128-block weight-gradient kernel


def _rows(t, name):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or (t.stride(1) != 1 and t.shape[1] != 1):
        raise RuntimeError("%s must be a (rows, channels) float32 device tensor with contiguous channels" % name)
    return t


def _ws(nbytes, device):
    return torch.empty((max(1, (int(nbytes) + 7) // 8),), dtype=torch.float64, device=device)


def _bn_tail(bn, C, device):
    """ptt_bn_train_tail for the training-mode BatchNorm module `bn` -> (structure, act_a, act_b): the launch that forms the
    batch statistics also writes the deferred activation's constants a = gamma * invstd, b = beta - mean * a and does the
    module's bookkeeping (running statistics, batch counter) in place."""
    a, b = (torch.empty((C,), dtype=torch.float32, device=device) for _ in range(2))
    for name in ("weight", "bias", "running_mean", "running_var"):
        t = getattr(bn, name)
        if t is None or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != C or t.device != device:
            raise RuntimeError("BatchNorm.%s must be a contiguous float32 (%d,) tensor on %s" % (name, C, device))
    tracked = bn.num_batches_tracked
    if tracked is not None and (tracked.dtype != torch.int64 or tracked.device != device):
        raise RuntimeError("BatchNorm.num_batches_tracked must be an int64 tensor on %s" % device)
    tail = _lib.BnTrainTail(gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(), act_a=a.data_ptr(), act_b=b.data_ptr(),
                            running_mean=bn.running_mean.data_ptr(), running_var=bn.running_var.data_ptr(),
                            num_batches_tracked=tracked.data_ptr() if tracked is not None else None, momentum=float(bn.momentum))
    return tail, a, b


def _bn_tail_done(bn):
    for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):      # the eval-mode parameter caches key on tensor versions
        if t is not None:
            torch.autograd.graph.increment_version(t)


def bn_stats(x, eps, bn=None):
    """Per-channel batch statistics of x (R,C): (mean, biased var, invstd) — ptt_bn_stats_f32. bn: the training-mode
    BatchNorm module these statistics belong to -> (mean, var, invstd, a, b) with the deferred activation's constants, and the
    module's running statistics / batch counter updated, all by the same two launches (ptt_bn_stats_train_f32)."""
    _rows(x, "x")
    R, C = x.shape
    mean, var, invstd = (torch.empty((C,), dtype=torch.float32, device=x.device) for _ in range(3))
    nb = _lib.lib().ptt_bn_stats_workspace(R, C)
    ws = _ws(nb, x.device)
    if bn is not None:
        tail, a, b = _bn_tail(bn, C, x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ptt_bn_stats_train_f32(_ptr(x), R, C, x.stride(0), float(eps), _ptr(mean), _ptr(var), _ptr(invstd),
                                                         _ptr(ws), ws.numel() * 8, ctypes.byref(tail), _stream()),
                       "ptt_bn_stats_train_f32")
        _bn_tail_done(bn)
        return mean, var, invstd, a, b
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_bn_stats_f32(_ptr(x), R, C, x.stride(0), float(eps), _ptr(mean), _ptr(var), _ptr(invstd),
                                               _ptr(ws), ws.numel() * 8, _stream()), "ptt_bn_stats_f32")
    return mean, var, invstd


def bn_apply(z, mean, invstd, gamma, beta, relu=True, out=None):
    """relu?((z - mean) * invstd * gamma + beta) over rows — ptt_bn_apply_f32."""
    _rows(z, "z")
    R, C = z.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_apply_f32(_ptr(z), z.stride(0), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), R, C,
                                               int(bool(relu)), _ptr(out), out.stride(0), _stream()), "ptt_bn_apply_f32")
    return out


def bn_bwd(g, act, z, mean, invstd, gamma, out=None, act_scale=None, act_shift=None):
    """Backward of BatchNorm(train) + ReLU on rows: -> (dz, dgamma, dbeta). `out` may alias g (in place). act = the
    layer's output, or None with act_scale / act_shift: the ReLU mask is then z * act_scale + act_shift > 0."""
    _rows(g, "g"); _rows(z, "z")
    if act is not None:
        _rows(act, "act")
    R, C = z.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=z.device)
    dgamma = torch.empty((C,), dtype=torch.float32, device=z.device)
    dbeta = torch.empty((C,), dtype=torch.float32, device=z.device)
    ws = _ws(_lib.lib().ptt_bn_stats_workspace(R, C), z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_f32(_ptr(g), g.stride(0), _ptr(act), act.stride(0) if act is not None else 0, _ptr(z),
                                             z.stride(0), _ptr(mean), _ptr(invstd), _ptr(gamma), R, C, 1, _ptr(out),
                                             out.stride(0), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel() * 8,
                                             _ptr(act_scale), _ptr(act_shift), _stream()), "ptt_bn_bwd_f32")
    return out, dgamma, dbeta


def bn_sums(x):
    """SyncBatchNorm forward, local part: 2C + 1 float64 = per-channel sum, sum of squares of x (R,C), then the row count
    (one all-reduce carries all three) — ptt_bn_sums_f64."""
    _rows(x, "x")
    R, C = x.shape
    sums = torch.empty((2 * C + 1,), dtype=torch.float64, device=x.device)
    ws = _ws(_lib.lib().ptt_bn_stats_workspace(R, C), x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_bn_sums_f64(_ptr(x), R, C, x.stride(0), _ptr(sums), _ptr(ws), ws.numel() * 8, _stream()),
                   "ptt_bn_sums_f64")
    return sums


def bn_finish(sums, eps):
    """(mean, biased var, invstd) from the (all-reduced) output of bn_sums; the row count is its last element and stays on
    the device — ptt_bn_finish_f64."""
    C = (sums.numel() - 1) // 2
    mean, var, invstd = (torch.empty((C,), dtype=torch.float32, device=sums.device) for _ in range(3))
    with torch.cuda.device(sums.device):
        _lib.check(_lib.lib().ptt_bn_finish_f64(_ptr(sums), C, float(eps), _ptr(mean), _ptr(var), _ptr(invstd), _stream()),
                   "ptt_bn_finish_f64")
    return mean, var, invstd


def bn_bwd_sums(g, act, z, mean, invstd, act_scale=None, act_shift=None):
    """SyncBatchNorm backward, local part: (2, C) float64 = (sum dy, sum dy * xhat) with dy = g under the ReLU mask."""
    _rows(g, "g"); _rows(z, "z")
    R, C = z.shape
    sums = torch.empty((2, C), dtype=torch.float64, device=z.device)
    ws = _ws(_lib.lib().ptt_bn_stats_workspace(R, C), z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_sums_f64(_ptr(g), g.stride(0), _ptr(act), act.stride(0) if act is not None else 0, _ptr(z),
                                                  z.stride(0), _ptr(mean), _ptr(invstd), R, C, _ptr(sums), _ptr(ws),
                                                  ws.numel() * 8, _ptr(act_scale), _ptr(act_shift), _stream()),
                   "ptt_bn_bwd_sums_f64")
    return sums


def bn_bwd_apply(g, act, z, mean, invstd, gamma, sum_dy, sum_dy_xhat, count, out=None, act_scale=None, act_shift=None):
    """dz of BatchNorm(train) + ReLU from the (global) float32 sums; count = a one-element float64 device tensor (the global
    row count, e.g. bn_sums()[-1:] after the all-reduce); `out` may alias g."""
    _rows(g, "g"); _rows(z, "z")
    R, C = z.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_apply_f32(_ptr(g), g.stride(0), _ptr(act), act.stride(0) if act is not None else 0, _ptr(z),
                                                   z.stride(0), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sum_dy),
                                                   _ptr(sum_dy_xhat), _ptr(count), R, C, _ptr(out), out.stride(0),
                                                   _ptr(act_scale), _ptr(act_shift), _stream()), "ptt_bn_bwd_apply_f32")
    return out


def bn_update_running(bn, mean, var, count):
    """nn.BatchNorm's training-mode bookkeeping for module `bn` from the batch statistics, one launch
    (ptt_bn_update_running_f32); count: float64 (1,) device tensor = rows the statistics were taken over."""
    tracked = bn.num_batches_tracked
    with torch.cuda.device(mean.device):
        _lib.check(_lib.lib().ptt_bn_update_running_f32(_ptr(mean), _ptr(var), _ptr(count), float(bn.momentum), mean.numel(),
                                                        _ptr(bn.running_mean), _ptr(bn.running_var),
                                                        _ptr(tracked) if tracked is not None else None, _stream()),
                   "ptt_bn_update_running_f32")
    for t in (bn.running_mean, bn.running_var, tracked):          # the eval-mode parameter caches key on tensor versions
        if t is not None:
            torch.autograd.graph.increment_version(t)


def xcorr_z0(P, cos_t, w_sim, want_stats=False):
    """z0[b,j,i,:] = P[b,i,:] + cos_t[b,j,i] * w_sim — (B*n2*n1, C0) rows ordered (b, j, i) — ptt_xcorr_z0_f32. want_stats:
    -> (z0, the float64 partial sums (chunks, 2, C0) of z0's BatchNorm statistics summed by the same launch, or None)."""
    B, n1, C = P.shape
    n2 = cos_t.shape[1]
    z0 = torch.empty((B * n2 * n1, C), dtype=torch.float32, device=P.device)
    chunks = _lib.lib().ptt_xcorr_z0_stat_chunks(B, n2, n1, C) if want_stats else 0
    with torch.cuda.device(P.device):
        if chunks:
            part = torch.empty((chunks, 2, C), dtype=torch.float64, device=P.device)
            _lib.check(_lib.lib().ptt_xcorr_z0_stats_f32(_ptr(P), _ptr(cos_t), _ptr(w_sim), B, n2, n1, C, _ptr(z0), _ptr(part), part.numel(),
                                                         _stream()), "ptt_xcorr_z0_stats_f32")
            return z0, part
        _lib.check(_lib.lib().ptt_xcorr_z0_f32(_ptr(P), _ptr(cos_t), _ptr(w_sim), B, n2, n1, C, _ptr(z0), _stream()), "ptt_xcorr_z0_f32")
    return (z0, None) if want_stats else z0


def xcorr_z0_bwd(dz0, cos_t, w_sim, B, n2, n1):
    """-> (dP (B,n1,C), dcos (B,n2,n1), dw (C,)) in one pass over dz0 (B*n2*n1, C) — ptt_xcorr_z0_bwd_f32."""
    C = dz0.shape[1]
    dP = torch.empty((B, n1, C), dtype=torch.float32, device=dz0.device)
    dcos = torch.empty((B, n2, n1), dtype=torch.float32, device=dz0.device)
    dw = torch.empty((C,), dtype=torch.float32, device=dz0.device)
    ws = _ws(_lib.lib().ptt_xcorr_z0_bwd_workspace(B, n1, C), dz0.device)
    with torch.cuda.device(dz0.device):
        _lib.check(_lib.lib().ptt_xcorr_z0_bwd_f32(_ptr(dz0), _ptr(cos_t), _ptr(w_sim), B, n2, n1, C, _ptr(dP), _ptr(dcos), _ptr(dw),
                                                   _ptr(ws), ws.numel() * 8, _stream()), "ptt_xcorr_z0_bwd_f32")
    return dP, dcos, dw


def xcorr_z0_bnbwd(part, g, P, cos_t, w_sim, mean, invstd, gamma, act_scale, act_shift):
    """CosineSimAug's layer 0 backward from the gradient g (B*n2*n1, C) of its ACTIVATED output and the BatchNorm-backward partial
    sums rows_gemm_bnbwd took with it: -> (dP (B,n1,C), dcos (B,n2,n1), dw (C,), dgamma, dbeta) in one pass over g — the apply pass
    and the pass over dz0 of xcorr_z0_bwd folded together, z0 recomputed — ptt_xcorr_z0_bnbwd_f32."""
    B, n1, C = P.shape
    n2 = cos_t.shape[1]
    dev = g.device
    dP = torch.empty((B, n1, C), dtype=torch.float32, device=dev)
    dcos = torch.empty((B, n2, n1), dtype=torch.float32, device=dev)
    dw, dgamma, dbeta = (torch.empty((C,), dtype=torch.float32, device=dev) for _ in range(3))
    ws = _ws(_lib.lib().ptt_xcorr_z0_bwd_workspace(B, n1, C), dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ptt_xcorr_z0_bnbwd_f32(_ptr(part), part.shape[0], _ptr(g), _ptr(P), _ptr(cos_t), _ptr(w_sim), _ptr(mean), _ptr(invstd),
                                                     _ptr(gamma), _ptr(act_scale), _ptr(act_shift), B, n2, n1, C, _ptr(dP), _ptr(dcos), _ptr(dw),
                                                     _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel() * 8, _stream()), "ptt_xcorr_z0_bnbwd_f32")
    return dP, dcos, dw, dgamma, dbeta


def bn_bwd_pooled(dpooled, arg, ns, z, mean, invstd, gamma, act_scale, act_shift, out=None):
    """bn_bwd for the last layer of a SharedMLP + max-pool stage with the gradient still pooled (dpooled (G,C), arg (G,C)
    int32 from pool_rows): -> (dz (G*ns,C), dgamma, dbeta) — ptt_bn_bwd_pooled_f32."""
    _rows(dpooled, "dpooled"); _rows(z, "z")
    R, C = z.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=z.device)
    dgamma = torch.empty((C,), dtype=torch.float32, device=z.device)
    dbeta = torch.empty((C,), dtype=torch.float32, device=z.device)
    ws = _ws(_lib.lib().ptt_bn_stats_workspace(R, C), z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_pooled_f32(_ptr(dpooled), dpooled.stride(0), _ptr(arg), int(ns), _ptr(z), z.stride(0), _ptr(mean),
                                                    _ptr(invstd), _ptr(gamma), R, C, _ptr(out), out.stride(0), _ptr(dgamma), _ptr(dbeta),
                                                    _ptr(ws), ws.numel() * 8, _ptr(act_scale), _ptr(act_shift), _stream()),
                   "ptt_bn_bwd_pooled_f32")
    return out, dgamma, dbeta


def bn_bwd_pooled_sums(dpooled, arg, ns, z, mean, invstd, act_scale, act_shift):
    """SyncBatchNorm backward, local part, pooled gradient: (2, C) float64 = (sum dy, sum dy * xhat)."""
    _rows(dpooled, "dpooled"); _rows(z, "z")
    R, C = z.shape
    sums = torch.empty((2, C), dtype=torch.float64, device=z.device)
    ws = _ws(_lib.lib().ptt_bn_stats_workspace(R, C), z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_pooled_sums_f64(_ptr(dpooled), dpooled.stride(0), _ptr(arg), int(ns), _ptr(z), z.stride(0),
                                                         _ptr(mean), _ptr(invstd), R, C, _ptr(sums), _ptr(ws), ws.numel() * 8,
                                                         _ptr(act_scale), _ptr(act_shift), _stream()), "ptt_bn_bwd_pooled_sums_f64")
    return sums


def bn_bwd_pooled_apply(dpooled, arg, ns, z, mean, invstd, gamma, sum_dy, sum_dy_xhat, count, act_scale, act_shift, out=None):
    """dz from the (global) sums, pooled gradient — ptt_bn_bwd_pooled_apply_f32."""
    _rows(dpooled, "dpooled"); _rows(z, "z")
    R, C = z.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_pooled_apply_f32(_ptr(dpooled), dpooled.stride(0), _ptr(arg), int(ns), _ptr(z), z.stride(0),
                                                          _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sum_dy), _ptr(sum_dy_xhat),
                                                          _ptr(count), R, C, _ptr(out), out.stride(0), _ptr(act_scale),
                                                          _ptr(act_shift), _stream()), "ptt_bn_bwd_pooled_apply_f32")
    return out


def pool_rows(x, ns, act_scale=None, act_shift=None):
    """max over every ns consecutive rows: (G*ns, C) -> (G, C) and the int32 arg-max (first among equals); with
    act_scale / act_shift the rows are relu(x * scale + shift), applied on the fly."""
    _rows(x, "x")
    R, C = x.shape
    G = R // int(ns)
    out = torch.empty((G, C), dtype=torch.float32, device=x.device)
    arg = torch.empty((G, C), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_pool_rows_f32(_ptr(x), x.stride(0), G, int(ns), C, _ptr(out), C, _ptr(arg), _ptr(act_scale),
                                                _ptr(act_shift), _stream()), "ptt_pool_rows_f32")
    return out, arg


def pool_rows_bwd(dout, arg, ns):
    _rows(dout, "dout")
    G, C = dout.shape
    dx = torch.empty((G * int(ns), C), dtype=torch.float32, device=dout.device)
    with torch.cuda.device(dout.device):
        _lib.check(_lib.lib().ptt_pool_rows_bwd_f32(_ptr(dout), dout.stride(0), _ptr(arg), G, int(ns), C, _ptr(dx), C, _stream()),
                   "ptt_pool_rows_bwd_f32")
    return dx


def linear_act_in(x, in_scale, in_shift, wpacked, cout):
    """relu(x * in_scale + in_shift) @ W^T with the transform applied while x is staged — ptt_linear_act_in_f32."""
    _rows(x, "x")
    rows, K = x.shape
    out = torch.empty((rows, int(cout)), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed('ptt_linear_f32'):
        _lib.check(_lib.lib().ptt_linear_act_in_f32(_ptr(x), rows, K, x.stride(0), _ptr(in_scale), _ptr(in_shift), _ptr(wpacked),
                                                    int(cout), _ptr(out), int(cout), _stream()), "ptt_linear_act_in_f32")
    return out


def rows_gemm_supported(rows, K, N, ldx=None, ldo=None, x=None):
    """True when ptt_rows_gemm_f32 takes the shape (K and N multiples of 64, ...): the persistent row GEMM of the training step.
    `x`: the input tensor, when the caller has it — the entry point also wants 16-byte aligned rows, and a caller that asks here
    first falls back to the linear kernel instead of meeting PTT_EINVAL."""
    if x is not None and x.data_ptr() % 16:
        return False
    return bool(_lib.lib().ptt_rows_gemm_supported(int(rows), int(K), int(N), int(ldx if ldx is not None else K),
                                                   int(ldo if ldo is not None else N)))


def rows_gemm(x, wpacked, cout, in_scale=None, in_shift=None, bias=None, relu=False, residual=None, want_stats=False, out=None):
    """out = relu?(act_in(x) @ W^T + bias) (+ residual) over (rows, K) activations — ptt_rows_gemm_f32, the persistent
    software-pipelined row GEMM of the training step. act_in(x) = relu(x * in_scale + in_shift) when given.
    want_stats: also return the float64 partial column sums / sums of squares of `out`, (chunks, 2, cout), for
    bn_finish_partials / bn_sums_partials (the BatchNorm statistics without a second pass over the output)."""
    _rows(x, "x")
    rows, K = x.shape
    cout = int(cout)
    if out is None:
        out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        chunks = _lib.lib().ptt_rows_gemm_stat_chunks(rows, K, cout)
        stats = torch.empty((max(1, chunks), 2, cout), dtype=torch.float64, device=x.device)
    r2 = None
    if residual is not None:
        r2 = _rows(residual, "residual")
    with torch.cuda.device(x.device), _timed('ptt_rows_gemm_f32'):
        _lib.check(_lib.lib().ptt_rows_gemm_f32(_ptr(x), rows, K, x.stride(0), _ptr(in_scale), _ptr(in_shift), _ptr(wpacked), cout,
                                                _ptr(bias), 1 if relu else 0, _ptr(r2), r2.stride(0) if r2 is not None else cout,
                                                _ptr(out), out.stride(0), _ptr(stats), stats.numel() if stats is not None else 0,
                                                _stream()), "ptt_rows_gemm_f32")
    return (out, stats) if want_stats else out


def rows_gemm_pool_supported(rows, K, N, ldx, ns, x=None):
    if x is not None and x.data_ptr() % 16:
        return False
    return bool(_lib.lib().ptt_rows_gemm_pool_supported(int(rows), int(K), int(N), int(ldx), int(ns)))


def rows_gemm_pool(x, wpacked, cout, in_scale, in_shift, ns):
    """ptt_rows_gemm_pool_f32: z = relu(x * in_scale + in_shift) @ W^T with the float64 statistics partials of z AND, per group of
    `ns` rows and column, (max z, min z, first row of each) — the last layer of a SharedMLP + max-pool stage without a pooling
    pass over z. -> (z, stats partials, (pmax, pmin, amax, amin)); pool_select() finishes once a, b of z's BatchNorm exist."""
    _rows(x, "x")
    rows, K = x.shape
    cout, ns = int(cout), int(ns)
    G = rows // ns
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    chunks = _lib.lib().ptt_rows_gemm_stat_chunks(rows, K, cout)
    stats = torch.empty((max(1, chunks), 2, cout), dtype=torch.float64, device=x.device)
    pmax = torch.empty((G, cout), dtype=torch.float32, device=x.device)
    pmin = torch.empty((G, cout), dtype=torch.float32, device=x.device)
    amax = torch.empty((G, cout), dtype=torch.int32, device=x.device)
    amin = torch.empty((G, cout), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device), _timed('ptt_rows_gemm_f32'):
        _lib.check(_lib.lib().ptt_rows_gemm_pool_f32(_ptr(x), rows, K, x.stride(0), _ptr(in_scale), _ptr(in_shift), _ptr(wpacked), cout,
                                                     _ptr(out), cout, _ptr(stats), stats.numel(), ns, _ptr(pmax), _ptr(pmin), _ptr(amax),
                                                     _ptr(amin), _stream()), "ptt_rows_gemm_pool_f32")
    return out, stats, (pmax, pmin, amax, amin)


def pool_select(extrema, act_scale, act_shift):
    """(pmax, pmin, amax, amin) of rows_gemm_pool + the BatchNorm's a, b -> (pooled relu(a z + b) (G,C), arg-max (G,C) int32)."""
    pmax, pmin, amax, amin = extrema
    G, C = pmax.shape
    out = torch.empty((G, C), dtype=torch.float32, device=pmax.device)
    arg = torch.empty((G, C), dtype=torch.int32, device=pmax.device)
    with torch.cuda.device(pmax.device):
        _lib.check(_lib.lib().ptt_pool_select_f32(_ptr(pmax), _ptr(pmin), _ptr(amax), _ptr(amin), _ptr(act_scale), _ptr(act_shift), G, C,
                                                  _ptr(out), _ptr(arg), _stream()), "ptt_pool_select_f32")
    return out, arg


def rows_gemm_masked(x, wpacked, cout, mask, want_colsum=False):
    """out = (mask > 0) ? x @ W^T : 0 — the input gradient of the layer behind a ReLU whose output is `mask`
    (ptt_rows_gemm_masked_f32). want_colsum: also the column sums of the masked result (float32 (cout,)), i.e. the bias
    gradient of the Linear in front of that ReLU, from the GEMM's own epilogue."""
    _rows(x, "x"); _rows(mask, "mask")
    rows, K = x.shape
    cout = int(cout)
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    stats = None
    if want_colsum:
        chunks = _lib.lib().ptt_rows_gemm_stat_chunks(rows, K, cout)
        stats = torch.empty((max(1, chunks), 2, cout), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device), _timed('ptt_rows_gemm_f32'):
        _lib.check(_lib.lib().ptt_rows_gemm_masked_f32(_ptr(x), rows, K, x.stride(0), _ptr(wpacked), cout, _ptr(mask), mask.stride(0),
                                                       _ptr(out), cout, _ptr(stats), stats.numel() if stats is not None else 0,
                                                       _stream()), "ptt_rows_gemm_masked_f32")
    if want_colsum:
        return out, bn_sums_partials(stats, rows)[:cout].float()
    return out


def rows_gemm_bnbwd(x, wpacked, cout, z, mean, invstd, act_scale, act_shift):
    """g = x @ W^T (the input gradient of a layer whose input is relu(BatchNorm(z))) and, from the same launch's epilogue, the
    float64 partials (chunks, 2, cout) of that BatchNorm's backward sums (sum dy, sum dy * xhat) — ptt_rows_gemm_bnbwd_f32."""
    _rows(x, "x"); _rows(z, "z")
    rows, K = x.shape
    cout = int(cout)
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    chunks = _lib.lib().ptt_rows_gemm_stat_chunks(rows, K, cout)
    part = torch.empty((max(1, chunks), 2, cout), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device), _timed('ptt_rows_gemm_f32'):
        _lib.check(_lib.lib().ptt_rows_gemm_bnbwd_f32(_ptr(x), rows, K, x.stride(0), _ptr(wpacked), cout, _ptr(z), z.stride(0), _ptr(mean),
                                                      _ptr(invstd), _ptr(act_scale), _ptr(act_shift), _ptr(out), cout, _ptr(part),
                                                      part.numel(), _stream()), "ptt_rows_gemm_bnbwd_f32")
    return out, part


def bn_bwd_from_partials(part, g, z, mean, invstd, gamma, act_scale, act_shift, out=None):
    """(dz, dgamma, dbeta) from the partial sums rows_gemm_bnbwd took — ptt_bn_bwd_from_partials_f32; `out` may alias g."""
    R, C = z.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=z.device)
    dgamma = torch.empty((C,), dtype=torch.float32, device=z.device)
    dbeta = torch.empty((C,), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_from_partials_f32(_ptr(part), part.shape[0], _ptr(g), g.stride(0), _ptr(z), z.stride(0), _ptr(mean),
                                                           _ptr(invstd), _ptr(gamma), R, C, _ptr(out), out.stride(0), _ptr(dgamma),
                                                           _ptr(dbeta), _ptr(act_scale), _ptr(act_shift), _stream()),
                   "ptt_bn_bwd_from_partials_f32")
    return out, dgamma, dbeta


def bn_bwd_consts(part, mean, invstd, gamma, rows):
    """(dgamma, dbeta, (k1, c0, c1)) from rows_gemm_bnbwd's partials: the layer's BatchNorm + ReLU backward as three per-channel
    constants, dz = c0 + c1 (z - mean) + (mask ? k1 g : 0), for the consumer that applies it (rows_gemm_bnbwd_fused) —
    ptt_bn_bwd_consts_f32."""
    chunks, _, C = part.shape
    # dgamma / dbeta become .grad of their parameters: tensors of their own, not rows of the constants' buffer
    dgamma, dbeta = (torch.empty((C,), dtype=torch.float32, device=part.device) for _ in range(2))
    out = torch.empty((3, C), dtype=torch.float32, device=part.device)
    with torch.cuda.device(part.device):
        _lib.check(_lib.lib().ptt_bn_bwd_consts_f32(_ptr(part), chunks, _ptr(mean), _ptr(invstd), _ptr(gamma), int(rows), C, _ptr(dgamma),
                                                    _ptr(dbeta), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream()), "ptt_bn_bwd_consts_f32")
    return dgamma, dbeta, (out[0], out[1], out[2])


def bn_bwd_pooled_consts(dpooled, arg, ns, z, mean, invstd, gamma, act_scale, act_shift):
    """The same for the last layer of a SharedMLP + max-pool stage, the gradient still pooled — ptt_bn_bwd_pooled_consts_f32."""
    _rows(dpooled, "dpooled"); _rows(z, "z")
    R, C = z.shape
    dgamma, dbeta = (torch.empty((C,), dtype=torch.float32, device=z.device) for _ in range(2))
    out = torch.empty((3, C), dtype=torch.float32, device=z.device)
    ws = _ws(_lib.lib().ptt_bn_stats_workspace(R, C), z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().ptt_bn_bwd_pooled_consts_f32(_ptr(dpooled), dpooled.stride(0), _ptr(arg), int(ns), _ptr(z), z.stride(0), _ptr(mean),
                                                           _ptr(invstd), _ptr(gamma), R, C, _ptr(dgamma), _ptr(dbeta), _ptr(out[0]), _ptr(out[1]),
                                                           _ptr(out[2]), _ptr(ws), ws.numel() * 8, _ptr(act_scale), _ptr(act_shift), _stream()),
                   "ptt_bn_bwd_pooled_consts_f32")
    return dgamma, dbeta, (out[0], out[1], out[2])


def rows_gemm_bnbwd_fused_supported(rows, K, cout, ns, *tensors):
    """ptt_rows_gemm_bnbwd_fused_f32 takes the shape and the layouts: contiguous 16-byte aligned rows (row stride == K)."""
    if not _lib.lib().ptt_rows_gemm_bnbwd_fused_supported(int(rows), int(K), int(cout), int(ns)):
        return False
    return all(t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
               for t in tensors if t is not None)


def rows_gemm_bnbwd_fused(g, arg, ns, z, consts, mean, act_a, act_b, wpacked, cout, zp, mp, ip, ap, bp, want_dz=True):
    """The input gradient of a layer whose own BatchNorm + ReLU backward is formed while the rows are staged: g_below = dz @ W^T with
    dz = c0 + c1 (z - mean) + (mask ? k1 g : 0), g dense (rows, K) (arg None, ns 0) or pooled (rows / ns, K) with the arg-max rows;
    the backward sums of the layer BELOW (zp, mp, ip, ap, bp as rows_gemm_bnbwd) out of the epilogue; want_dz: dz written out once
    for the layer's weight gradient. -> (g_below, partials, dz | None) — ptt_rows_gemm_bnbwd_fused_f32."""
    rows, K = z.shape
    cout = int(cout)
    dev = z.device
    out = torch.empty((rows, cout), dtype=torch.float32, device=dev)
    dz = torch.empty((rows, K), dtype=torch.float32, device=dev) if want_dz else None
    chunks = _lib.lib().ptt_rows_gemm_stat_chunks(rows, K, cout)
    part = torch.empty((max(1, chunks), 2, cout), dtype=torch.float64, device=dev)
    d = _lib.BnBwdInput(g=g.data_ptr(), ldg=g.stride(0), arg=arg.data_ptr() if arg is not None else None, ns=int(ns), z=z.data_ptr(),
                        ldz=z.stride(0), k1=consts[0].data_ptr(), c0=consts[1].data_ptr(), c1=consts[2].data_ptr(), mean=mean.data_ptr(),
                        act_a=act_a.data_ptr(), act_b=act_b.data_ptr(), dz_out=dz.data_ptr() if dz is not None else None,
                        ldd=dz.stride(0) if dz is not None else 0)
    with torch.cuda.device(dev), _timed('ptt_rows_gemm_f32'):
        _lib.check(_lib.lib().ptt_rows_gemm_bnbwd_fused_f32(ctypes.byref(d), rows, K, _ptr(wpacked), cout, _ptr(zp), zp.stride(0), _ptr(mp),
                                                            _ptr(ip), _ptr(ap), _ptr(bp), _ptr(out), cout, _ptr(part), part.numel(), _stream()),
                   "ptt_rows_gemm_bnbwd_fused_f32")
    return out, part, dz


def bn_bwd_sums_from_partials(part):
    """The (2, C) float64 sums of bn_bwd_sums from rows_gemm_bnbwd's partials (SyncBatchNorm: all-reduce them, then bn_bwd_apply)."""
    chunks, _, C = part.shape
    sums = torch.empty((2, C), dtype=torch.float64, device=part.device)
    with torch.cuda.device(part.device):
        _lib.check(_lib.lib().ptt_bn_bwd_sums_partials_f64(_ptr(part), chunks, C, _ptr(sums), _stream()), "ptt_bn_bwd_sums_partials_f64")
    return sums


def bn_finish_partials(partials, rows, eps, bn=None):
    """(mean, biased var, invstd) from rows_gemm's partial sums (chunks, 2, C), combined in chunk order. bn: as bn_stats —
    (mean, var, invstd, a, b) plus the module's bookkeeping in the same launch (ptt_bn_finish_partials_train_f32)."""
    chunks, _, C = partials.shape
    mean, var, invstd = (torch.empty((C,), dtype=torch.float32, device=partials.device) for _ in range(3))
    if bn is not None:
        tail, a, b = _bn_tail(bn, C, partials.device)
        with torch.cuda.device(partials.device):
            _lib.check(_lib.lib().ptt_bn_finish_partials_train_f32(_ptr(partials), chunks, C, int(rows), float(eps), _ptr(mean),
                                                                   _ptr(var), _ptr(invstd), ctypes.byref(tail), _stream()),
                       "ptt_bn_finish_partials_train_f32")
        _bn_tail_done(bn)
        return mean, var, invstd, a, b
    with torch.cuda.device(partials.device):
        _lib.check(_lib.lib().ptt_bn_finish_partials_f32(_ptr(partials), chunks, C, int(rows), float(eps), _ptr(mean), _ptr(var),
                                                         _ptr(invstd), _stream()), "ptt_bn_finish_partials_f32")
    return mean, var, invstd


def bn_sums_partials(partials, rows):
    """rows_gemm's partial sums -> the 2C + 1 float64 vector of bn_sums (sum, sum of squares, row count) for SyncBatchNorm."""
    chunks, _, C = partials.shape
    sums = torch.empty((2 * C + 1,), dtype=torch.float64, device=partials.device)
    with torch.cuda.device(partials.device):
        _lib.check(_lib.lib().ptt_bn_sums_partials_f64(_ptr(partials), chunks, C, int(rows), _ptr(sums), _stream()),
                   "ptt_bn_sums_partials_f64")
    return sums


def colsum(x, out=None):
    """(rows, C) -> (C,) column sums in a fixed order (ptt_colsum_f32): the bias gradient of a row-wise layer (float4 loads where
    the rows allow them, scalar loads otherwise)."""
    _rows(x, "x")
    R, C = x.shape
    if R == 0:
        return x.sum(0) if out is None else out.zero_()
    if out is None:
        out = torch.empty((C,), dtype=torch.float32, device=x.device)
    ws = _ws(_lib.lib().ptt_colsum_workspace(R, C), x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_colsum_f32(_ptr(x), R, C, x.stride(0), _ptr(out), _ptr(ws), ws.numel() * 8, _stream()), "ptt_colsum_f32")
    return out


def linear_wgrad(dz, x, out=None, accumulate=False, x_scale=None, x_shift=None):
    """dW (Cout,Cin) = dz^T x over the rows, on fp32 MFMA — ptt_linear_wgrad2_f32 (up to 256 x 256 outputs per workgroup)
    where it applies, else ptt_linear_wgrad_f32; with x_scale / x_shift the rows of x are relu(x * scale + shift), applied
    while they are staged."""
    _rows(dz, "dz"); _rows(x, "x")
    R, Cout = dz.shape
    Cin = x.shape[1]
    if out is None:
        out = torch.empty((Cout, Cin), dtype=torch.float32, device=dz.device)
    # ptt_linear_wgrad2_f32's own entry checks, asked here so that a shape / layout it refuses goes to the older kernel:
    # float4 rows (stride % 4, 16-byte aligned) and 32-bit element offsets (R * stride < 2^29)
    ok2 = (dz.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dz.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
           and R * max(dz.stride(0), x.stride(0)) < (1 << 29))
    nb2 = _lib.lib().ptt_linear_wgrad2_workspace(R, Cout, Cin) if ok2 else 0
    if nb2 and WGRAD2:
        ws = _ws(nb2, dz.device)
        with torch.cuda.device(dz.device), _timed('ptt_linear_wgrad_f32'):
            _lib.check(_lib.lib().ptt_linear_wgrad2_f32(_ptr(dz), dz.stride(0), _ptr(x), x.stride(0), R, Cout, Cin, _ptr(out),
                                                        int(bool(accumulate)), _ptr(ws), ws.numel() * 8, _ptr(x_scale), _ptr(x_shift),
                                                        _stream()), "ptt_linear_wgrad2_f32")
        return out
    ws = _ws(_lib.lib().ptt_linear_wgrad_workspace(R, Cout, Cin), dz.device)
    with torch.cuda.device(dz.device), _timed('ptt_linear_wgrad_f32'):
        _lib.check(_lib.lib().ptt_linear_wgrad_f32(_ptr(dz), dz.stride(0), _ptr(x), x.stride(0), R, Cout, Cin, _ptr(out),
                                                   int(bool(accumulate)), _ptr(ws), ws.numel() * 8, _ptr(x_scale), _ptr(x_shift),
                                                   _stream()), "ptt_linear_wgrad_f32")
    return out


def linear_wgrad_partials(dz, x, x_scale=None, x_shift=None):
    """linear_wgrad without its finishing launch: -> (workspace holding the row-chunk partials [nchunks][Cout * Cin] float32,
    nchunks) — ptt_linear_wgrad2_partials_f32 / ptt_linear_wgrad_partials_f32; GradFinishPlan sums them later."""
    _rows(dz, "dz"); _rows(x, "x")
    R, Cout = dz.shape
    Cin = x.shape[1]
    ok2 = (dz.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dz.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
           and R * max(dz.stride(0), x.stride(0)) < (1 << 29))
    nb2 = _lib.lib().ptt_linear_wgrad2_workspace(R, Cout, Cin) if ok2 else 0
    nch = ctypes.c_int(0)
    two = bool(nb2 and WGRAD2)
    ws = _ws(nb2 if two else _lib.lib().ptt_linear_wgrad_workspace(R, Cout, Cin), dz.device)
    fn = _lib.lib().ptt_linear_wgrad2_partials_f32 if two else _lib.lib().ptt_linear_wgrad_partials_f32
    with torch.cuda.device(dz.device), _timed('ptt_linear_wgrad_f32'):
        _lib.check(fn(_ptr(dz), dz.stride(0), _ptr(x), x.stride(0), R, Cout, Cin, _ptr(ws), ws.numel() * 8, _ptr(x_scale), _ptr(x_shift),
                      ctypes.byref(nch), _stream()), "ptt_linear_wgrad_partials_f32")
    return ws, nch.value


def colsum_partials(x):
    """colsum without its finishing launch: -> (workspace holding [nchunks][C] float32 partial column sums, nchunks)."""
    _rows(x, "x")
    R, C = x.shape
    ws = _ws(_lib.lib().ptt_colsum_workspace(R, C), x.device)
    nch = ctypes.c_int(0)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_colsum_partials_f32(_ptr(x), R, C, x.stride(0), _ptr(ws), ws.numel() * 8, ctypes.byref(nch), _stream()),
                   "ptt_colsum_partials_f32")
    return ws, nch.value


_capture_uploads = []       # (device table, host contents) of launches recorded into a hipGraph, pending finish_capture_uploads()


def finish_capture_uploads():
    """After a stream capture ended: upload the device tables its launches read (their contents were only known while the
    capture ran, and no host-to-device copy is recorded into the graph). Whoever captures a training step calls this before
    the first replay; returns the tables (the caller keeps them alive with the graph)."""
    done = []
    while _capture_uploads:
        table, host, keep = _capture_uploads.pop(0)
        table.copy_(host)
        done.append((table, keep))
    return done


class GradFinishPlan(object):
    """ptt_grad_finish_f32 over the contributions one backward pass left behind: `jobs` = [(dst, cols, ld, n, data_ptr, nchunks)] in the
    order they were issued (dst = first element inside the flat buffer; the n elements form rows of `cols` with row stride `ld`).
    The segment / workgroup tables depend only on the jobs' shapes and are rebuilt when those change (a training step repeats
    them); the partials' addresses are uploaded every call through one pinned copy."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.signature = None
        self.capture_table = None
        self.uploaded = torch.cuda.Event()

    def prepare_capture(self):
        """Before a stream capture that will record run(): the job table of that capture (device + host side), allocated while
        allocations are still ordinary ones."""
        if self.signature is None:
            raise RuntimeError("GradFinishPlan: a step must have run eagerly before it is captured")
        self.capture_table = (torch.empty_like(self.table), torch.zeros(tuple(self.host.shape), dtype=torch.uint8))

    @staticmethod
    def _outputs_per_group(total_chunks):
        # a function of the chunk count only: the summation tree of a destination is fixed by its job list
        return 4 if total_chunks > 1024 else 8 if total_chunks > 256 else 16 if total_chunks > 64 else 32 if total_chunks > 8 else 64 if total_chunks > 2 else 256

    def _build(self, sig):
        by_dst, order = {}, []
        for k, (dst, cols, ld, n, nch) in enumerate(sig):
            key = (dst, cols, ld, n)
            if key not in by_dst:
                by_dst[key] = []
                order.append(key)
            by_dst[key].append(k)
        nseg = len(order)
        segs = (_lib.GradSegment * nseg)()
        blocks, self.job_order = [], []
        for si, key in enumerate(order):
            dst, cols, ld, n = key
            ks = by_dst[key]
            total = sum(sig[k][4] for k in ks)
            vec = int(n % 4 == 0 and cols % 4 == 0 and ld % 4 == 0 and dst % 4 == 0)
            out = self._outputs_per_group(total)
            segs[si] = _lib.GradSegment(dst=dst, n=n, cols=cols, ld=ld, job0=len(self.job_order), njobs=len(ks), out=out, vec=vec, reserved=0)
            self.job_order += ks
            units = n // 4 if vec else n
            for u in range(0, units, out):
                blocks += [si, u]
        # two segments may not touch the same elements: the launch adds into flat[] without atomics, one workgroup per slice of
        # a segment (e.g. a weight used whole in one branch and through a column slice in another: not a layout this plan takes)
        spans = sorted((dst, dst + (n // cols - 1) * ld + cols, cols, ld) for dst, cols, ld, n in order)
        for (a0, a1, ac, al), (b0, b1, bc, bl) in zip(spans, spans[1:]):
            if b0 < a1 and not (al == bl and (b0 - a0) % al >= ac and (b0 - a0) % al + bc <= al):
                raise ValueError("GradFinishPlan: two gradient destinations overlap inside the flat buffer "
                                 "(elements %d..%d and %d..%d): the same parameter was handed over both whole and as a slice" % (a0, a1, b0, b1))
        self.vec_jobs = [bool(segs[si].vec) for si, key in enumerate(order) for _ in by_dst[key]]
        self.n_blocks = len(blocks) // 2
        self.segs = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(self.device)
        self.blocks = torch.tensor(blocks, dtype=torch.int32, device=self.device)
        nj = len(self.job_order)
        self.host = torch.zeros((nj, ctypes.sizeof(_lib.GradJob)), dtype=torch.uint8).pin_memory()
        self.rows = (_lib.GradJob * nj).from_address(self.host.data_ptr())
        self.table = torch.empty_like(self.host, device=self.device)
        self.signature = sig

    def run(self, jobs, flat):
        if not jobs:
            return
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self.device or flat.data_ptr() % 16:
            raise ValueError("GradFinishPlan: a contiguous, 16-byte aligned float32 gradient buffer on %s expected" % self.device)
        total = flat.numel()
        sig = tuple((int(d), int(c), int(l), int(n), int(nch)) for d, c, l, n, _, nch in jobs)
        if sig != self.signature:
            for d, c, l, n, nch in sig:
                if n <= 0 or c <= 0 or n % c or l < c or nch <= 0 or d < 0 or d + (n // c - 1) * l + c > total:
                    raise ValueError("GradFinishPlan: job (dst %d, cols %d, ld %d, n %d, chunks %d) outside the %d-element buffer" % (d, c, l, n, nch, total))
            self._build(sig)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # a launch recorded into a hipGraph: the partial sums live at the addresses of the graph's memory pool for as long as the
            # graph does, so the table is written ONCE — into a table of the capture's own (a later eager run of this plan must not
            # overwrite it), uploaded after the capture ends (finish_capture_uploads: no copy is recorded into the graph)
            if sig != self.signature:
                raise RuntimeError("GradFinishPlan: a step must have run eagerly before it is captured (the segment tables are built then)")
            if self.capture_table is None:
                raise RuntimeError("GradFinishPlan: prepare_capture() must be called before the stream starts capturing")
            table, host = self.capture_table
            self.capture_table = None
            rows = (_lib.GradJob * len(self.job_order)).from_address(host.data_ptr())
            # the segment / workgroup tables the recorded launch reads stay alive with the capture: a later eager step of another
            # shape REPLACES this plan's tables (never rewrites them)
            _capture_uploads.append((table, host, (self.segs, self.blocks)))
        else:
            self.uploaded.synchronize()                        # the previous call's upload has left the pinned table
            rows, table = self.rows, self.table
        for r, k, vec in zip(rows, self.job_order, self.vec_jobs):
            ptr = int(jobs[k][4])
            if vec and ptr % 16:
                raise ValueError("GradFinishPlan: partial sums of a float4 segment must be 16-byte aligned")
            r.partial, r.nchunks = ptr, sig[k][4]
        if not capturing:
            self.table.copy_(self.host, non_blocking=True)
            self.uploaded.record()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ptt_grad_finish_f32(_ptr(self.segs), _ptr(table), _ptr(self.blocks), self.n_blocks, _ptr(flat), _stream()),
                       "ptt_grad_finish_f32")


# --------------------------------------------------------------------------- T-opt: dense attention as batched MFMA GEMMs
def pack_weight_strided(src, cout, k, stride_out, stride_k, batch, stride_batch):
    """`batch` (cout x k) matrices addressed by element strides inside the float32 device tensor `src` -> packed
    MFMA B-fragment buffers, (batch, ptt_packed_weight_elems(cout, k)) — ptt_pack_weight_strided_f32."""
    if not src.is_cuda or src.dtype != torch.float32:
        raise RuntimeError("src must be a float32 device tensor")
    n = _lib.lib().ptt_packed_weight_elems(int(cout), int(k))
    out = torch.empty((int(batch), n), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().ptt_pack_weight_strided_f32(_ptr(src), int(cout), int(k), int(stride_out), int(stride_k), int(batch),
                                                          int(stride_batch), _ptr(out), _stream()), "ptt_pack_weight_strided_f32")
    return out


def pack_jobs_table(jobs, device):
    """[(data_ptr, float offset in the arena, stride_out, stride_k, Cout, K)] -> the device job table of pack_weights."""
    arr = (_lib.PackJob * len(jobs))()
    for n, (ptr, off, so, sk, cout, k) in enumerate(jobs):
        arr[n] = _lib.PackJob(W=ptr, out_offset=off, stride_out=so, stride_k=sk, Cout=cout, K=k)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


def pack_weights(table, n_jobs, arena):
    """Packs the n_jobs weights of a pack_jobs_table into `arena` (float32, 1-D) in one launch — ptt_pack_weights_f32."""
    if not arena.is_cuda or arena.dtype != torch.float32 or not arena.is_contiguous():
        raise RuntimeError("arena must be a contiguous float32 device tensor")
    with torch.cuda.device(arena.device):
        _lib.check(_lib.lib().ptt_pack_weights_f32(_ptr(table), int(n_jobs), _ptr(arena), _stream()), "ptt_pack_weights_f32")
    return arena


def linear_batched(x, wpacked, cout, residual=None):
    """out[b] = x[b] @ W[b]^T (+ residual[b]) for b < batch: x (batch, rows, K) with contiguous last dim (any row /
    batch strides, e.g. a column slice of a wider buffer), wpacked (batch, packed elems) -> (batch, rows, cout)."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 or x.stride(2) != 1:
        raise RuntimeError("x must be a (batch, rows, K) float32 device tensor with contiguous channels")
    Bt, rows, K = x.shape
    out = torch.empty((Bt, rows, int(cout)), dtype=torch.float32, device=x.device)
    r = residual
    if r is not None and (r.dim() != 3 or r.stride(2) != 1):
        raise RuntimeError("residual must be (batch, rows, cout) with contiguous channels")
    with torch.cuda.device(x.device), _timed('ptt_linear_f32'):
        _lib.check(_lib.lib().ptt_linear_batched_f32(
            _ptr(x), rows, K, x.stride(1), x.stride(0), _ptr(wpacked), wpacked.stride(0), int(cout), None, None, 0,
            _ptr(r), r.stride(1) if r is not None else int(cout), r.stride(0) if r is not None else 0,
            _ptr(out), int(cout), rows * int(cout), Bt, _stream()), "ptt_linear_batched_f32")
    return out


def softmax_rows_(x, scale):
    """In-place softmax(scale * x) along the last dim of a contiguous float32 device tensor."""
    _chk(x, "x", torch.float32)
    n = x.shape[-1]
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_softmax_rows_f32(_ptr(x), x.numel() // n, n, n, float(scale), _stream()), "ptt_softmax_rows_f32")
    return x


def gather_rows(src, idx):
    """src (B,N,C) point-major rows, idx (B,E) int32 -> (B,E,C): out[b,e] = src[b, idx[b,e]] — ptt_gather_rows_f32."""
    _chk(src, "src", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 2)
    B, N, C = src.shape
    E = idx.shape[1]
    out = torch.empty((B, E, C), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().ptt_gather_rows_f32(_ptr(src), _ptr(idx), B, N, E, C, _ptr(out), _stream()), "ptt_gather_rows_f32")
    return out


def sa_z0_rows(xyz, new_xyz, idx, term, wx, radius, normalize_xyz, want_stats=False):
    """Layer 0 of a hoisted SA level per (centre, neighbour) row in one pass — ptt_sa_z0_rows_f32: xyz (B,N,3), new_xyz (B,M,3),
    idx (B,M,ns) int32, term (B,N,C) | None, wx (C,3) -> (z0 (B*M*ns, C), rel rows (B*M*ns, 3)[, stats partials (chunks,2,C) f64 —
    want_stats: the BatchNorm statistics of z0 summed by the same launch, for bn_finish_partials; None when C > 1024])."""
    _chk(xyz, "xyz", torch.float32, 3)
    _chk(new_xyz, "new_xyz", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 3)
    if not (wx.is_cuda and wx.dtype == torch.float32 and wx.dim() == 2 and wx.stride(1) == 1):
        raise RuntimeError("wx must be a float32 (C,3) device tensor with unit column stride (a column slice of the weight is fine)")
    B, N, _ = xyz.shape
    _, M, ns = idx.shape
    C = wx.shape[0]
    if term is not None:
        _chk(term, "term", torch.float32, 3)
        if tuple(term.shape) != (B, N, C):
            raise RuntimeError("term must be (B,N,C) = %s, got %s" % ((B, N, C), tuple(term.shape)))
    if wx.shape[1] != 3 or new_xyz.shape[1] != M:
        raise RuntimeError("wx must be (C,3) and new_xyz (B,M,3)")
    z0 = torch.empty((B * M * ns, C), dtype=torch.float32, device=xyz.device)
    rel = torch.empty((B * M * ns, 3), dtype=torch.float32, device=xyz.device)
    chunks = _lib.lib().ptt_sa_z0_rows_stat_chunks(B, M, ns, C) if want_stats else 0
    if chunks:
        part = torch.empty((chunks, 2, C), dtype=torch.float64, device=xyz.device)
        with torch.cuda.device(xyz.device):
            _lib.check(_lib.lib().ptt_sa_z0_rows_stats_f32(_ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(term), _ptr(wx), wx.stride(0), B, N, M, ns, C,
                                                           float(radius), int(bool(normalize_xyz)), _ptr(z0), _ptr(rel), _ptr(part), part.numel(),
                                                           _stream()), "ptt_sa_z0_rows_stats_f32")
        return z0, rel, part
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().ptt_sa_z0_rows_f32(_ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(term), _ptr(wx), wx.stride(0), B, N, M, ns, C, float(radius),
                                                 int(bool(normalize_xyz)), _ptr(z0), _ptr(rel), _stream()), "ptt_sa_z0_rows_f32")
    return (z0, rel, None) if want_stats else (z0, rel)


def sa_z0_bnbwd(part, g, z0, rel, mean, invstd, gamma, act_scale, act_shift, want_dz, dwx_partials=False):
    """Layer 0 of a hoisted SA level, backward, from the gradient g (R, C) of its ACTIVATED output and the BatchNorm-backward partial
    sums rows_gemm_bnbwd took with it: -> (dz0 (written over g) | None, d_wx (C,3), dgamma, dbeta) in one pass in row order over g and
    z0 — the apply pass and the K = 3 weight gradient folded together; dz0 is written only when want_dz (a level with point features:
    its row scatter reads it) — ptt_sa_z0_bnbwd_f32. dwx_partials: d_wx is returned as (workspace holding its [nchunks][C][3] partial
    sums, nchunks) for GradFinishPlan instead of the finished tensor."""
    _rows(g, "g"); _rows(z0, "z0")
    R, C = z0.shape
    dev = g.device
    d_wx = torch.empty((C, 3), dtype=torch.float32, device=dev) if not dwx_partials else None
    dgamma, dbeta = (torch.empty((C,), dtype=torch.float32, device=dev) for _ in range(2))
    ws = _ws(_lib.lib().ptt_sa_z0_bnbwd_workspace(R, C), dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ptt_sa_z0_bnbwd_f32(_ptr(part), part.shape[0], _ptr(g), _ptr(z0), _ptr(rel), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                                  _ptr(act_scale), _ptr(act_shift), R, C, _ptr(g) if want_dz else None, _ptr(d_wx), _ptr(dgamma),
                                                  _ptr(dbeta), _ptr(ws), ws.numel() * 8, _stream()), "ptt_sa_z0_bnbwd_f32")
    if dwx_partials:
        d_wx = (ws, _lib.lib().ptt_sa_z0_bnbwd_workspace(R, C) // (12 * C))
    return (g if want_dz else None), d_wx, dgamma, dbeta


def scatter_csr(idx, N):
    """(order (B,E), start (B,N+1)) of ptt_scatter_csr_i32 for idx (B,E) into N bins: the entries of every cloud sorted by (bin,
    entry). Reusable by every scatter_rows_det over the same indices (the k and v gathers of a TransformerBlock share its kNN)."""
    _chk(idx, "idx", torch.int32, 2)
    B, E = idx.shape
    order = torch.empty((B, E), dtype=torch.int32, device=idx.device)
    start = torch.empty((B, int(N) + 1), dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        _lib.check(_lib.lib().ptt_scatter_csr_i32(_ptr(idx), B, int(N), E, _ptr(order), _ptr(start), _stream()), "ptt_scatter_csr_i32")
    return order, start


def scatter_rows_det(g, idx, N, csr=None, minuend=None, negate=False):
    """The adjoint of gather_rows in a fixed summation order: g (B,E,C), idx (B,E) -> (B,N,C). csr: scatter_csr(idx, N) if the
    caller already has it. minuend (B,N,C) / negate: the result is minuend - (the sums), resp. their negative —
    ptt_scatter_rows_csr_sub_f32."""
    _chk(g, "g", torch.float32, 3)
    _chk(idx, "idx", torch.int32, 2)
    B, E, C = g.shape
    order, start = csr if csr is not None else scatter_csr(idx, N)
    out = torch.empty((B, int(N), C), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        if minuend is None and not negate:
            _lib.check(_lib.lib().ptt_scatter_rows_csr_f32(_ptr(g), _ptr(order), _ptr(start), B, int(N), E, C, _ptr(out), _stream()),
                       "ptt_scatter_rows_csr_f32")
        else:
            if minuend is not None:
                _chk(minuend, "minuend", torch.float32, 3)
                if tuple(minuend.shape) != (B, int(N), C):
                    raise ValueError("minuend: (B,N,C) expected")
            _lib.check(_lib.lib().ptt_scatter_rows_csr_sub_f32(_ptr(g), _ptr(order), _ptr(start), B, int(N), E, C, _ptr(minuend), _ptr(out),
                                                               _stream()), "ptt_scatter_rows_csr_sub_f32")
    return out


def rows_gemm_rsum16_supported(x, K, N):
    return bool(x.dim() == 2 and x.stride(1) == 1 and x.data_ptr() % 16 == 0
                and _lib.lib().ptt_rows_gemm_rsum16_supported(x.shape[0], int(K), int(N), x.stride(0)))


def rows_gemm_rsum16(x, wpacked, N, residual):
    """-> (plain = x @ W^T (rows, N), out = plain + residual, gsum (rows / 16, N) = the sums of plain over groups of 16 consecutive
    rows) from one launch — ptt_rows_gemm_rsum16_f32."""
    _rows(x, "x"); _rows(residual, "residual")
    rows, K = x.shape
    if tuple(residual.shape) != (rows, int(N)):
        raise ValueError("rows_gemm_rsum16: residual (rows, N) expected")
    out, plain = (torch.empty((rows, int(N)), dtype=torch.float32, device=x.device) for _ in range(2))
    gsum = torch.empty((rows // 16, int(N)), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed('ptt_linear_f32'):
        _lib.check(_lib.lib().ptt_rows_gemm_rsum16_f32(_ptr(x), rows, K, x.stride(0), _ptr(wpacked), int(N), _ptr(residual), residual.stride(0),
                                                       _ptr(out), int(N), _ptr(plain), int(N), _ptr(gsum), int(N), _stream()),
                   "ptt_rows_gemm_rsum16_f32")
    return plain, out, gsum


# --------------------------------------------------------------------------- Point-Transformer block, training mode
def pt_pair_input(q, kf, knn, pos):
    """t = q_i - kf[knn_ij] + pos_ij : (B,N,D), (B,N,D), (B,N,k) i32, (B,N,k,D) -> (B,N,k,D)."""
    B, N, D = q.shape
    k = knn.shape[2]
    t = torch.empty((B, N, k, D), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().ptt_pt_pair_input_f32(_ptr(q), _ptr(kf), _ptr(knn), _ptr(pos), B, N, k, D, _ptr(t), _stream()),
                   "ptt_pt_pair_input_f32")
    return t


def pt_pair_input_qkv(qkv, knn, pos, D):
    """t = q_i - k[knn_ij] + pos_ij with q, k taken as column slices of the stacked (B,N,3D) projection (no copies)."""
    B, N, _ = qkv.shape
    k = knn.shape[2]
    t = torch.empty((B, N, k, D), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.lib().ptt_pt_pair_input_ld_f32(qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, 3 * D, _ptr(knn), _ptr(pos), B, N, k,
                                                       D, _ptr(t), _stream()), "ptt_pt_pair_input_ld_f32")
    return t


def pt_attn_fwd_qkv(a, qkv, knn, pos, D, scale, want_attn=False):
    """attn = softmax_j(a * scale), res = sum_j attn * (v[knn] + pos), v = the third column block of the stacked projection;
    the (B,N,k,D) attention tensor is written only when want_attn. -> (res (B,N,D), attn | None)."""
    B, N, k, _ = a.shape
    res = torch.empty((B, N, D), dtype=torch.float32, device=a.device)
    attn = torch.empty_like(a) if want_attn else None
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().ptt_pt_attn_fwd_ld_f32(_ptr(a), qkv.data_ptr() + 8 * D, 3 * D, _ptr(knn), _ptr(pos), B, N, k, D, float(scale),
                                                     _ptr(attn), _ptr(res), _stream()), "ptt_pt_attn_fwd_ld_f32")
    return res, attn


def pt_attn_train_fwd(a, vf, knn, pos, scale):
    """attn = softmax_j(a * scale), res = sum_j attn * (vf[knn] + pos) -> (attn (B,N,k,D), res (B,N,D))."""
    B, N, k, D = a.shape
    attn = torch.empty_like(a)
    res = torch.empty((B, N, D), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().ptt_pt_attn_train_fwd_f32(_ptr(a), _ptr(vf), _ptr(knn), _ptr(pos), B, N, k, D, float(scale), _ptr(attn),
                                                        _ptr(res), _stream()), "ptt_pt_attn_train_fwd_f32")
    return attn, res


def pt_attn_train_bwd(attn, vf, knn, pos, dres, scale):
    """-> (da, dvp), both (B,N,k,D)."""
    B, N, k, D = attn.shape
    da, dvp = torch.empty_like(attn), torch.empty_like(attn)
    with torch.cuda.device(attn.device):
        _lib.check(_lib.lib().ptt_pt_attn_train_bwd_f32(_ptr(attn), _ptr(vf), _ptr(knn), _ptr(pos), _ptr(dres), B, N, k, D, float(scale),
                                                        _ptr(da), _ptr(dvp), _stream()), "ptt_pt_attn_train_bwd_f32")
    return da, dvp


# --------------------------------------------------------------------------- row jobs (one tracklet frame's launch chain)
def _rows2(t, name):
    """(..., C) float32 device tensor -> its (rows, C) view with unit column stride and a uniform row stride."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("%s must be a float32 device tensor" % name)
    t2 = t.reshape(-1, t.shape[-1])
    if t2.shape[1] > 1 and t2.stride(1) != 1:
        raise RuntimeError("%s must have unit column stride" % name)
    return t2


def _ld(t2):
    return int(t2.stride(0)) if t2.shape[0] > 1 else max(int(t2.shape[1]), int(t2.stride(0)))


def row_job(wpacked, cout, x=None, x2=None, xmax=None, scale=None, shift=None, act=0, res=None, res2=None, res_split=0, out=None,
            out2=None, out_split=0, out_col0=0, raw=None, rel=None, w1=None, qkv=None, knn=None, pos=None, q_off=0,
            k_off=0, v_off=0, N=0, sm_scale=1.0, prologue=0, epilogue=0, K=None, col_tiles=0, idx=None, xyz=None, centres=None,
            wx=None, radius=1.0, ns=0, M=0, normalize_xyz=False, pro_relu=False):
    """One job of ptt_row_jobs_f32 (include/ptt_hip.h: ptt_row_job) from torch tensors; `out` etc. are written in place.
    Returns (RowJob, tensors kept alive until the launch is enqueued)."""
    j = _lib.RowJob()
    keep = [wpacked, scale, shift]
    j.Wpacked, j.scale, j.shift = wpacked.data_ptr(), (scale.data_ptr() if scale is not None else None), \
        (shift.data_ptr() if shift is not None else None)
    j.Cout, j.act, j.prologue, j.epilogue, j.col_tiles = int(cout), int(act), int(prologue), int(epilogue), int(col_tiles)
    if prologue == 0:
        x_ = _rows2(x, "x")
        j.X, j.ldx, j.K1, j.rows = x_.data_ptr(), _ld(x_), x_.shape[1], x_.shape[0]
        j.K = j.K1
        if x2 is not None:
            y_ = _rows2(x2, "x2")
            if y_.shape[0] != x_.shape[0]:
                raise RuntimeError("x and x2 must have the same number of rows")
            j.X2, j.ldx2, j.K = y_.data_ptr(), _ld(y_), j.K1 + y_.shape[1]
            keep.append(y_)
        if xmax is not None:
            m_ = _rows2(xmax, "xmax")
            if m_.shape != x_.shape or _ld(m_) != _ld(x_):
                raise RuntimeError("xmax must have x's shape and row stride")
            j.Xmax = m_.data_ptr()
            keep.append(m_)
        keep.append(x_)
    elif prologue == 1:
        r_ = _rows2(rel, "rel")
        if r_.shape[1] != 3 or not r_.is_contiguous() or w1.shape != (int(K), 4) or not w1.is_contiguous():
            raise RuntimeError("prologue 1: rel (rows,3) and w1 (K,4) contiguous")
        j.rel, j.w1, j.rows, j.K, j.K1 = r_.data_ptr(), w1.data_ptr(), r_.shape[0], int(K), int(K)
        keep += [r_, w1]
    elif prologue == 2:
        j.rows, j.K, j.K1 = pos.reshape(-1, pos.shape[-1]).shape[0], int(K), int(K)
    elif prologue == 3:
        x_ = _rows2(x, "x (per-point term)")
        if (idx.dtype != torch.int32 or not idx.is_contiguous() or not xyz.is_contiguous() or not centres.is_contiguous()
                or not wx.is_contiguous() or tuple(wx.shape) != (3, x_.shape[1])):
            raise RuntimeError("prologue 3: idx int32, xyz / centres / wx (3,K) contiguous")
        j.X, j.ldx, j.K, j.K1, j.rows = x_.data_ptr(), _ld(x_), x_.shape[1], x_.shape[1], idx.numel()
        j.idx, j.xyz, j.centres, j.wx = idx.data_ptr(), xyz.data_ptr(), centres.data_ptr(), wx.data_ptr()
        j.radius, j.normalize_xyz, j.pro_relu, j.N = float(radius), int(bool(normalize_xyz)), int(bool(pro_relu)), int(N)
        keep += [x_, idx, xyz, centres, wx]
    j.ns, j.M = int(ns), int(M)
    if prologue == 2 or epilogue == 1:
        q_, p_ = _rows2(qkv, "qkv"), _rows2(pos, "pos")
        if knn.dtype != torch.int32 or not knn.is_contiguous() or knn.shape[-1] != 16:
            raise RuntimeError("knn must be a contiguous (points,16) int32 tensor")
        j.qkv, j.ldq, j.knn, j.pos, j.ldp = q_.data_ptr(), _ld(q_), knn.data_ptr(), p_.data_ptr(), _ld(p_)
        j.q_off, j.k_off, j.v_off, j.N, j.sm_scale = int(q_off), int(k_off), int(v_off), int(N), float(sm_scale)
        keep += [q_, p_, knn]
    o_ = _rows2(out, "out")
    j.out, j.ldo, j.out_split, j.out_col0 = o_.data_ptr(), _ld(o_), int(out_split), int(out_col0)
    keep.append(o_)
    if out2 is not None:
        o2 = _rows2(out2, "out2")
        j.out2, j.ldo2 = o2.data_ptr(), _ld(o2)
        keep.append(o2)
    if raw is not None:
        r2 = _rows2(raw, "raw")
        j.raw, j.ldraw = r2.data_ptr(), _ld(r2)
        keep.append(r2)
    j.res_split = int(res_split)
    if res is not None:
        r_ = _rows2(res, "res")
        j.res, j.ldr = r_.data_ptr(), _ld(r_)
        keep.append(r_)
    if res2 is not None:
        r_ = _rows2(res2, "res2")
        j.res2, j.ldr2 = r_.data_ptr(), _ld(r_)
        keep.append(r_)
    return j, keep


def row_jobs(jobs):
    """Launch up to 4 independent row jobs (from row_job) as ONE kernel on the current stream."""
    n = len(jobs)
    arr = (_lib.RowJob * n)(*[j for j, _ in jobs])
    dev = jobs[0][1][0].device
    with torch.cuda.device(dev), _timed('ptt_linear_f32'):
        _lib.check(_lib.lib().ptt_row_jobs_f32(arr, n, _stream()), "ptt_row_jobs_f32")


def sa_levels_point_jobs(xyz, inds0, npoints, radii, nsamples, knn_k=0):
    """The centre selections + ball queries of three set-abstraction levels whose levels 1 and 2 take the FIRST npoints of the
    level below ('sequence' sampling), and optionally the kNN of the last level's centres, in ONE launch (ptt_point_jobs_f32).
    xyz (B,N,3), inds0 (B,npoints[0]) int32 level-0 sample indices. Returns ([(new_xyz, idx), ...] per level, inds64 of level 0,
    (knn_idx, rel) | None) — the tensors ptt_centres_ball_query_f32 / ptt_knn_rel_f32 would produce level by level."""
    _chk(xyz, "xyz", torch.float32, 3)
    _chk(inds0, "inds0", torch.int32, 2)
    B, N, _ = xyz.shape
    dev = xyz.device
    n_jobs = len(npoints) + (1 if knn_k else 0)
    arr = (_lib.PointJob * n_jobs)()
    levels = []
    inds64 = torch.empty((B, npoints[0]), dtype=torch.int64, device=dev)
    n_pts = N
    for l, (M, r, ns) in enumerate(zip(npoints, radii, nsamples)):
        new_xyz = torch.empty((B, M, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((B, M, ns), dtype=torch.int32, device=dev)
        j = arr[l]
        j.xyz, j.centre_sel, j.point_sel = xyz.data_ptr(), inds0.data_ptr(), (inds0.data_ptr() if l > 0 else None)
        j.new_xyz, j.idx64_out, j.idx_out = new_xyz.data_ptr(), (inds64.data_ptr() if l == 0 else None), idx.data_ptr()
        j.kind, j.sel_ld, j.B, j.Nraw, j.Npts, j.M, j.nsample, j.radius = 0, inds0.shape[1], B, N, n_pts, M, ns, float(r)
        levels.append((new_xyz, idx))
        n_pts = M
    knn = None
    if knn_k:
        M = npoints[-1]
        kidx = torch.empty((B, M, knn_k), dtype=torch.int32, device=dev)
        rel = torch.empty((B, M, knn_k, 3), dtype=torch.float32, device=dev)
        j = arr[len(npoints)]
        j.xyz, j.centre_sel, j.point_sel = xyz.data_ptr(), inds0.data_ptr(), inds0.data_ptr()
        j.idx_out, j.rel_out = kidx.data_ptr(), rel.data_ptr()
        j.kind, j.sel_ld, j.B, j.Nraw, j.Npts, j.M, j.nsample, j.radius = 1, inds0.shape[1], B, N, M, M, int(knn_k), 0.0
        knn = (kidx, rel)
    with torch.cuda.device(dev), _timed('ptt_ball_query_f32'):
        _lib.check(_lib.lib().ptt_point_jobs_f32(arr, n_jobs, _stream()), "ptt_point_jobs_f32")
    return levels, inds64, knn


def fps_ball_knn(xyz, npoint, radius, nsample, k=0):
    """ptt_fps_ball_knn_f32: furthest point sampling, centre selection, ball query and (k > 0) the kNN of the centres among
    themselves in ONE launch; xyz (B,N<=256,3), npoint <= 128. -> (inds i32 (B,M), inds64, new_xyz, idx (B,M,nsample),
    (knn (B,M,k) i32, rel (B,M,k,3)) | None) — bit for bit what furthest_point_sampling + centres_ball_query + knn return."""
    _chk(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    M, dev = int(npoint), xyz.device
    inds = torch.empty((B, M), dtype=torch.int32, device=dev)
    inds64 = torch.empty((B, M), dtype=torch.int64, device=dev)
    new_xyz = torch.empty((B, M, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((B, M, int(nsample)), dtype=torch.int32, device=dev)
    knn = torch.empty((B, M, int(k)), dtype=torch.int32, device=dev) if k else None
    rel = torch.empty((B, M, int(k), 3), dtype=torch.float32, device=dev) if k else None
    with torch.cuda.device(dev), _timed('ptt_fps_f32'):
        _lib.check(_lib.lib().ptt_fps_ball_knn_f32(_ptr(xyz), B, N, M, float(radius), int(nsample), int(k), _ptr(inds), _ptr(inds64),
                                                   _ptr(new_xyz), _ptr(idx), _ptr(knn), _ptr(rel), _stream()), "ptt_fps_ball_knn_f32")
    return inds, inds64, new_xyz, idx, ((knn, rel) if k else None)


# --------------------------------------------------------------------------- the ends of the training step
def _track_loss_desc(seed_cls, cls_label, search_inds, votes, reg_label, box_data, centres, pw_seed, pw_box, weights):
    B, N = seed_cls.shape
    M = box_data.shape[1]
    for t, name, dt in ((seed_cls, "seed_cls", torch.float32), (cls_label, "cls_label", torch.float32), (votes, "votes", torch.float32),
                        (reg_label, "reg_label", torch.float32), (box_data, "box_data", torch.float32), (centres, "centres", torch.float32),
                        (pw_seed, "pos_weight_seed", torch.float32), (pw_box, "pos_weight_box", torch.float32)):
        if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
            raise ValueError("%s: contiguous %s tensor on a HIP device expected" % (name, dt))
    if search_inds is not None and (search_inds.dtype != torch.int64 or not search_inds.is_contiguous() or tuple(search_inds.shape) != (B, N)):
        raise ValueError("search_inds: contiguous (B,N) int64 expected")
    if tuple(votes.shape) != (B, N, 3) or tuple(box_data.shape) != (B, M, 5) or tuple(centres.shape) != (B, M, 3) or reg_label.shape[0] != B \
            or reg_label.dim() != 2 or reg_label.shape[1] < 4 or cls_label.shape[0] != B or (search_inds is None and cls_label.shape[1] != N):
        raise ValueError("track losses: seed_cls (B,N), votes (B,N,3), box_data (B,M,5), centres (B,M,3), reg_label (B,>=4)")
    d = _lib.TrackLossDesc()
    d.seed_cls, d.cls_label, d.search_inds, d.votes = seed_cls.data_ptr(), cls_label.data_ptr(), (search_inds.data_ptr() if search_inds is not None else None), votes.data_ptr()
    d.reg_label, d.box_data, d.centres = reg_label.data_ptr(), box_data.data_ptr(), centres.data_ptr()
    d.pos_weight_seed, d.pos_weight_box = pw_seed.data_ptr(), pw_box.data_ptr()
    d.B, d.N, d.Ns, d.M, d.ld_reg = B, N, cls_label.shape[1], M, reg_label.shape[1]
    d.w_seed_cls, d.w_seed_reg, d.w_box_cls, d.w_box_reg = (float(w) for w in weights)
    return d


def track_losses(seed_cls, cls_label, search_inds, votes, reg_label, box_data, centres, pw_seed, pw_box, weights):
    """The four tracking losses of a training step in one launch (ptt_track_losses_f32; reference centroids_voting_head.py:29-62,
    box_voting_head.py:33-66,96-104) -> (total (), out (8,): [total, seed cls, seed reg, proposal cls, proposal reg, three label sums]).
    weights = (centroids_cls_weight, centroids_reg_weight, boxes_cls_weight, boxes_reg_weight)."""
    d = _track_loss_desc(seed_cls, cls_label, search_inds, votes, reg_label, box_data, centres, pw_seed, pw_box, weights)
    out = torch.empty((8,), dtype=torch.float32, device=seed_cls.device)
    total = torch.empty((), dtype=torch.float32, device=seed_cls.device)
    with torch.cuda.device(seed_cls.device):
        _lib.check(_lib.lib().ptt_track_losses_f32(ctypes.byref(d), _ptr(out), _ptr(total), _stream()), "ptt_track_losses_f32")
    return total, out


def track_losses_bwd(out8, upstream, seed_cls, cls_label, search_inds, votes, reg_label, box_data, centres, pw_seed, pw_box, weights):
    """d total / d (seed_cls, votes, box_data) times the scalar tensor `upstream` (None: 1), one launch."""
    d = _track_loss_desc(seed_cls, cls_label, search_inds, votes, reg_label, box_data, centres, pw_seed, pw_box, weights)
    if upstream is not None and (upstream.dtype != torch.float32 or upstream.numel() != 1 or not upstream.is_cuda):
        raise ValueError("upstream: one float32 on the device expected")
    g_cls, g_votes, g_box = torch.empty_like(seed_cls), torch.empty_like(votes), torch.empty_like(box_data)
    with torch.cuda.device(seed_cls.device):
        _lib.check(_lib.lib().ptt_track_losses_bwd_f32(ctypes.byref(d), _ptr(out8), _ptr(upstream), _ptr(g_cls), _ptr(g_votes), _ptr(g_box),
                                                       _stream()), "ptt_track_losses_bwd_f32")
    return g_cls, g_votes, g_box


class AdamTable(object):
    """The device table ptt_adam_clip_step_f32 walks: parameters, their moments and the chunk map are fixed, the gradient
    pointers are refreshed every step (autograd allocates new .grad tensors after zero_grad(set_to_none=True)) through one
    pinned host copy of the table."""

    def __init__(self, params, exp_avgs, exp_avg_sqs):
        self.device = params[0].device
        chunk = _lib.lib().ptt_adam_chunk_elems()
        n = len(params)
        self.host = torch.zeros((n, ctypes.sizeof(_lib.AdamTensor)), dtype=torch.uint8).pin_memory()
        self.rows = (_lib.AdamTensor * n).from_address(self.host.data_ptr())
        which, first = [], []
        for k, (p, m, v) in enumerate(zip(params, exp_avgs, exp_avg_sqs)):
            for t in (p, m, v):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                    raise ValueError("AdamTable: contiguous float32 tensors on one device expected")
            r = self.rows[k]
            r.param, r.exp_avg, r.exp_avg_sq, r.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
            for e in range(0, p.numel(), chunk):
                which.append(k)
                first.append(e)
        self.n_chunks = len(which)
        self.which = torch.tensor(which, dtype=torch.int32, device=self.device)
        self.first = torch.tensor(first, dtype=torch.int64, device=self.device)
        self.table = torch.empty_like(self.host, device=self.device)
        self.partial = torch.empty((max(1, self.n_chunks),), dtype=torch.float64, device=self.device)
        self.norm = torch.zeros((1,), dtype=torch.float32, device=self.device)
        self.keep = (list(params), list(exp_avgs), list(exp_avg_sqs))
        self.uploaded = torch.cuda.Event()
        self.grad_ptrs = None

    def step(self, grads, beta1, beta2, eps, step_size, bias2_sqrt, weight_decay=0.0, max_norm=0.0, write_clipped=True):
        for g, p in zip(grads, self.keep[0]):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != p.numel() or g.device != self.device:
                raise ValueError("AdamTable.step: contiguous float32 gradients of the parameters' sizes expected")
        ptrs = [g.data_ptr() for g in grads]
        if ptrs != self.grad_ptrs:                             # gradients that live in one place (train_ops.GradSink's views): one upload ever
            self.uploaded.synchronize()                        # the previous step's upload has left the pinned table
            for r, ptr in zip(self.rows, ptrs):
                r.grad = ptr
            self.table.copy_(self.host, non_blocking=True)
            self.uploaded.record()
            self.grad_ptrs = ptrs
        h = self.hyper(beta1, beta2, eps, step_size, bias2_sqrt, weight_decay, max_norm, write_clipped)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ptt_adam_clip_step_f32(_ptr(self.table), _ptr(self.which), _ptr(self.first), self.n_chunks, ctypes.byref(h),
                                                         _ptr(self.partial), self.partial.numel(), _ptr(self.norm), _stream()),
                       "ptt_adam_clip_step_f32")
        return self.norm

    @staticmethod
    def hyper(beta1, beta2, eps, step_size, bias2_sqrt, weight_decay=0.0, max_norm=0.0, write_clipped=True):
        return _lib.AdamHyper(float(beta1), float(beta2), 1.0 - float(beta1), 1.0 - float(beta2), float(eps), float(step_size), float(bias2_sqrt),
                              float(weight_decay), float(max_norm), 1 if write_clipped else 0)

    def check_grads_in_place(self, grads):
        """The table as uploaded addresses exactly these gradients (True once a step() saw them and they have not moved)."""
        return self.grad_ptrs is not None and [g.data_ptr() for g in grads] == self.grad_ptrs

    def step_device_hyper(self, hyper_device, clip):
        """The two launches of step() with the hyper-parameters read from `hyper_device` (AdamHyperRing.dev) when they RUN: what
        a captured training step records. The gradients are the ones the last step() uploaded (check_grads_in_place)."""
        if self.grad_ptrs is None:
            raise RuntimeError("AdamTable.step_device_hyper: step() must have run once (the gradient addresses are uploaded then)")
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ptt_adam_clip_step_dev_f32(_ptr(self.table), _ptr(self.which), _ptr(self.first), self.n_chunks, _ptr(hyper_device),
                                                             1 if clip else 0, _ptr(self.partial), self.partial.numel(), _ptr(self.norm), _stream()),
                       "ptt_adam_clip_step_dev_f32")
        return self.norm


class AdamHyperRing(object):
    """ptt_adam_hyper of the step about to be replayed, handed to the device: a ring of pinned host slots (the host may run
    `slots` steps ahead of the device; the slot's previous upload is waited for before it is rewritten) copied into ONE device
    struct on the stream the step runs on."""

    def __init__(self, device, slots=64):
        n = ctypes.sizeof(_lib.AdamHyper)
        self.host = torch.zeros((slots, n), dtype=torch.uint8).pin_memory()
        self.dev = torch.zeros((n,), dtype=torch.uint8, device=device)
        self.events = [None] * slots
        self.k = 0

    def upload(self, h):
        s = self.k % len(self.events)
        if self.events[s] is not None:
            self.events[s].synchronize()
        ctypes.memmove(self.host[s].data_ptr(), ctypes.byref(h), ctypes.sizeof(_lib.AdamHyper))
        self.dev.copy_(self.host[s], non_blocking=True)
        ev = self.events[s] or torch.cuda.Event()
        ev.record()
        self.events[s] = ev
        self.k += 1


def unit_rows_eps(x, eps):
    """(B,C,n) float32 in any strides -> (unit (B,n,C) = x / max(|x|, eps) over the channel axis as point-major rows,
    nrm (B,n) = max(|x|, eps), negative where the clamp is active) — ptt_unit_rows_f32."""
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3:
        raise ValueError("x: (B,C,n) float32 on a HIP device expected")
    B, C, n = x.shape
    unit = torch.empty((B, n, C), dtype=torch.float32, device=x.device)
    nrm = torch.empty((B, n), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ptt_unit_rows_f32(_ptr(x), x.stride(0), x.stride(2), x.stride(1), B, n, C, float(eps), _ptr(unit), _ptr(nrm),
                                                _stream()), "ptt_unit_rows_f32")
    return unit, nrm


def cos_bwd_rows(A, unit, nrm, G, cosm, own_is_row, like):
    """The gradient of the cosine map w.r.t. one side's features, in the layout of `like` ((B,C,n), any strides):
    dx[b,:,j] = (A[b,j,:] - (sum_i G cos) unit[b,j,:]) / |nrm[b,j]|. G / cos (B,n2,n1) contiguous; own_is_row: this side indexes
    the maps' rows (the search side), else their columns (the template side); like = (size, strides) of the (B,C,n) input — ptt_cos_bwd_rows_f32."""
    B, n, C = unit.shape
    n2, n1 = G.shape[1], G.shape[2]
    for t in (A, unit, nrm, G, cosm):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("cos_bwd_rows: contiguous float32 device tensors expected")
    if n != (n2 if own_is_row else n1) or tuple(A.shape) != (B, n, C) or tuple(cosm.shape) != tuple(G.shape):
        raise ValueError("cos_bwd_rows: shapes")
    dx = torch.empty_strided(like[0], like[1], dtype=torch.float32, device=unit.device)
    own, other, m = (n1, 1, n1) if own_is_row else (1, n1, n2)
    with torch.cuda.device(unit.device):
        _lib.check(_lib.lib().ptt_cos_bwd_rows_f32(_ptr(A), _ptr(unit), _ptr(nrm), _ptr(G), _ptr(cosm), n2 * n1, own, other, m, B, n, C, _ptr(dx),
                                                   dx.stride(0), dx.stride(2), dx.stride(1), _stream()), "ptt_cos_bwd_rows_f32")
    return dx

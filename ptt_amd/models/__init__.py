"""Import surface of `ptt.models` that the reference's tools use (tools/train_tracking.py:14,
tools/demo_tracking.py:6; reference ptt/models/__init__.py:9-39): `build_network`, `load_data_to_gpu`,
`model_fn_decorator`. Signatures and the `(loss, tb_dict, disp_dict)` result are the contract; the bodies are this
package's own:

* `load_data_to_gpu` moves a batch to the HIP device the calling rank is bound to (one process per GPU:
  `torch.cuda.current_device()`), with pinned non-blocking copies, and converts only what the model consumes as
  float32 — numeric numpy arrays and floating / integer tensors. Strings, object arrays, scalars and anything
  else stay untouched; nothing is swallowed by a blanket `except`.
* `model_fn_decorator` unwraps DistributedDataParallel explicitly to advance the tracker's step counter.
"""
from typing import NamedTuple

import numpy as np
import torch

from .trackers import build_tracker


def build_network(model_cfg, num_class, dataset):
    return build_tracker(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def _to_device_f32(val, device):
    """float32 device tensor for numeric arrays / tensors, None for values that are not batch data."""
    if isinstance(val, np.ndarray):
        if val.dtype.kind not in 'fiub':            # object / string arrays (frame ids, paths) stay on the host
            return None
        val = torch.from_numpy(np.ascontiguousarray(val))
    elif not isinstance(val, torch.Tensor):
        return None
    if val.is_complex():
        return None
    if val.device.type == 'cpu' and device.type == 'cuda' and val.numel() > 0:
        val = val.pin_memory()
    return val.to(device=device, dtype=torch.float32, non_blocking=True)


def load_data_to_gpu(batch_dict, device=None):
    """In place: every numeric array / tensor of `batch_dict` becomes a float32 tensor on `device` (default: this
    process's current HIP device). Mirrors the reference's effect on the keys the tracker reads
    (search_points, template_points, cls_label, reg_label, ...)."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("load_data_to_gpu: no HIP device is visible to this process")
        device = torch.device('cuda', torch.cuda.current_device())
    for key in list(batch_dict.keys()):
        moved = _to_device_f32(batch_dict[key], device)
        if moved is not None:
            batch_dict[key] = moved
    return batch_dict


class ModelReturn(NamedTuple):
    loss: torch.Tensor
    tb_dict: dict
    disp_dict: dict


def model_fn_decorator():
    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        wrappers = (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)
        tracker = model.module if isinstance(model, wrappers) else model
        tracker.update_global_step()
        return ModelReturn(ret_dict['loss'].mean(), tb_dict, disp_dict)

    return model_func

"""ctypes/numpy front-end of oracle/ptt_oracle.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by ptt_amd/. Each function restates one reference call site (see ptt_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libptt_oracle.so")
_lib = None


def _build():
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-fopenmp",
                           os.path.join(_HERE, "ptt_oracle.c"), "-o", _SO, "-lm"])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "ptt_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            _build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_threads(n):
    """OpenMP threads of the C oracle (its libgomp is separate from torch's)."""
    lib().oracle_set_threads(int(n))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fps(xyz, npoint):
    """xyz (B,N,3) f32 -> (B,npoint) i32   [pointnet2_utils.py:78]"""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), np.int32)
    lib().oracle_fps(_p(xyz), B, N, int(npoint), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """centres first, like the extension call [pointnet2_utils.py:287] -> (B,M,nsample) i32"""
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    out = np.zeros((B, M, nsample), np.int32)
    lib().oracle_ball_query(_p(new_xyz), _p(xyz), B, M, N, ctypes.c_float(float(radius)), int(nsample), _p(out))
    return out


def gather(feat, idx):
    feat, idx = _f32(feat), _i32(idx)
    B, C, N = feat.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), np.float32)
    lib().oracle_gather(_p(feat), _p(idx), B, C, N, M, _p(out))
    return out


def gather_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M = grad_out.shape
    out = np.empty((B, C, N), np.float32)
    lib().oracle_gather_grad(_p(grad_out), _p(idx), B, C, int(N), M, _p(out))
    return out


def group(feat, idx):
    feat, idx = _f32(feat), _i32(idx)
    B, C, N = feat.shape
    _, M, ns = idx.shape
    out = np.empty((B, C, M, ns), np.float32)
    lib().oracle_group(_p(feat), _p(idx), B, C, N, M, ns, _p(out))
    return out


def group_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M, ns = grad_out.shape
    out = np.empty((B, C, N), np.float32)
    lib().oracle_group_grad(_p(grad_out), _p(idx), B, C, int(N), M, ns, _p(out))
    return out


def knn(xyz, k):
    """(B,N,3) -> (B,N,k) i32 ascending by (distance, index) [variants.py:150-151]"""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.empty((B, N, k), np.int32)
    lib().oracle_knn(_p(xyz), B, N, int(k), _p(out))
    return out

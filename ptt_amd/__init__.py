"""ptt_amd — MI355X-native (gfx950) implementation of PTT's per-frame point-feature hot path.

Layout
  csrc/      hand-written HIP kernels + the C ABI (include/ptt_hip.h)
  _lib.py    ctypes binding of libptt_hip.so (no fallback: missing library => RuntimeError)
  ops.py     operator boundary (names/semantics of the reference's pointnet2_ops._ext calls)
  models/    host-side mirror of the reference's module interface for the hot path
  synth.py   seeded synthetic tracklet frames
"""
__version__ = "0.1.0"

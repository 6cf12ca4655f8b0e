"""In-tree build of libptt_hip.so (gfx950) and of the CPU oracle used by the tests.

`hipcc` cross-compiles for gfx950 without a GPU, so this runs in the build container;
the resulting .so files are git-ignored but travel to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ptt_amd", "csrc")
LIBDIR = os.path.join(ROOT, "ptt_amd", "lib")
LIB = os.path.join(LIBDIR, "libptt_hip.so")

HIP_SOURCES = ["errors.hip", "point_ops.hip", "mfma_ops.hip", "track_ops.hip", "train_ops.hip", "gemm_ops.hip", "rowjobs.hip", "step_ops.hip", "wgrad_stream.hip"]
# FPS / ball query / kNN index parity needs un-fused fp32 arithmetic (see point_ops.hip header)
EXTRA_FLAGS = {"point_ops.hip": ["-ffp-contract=off"], "track_ops.hip": ["-ffp-contract=off"], # -fno-honor-nans: without it every fmaxf on an MFMA result costs a second v_max (sNaN canonicalisation), and
               # vector-ALU instructions next to fp32 MFMAs are paid in matrix time (a third of the epilogue instructions)
               "mfma_ops.hip": ["-fno-honor-nans"] + os.environ.get("PTT_MFMA_FLAGS", "").split(),
               "gemm_ops.hip": ["-fno-honor-nans"] + os.environ.get("PTT_GEMM_FLAGS", "").split(),
               "rowjobs.hip": ["-fno-honor-nans"],
               # accumulators in vector registers: see the file's header (one wave per SIMD otherwise)
               "wgrad_stream.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# build-time only: flags for every source, e.g. PTT_HIP_FLAGS="-DPTT_DEV" for a developer build that reads the PTT_*
# A/B switches from the environment and keeps the kernels' cycle-stamp hooks (a release build has neither)
COMMON_FLAGS = os.environ.get("PTT_HIP_FLAGS", "").split()


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    if force and os.path.isdir(LIBDIR):
        # a forced build starts from an empty directory: whatever an earlier toolchain invocation left there (--save-temps
        # bundles, objects of sources that no longer exist) would otherwise keep travelling to every GPU box
        for f in os.listdir(LIBDIR):
            path = os.path.join(LIBDIR, f)
            if os.path.isfile(path):
                os.remove(path)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "ptt_hip.h"))
    objs, running = [], []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
            cmd += EXTRA_FLAGS.get(src, []) + COMMON_FLAGS
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))        # the sources are independent: compile them side by side
    for cmd, proc in running:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    _write_build_id()
    return LIB


def _write_build_id():
    """ptt_amd/lib/BUILD_ID = the source revision the library was built from (the GPU boxes get the tree without .git;
    profile summaries record this string)."""
    try:
        rev = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.check_output(["git", "status", "--porcelain", "--", "ptt_amd/csrc", "include"], cwd=ROOT,
                                        stderr=subprocess.DEVNULL).decode().strip()
        with open(os.path.join(LIBDIR, "BUILD_ID"), "w") as fh:
            fh.write("git %s%s\n" % (rev, " + uncommitted kernel changes" if dirty else ""))
    except (OSError, subprocess.CalledProcessError):
        pass


def build_oracle(force=False, verbose=False):
    odir = os.path.join(ROOT, "oracle")
    src = os.path.join(odir, "ptt_oracle.c")
    out = os.path.join(odir, "build", "libptt_oracle.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if force or _stale(out, [src]):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-fopenmp", src, "-o", out, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force=force, verbose=True))
    if os.path.exists(os.path.join(ROOT, "oracle", "ptt_oracle.c")):
        print(build_oracle(force=force, verbose=True))

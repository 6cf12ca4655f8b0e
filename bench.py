#!/usr/bin/env python
"""bench.py — tracklet frames/sec of PTT's per-frame hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload car|ped|stress|train]

A *step* = one pass of the hot path (ptt_amd.hot_path.FrameHotPath, eval mode) over one batch of synthetic frames
already resident in HBM. Workloads (BASELINE.json `configs`):
  car    configs[1]  KITTI-Car shaped, batch 48 per GPU, 2048 search + 1024 template points   (default, the headline)
  ped    configs[2]  KITTI-Pedestrian sparse: 60 / 40 unique points resampled to 2048 / 1024, one all-zero frame in 48
  stress configs[4]  16384-pt search / 4096-pt template, 3 SA levels [8192,4096,2048] / [2048,1024,512], 32 per GPU
  train  configs[3]  nuScenes-Car shaped training step of the FULL tracker: fwd + bwd + clip + Adam, batch 48 per GPU,
                     DistributedDataParallel gradient all-reduce (19.6 MB, one bucket) over RCCL when N > 1
Frames are independent, so for car/ped/stress N GPUs = N ranks each running its own batch with NO data-path
collective (weak scaling); torch.distributed (RCCL) carries only the barrier, the max-over-ranks of the elapsed time
and a one-element all-reduce of ones (`rccl_ranks_seen`).

Launch: under `python -m torch.distributed.run ...` (WORLD_SIZE set) each process is one rank. With WORLD_SIZE unset
and --gpus N > 1 this script re-executes ITSELF under torch.distributed.run with N ranks on 127.0.0.1 (what
scripts/train_ddp.sh:9 does for the reference's tools/train_tracking.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with, besides the contract keys:
  roofline     the dominant kernel (pt_attn_pair_kernel, fp32 MFMA): algorithmic FLOPs per launch
               / mean launch duration measured live with HIP events on the launch stream, against the 157.3 TFLOP/s
               dense fp32-MFMA peak. `traffic` is null: HBM bytes need rocprofv3 --pmc passes, which this process
               cannot collect on itself (profiles/ holds them).
  cpu_baseline the CPU oracle path (oracle/: C index ops + torch-CPU dense restatement, kind "port"; the reference has
               no CPU implementation of FPS/ball-query/group) timed on this host's cores over a bounded sample.
  sustained    the same step replayed for >= --sustain seconds right after the K-step timed region (K steps are only
               tens of milliseconds; this is the number at sustained clocks).
  latency_b1   (car, N=1) per-frame latency of ONE frame (B = 1) — the reference's sequential tracking loop
               (tools/eval_utils/eval_tracking_utils.py:140-152) runs the model that way.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBS = 8000.0               # spec; ~6300 achievable

WORKLOADS = {
    "car": dict(ref="BASELINE.json configs[1]", batch=48, ns=2048, nt=1024, K_s=600, K_t=300, kind="car", zero=0,
                npoints_s=[512, 256, 128], npoints_t=[256, 128, 64],
                text="KITTI-Car shaped frames"),
    "ped": dict(ref="BASELINE.json configs[2]", batch=48, ns=2048, nt=1024, K_s=60, K_t=40, kind="ped", zero=1,
                npoints_s=[512, 256, 128], npoints_t=[256, 128, 64],
                text="KITTI-Pedestrian shaped sparse frames (60 / 40 unique points resampled with replacement, one "
                     "all-zero frame per 48)"),
    "stress": dict(ref="BASELINE.json configs[4]", batch=32, ns=16384, nt=4096, K_s=16384, K_t=4096, kind="dense", zero=0,
                   npoints_s=[8192, 4096, 2048], npoints_t=[2048, 1024, 512],
                   text="roofline stress frames (no duplicate points)"),
    "train": dict(ref="BASELINE.json configs[3]", batch=48, ns=1024, nt=512, K_s=200, K_t=100, kind="car", zero=0,
                  npoints_s=[512, 256, 128], npoints_t=[256, 128, 64],
                  text="nuScenes-Car shaped training batches (200 / 100 unique points)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="car")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: the workload's)")
    ap.add_argument("--ns", type=int, default=None, help="search points per frame (default: the workload's)")
    ap.add_argument("--nt", type=int, default=None, help="template points per frame (default: the workload's)")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of extra replays for the sustained-clock figure (0: off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--serial", action="store_true",
                    help="eager, ONE stream, no pipelining: the form to put under rocprofv3 --kernel-trace (per-kernel "
                         "durations then equal the HIP-event figures of the roofline object)")
    ap.add_argument("--no-full-model", action="store_true", help="skip the secondary full PTT.forward measurement")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 latency measurement")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the FPS of batch n+1 with the dense kernels of batch n")
    ap.add_argument("--ways", type=int, default=0,
                    help="independent batches in flight (InterleavedHotPath: that many pipelined graphs on their own "
                         "streams, replayed round-robin); 1 = a single pipelined graph; 0 = 3 for clouds up to 4096 points, "
                         "1 above (the long FPS chains of two 16384-point batches compete: measured 1147 vs 1177 frames/s)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--no-workloads", action="store_true",
                    help="default car run at N = 1: do not append the short ped / stress / train runs (`workloads` object)")
    ap.add_argument("--workloads", default="ped,stress,train", help="which short side runs the default car line carries")
    ap.add_argument("--no-train-launch", action="store_true",
                    help="N > 1 car line: do not follow it with the training launch (set by bench.py itself when it spawned the ranks: it "
                         "runs that launch from the parent)")
    ap.add_argument("--no-extras", action="store_true",
                    help="train: only the warm-up and the timed steps (no host-issue measurement, no 8-frame run): the form to put under a profiler")
    ap.add_argument("--force-collective", action="store_true",
                    help="take the multi-rank branches (RCCL process group, ranks-seen all-reduce, DistributedDataParallel, the "
                         "no_sync() exposure measurement) at ANY world size, 1 included: how a one-GPU box runs the code an 8-GPU launch runs")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


TRAIN_LAUNCH_TIMEOUT_S = int(os.environ.get("PTT_BENCH_TRAIN_TIMEOUT", "420"))     # the follow-up training launch at N > 1
LAUNCHER_ENV = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")


def run_captured(cmd, env, timeout):
    """cmd in a process group of its own -> (exit code, stdout bytes). Past `timeout` seconds the whole group is killed (a launcher and
    the ranks it started: a collective that never returns must not take the caller's own line with it) and the exit code is -9."""
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
        return proc.returncode, out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # proc.pid is the id of the group this call created
        except OSError:
            proc.kill()
        out, _ = proc.communicate()
        print("[bench] launch killed after %d s: %s" % (timeout, " ".join(cmd[-8:])), file=sys.stderr, flush=True)
        return -9, out


def launch_ranks(n, extra, capture, timeout=None):
    """`python -m torch.distributed.run --nproc-per-node n bench.py extra...` in an environment without a launcher's variables
    -> (exit code, the JSON line of its rank 0 | None)."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV and not k.startswith("TORCHELASTIC_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + extra
    print("[bench] spawn: %d ranks: bench.py %s" % (n, " ".join(extra)), file=sys.stderr, flush=True)
    if not capture:
        return subprocess.call(cmd, env=env), None
    rc, out = run_captured(cmd, env, timeout)
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
    try:
        return rc, (json.loads(lines[-1]) if lines else None)
    except ValueError:
        return rc, None


def train_launch(n):
    """The DDP step of configs[3] on n ranks of their own -> the `workloads.train` record (an `error` record if the launch fails:
    it must not take the headline line down)."""
    flags = ["--gpus", str(n), "--workload", "train", "--steps", "10", "--warmup", "5", "--sustain", "0", "--no-cpu-baseline", "--no-workloads"]
    # bounded: this is the first place a run with several RCCL ranks can hang, and the caller still has its own line to print
    rc, train = launch_ranks(n, flags + (["--force-collective"] if n == 1 else []), True, timeout=TRAIN_LAUNCH_TIMEOUT_S)
    if train is not None and rc == 0:
        return side_record(train, "train")
    return {"error": "the train launch on %d ranks exited with code %d" % (n, rc)}


def wants_train_launch(args):
    return (args.workload == "car" and not args.no_workloads and "train" in args.workloads.split(",") and args.batch is None
            and args.ns is None and args.nt is None and not args.serial and not args.no_graph)


def spawn_ranks(n):
    """WORLD_SIZE unset and --gpus n > 1: run this very command line as n ranks under torch.distributed.run — and, for the
    default car line, run the n ranks a SECOND time on `--workload train` (BASELINE.json configs[3]: the DDP step whose
    gradient all-reduce over RCCL / xGMI is the only collective of the whole path, tools/train_tracking.py:158-159,
    scripts/train_ddp.sh:9) and carry that line as `workloads.train` of the one JSON line printed: the car workload
    shards frames with no data-path collective, so without the second launch the one command a driver runs at N > 1 would
    never meet RCCL beyond a barrier."""
    argv = sys.argv[1:]
    args = parse_args()
    if not wants_train_launch(args):
        return launch_ranks(n, argv, False)[0]
    rc, line = launch_ranks(n, argv + ["--no-train-launch"], True)
    train = train_launch(n)                               # whatever became of the headline launch
    if line is not None:
        line.setdefault("workloads", {})["train"] = train
        print(json.dumps(compact_line(line)), flush=True)
    return rc


def pair_kernel_flops(B, N, k=16, D=512):
    """Algorithmic FLOPs of one pt_attn_pair_kernel launch: per (point,neighbour) row fc_delta[0] (3xD) + fc_delta[2]
    + fc_gamma[0] + fc_gamma[2] (DxD each), 2 FLOP per MAC (softmax / weighted sum not counted)."""
    return 2.0 * B * N * k * (3 * D + 3 * D * D)


def pair_alg_bytes(B, N, k=16, D=512):
    """Compulsory bytes of one pt_attn_pair_kernel launch: q|k|v rows in (3 D floats per point), neighbour indices and
    relative coordinates (k x (4 + 12) bytes per point), res out (D floats per point), the three D x D weights + small terms once."""
    return B * N * (3 * D * 4 + k * 16 + D * 4) + 3 * D * D * 4 + 6 * D * 4


def hot_path_flops_per_frame(NPS, NPT, n_seeds, executed=False):
    """Algorithmic FLOPs of one frame of the hot path as the reference executes it (3.65 GFLOP of grouped
    MLPs + 5.24 GFLOP of transformer blocks at the shipped cfg): every SharedMLP layer on every (centre, neighbour)
    row, fc1 / q,k,v / fc2 per point and the three 512 x 512 layers per (point, neighbour) pair."""
    def sa(M, ns, spec):
        return M * ns * sum(ci * co for ci, co in zip(spec[:-1], spec[1:]))
    specs = ([3, 64, 64, 128], [131, 128, 128, 256], [259, 128, 128, 256])
    mac = sum(sa(M, 32, sp) for M, sp in zip(NPS, specs)) + sum(sa(M, 32, sp) for M, sp in zip(NPT, specs))
    mac += sa(64, 16, [259, 256, 256, 256])                                    # vote aggregation
    for N in (n_seeds, 64):                                                     # the two TransformerBlocks
        mac += N * (256 * 512 + 3 * 512 * 512 + 512 * 256) + N * 16 * (3 * 512 + 3 * 512 * 512)
    if executed:
        # what the kernels execute: layer 0 of every SA level that has features (SA1, SA2 of both branches, vote aggregation) is
        # evaluated per POINT of the level below (C x Cout per point) plus the 3 relative coordinates per (centre, neighbour) row,
        # instead of (3 + C) x Cout per row
        def hoist(points, M, ns, cin, cout):
            return M * ns * cin * cout - (points * (cin - 3) * cout + M * ns * 3 * cout)
        mac -= sum(hoist(P[0], P[1], 32, specs[1][0], specs[1][1]) + hoist(P[1], P[2], 32, specs[2][0], specs[2][1]) for P in (NPS, NPT))
        mac -= hoist(n_seeds, 64, 16, 259, 256)
    return 2.0 * mac


def fps_bytes(B, N, m):
    return B * (12.0 * N + 4.0 * m)                 # compulsory bytes per cloud


def ball_query_bytes(B, N, M, ns):
    return B * (12.0 * N + 12.0 * M + 4.0 * M * ns)


def committed_traffic(kernel_prefix, tag=None):
    """HBM-side bytes per launch of a kernel from the newest committed rocprofv3 --pmc summary under profiles/ (a process
    cannot collect PMC passes on itself: scripts/pmc_passes.sh does, in separate counter-only runs, and
    scripts/pmc_summary.py applies the guide's gfx950 corrections: read bytes = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB).
    tag: None = the passes on the car shapes (B = 48); "stress" = the directory whose name ends in _stress (B = 32, 2048 seeds).
    -> (bytes per launch averaged over the kernel's launch shapes, source dict) or (None, None)."""
    import glob
    files = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "*pmc*", "pmc_summary.json"))
                   if (os.path.basename(os.path.dirname(p)).endswith("_" + tag) if tag
                       else not os.path.basename(os.path.dirname(p)).endswith(("_stress", "_stress_sampling_order", "_train_step"))))
    for path in reversed(files):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        rows = [(k, e) for k, e in d.get("kernels", {}).items() if k.startswith(kernel_prefix) and "read_bytes" in e and "write_bytes" in e]
        if not rows:
            continue
        per = {k: e["read_bytes"] + e["write_bytes"] for k, e in rows}
        src = {"file": os.path.relpath(path, ROOT), "build": d.get("build", "see the file's directory name (round / session)"),
               "per_launch_shape": {k: {"read_bytes": e["read_bytes"], "write_bytes": e["write_bytes"],
                                        "l2_hit_rate": round(e.get("l2_hit_rate", 0.0), 4)} for k, e in rows},
               "how": d.get("note", "rocprofv3 --pmc passes in separate counter-only runs") + " — committed; NOT collected by this process"}
        return sum(per.values()) / len(per), src
    return None, None


def note(msg):
    """Progress marks on stderr (PTT_BENCH_VERBOSE=1): where a long default run is."""
    if os.environ.get("PTT_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def timed_loop(step, steps, sync_all):
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    return time.perf_counter() - t0


def cpu_baseline(model, cfg, s_np, t_np, frames, ns, nt):
    """rank 0, N = 1 only: the oracle path on a bounded sample of the same workload (about 10-20 s of CPU work)."""
    import torch
    from oracle import frame_ref
    from oracle import index_ops as oracle_index_ops
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    # the cores this process may really use: the cgroup CPU quota when there is one (on the GPU boxes 256 cores are visible
    # and cpu.max grants 16 — 256 OpenMP threads on 16 cores take 28 s per frame instead of 0.05)
    ncpu, quota_note = visible, "no cgroup CPU quota"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            ncpu = max(1, min(visible, int(-(-int(q) // int(per)))))
            quota_note = "cgroup cpu.max %s/%s" % (q, per)
    except (OSError, ValueError):
        pass
    nf = max(1, min(frames, s_np.shape[0]))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sc, tc = torch.from_numpy(s_np[:nf]), torch.from_numpy(t_np[:nf])

    def run_cpu(threads, n):
        torch.set_num_threads(threads)
        oracle_index_ops.set_threads(threads)
        t1 = time.perf_counter()
        with torch.no_grad():
            frame_ref.frame(sd, cfg, sc[:n], tc[:n])
        return time.perf_counter() - t1

    # torch.set_num_threads(all visible cores), 3 warm-ups, >= 20 timed iterations, frames/s = B / median —
    # inside a budget of ~30 s of CPU work: when the visible core count over-subscribes what the container may really use an
    # iteration takes seconds, and the iteration count is cut (never below 3; the count used is reported)
    t_begin = time.perf_counter()
    w0 = run_cpu(ncpu, 1)                                   # first touch, library init
    note("cpu baseline: first frame on %d threads %.2f s" % (ncpu, w0))
    w1 = run_cpu(ncpu, 1)
    nf = int(max(1, min(nf, 1.5 / max(w1, 1e-3))))          # frames per iteration: about a second of work at most
    run_cpu(ncpu, nf)                                       # third warm-up, at the sample size
    t_iter = run_cpu(ncpu, nf)
    n_iter = int(max(3, min(20, 12.0 / max(t_iter, 1e-3))))
    times = [t_iter] + [run_cpu(ncpu, nf) for _ in range(n_iter - 1)]
    note("cpu baseline: %d iterations of %d frames, %.2f s each" % (n_iter, nf, t_iter))
    dt = float(np.median(times))
    out = {"value": round(nf / dt, 3), "unit": "frames/s", "cores": ncpu, "kind": "port",
           "sample": "%d frames (%d+%d pts) per iteration, median of %d iterations, oracle/frame_ref.py on %d threads" % (nf, ns, nt, len(times), ncpu),
           "sample_detail": "%d frames (%d+%d pts) per iteration, median of %d iterations after 3 warm-ups, through oracle/frame_ref.py "
                     "(C index ops with OpenMP + torch-CPU dense path) with torch / OpenMP threads = all %d usable cores (%d "
                     "visible, %s) (BASELINE.md section 3 protocol; 20 iterations unless they exceed a 12 s budget)"
                     % (nf, ns, nt, len(times), ncpu, visible, quota_note)}
    # second figure: the visible core count can exceed what the container may really use — the fastest thread count
    trial = {}
    for cnt in (8, 16, 32, 64, 128):
        if cnt < ncpu and time.perf_counter() - t_begin < 30.0:
            run_cpu(cnt, min(2, nf))
            trial[cnt] = min(run_cpu(cnt, nf) for _ in range(2))
    note("cpu baseline: thread-count trials %s" % {k: round(v, 2) for k, v in trial.items()})
    if trial:
        best = min(trial, key=trial.get)
        out["best_thread_count"] = {"threads": best, "value": round(nf / trial[best], 3),
                                    "note": "fastest of %s threads, best of 2 iterations" % sorted(trial)}
    return out


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    import torch
    from ptt_amd import ops, synth
    from ptt_amd.hot_path import (FrameHotPath, GraphedHotPath, InterleavedHotPath, PipelinedHotPath, TrackerThroughput,
                                  kitti_model_cfg, randomize_)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; the process group decides: %d ranks" % (args.gpus, world, world),
              file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU: each rank gets its own share of the host's cores (NUMA-local when sysfs tells), whoever launched it
    from ptt_amd import affinity
    bound = affinity.bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    dist = None
    ranks_seen = 1
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))             # only unset outside torch.distributed.run (one rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                                # every rank adds 1 over RCCL
        ranks_seen = int(one.item())

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    env = dict(torch=torch, ops=ops, synth=synth, dev=dev, dist=dist, world=world, rank=rank, ranks_seen=ranks_seen, sync_all=sync_all,
               hp=(FrameHotPath, GraphedHotPath, InterleavedHotPath, PipelinedHotPath, TrackerThroughput, kitti_model_cfg, randomize_))
    out = run_workload(args, env)
    from ptt_amd import graph_policy
    # parallel branches inside this process's captured graphs (ptt_amd/graph_policy.py): a headline taken with serialised graphs says so
    out["graph_policy"] = {"mode": graph_policy.graph_mode(), "forked_captures": graph_policy.forked_captures, "budget": graph_policy.FORK_BUDGET,
                           "serialised": graph_policy.graph_mode() == "safe" or (graph_policy.graph_mode() == "auto" and graph_policy._warned)}
    out["cpu_affinity"] = ({"cores_per_rank": len(bound["cores"]), "rank0_cores": bound["cores"][:16], "numa_node": bound["numa_node"],
                            "allowed": bound["allowed"]} if bound else {"bound": False, "allowed": len(os.sched_getaffinity(0))})
    # BASELINE.json configs[2], [4], [3] beside the headline: the default `python bench.py` line carries a short run of each
    # (value, ms_per_step, dominant-kernel roofline), so that one driver invocation observes every GPU config
    if (args.workload == "car" and world == 1 and not args.no_workloads and not args.serial and not args.no_graph
            and args.batch is None and args.ns is None and args.nt is None):
        out["workloads"] = {}
        for name, steps, warm in (("ped", 20, 5), ("stress", 8, 3), ("train", 20, 5)):
            if name not in args.workloads.split(","):
                continue
            note("workload %s" % name)
            out["workloads"][name] = side_workload(name, steps, warm)
    note("done")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # N > 1 started by a launcher (`python -m torch.distributed.run ... bench.py --gpus N`, what a driver runs): the car workload
    # shards frames with no data-path collective, so this command would never meet RCCL beyond a barrier. Once every rank is
    # through, rank 0 runs the n ranks a second time — on `--workload train`, BASELINE.json configs[3], whose gradient all-reduce
    # over RCCL / xGMI is the only collective of the whole path (tools/train_tracking.py:158-159, scripts/train_ddp.sh:9) — and
    # carries that line as workloads.train. In ranks of their own: the graphs and pools of the car run stay out of its way.
    follow = (world > 1 or os.environ.get("PTT_BENCH_TRAIN_AFTER") == "1") and "WORLD_SIZE" in os.environ and wants_train_launch(args) \
        and not args.no_train_launch
    if rank == 0:
        if follow:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out.setdefault("workloads", {})["train"] = train_launch(world)
        print("[bench detail] " + json.dumps(out), file=sys.stderr, flush=True)     # every key, with its prose
        print(json.dumps(compact_line(out)), flush=True)


# keys that hold prose (how a number was taken): DESIGN.md section 7 says it once; the stdout line carries numbers and names
PROSE_KEYS = ("how", "split", "note", "traffic_note", "per_frame", "launch", "timing", "process", "per_launch_shape", "sample_detail",
              "sharding_detail")
LINE_ORDER = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "speedup_vs_cpu_baseline", "rccl_ranks_seen", "latency_b1", "workloads",
              "sustained", "full_model", "whole_step", "index_ops", "loss", "grad_bytes_allreduced_per_step", "allreduce",
              "host_issue_ms_per_step", "small_batch", "cpu_affinity", "graph_policy", "kernel_ms_per_step")
LINE_LIMIT = 6144            # the driver keeps a tail of its child's stdout: the ONE line must fit it whole


def strip_prose(v):
    if isinstance(v, dict):
        return {k: strip_prose(x) for k, x in v.items() if k not in PROSE_KEYS and x is not None or k in ("traffic", "vs_baseline", "cpu_baseline")}
    if isinstance(v, float):
        return float("%.6g" % v)
    return v


def compact_line(out):
    """The stdout line: `out` without its prose, contract keys first, then the objects a reader looks for (roofline, cpu_baseline,
    latency_b1, workloads) and the per-kernel tables last; at most LINE_LIMIT characters (the tables go first if it is not)."""
    d = strip_prose(out)
    d = {k: d[k] for k in LINE_ORDER if k in d}
    d.update({k: v for k, v in strip_prose(out).items() if k not in d})
    for drop in ("kernel_ms_per_step", "index_ops", "whole_step", "full_model", "sustained"):
        if len(json.dumps(d)) <= LINE_LIMIT:
            break
        d.pop(drop, None)
    return d


def side_workload(name, steps, warmup):
    """One short run of another workload in its OWN process (`python bench.py --workload name ...`), reduced to the keys a
    reader needs. A fresh process per workload: the graphs, streams and memory pools of the headline run stay out of its
    way, and a failing side workload cannot take the headline line down with it. (An in-process sequence car -> ped -> stress
    crashes inside hipGraphLaunch on ROCm 7.2 when two parallel branches of a graph are given the same hardware queue — DESIGN.md
    section 6, scripts/probes/graph_queue_repro.py; GPU_MAX_HW_QUEUES=8 avoids it at the price of slower graph replays.)"""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", str(warmup),
           "--sustain", "0", "--no-cpu-baseline", "--no-full-model", "--no-latency", "--no-workloads"]
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "exit code %d: %s" % (p.returncode, p.stderr.decode()[-400:])}
        sub = json.loads(line[-1])
    except (subprocess.TimeoutExpired, ValueError, OSError) as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    return side_record(sub, name, "own process: " + " ".join(cmd[1:]), round(time.perf_counter() - t0, 2))


def side_record(sub, name, process=None, wall_s=None):
    """A side workload's line reduced to what the headline line carries of it: the contract numbers and the roofline's numbers."""
    keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "loss", "rccl_ranks_seen",
            "grad_bytes_allreduced_per_step", "allreduce")
    rec = {k: sub[k] for k in keep if sub.get(k) is not None}
    r = sub.get("roofline") or {}
    rec["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "alg_bytes", "traffic_over_alg_bytes") if k in r}
    if (sub.get("whole_step") or {}).get("frac_of_mfma_peak") is not None:
        rec["whole_step_frac_of_mfma_peak"] = sub["whole_step"]["frac_of_mfma_peak"]
        rec["whole_step_frac_of_mfma_peak_executed"] = sub["whole_step"].get("frac_of_mfma_peak_executed")
    for k in ("host_issue_ms_per_step", "small_batch", "cpu_affinity"):
        if sub.get(k) is not None:
            rec[k] = sub[k]
    rec["config"] = {"workload": sub["config"]["workload"], "name": name, "ref": WORKLOADS[name]["ref"]}
    if sub["config"].get("sharding") and name == "train":
        rec["config"]["sharding"] = sub["config"]["sharding"]
        rec["config"]["graphs_per_step"] = sub["config"].get("graphs_per_step")
    if process:
        rec["process"] = process
    if wall_s is not None:
        rec["wall_s"] = wall_s
    return rec


def run_workload(args, env):
    """One workload of WORKLOADS on this rank's device -> the result dict of the contract."""
    torch, ops, synth, dev, dist = env["torch"], env["ops"], env["synth"], env["dev"], env["dist"]
    world, rank, ranks_seen, sync_all = env["world"], env["rank"], env["ranks_seen"], env["sync_all"]
    FrameHotPath, GraphedHotPath, InterleavedHotPath, PipelinedHotPath, TrackerThroughput, kitti_model_cfg, randomize_ = env["hp"]
    W = WORKLOADS[args.workload]
    B = args.batch or W["batch"]
    NS, NT = args.ns or W["ns"], args.nt or W["nt"]
    ways = args.ways if args.ways > 0 else (3 if NS <= 4096 else 1)
    if args.workload == "train":
        return run_train(args, torch, dev, dist, world, rank, ranks_seen, sync_all, B, NS, NT, W)
    s_np, t_np = synth.frames(1000 + rank, B, NS, NT, K_s=min(W["K_s"], NS), K_t=min(W["K_t"], NT), kind=W["kind"],
                              zero_clouds=W["zero"])

    def throughput_graph(m, a, b):             # the pipelined form: one graph, or `--ways` of them round-robin
        return PipelinedHotPath(m, a, b) if ways <= 1 else InterleavedHotPath(m, a, b, ways=ways)
    args.ways_used = ways
    return run_infer(args, torch, ops, dev, dist, world, rank, ranks_seen, sync_all, B, NS, NT, W, s_np, t_np,
                     FrameHotPath, GraphedHotPath, throughput_graph, TrackerThroughput, kitti_model_cfg, randomize_)


def reduce_max(torch, dist, dev, seconds):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_infer(args, torch, ops, dev, dist, world, rank, ranks_seen, sync_all, B, NS, NT, W, s_np, t_np,
              FrameHotPath, GraphedHotPath, PipelinedHotPath, TrackerThroughput, kitti_model_cfg, randomize_):
    cfg = kitti_model_cfg()
    cfg.BACKBONE_3D.SA_CONFIG.NPOINTS_SEARCH = list(W["npoints_s"])
    cfg.BACKBONE_3D.SA_CONFIG.NPOINTS_TEMPLATE = list(W["npoints_t"])
    model = randomize_(FrameHotPath(cfg), seed=0).to(dev).eval()
    search = torch.from_numpy(s_np).to(dev)
    template = torch.from_numpy(t_np).to(dev)
    eager = args.no_graph or args.serial
    pipelined = not (args.no_pipeline or eager)
    if args.serial:
        model.overlap_branches = False

    def eager_step():
        with torch.no_grad():
            return model(search, template)

    graphed = None
    if not eager:
        graphed = (PipelinedHotPath if pipelined else GraphedHotPath)(model, search, template)
    # pipelined: replay k runs the dense stage of batch k and the sampling stage of batch k+1; K replays
    # therefore execute K full batches' worth of every kernel (the warm-up replays prime the pipeline)
    step = eager_step if graphed is None else (lambda: graphed())

    for _ in range(args.warmup):
        step()
    timed = ["ptt_pt_attn_pair_f32", "ptt_fps_f32", "ptt_ball_query_f32", "ptt_sa_fused_fwd_f32", "ptt_linear_f32",
             "ptt_knn_f32"]
    if graphed is None:
        ops.start_kernel_timing(timed)          # HIP events around each launch, inside the timed region
    elapsed = timed_loop(step, args.steps, sync_all)
    ktimes = ops.stop_kernel_timing() if graphed is None else None
    elapsed = reduce_max(torch, dist, dev, elapsed)

    # ---- the same step at sustained clocks: replay for >= --sustain seconds right after the timed region ----
    sustained = None
    if args.sustain > 0:
        n_sus = max(args.steps, int(args.sustain / max(elapsed / args.steps, 1e-6)) + 1)
        n_sus = min(n_sus, 200000)
        dt = reduce_max(torch, dist, dev, timed_loop(step, n_sus, sync_all))
        sustained = {"steps": n_sus, "seconds": round(dt, 3), "value": round(B * world * n_sus / dt, 2),
                     "ms_per_step": round(dt / n_sus * 1e3, 4)}

    if ktimes is None:
        # events cannot be recorded inside a replayed graph: the per-kernel durations come from the same kernels, same
        # inputs, launched eagerly on ONE stream right after the timed region (python bench.py --serial under
        # rocprofv3 --kernel-trace reproduces them: profiles/)
        model.overlap_branches = False
        for _ in range(2):
            eager_step()
        ops.start_kernel_timing(timed)
        for _ in range(args.steps):
            eager_step()
        ktimes = ops.stop_kernel_timing()
        model.overlap_branches = True

    frames_total = float(B) * world * args.steps
    value = frames_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (rank 0's launches) ----
    n_seeds = W["npoints_s"][-1]
    pair_ms = ktimes["ptt_pt_attn_pair_f32"]
    n_launch = len(pair_ms)
    flops_per_step = pair_kernel_flops(B, n_seeds) + pair_kernel_flops(B, 64)     # seeds, then the 64 proposals
    pair_avg_ms = sum(pair_ms) / max(n_launch, 1)
    flops_per_launch = flops_per_step / 2.0
    achieved = flops_per_launch / (pair_avg_ms * 1e-3) / 1e12 if n_launch else 0.0
    roofline = {"kernel": "pt_attn_pair_kernel<512>", "bound": "mfma", "achieved": round(achieved, 2),
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": None,
                "traffic_note": "HBM bytes need rocprofv3 --pmc passes (separate runs): none committed under profiles/",
                "avg_launch_ms": round(pair_avg_ms, 4), "launches": n_launch,
                "timing": "HIP events on the launch stream" + ("" if graphed is None else
                                                               ", eager single-stream pass of the same kernels right after the graphed timed region"),
                "alg_flops_per_launch": flops_per_launch}

    if (B == 48 and n_seeds == 128) or (B == 32 and n_seeds == 2048):   # the committed passes profile exactly these launch shapes
        tr, src = committed_traffic("pt_attn_pair_kernel<512>", "stress" if n_seeds == 2048 else None)
        if tr is not None:
            roofline["traffic"] = tr
            roofline["traffic_source"] = src
            roofline["traffic_note"] = ("bytes per launch (read + write at the fabric side of L2), mean of the seed and the proposal "
                                        "launches, from the committed PMC passes named in traffic_source")
            roofline["alg_bytes_per_launch"] = 0.5 * (pair_alg_bytes(B, n_seeds) + pair_alg_bytes(B, 64))
    # secondary: FPS + ball-query algorithmic HBM GB/s vs peak (BASELINE.json metric, second half)
    def gbs(name, nbytes_per_step):
        ms = sum(ktimes[name]) / args.steps
        return round(nbytes_per_step / (ms * 1e-3) / 1e9, 3) if ms > 0 else None, round(ms, 4)

    ps, pt = W["npoints_s"], W["npoints_t"]
    fps_b = fps_bytes(B, NS, ps[0]) + fps_bytes(B, NT, pt[0]) + fps_bytes(B, n_seeds, 64)
    bq_b = (ball_query_bytes(B, NS, ps[0], 32) + ball_query_bytes(B, ps[0], ps[1], 32) + ball_query_bytes(B, ps[1], ps[2], 32)
            + ball_query_bytes(B, NT, pt[0], 32) + ball_query_bytes(B, pt[0], pt[1], 32) + ball_query_bytes(B, pt[1], pt[2], 32)
            + ball_query_bytes(B, n_seeds, 64, 16))
    fps_gbs, fps_ms = gbs("ptt_fps_f32", fps_b)
    bq_gbs, bq_ms = gbs("ptt_ball_query_f32", bq_b)
    kernel_ms = {k.replace("ptt_", "").replace("_f32", ""): round(sum(v) / args.steps, 4) for k, v in ktimes.items()}
    # FPS is a dependent arg-max chain (latency bound), ball query an L2-resident sweep — beside the HBM
    # fraction report what they are actually limited by: FPS iterations per second per cloud, pair tests per second
    fps_its = (ps[0] - 1) + (pt[0] - 1) + 63                # chained iterations of one frame (search, template, proposals)
    bq_tests = B * (NS * ps[0] + ps[0] * ps[1] + ps[1] * ps[2] + NT * pt[0] + pt[0] * pt[1] + pt[1] * pt[2] + n_seeds * 64)
    index_ops = {"fps": {"alg_GBps": fps_gbs, "ms_per_step": fps_ms, "frac_of_hbm_peak": (fps_gbs or 0) / PEAK_HBM_GBS,
                         "iterations_per_s_per_cloud": round(fps_its / (fps_ms * 1e-3), 0) if fps_ms else None,
                         "note": "all clouds of a launch advance together, one workgroup each; the search, template and "
                                 "proposal chains are serialised in this eager measurement"},
                 "ball_query": {"alg_GBps": bq_gbs, "ms_per_step": bq_ms, "frac_of_hbm_peak": (bq_gbs or 0) / PEAK_HBM_GBS,
                                "max_pair_tests_per_s": round(bq_tests / (bq_ms * 1e-3), 0) if bq_ms else None,
                                "note": "upper bound: a centre's sweep stops at nsample hits"}}
    # the whole step against the matrix roof: reference-algorithm FLOPs of B frames / step time. The hoisted layer 0 of
    # SA1 / SA2 / vote aggregation executes fewer FLOPs than the reference's per-row form, so this is an effective rate
    step_flops = B * hot_path_flops_per_frame(ps, pt, n_seeds)
    exec_flops = B * hot_path_flops_per_frame(ps, pt, n_seeds, executed=True)
    whole_step = {"alg_gflop_per_frame": round(step_flops / B / 1e9, 3),
                  "achieved_tflops": round(step_flops * world / (elapsed / args.steps) / 1e12 / world, 2),
                  "frac_of_mfma_peak": round(step_flops / (elapsed / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                  "executed_gflop_per_frame": round(exec_flops / B / 1e9, 3),
                  "frac_of_mfma_peak_executed": round(exec_flops / (elapsed / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                  "note": "per GPU, index ops and launch gaps included. frac_of_mfma_peak divides the REFERENCE algorithm's FLOPs "
                          "(SURVEY.md 8d) by the step time: an effective rate, not a hardware fraction; frac_of_mfma_peak_executed "
                          "divides what the kernels execute (layer 0 of SA1 / SA2 / vote aggregation evaluated per point): "
                          "what the hardware did"}

    solo = rank == 0 and world == 1
    note("%s: headline done (%.3f ms per step)" % (args.workload, ms_per_step))
    # ---- secondary line: the FULL tracker forward (hot path + CosineSimAug + both heads), same batch, graph replay ----
    full = None
    tracker = None
    if solo and args.workload == "car" and not args.no_full_model and not eager:
        from ptt_amd.config import StubDataset, ptt_model_cfg
        from ptt_amd.models import build_network
        tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=0).to(dev).eval()
        gfull = (PipelinedHotPath if pipelined else GraphedHotPath)(TrackerThroughput(tracker), search, template)
        for _ in range(3):
            gfull()
        dtf = timed_loop(gfull, args.steps, sync_all)
        full = {"metric": "full PTT.forward frames/sec (eval; backbone + CosineSimAug + centroid and box heads)",
                "value": round(B * args.steps / dtf, 2), "ms_per_step": round(dtf / args.steps * 1e3, 4),
                "launch": "hipGraph replay, two-stream branches" + (", FPS of the next batch pipelined as in the headline" if pipelined else "")}
        del gfull

    # ---- B = 1 latency: what one frame of a sequential tracklet costs (hipGraph replay of one frame) ----
    latency = None
    if solo and args.workload == "car" and not args.no_latency and not eager:
        latency = latency_b1(torch, dev, model, tracker, search, template, GraphedHotPath, TrackerThroughput, sync_all)

    cpu = None
    note("%s: full model / latency done" % args.workload)
    if solo and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, cfg, s_np, t_np, args.cpu_frames if args.workload != "stress" else 1, NS, NT)

    out = {
        "metric": "tracklet frames/sec (hot path: PointNet++ SA stack + Point-Track-Transformer blocks)",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s (%s): batch %d per GPU, %d search + %d template points, SA centres %s / %s, eval"
                               % (args.workload, W["ref"], B, NS, NT, W["npoints_s"], W["npoints_t"]),
                   "sample_detail": "%s; 3 SA levels x 2 branches + vote-aggregation SA + 2 TransformerBlocks (d_model 512, k 16), "
                                    "random-init weights" % W["text"],
                   "name": args.workload, "frames_per_gpu_per_step": B, "search_points": NS, "template_points": NT,
                   "sharding": "frames across ranks, no collective",
                   "launch": ("eager, one stream" if args.serial else "eager" if graphed is None else
                              "hipGraph replay, template branch on a second stream" +
                              ("; software-pipelined across batches: FPS of batch n+1 runs on a side stream during the "
                               "dense kernels of batch n (every batch still executes every kernel)" +
                               ("; %d such pipelines on their own streams replayed round-robin (%d independent batches in "
                                "flight)" % (args.ways_used, args.ways_used) if args.ways_used > 1 else "") if pipelined else ""))},
        "rccl_ranks_seen": ranks_seen,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "sustained": sustained,
        "index_ops": index_ops,
        "whole_step": whole_step,
        "full_model": full,
        "latency_b1": latency,
        "kernel_ms_per_step": kernel_ms,
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    return out


def latency_b1(torch, dev, model, tracker, search, template, GraphedHotPath, TrackerThroughput, sync_all, n=300):
    """One frame at a time (B = 1): hipGraph replay latency of the hot path and of the full tracker forward."""
    s1, t1 = search[:1].contiguous(), template[:1].contiguous()
    res = {"frames": n, "search_points": int(s1.shape[1]), "template_points": int(t1.shape[1]),
           "launch": "hipGraph replay of one frame, template branch on a second stream; host waits for each frame"}

    def per_frame(g):
        for _ in range(20):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g()
            torch.cuda.synchronize()            # a tracklet needs frame i's box before it can crop frame i+1
        return (time.perf_counter() - t0) / n * 1e3

    g = GraphedHotPath(model, s1, t1)
    res["hot_path_ms_per_frame"] = round(per_frame(g), 4)
    del g
    if tracker is not None:
        g = GraphedHotPath(TrackerThroughput(tracker), s1, t1)
        ms = per_frame(g)
        res["full_tracker_ms_per_frame"] = round(ms, 4)
        res["full_tracker_frames_per_s"] = round(1e3 / ms, 1)
        del g
        res["tracklet_loop"] = tracklet_loop(torch, dev, tracker)
    return res


def tracklet_loop(torch, dev, tracker, frames_b1=120, frames_b48=30):
    """The reference's actual "tracklet frames/sec" mode (tools/eval_utils/eval_tracking_utils.py:140-152): crop around
    the previous result box, resample, infer, move the box — frame i needs frame i-1. ptt_amd.tracklet_runner keeps the
    clouds on the device (crop + resample kernels, model graph, one 5-float read-back per frame); shipped sizes
    1024 + 512 points. One tracklet alone (B = 1), and 48 tracklets advanced in lockstep."""
    from ptt_amd import synth
    from ptt_amd.tracklet_runner import TrackletRunner
    out = {"search_points": 1024, "template_points": 512,
           "per_frame": "host: float64 crop bounds + job upload; device: crop/compact, resample, tracker graph, box "
                        "selection; host: 5-float read-back, float64 box update"}
    for B, T, key in ((1, frames_b1, "b1"), (48, frames_b48, "b48")):
        tracklets = [synth.tracklet(9000 + k, T) for k in range(B)]
        runner = TrackletRunner(tracker, dev, batch=B)
        runner.run([(c[:4], b[:4]) for c, b in tracklets])          # warm-up: graph capture, allocator
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(tracklets)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = B * (T - 1)                                             # frame 0 only initialises (:96-100)
        out[key] = {"tracklets": B, "frames_tracked": n, "ms_per_step": round(dt / (T - 1) * 1e3, 4),
                    "frames_per_s": round(n / dt, 1)}
        if B == 1:
            # where a frame's time is (the reference prints the same split per frame: pre-process / model / post-process,
            # tools/eval_utils/eval_tracking_utils.py:140-152, ptt/utils/timer_utils.py:104-151): a second, instrumented pass
            # (HIP events around the frame's device work, host clocks around the host's) — its own wall time is NOT the figure above
            runner.profile = {}
            runner.run(tracklets[:1])
            torch.cuda.synchronize()
            med = lambda v: round(float(sorted(v)[len(v) // 2]), 4)
            p = runner.profile
            out[key].update({
                "host_pre_ms": med(p["host_pre_ms"]), "device_ms": med(p["device_ms"]), "host_post_ms": med(p["host_post_ms"]),
                "split": "medians over the frames of an instrumented pass: host_pre = float64 crop bounds written into the pinned "
                         "job table + the launch of the frame's hipGraph; device = that graph (crop + resample, tracker, read-back) "
                         "between two HIP events on the launch stream; host_post = arg-max of the proposal scores + float64 box "
                         "update; the host waits for the device in between, so ms_per_step ~ host_pre + device + host_post minus "
                         "the launch / execution overlap (the two event records of the instrumented pass cost ~0.1 ms of host_pre)",
                "launches_per_frame": launches_per_frame(torch, dev, tracker, runner)})
            runner.profile = None
        del runner
    return out


def launches_per_frame(torch, dev, tracker, runner):
    """Kernel launches of ONE tracklet frame = the kernels of one eager tracker forward at the runner's sizes (counted by
    torch.profiler; the same launches the runner's hipGraph replays) + crop, resample and box selection."""
    if os.environ.get("PTT_BENCH_NO_PROFILER"):             # scripts/probes/graph_sequence_probe.py: bisecting a crash
        return {"skipped": "PTT_BENCH_NO_PROFILER"}
    try:
        from torch.profiler import ProfilerActivity, profile
        s = torch.zeros((1, runner.S, 3), device=dev)
        t = torch.zeros((1, runner.T, 3), device=dev)
        s[0, :, 0] = torch.linspace(0.2, 2.0, runner.S, device=dev)          # distinct, off-origin points
        t[0, :, 1] = torch.linspace(0.2, 1.0, runner.T, device=dev)
        with torch.no_grad():
            for _ in range(2):
                runner._model(s, t)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                runner._model(s, t)
                torch.cuda.synchronize()
        n = sum(1 for e in prof.events() if e.device_type.name in ("CUDA", "PrivateUse1") and "memcpy" not in e.name.lower()
                and "memset" not in e.name.lower())
        whole = runner.few and runner.use_graph
        pre, copies = (1, 1) if whole else (2, 2 if runner.few else 3)
        return {"model_graph": n, "crop_resample": pre, "copies": copies, "total": n + pre + copies,
                "how": "torch.profiler kernel events of one eager tracker forward at 1024 + 512 points (the launches the runner's "
                       "hipGraph replays) + per frame one crop-and-resample launch (its job table read from pinned host memory) and "
                       "one read-back (proposals + resampling counts in one buffer; the host takes the arg-max of the scores): the "
                       "whole frame is ONE hipGraph replay"}
    except Exception as e:                                  # the count is a diagnostic: never take the line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def train_alg_bytes(B, npoints_s=(512, 256, 128), npoints_t=(256, 128, 64), n_params=4903113):
    """A denominator for the training step's HBM traffic: the bytes a step moves if every dense layer's output (the saved
    activation) is written once and read once in the forward pass, read once more in the backward pass, and its gradient is
    written once and read once (5 x 4 bytes per output element of every layer of SURVEY.md 8a's layer list + N1 + the heads),
    plus the parameters: read by the forward and by the backward pass, gradient written, Adam's read of p, g, m, v and write of
    p, m, v (10 x 4 bytes per parameter). Index tensors and coordinates are negligible beside it. Not a lower bound (a kernel that
    fuses two layers moves less) — the figure `traffic` is to be read against."""
    def sa(M, ns, couts, points=0):
        return M * ns * sum(couts) + points * couts[0] + M * couts[-1]          # rows of the three layers, hoisted point term, pooled
    def block(N, D=512, k=16):                                                  # TransformerBlock: per point + per (point, neighbour)
        return N * (D + 3 * D + D + 256) + N * k * 6 * D                        # fc1, q|k|v, aggregate, fc2; delta0, delta, t, gamma0, gamma, attn
    e = 0
    for P, first in ((npoints_s, 0), (npoints_t, 0)):
        e += sa(P[0], 32, (64, 64, 128)) + sa(P[1], 32, (128, 128, 256), P[0]) + sa(P[2], 32, (128, 128, 256), P[1]) + P[2] * 256
    ns_, nt_ = npoints_s[2], npoints_t[2]
    e += ns_ * nt_ * (1 + 3 * 256) + ns_ * 256 * 3                              # CosineSimAug: cosine map, SharedMLP over the map, pool + two Conv1d
    e += block(ns_) + ns_ * (256 + 256 + 1 + 256 + 256 + 259)                   # centroid head
    e += sa(64, 16, (256, 256, 256), ns_) + block(64) + 64 * (256 + 256 + 5)    # box head
    return B * e * 4 * 5 + n_params * 4 * 10


def train_roofline(achieved, flops, B, NS, NT):
    """The training step has no single dominant kernel: the roofline object is the whole step against the fp32-MFMA peak;
    `traffic` = HBM-side bytes of one step (every launch) from the committed PMC passes of scripts/pmc_train_step.sh, taken at the
    shipped shape (48 frames of 1024 + 512 points)."""
    r = {"kernel": "whole training step (no single dominant kernel)", "bound": "mfma", "achieved": round(achieved, 2),
         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
         "alg_flops_per_step": flops, "timing": "wall clock of the timed steps (3 x 12.3 GFLOP per frame dense fp32)"}
    r["alg_bytes"] = train_alg_bytes(B)
    if (B, NS, NT) == (48, 1024, 512):
        tr, src = committed_traffic("whole training step", "train_step")
        if tr is not None:
            r["traffic"], r["traffic_source"] = tr, src
            r["traffic_over_alg_bytes"] = round(tr / r["alg_bytes"], 3)
    return r


def run_train(args, torch, dev, dist, world, rank, ranks_seen, sync_all, B, NS, NT, W):
    """configs[3]: forward + backward + clip + Adam of the full tracker, one gradient all-reduce per step when world > 1."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from ptt_amd.train_step import GRAD_ELEMS, DataParallelTrainer, synthetic_train_batch
    torch.manual_seed(1)                                    # tools/train_tracking.py:73-79
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    collective = dist is not None                          # world > 1, or --force-collective at world 1
    trainer = DataParallelTrainer(model, dev, force_ddp=collective, reducer=os.environ.get("PTT_TRAIN_REDUCER") or None)   # dev A/B: flat | ddp
    batch = synthetic_train_batch(100 + rank, B, dev, NS, NT, K_s=W["K_s"], K_t=W["K_t"])
    last = {}

    def step():
        last["loss"] = trainer.step(batch)

    # the trainer captures the step after its first eager steps: the warm-up must reach the first REPLAY, so that the timed
    # region holds replays only (the line reports the warm-up actually run)
    args.warmup = max(args.warmup, trainer.graph_warmup + 2) if trainer.graph_mode else args.warmup
    for _ in range(args.warmup):
        step()
    elapsed = reduce_max(torch, dist, dev, timed_loop(step, args.steps, sync_all))
    sustained = None
    if args.sustain > 0:
        n_sus = max(args.steps, int(args.sustain / max(elapsed / args.steps, 1e-6)) + 1)
        dt = reduce_max(torch, dist, dev, timed_loop(step, n_sus, sync_all))
        sustained = {"steps": n_sus, "seconds": round(dt, 3), "value": round(B * world * n_sus / dt, 2),
                     "ms_per_step": round(dt / n_sus * 1e3, 4)}
    value = B * world * args.steps / elapsed
    graphed = trainer.captured is not None

    def host_issue(fn, n=20):
        """Host time to QUEUE a step (n steps issued back to back, no synchronisation in between), per step."""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        return round(dt / n * 1e3, 4)
    issue_ms = None if args.no_extras else host_issue(step)
    small = None
    if world == 1 and not collective and B > 8 and (NS, NT) == (W["ns"], W["nt"]) and not args.no_extras:
        # 8 frames per GPU: the step the HOST used to bound (12.4 ms of Python / autograd per step whatever the batch)
        torch.manual_seed(1)
        m8 = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
        t8 = DataParallelTrainer(m8, dev)
        b8 = synthetic_train_batch(100, 8, dev, NS, NT, K_s=W["K_s"], K_t=W["K_t"])
        for _ in range(t8.graph_warmup + 2):
            t8.step(b8)
        dt8 = timed_loop(lambda: t8.step(b8), 50, sync_all)
        small = {"frames_per_step": 8, "ms_per_step": round(dt8 / 50 * 1e3, 4), "value": round(8 * 50 / dt8, 2),
                 "host_issue_ms_per_step": host_issue(lambda: t8.step(b8)), "graphs_per_step": 1 if t8.captured is not None else 0}
        del t8, m8
    allreduce = None
    if collective:
        # what the gradient all-reduce costs the step: the same steps with DDP's synchronisation switched off
        # (model.no_sync(): gradients stay local; the replicas drift apart, which is why this runs last)
        def step_local():
            with trainer.no_sync():
                last["loss"] = trainer.step(batch)
        for _ in range(2):
            step_local()
        local = reduce_max(torch, dist, dev, timed_loop(step_local, args.steps, sync_all))
        allreduce = {"ms_per_step_without_allreduce": round(local / args.steps * 1e3, 4),
                     "exposed_ms_per_step": round((elapsed - local) / args.steps * 1e3, 4),
                     "how": "step time minus the same steps with the gradient all-reduce switched off (trainer.no_sync()): the "
                            "one %.1f MB all-reduce over the flat gradient buffer runs between the backward pass and the optimiser" % (GRAD_ELEMS * 4 / 1e6)}
    # dense FLOPs of one training step: forward 12.3 GFLOP per frame at 1024+512 (SURVEY.md §8a totals) x 3 (the
    # backward of a linear layer is two GEMMs of the forward's size)
    flops = 3.0 * 12.3e9 * B
    achieved = flops / (elapsed / args.steps) / 1e12
    return {
        "metric": "training frames/sec (full tracker: forward + backward + clip_grad_norm + Adam)",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "train (%s): batch %d per GPU, %d search + %d template points, full tracker, fwd + bwd + clip + Adam"
                               % (W["ref"], B, NS, NT),
                   "sample_detail": "%s; train mode (batch-statistics BatchNorm), Adam lr 1e-3 betas .5/.999 eps 1e-6, clip 10" % W["text"],
                   "name": "train", "frames_per_gpu_per_step": B, "search_points": NS, "template_points": NT,
                   "sharding": "batch across ranks, one gradient all-reduce per step over RCCL (%s)" % trainer.reducer if collective else "single rank, no collective",
                   "launch": ("hipGraph replay: forward + backward + gradient finish, %s" % (
                       "the all-reduce issued between it and a second graph (clip + Adam)" if collective else "clip + Adam in the same graph")
                       if graphed else "eager"),
                   "graphs_per_step": (2 if trainer.captured.second is not None else 1) if graphed else 0},
        "rccl_ranks_seen": ranks_seen,
        "host_issue_ms_per_step": issue_ms,
        "small_batch": small,
        "roofline": train_roofline(achieved, flops, B, NS, NT),
        "cpu_baseline": None,
        "sustained": sustained,
        "loss": float(last["loss"].detach()),
        "grad_bytes_allreduced_per_step": trainer.grad_bytes_allreduced() if collective else 0,
        "allreduce": allreduce,
    }


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Repro hunt: car -> ped -> stress hipGraph drivers in ONE process (docs/experiments.md: "was seen to crash inside hipGraphLaunch").

    python scripts/probes/graph_sequence_probe.py MODE [ORDER]
      MODE   drop   : build a workload's model + graph driver, replay, delete it (gc), next workload   (what a loop over workloads does)
             alive  : keep every model and driver alive; after building all, replay each again, round-robin
             empty  : as drop, plus torch.cuda.empty_cache() after each delete
             eager  : no graphs at all (the same kernels, eager)
             bench  : bench.run_workload itself for every workload of ORDER in this one process (the headline's pipelined graphs,
                      the eager timing pass, for car also the full-tracker graph, the B = 1 latency graphs and the tracklet loop;
                      `train` = the training step) — what `python bench.py` did in one process before its side workloads moved
                      into processes of their own
      ORDER  comma list of car,ped,stress (default car,ped,stress)
Prints one line per stage; a crash shows as the last line printed + the exit code (faulthandler dumps the Python stack)."""
import faulthandler
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                             # noqa: E402
from ptt_amd import synth                                                # noqa: E402
from ptt_amd.hot_path import FrameHotPath, InterleavedHotPath, PipelinedHotPath, kitti_model_cfg, randomize_    # noqa: E402

faulthandler.enable()


def build(name, dev, graphs=True):
    W = bench.WORKLOADS[name]
    cfg = kitti_model_cfg()
    cfg.BACKBONE_3D.SA_CONFIG.NPOINTS_SEARCH = list(W["npoints_s"])
    cfg.BACKBONE_3D.SA_CONFIG.NPOINTS_TEMPLATE = list(W["npoints_t"])
    model = randomize_(FrameHotPath(cfg), seed=0).to(dev).eval()
    s, t = synth.frames(1000, W["batch"], W["ns"], W["nt"], K_s=min(W["K_s"], W["ns"]), K_t=min(W["K_t"], W["nt"]), kind=W["kind"],
                        zero_clouds=W["zero"])
    s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
    if not graphs:
        def step():
            with torch.no_grad():
                return model(s, t)
        return model, step
    ways = 3 if W["ns"] <= 4096 else 1
    model.overlap_branches = os.environ.get("PROBE_OVERLAP", "1") != "0"     # template branch of the backbone on its own stream
    kind = os.environ.get("PROBE_DRIVER", "")                                # graphed | pipelined | (default) what bench.py uses
    if kind == "graphed":
        from ptt_amd.hot_path import GraphedHotPath
        return model, GraphedHotPath(model, s, t)
    drv = PipelinedHotPath(model, s, t) if (ways == 1 or kind == "pipelined") else InterleavedHotPath(model, s, t, ways=ways)
    return model, drv


def bench_mode(order, rounds):
    from ptt_amd import ops
    from ptt_amd.hot_path import GraphedHotPath, TrackerThroughput
    dev = torch.device("cuda:0")

    def sync_all():
        torch.cuda.synchronize()
    env = dict(torch=torch, ops=ops, synth=synth, dev=dev, dist=None, world=1, rank=0, ranks_seen=1, sync_all=sync_all,
               hp=(FrameHotPath, GraphedHotPath, InterleavedHotPath, PipelinedHotPath, TrackerThroughput, kitti_model_cfg, randomize_))
    for rnd in range(rounds):
        for name in order:
            sys.argv = ["bench.py", "--workload", name, "--steps", "5", "--warmup", "2", "--sustain", "0.3", "--no-cpu-baseline", "--no-workloads"]
            if name == "car":
                sys.argv += os.environ.get("PROBE_CAR_FLAGS", "").split()
            args = bench.parse_args()
            print("[probe] bench round %d: %s" % (rnd, name), flush=True)
            out = bench.run_workload(args, env)
            torch.cuda.synchronize()
            from ptt_amd import graph_policy
            print("[probe] bench round %d: %s ok, %.1f frames/s, reserved %.2f GB, forked captures so far %d%s" % (
                rnd, name, out["value"], torch.cuda.memory_reserved() / 2**30, graph_policy.forked_captures,
                " (serialising)" if graph_policy._warned else ""), flush=True)
            if os.environ.get("PROBE_EMPTY_CACHE"):
                del out
                gc.collect()
                torch.cuda.empty_cache()
    print("[probe] bench %s x%d: PASSED" % (",".join(order), rounds), flush=True)


KEPT = []


def tracklet_stage(dev):
    """The tracklet loop of bench.latency_b1 alone: PROBE_TL = which runners (b1: whole-frame graph with a pinned-host job table and a
    device-to-host copy node; b48: model graph + separate copies), PROBE_KEEP = 1 keeps the runners alive, PROBE_NOGRAPH = 1 runs
    them without hipGraphs, PROBE_LAT = 1 adds the B = 1 GraphedHotPath of the tracker (bench.latency_b1's other graph)."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.hot_path import GraphedHotPath, TrackerThroughput
    from ptt_amd.models import build_network
    from ptt_amd.tracklet_runner import TrackletRunner
    tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=0).to(dev).eval()
    if os.environ.get("PROBE_LAT"):
        s, t = synth.frames(1000, 1, 2048, 1024)
        g = GraphedHotPath(TrackerThroughput(tracker), torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev))
        for _ in range(20):
            g()
        torch.cuda.synchronize()
        print("[probe] tracklet stage: B = 1 tracker graph ok", flush=True)
        if os.environ.get("PROBE_KEEP"):
            KEPT.append(g)
        del g
    for key in os.environ.get("PROBE_TL", "b1,b48").split(","):
        if not key:
            continue
        B, T = (1, 40) if key == "b1" else (48, 12)
        tracklets = [synth.tracklet(9000 + k, T) for k in range(B)]
        runner = TrackletRunner(tracker, dev, batch=B, use_graph=not os.environ.get("PROBE_NOGRAPH"))
        runner.run([(c[:4], b[:4]) for c, b in tracklets])
        runner.run(tracklets)
        torch.cuda.synchronize()
        print("[probe] tracklet stage: runner %s ok" % key, flush=True)
        if os.environ.get("PROBE_KEEP"):
            KEPT.append(runner)
        del runner
    if os.environ.get("PROBE_KEEP"):
        KEPT.append(tracker)
    del tracker
    gc.collect()


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "drop"
    order = (sys.argv[2] if len(sys.argv) > 2 else "car,ped,stress").split(",")
    if mode.startswith("streams"):               # streamsN: N throw-away torch.cuda.Stream() objects first (torch hands out 32 pool
        n = int(mode[7:])                        # streams per device round-robin: creation 33 IS creation 1), then `drop`
        dev = torch.device("cuda:0")
        for k in range(n):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                torch.zeros(8, device=dev).add_(1)
            del st
        torch.cuda.synchronize()
        print("[probe] %d streams created and dropped" % n, flush=True)
        mode = "drop"
    if mode == "tracklet":                       # the tracklet loop, then the workloads of ORDER as `drop` does them
        tracklet_stage(torch.device("cuda:0"))
        mode = "drop"
    if mode == "bench":
        return bench_mode(order, int(os.environ.get("PROBE_ROUNDS", "1")))
    reps = int(os.environ.get("PROBE_REPS", "12"))
    dev = torch.device("cuda:0")
    kept = []
    for rnd in range(int(os.environ.get("PROBE_ROUNDS", "1"))):
        for name in order:
            print("[probe] %s round %d: build %s" % (mode, rnd, name), flush=True)
            model, drv = build(name, dev, graphs=mode != "eager")
            torch.cuda.synchronize()
            print("[probe] %s: replay %s x%d" % (mode, name, reps), flush=True)
            for _ in range(reps):
                out = drv()
            torch.cuda.synchronize()
            o = out[-1] if isinstance(out, list) else out
            chk = float(o["box_feats"].abs().sum()) if o is not None else float("nan")
            print("[probe] %s: %s ok, |box_feats| = %.6g, reserved %.2f GB" % (mode, name, chk, torch.cuda.memory_reserved() / 2**30), flush=True)
            if mode == "alive":
                kept.append((name, model, drv))
            else:
                del model, drv, out, o
                gc.collect()
                if mode == "empty":
                    torch.cuda.empty_cache()
        for name, model, drv in kept:
            for _ in range(reps):
                drv()
            torch.cuda.synchronize()
            print("[probe] alive: second pass %s ok" % name, flush=True)
    print("[probe] %s %s: PASSED" % (mode, ",".join(order)), flush=True)


if __name__ == "__main__":
    main()

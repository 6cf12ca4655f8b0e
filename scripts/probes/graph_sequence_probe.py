#!/usr/bin/env python
"""Repro hunt: car -> ped -> stress hipGraph drivers in ONE process ("was seen to crash inside hipGraphLaunch").

    python scripts/probes/graph_sequence_probe.py MODE [ORDER]
      MODE   drop   : build a workload's model + graph driver, replay, delete it (gc), next workload   (what a loop over workloads does)
             alive  : keep every model and driver alive; after building all, replay each again, round-robin
             empty  : as drop, plus torch.cuda.empty_cache() after each delete
             eager  : no graphs at all (the same kernels, eager)
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
    drv = PipelinedHotPath(model, s, t) if ways == 1 else InterleavedHotPath(model, s, t, ways=ways)
    return model, drv


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "drop"
    order = (sys.argv[2] if len(sys.argv) > 2 else "car,ped,stress").split(",")
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

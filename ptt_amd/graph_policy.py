"""Parallel branches inside captured hipGraphs: allowed or serialised — a process-wide policy.

Why: on ROCm 7.2 `hipGraphLaunch` segfaults when two parallel branches of ONE instantiated graph were mapped to the same hardware
queue (DESIGN.md section 5 "Forked graphs per process", docs/experiments.md; pure-PyTorch reproduction in scripts/probes/graph_queue_repro.py). With the runtime's default of 4
queues that takes a process which has instantiated a few dozen forked graphs (the car -> ped -> stress sequence in one process);
a graph WITHOUT parallel branches never crashes. The drivers of ptt_amd.hot_path and the backbone fork inside their captures for
speed (the template branch beside the search branch, the next batch's sampling beside the dense stage), so:

    mode "fast"   always fork (what every number in bench.py is taken with, in processes of a single workload)
    mode "safe"   never fork inside a capture: linear graphs, a few per cent slower, cannot hit the bug
    mode "auto"   (default) fork for the first FORK_BUDGET forked captures of the process, then serialise — with ONE RuntimeWarning
                  that names the way out. A process that exported GPU_MAX_HW_QUEUES >= 8 before the runtime started (INTEGRATION.md,
                  "Known limits") is always "fast": with 8 queues the sequence was replayed 7 x without a crash.

FORK_BUDGET = 10: what the headline car run of bench.py instantiates (three pipelined graphs, the full tracker, the B = 1 graphs and
the tracklet loops) — every number of bench.py is taken with forked graphs — while the crash was met at the 14th forked capture of a
process (car 10 + ped 3 + the stress graph; scripts/probes/graph_sequence_probe.py prints the count). What serialising costs: car /
ped ~2 %, the stress frames (16384-point clouds: the next batch's FPS no longer hides beside the dense stage) 1335 -> 800 frames/s.
PTT_GRAPH_MODE / PTT_GRAPH_FORK_BUDGET override the defaults from the environment."""
import os
import warnings

import torch

_mode = os.environ.get("PTT_GRAPH_MODE", "auto")
FORK_BUDGET = int(os.environ.get("PTT_GRAPH_FORK_BUDGET", "10"))
forked_captures = 0          # captures of this process in which at least one fork was recorded
_warned = False
_capture_forked = None       # None outside a capture_scope(); inside: whether a fork was recorded so far


def set_graph_mode(mode):
    """"fast" | "safe" | "auto" (see the module docstring); returns the previous mode."""
    global _mode
    if mode not in ("fast", "safe", "auto"):
        raise ValueError("graph mode: 'fast', 'safe' or 'auto'")
    prev, _mode = _mode, mode
    return prev


def graph_mode():
    return _mode


def _many_queues():
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8
    except ValueError:
        return False


def fork_allowed():
    """May code that is being captured right now put work on a second stream? Eager code never needs to ask."""
    global _warned
    if _mode == "fast" or (_mode == "auto" and (_many_queues() or forked_captures < FORK_BUDGET)):
        return True
    if _mode == "auto" and not _warned:
        _warned = True
        warnings.warn("ptt_amd: this process has captured %d hipGraphs with parallel branches; further captures are serialised "
                      "(slower replays) because ROCm 7.2's hipGraphLaunch can segfault once many such graphs exist in one process. "
                      "Export GPU_MAX_HW_QUEUES=8 before starting the process, or call ptt_amd.hot_path.set_graph_mode('fast'), to keep "
                      "forking (INTEGRATION.md, 'Known limits')." % forked_captures, RuntimeWarning, stacklevel=3)
    return False


def note_fork():
    """Called by whoever recorded a fork into the capture in progress."""
    global forked_captures, _capture_forked
    if _capture_forked is None:
        forked_captures += 1                 # a capture not wrapped in capture_scope(): every fork counts
    elif not _capture_forked:
        _capture_forked = True
        forked_captures += 1


def branch(main, side):
    """Inside a capture: the stream to put a parallel branch on — `side` (made to wait for `main`) when forking is allowed, else
    `main` itself (the branch then simply runs in line). The caller joins with join(main, got)."""
    if torch.cuda.is_current_stream_capturing() and not fork_allowed():
        return main
    if torch.cuda.is_current_stream_capturing():
        note_fork()
    side.wait_stream(main)
    return side


def join(main, used):
    if used is not main:
        main.wait_stream(used)


class capture_scope(object):
    """Around ONE torch.cuda.graph(...) capture: its forks count as one forked capture."""

    def __enter__(self):
        global _capture_forked
        self._outer, _capture_forked = _capture_forked, False
        return self

    def __exit__(self, *exc):
        global _capture_forked
        _capture_forked = self._outer

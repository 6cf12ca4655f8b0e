"""CPU binding of the one-process-per-GPU ranks of a node (scripts/train_ddp.sh:9 launches them with no binding at all).

A rank's Python thread queues the step (two hipGraph replays and one all-reduce once the training step is captured; ~400 launches
through autograd while it is not) and RCCL runs a proxy thread per rank: eight unbound ranks on a box that shows 16 usable cores
migrate and share cores. bind_rank() gives local rank r an even share of the cores the process may use — the cores of the GPU's
own NUMA node when sysfs tells (PCI address of the device -> /sys/bus/pci/devices/<addr>/numa_node ->
/sys/devices/system/node/node<k>/cpulist), shared evenly among the local ranks whose GPUs sit on that node."""
import os


def _parse_cpulist(text):
    cores = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cores += list(range(int(lo), int(hi or lo) + 1))
    return cores


def gpu_numa_node(index):
    """NUMA node of HIP device `index` from sysfs, or None (unknown / single node / not readable)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        addr = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % addr) as fh:
            node = int(fh.read().strip())
        return node if node >= 0 else None
    except Exception:            # noqa: BLE001 — no sysfs, no such attribute, no device: an even split is the fallback
        return None


def node_cores(node):
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as fh:
            return _parse_cpulist(fh.read())
    except (OSError, ValueError):
        return []


def plan(local_rank, local_world, allowed, numa_of_rank=None, cores_of_node=None):
    """The cores local rank `local_rank` of `local_world` gets out of `allowed` (sorted core ids). numa_of_rank: [node | None] per
    local rank; cores_of_node: {node: [core ids]}. Pure function (tests/test_affinity_cpu.py)."""
    allowed = sorted(allowed)
    if local_world <= 1 or not allowed:
        return allowed, None
    if numa_of_rank and all(n is not None for n in numa_of_rank):
        # NUMA-local only when EVERY rank can have a local core of its own (one rule for all ranks: the shares stay disjoint)
        local = {n: [c for c in (cores_of_node or {}).get(n, []) if c in set(allowed)] for n in set(numa_of_rank)}
        if all(len(local[n]) >= numa_of_rank.count(n) for n in local):
            node = numa_of_rank[local_rank]
            peers = [r for r in range(local_world) if numa_of_rank[r] == node]
            k, share = peers.index(local_rank), len(local[node]) // len(peers)
            return local[node][k * share:(k + 1) * share], node
    share = max(1, len(allowed) // local_world)
    lo = (local_rank * share) % len(allowed)
    return allowed[lo:lo + share], None


def bind_rank(local_rank=None, local_world=None):
    """os.sched_setaffinity for this rank -> {"cores": [...], "numa_node": k | None, "allowed": n} (what bench.py prints), or None
    when nothing was bound (one rank, PTT_NO_BIND=1, no sched_setaffinity on this platform)."""
    if os.environ.get("PTT_NO_BIND", "0") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else local_world
    if local_world <= 1:
        return None
    allowed = sorted(os.sched_getaffinity(0))
    numa = [gpu_numa_node(r) for r in range(local_world)]
    nodes = {n: node_cores(n) for n in set(numa) if n is not None}
    cores, node = plan(local_rank, local_world, allowed, numa, nodes)
    if not cores:
        return None
    try:
        os.sched_setaffinity(0, cores)
    except OSError:
        return None
    try:
        import torch
        torch.set_num_threads(max(1, min(len(cores), torch.get_num_threads())))
    except Exception:            # noqa: BLE001
        pass
    return {"cores": cores, "numa_node": node, "allowed": len(allowed)}

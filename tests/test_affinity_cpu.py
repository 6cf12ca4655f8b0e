"""ptt_amd.affinity.plan: the per-rank core split of a one-process-per-GPU node (pure host logic)."""
from ptt_amd import affinity


def test_even_split_without_numa():
    allowed = list(range(16))
    got = [affinity.plan(r, 8, allowed)[0] for r in range(8)]
    assert got == [[2 * r, 2 * r + 1] for r in range(8)]
    assert sorted(c for g in got for c in g) == allowed                     # disjoint, everything used


def test_fewer_cores_than_ranks_still_binds_one_core_each():
    got = [affinity.plan(r, 8, [0, 1, 2, 3])[0] for r in range(8)]
    assert all(len(g) == 1 for g in got) and set(c for g in got for c in g) == {0, 1, 2, 3}


def test_numa_local_split():
    allowed = list(range(32))
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    nodes = {0: list(range(0, 16)), 1: list(range(16, 32))}
    got = [affinity.plan(r, 8, allowed, numa, nodes) for r in range(8)]
    assert [g[1] for g in got] == numa
    assert got[0][0] == [0, 1, 2, 3] and got[3][0] == [12, 13, 14, 15] and got[4][0] == [16, 17, 18, 19] and got[7][0] == [28, 29, 30, 31]


def test_numa_unknown_for_one_rank_falls_back_to_even_split():
    got = affinity.plan(1, 2, list(range(8)), [0, None], {0: list(range(8))})
    assert got == ([4, 5, 6, 7], None)


def test_numa_node_cores_outside_the_cgroup_are_not_used():
    # the container may use cores 0-7 only; node 1's cores are 16-31: nothing local is allowed -> even split of what is
    nodes = {0: list(range(0, 16)), 1: list(range(16, 32))}
    assert affinity.plan(1, 2, list(range(8)), [0, 1], nodes) == ([4, 5, 6, 7], None)
    assert affinity.plan(0, 2, list(range(8)), [0, 1], nodes) == ([0, 1, 2, 3], None)      # ... for EVERY rank: disjoint shares


def test_single_rank_is_left_alone():
    assert affinity.plan(0, 1, [0, 1, 2, 3]) == ([0, 1, 2, 3], None)
    assert affinity.bind_rank(0, 1) is None


def test_cpulist_parser():
    assert affinity._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]

"""Multi-rank ICP on ONE GPU: two handles sharded as rank 0/1 of a world of 2, driven from two threads with an
in-process all-reduce.  Exercises exactly the code path the 8-GPU run uses (slices, count and normal-equation
all-reduces, identical LM decisions on every rank) and checks it against the single-rank run and the oracle."""
import threading

import numpy as np
import pytest

from conftest import pose_error

pytestmark = pytest.mark.gpu


class _ThreadAllReduce:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = 0

    def make(self, rank):
        def allreduce(arr):
            self.slots[rank] = arr.copy()
            self.barrier.wait(timeout=60)
            total = sum(self.slots[r] for r in range(self.world))   # same order on every rank
            self.barrier.wait(timeout=60)
            arr[:] = total
            if rank == 0:
                self.calls += 1
        return allreduce


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_rank(e3d, ob, synth, world):
    scans = synth.make_scene(3, 30000, seed=11)
    clouds = [(s["xyz"].numpy(), s["normals"].numpy(), s["T_init"], i == 2) for i, s in enumerate(scans)]
    d, iters, thr = 0.12, 4, 1e-9

    ref = e3d.PointToPlaneICP()
    ids = [ref.add_point_cloud(*c) for c in clouds]
    ref.run(d, 0, iters, thr, False)

    ar = _ThreadAllReduce(world)
    handles, errors = [], []
    for r in range(world):
        h = e3d.PointToPlaneICP()
        for c in clouds:
            h.add_point_cloud(*c)
        h.set_shard(r, world, ar.make(r))
        handles.append(h)

    def work(h):
        try:
            h.run(d, 0, iters, thr, False)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)
            ar.barrier.abort()

    th = [threading.Thread(target=work, args=(h,)) for h in handles]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    assert ar.calls > iters          # counts + several LM passes per iteration went through the all-reduce

    ref_counts = [(r[0], r[1], r[2], r[3]) for r in ref.pair_records()]
    for h in handles:
        assert [(r[0], r[1], r[2], r[3]) for r in h.pair_records()] == ref_counts      # global counts on every rank
        for i in ids:
            if i >= 0:
                ang, tr = pose_error(h.get_result_global_T_cloud(i), ref.get_result_global_T_cloud(i))
                assert ang <= 1e-5 and tr <= 1e-4
    # every rank holds bit-identical poses (same reduced numbers -> same decisions)
    for i in ids:
        if i >= 0:
            assert all(np.array_equal(handles[0].get_result_global_T_cloud(i), h.get_result_global_T_cloud(i)) for h in handles)
    # local work is split: each rank handled ~1/world of the queries
    q0 = ref.iter_records()[0]["queries"]
    for h in handles:
        assert abs(h.iter_records()[0]["queries"] - q0 / world) <= 8
    # and the oracle agrees
    o = ob.OracleICP()
    for c in clouds:
        o.add_point_cloud(*c)
    o.run(d, 0, iters, thr, False)
    assert [(r[0], r[1], r[2], r[3]) for r in o.pair_records()] == ref_counts

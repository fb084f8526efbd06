"""world_size-2 gloo tests (CPU) of the multi-rank path: the slice formula, the all-reduce callback used by
bench.py, and that summing per-rank normal equations reproduces the single-rank system (the per-rank compute
here is the oracle -- there is no GPU in this container; the same logic runs on the HIP library in
tests/test_gpu_distributed.py)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        d = importlib.import_module("dataset-pipeline_amd.dist")
        rng = np.random.RandomState(0)                      # identical data on every rank
        S = rng.uniform(-1, 1, (300, 3)).astype(np.float32); Sn = rng.normal(size=(300, 3)).astype(np.float32)
        T = rng.uniform(-1, 1, (300, 3)).astype(np.float32); Tn = rng.normal(size=(300, 3)).astype(np.float32)
        iq = rng.randint(0, 300, 1001).astype(np.int32); im = rng.randint(0, 300, 1001).astype(np.int32)
        ident = ([1, 0, 0, 0], [0, 0, 0])
        b0, b1 = d.shard_slice(len(iq), rank, world)
        H, b, c = ob.pair_system(S, Sn, T, Tn, iq[b0:b1], im[b0:b1], *ident, *ident)
        buf = np.concatenate([H.ravel(), b, [c], [b1 - b0]])
        d.make_allreduce()(buf)
        Hf, bf, cf = ob.pair_system(S, Sn, T, Tn, iq, im, *ident, *ident)
        ok = (np.abs(buf[:144] - Hf.ravel()).max() <= 1e-12 * np.abs(Hf).max()
              and np.abs(buf[144:156] - bf).max() <= 1e-12 * np.abs(bf).max()
              and abs(buf[156] - cf) <= 1e-12 * cf and int(buf[157]) == len(iq))
        q.put((rank, bool(ok), buf[:8].tolist()))
    finally:
        dist.destroy_process_group()


def test_shard_slices_cover_everything():
    d = importlib.import_module("dataset-pipeline_amd.dist")
    for n in (0, 1, 7, 64, 1001, 50_000_000):
        for w in (1, 2, 3, 8):
            sl = [d.shard_slice(n, r, w) for r in range(w)]
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(sl[i][1] == sl[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in sl) - min(e - b for b, e in sl) <= 1


@pytest.mark.timeout(120)
def test_allreduce_of_normal_equations_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]          # identical reduced buffer on every rank


# ---- path (B): image sharding -- per-rank systems of the owned images sum to the single-rank system ---------------------------
def _reg_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.reg_driver import OracleRegProblem
        from oracle import reg_binding as rb
        from reg_util import make_multi_image_scene
        d = importlib.import_module("dataset-pipeline_amd.dist")
        M = make_multi_image_scene(n_points=1500, n_images=3, width=120, height=90, seed=3)
        O = OracleRegProblem(K=M["K"], image_scale_count=M["n_levels"])
        O.set_intrinsics(0, M["width"], M["height"], M["params"], 0, M["n_levels"])
        O.set_point_scale(0, M["pts"], M["point_radius"], M["nbr"], M["fixed_desc"]); O.set_splat_points(M["pts"])
        for i, im in enumerate(M["images"]):
            O.set_image(i, 0, im["pyr"]); O.set_image_pose(i, im["q_init"], im["t_init"])
        O.update_observations(1)
        S = O.scales[0]; I0 = O.intr[0]

        def block(i):
            im = O.images[i]; o = O.obs[(i, 0)]
            return rb.accumulate(S["pts"], float(S["radius"]), S["nbr"], O.K, S["fixed"], S["var"], S["counts"], I0["levels"][0], 0,
                                 im["pyr"], O._R(im), im["t"], o[:4], o[4], O.robust_type, O.robust_param, 1.0, 0.0)
        V = 4 + 6 * 3
        buf = np.zeros(V * V + V + 4)
        full = np.zeros(V * V + V + 4)
        for i in range(3):
            H, b, s2, c2 = block(i)
            g = list(range(4)) + list(range(4 + 6 * i, 10 + 6 * i))
            for tgt, mine in ((full, True), (buf, d.image_owner(i, world) == rank)):
                if not mine:
                    continue
                Hm = tgt[:V * V].reshape(V, V)
                for r in range(10):
                    for c in range(r, 10):
                        Hm[g[r], g[c]] += H[r, c]
                    tgt[V * V + g[r]] += b[r]
                tgt[V * V + V:V * V + V + 2] += s2; tgt[V * V + V + 2:] += c2
        d.make_allreduce()(buf)
        ok = np.abs(buf - full).max() <= 1e-12 * np.abs(full).max() and int(buf[-2]) == int(full[-2]) > 100
        q.put((rank, bool(ok), [d.image_owner(i, world) for i in range(5)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_reg_image_blocks_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_reg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    assert all(owners == [0, 1, 0, 1, 0] for _, _, owners in res)

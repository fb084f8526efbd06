"""Randomised differential run of the normal-estimation path (A') against the CPU oracle (test infrastructure, not product code).

    python tests/fuzz_normals_vs_oracle.py --trials 40 --seed 1

Each trial draws a cloud -- a room scan, a scanner-sampled scan, a noisy plane with tight clusters and far outliers, a lattice (every
neighbour distance tied), a cloud with duplicated points, a thin line of points -- a size and a neighbour count k in 3 .. 70, and
compares `normals_knn` with the oracle's kd-tree estimator: neighbour index lists equal, NaN pattern equal, normals and curvatures
bit for bit.  The library's switches are read once per process, so each switch set runs in its own interpreter (`--worker`)."""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCH_SETS = [{}, {"E3D_KNN_SINGLE": "0"}, {"E3D_KNN_EST": "0"}, {"E3D_KNN_WIDE_WAVE": "0", "E3D_KNN_SEED": "0"}]
KINDS = ("room", "scanner", "clusters", "lattice", "duplicates", "line")


def draw(rng):
    return {"kind": str(rng.choice(KINDS, p=[0.3, 0.25, 0.15, 0.1, 0.1, 0.1])), "n": int(rng.integers(2_000, 400_000)),
            "k": int(rng.choice([3, 4, 5, 8, 10, 11, 16, 24, 32, 33, 40, 60, 61, 70])), "seed": int(rng.integers(1, 100_000)),
            "scale": float(rng.choice([0.01, 1.0, 30.0])), "offset": float(rng.choice([0.0, 0.0, 250.0]))}


def cloud(job):
    synth = importlib.import_module("dataset-pipeline_amd.synth")
    rng = np.random.RandomState(job["seed"])
    n = job["n"]
    if job["kind"] == "room":
        P = synth.make_scene(1, n, seed=job["seed"])[0]["xyz"].numpy()
    elif job["kind"] == "scanner":
        origin, yaw = synth.SCAN_POSES[job["seed"] % len(synth.SCAN_POSES)]
        P = synth.make_scan_angular(n, origin, yaw, job["seed"])[0].numpy()
    elif job["kind"] == "clusters":
        dense = rng.normal(size=(n, 3)) * np.array([1, 1, 0.01])
        tight = rng.normal(size=(max(n // 8, 50), 3)) * 1e-3 + np.array([5, 5, 5])
        far = rng.uniform(-1000, 1000, size=(6, 3))
        P = np.vstack([dense, tight, far])
    elif job["kind"] == "lattice":
        m = max(int(round(min(n, 30_000) ** (1 / 3))), 3)
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(max(m // 4, 1)), indexing="ij"), -1).reshape(-1, 3)
        P = g.astype(np.float64) * 0.25
    elif job["kind"] == "duplicates":
        base = rng.uniform(-2, 2, size=(max(n // 3, 10), 3)) * np.array([1, 1, 0.02])
        P = np.vstack([base, base[rng.randint(0, len(base), size=2 * len(base))]])
    else:
        t = rng.uniform(0, 10, size=(min(n, 20_000), 1))
        P = np.hstack([t, 0.5 * t + rng.normal(size=t.shape) * 1e-4, rng.normal(size=t.shape) * 1e-4])
    return np.ascontiguousarray(P * job["scale"] + job["offset"], np.float32)


def worker(path):
    e3d = importlib.import_module("dataset-pipeline_amd")
    from oracle import binding as ob
    bad = []
    jobs = json.load(open(path))
    for t, job in enumerate(jobs):
        P = cloud(job)
        vp = (0.0, 0.0, 10.0)
        gn, gc, gk = e3d.normals_knn(P, job["k"], vp, return_knn=True)
        on, oc, ok = ob.normals(P, k=job["k"], viewpoint=vp, return_knn=True)
        nan_o = np.isnan(on[:, 0])
        v = ~nan_o
        why = None
        if not np.array_equal(gk, ok): why = "lists"
        elif not np.array_equal(np.isnan(gn[:, 0]), nan_o): why = "nan pattern"
        elif not np.array_equal(gn[v].view(np.uint32), on[v].view(np.uint32)): why = "normals"
        elif not np.array_equal(gc[v].view(np.uint32), oc[v].view(np.uint32)): why = "curvatures"
        if why:
            bad.append({"trial": t, "job": job, "points": int(len(P)), "why": why})
    print("RESULT" + json.dumps({"trials": len(jobs), "bad": bad}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--worker", default=None)
    args = ap.parse_args()
    if args.worker:
        return worker(args.worker)
    rng = np.random.default_rng(args.seed)
    jobs = [draw(rng) for _ in range(args.trials)]
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(jobs, f)
    failures = 0
    for env in SWITCH_SETS:
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", f.name], env=e, capture_output=True, text=True)
        if p.returncode != 0:
            print("switches %s: worker failed\n%s" % (env, p.stderr[-2000:]))
            failures += 1
            continue
        r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1][len("RESULT"):])
        print("switches %s: %d trials, %d mismatches" % (json.dumps(env), r["trials"], len(r["bad"])))
        for b in r["bad"]:
            print("   ", json.dumps(b))
        failures += len(r["bad"])
    os.unlink(f.name)
    print("FUZZ %s" % ("OK" if failures == 0 else "FAILED (%d)" % failures))
    sys.exit(0 if failures == 0 else 1)


if __name__ == "__main__":
    main()

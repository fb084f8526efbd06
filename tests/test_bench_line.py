"""bench.py's last stdout line is what the driver parses: it has to stay one small JSON object whatever the detail grows to
(round 4's single 33 KB line was not parsed).  Built here from recorded detail files of real runs (no GPU needed)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# (the files of rounds 1 and 2 predate the leg layout)
RECORDED = sorted(glob.glob(os.path.join(ROOT, "profiles", "round[3-9]_bench_default*.json")) + glob.glob(os.path.join(ROOT, "profiles", "round[3-9]_bench_detail*.json")))


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_compact_line_is_small_and_complete(path):
    detail = json.load(open(path))
    line = json.dumps(bench.compact_line(detail), separators=(",", ":"))
    assert "\n" not in line and len(line) < 4096, len(line)
    o = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in o, k
    assert isinstance(o["config"].get("workload"), str)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in o["roofline"], k
    assert abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-3
    if "cpu_baseline" in detail:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in o["cpu_baseline"], k
    # the headline numbers are the detail's (rounded to >= 4 significant digits)
    assert abs(o["value"] - detail["value"]) <= 1e-4 * abs(detail["value"])
    assert abs(o["ms_per_step"] - detail["ms_per_step"]) <= 1e-4 * detail["ms_per_step"]


def test_there_is_a_recorded_run():
    assert RECORDED, "profiles/ holds no recorded bench detail file"


def test_emit_prints_the_compact_line_last(tmp_path, capsys):
    detail = json.load(open(RECORDED[-1]))

    class A:
        detail = str(tmp_path / "d.json")
    bench.emit(detail, A)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < 4096
    assert json.loads(out[0])["metric"] == detail["metric"]
    assert json.load(open(A.detail))["value"] == detail["value"]


def test_check_scale8_reads_compact_lines(tmp_path):
    """tools/check_scale8.py: the reader of the first multi-GPU run works on the compact lines (no GPU)."""
    import subprocess
    detail = json.load(open(RECORDED[-1]))
    l1 = bench.compact_line(detail)
    l8 = json.loads(json.dumps(l1))
    l8["n_gpus"] = 8
    l8["value"] *= 7.0
    if "allpairs" in l8.get("legs", {}):
        l8["legs"]["allpairs"]["value"] *= 5.0
        l8["legs"]["allpairs"]["ms_per_iter"] /= 5.0
    (tmp_path / "n1.txt").write_text("some earlier output\n" + json.dumps(l1) + "\n")
    (tmp_path / "n8.txt").write_text(json.dumps(l8) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scale8.py"), str(tmp_path / "n1.txt"), str(tmp_path / "n8.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [ln.split() for ln in r.stdout.splitlines() if ln.strip().startswith(("1 ", "8 "))]
    assert len(rows) == 2 and abs(float(rows[1][3]) - 7.0 / 8.0) < 1e-3, r.stdout


class _FakeICP:
    """Run() of the tool's loop: converges in iteration `conv` (returns True there), records one iteration record per call."""

    def __init__(self, conv):
        self.conv, self.recs, self.calls = conv, [], []

    def run(self, d, it, n, thr, progress):
        import collections
        self.calls.append(it)
        r = collections.defaultdict(float)
        r["iteration"] = it
        r["correspondences"] = 1000
        r["queries"] = 2000
        self.recs.append(r)
        return it >= self.conv

    def iter_records(self):
        return list(self.recs)

    def clear_records(self):
        self.recs = []


class _FakeRanks:
    world, comm, dist = 1, None, None

    def barrier(self):
        pass

    def reduce(self, v, op="sum"):
        import numpy as np
        return np.asarray(v, np.float64)


@pytest.mark.parametrize("conv,expect_steps,expect_conv", [(100, 20, None), (24, 20, 24), (18, 14, 18), (5, 1, 5)])
def test_timed_loop_ends_with_the_converging_iteration(conv, expect_steps, expect_conv):
    """src/exe/icp_scan_aligner.cc:342-370: the iteration whose Run() reports convergence is the last one the tool runs -- bench.py
    times it and nothing after it (round 4 timed five no-op steps after convergence)."""
    icp = _FakeICP(conv)
    dt, tot, warm, recs, per_rank, steps_run, converged_at = bench.run_icp(None, _FakeRanks(), icp, 0.01, 1e-10, 5, 20)
    assert steps_run == expect_steps and converged_at == expect_conv
    assert icp.calls == list(range(0, 5 + expect_steps))
    assert len(warm) == 5 and len(recs) == expect_steps and tot[0] == 1000 * expect_steps


def test_a_run_that_converges_in_the_warm_up_times_nothing():
    icp = _FakeICP(3)
    dt, tot, warm, recs, per_rank, steps_run, converged_at = bench.run_icp(None, _FakeRanks(), icp, 0.01, 1e-10, 5, 20)
    assert steps_run == 0 and converged_at == 3 and icp.calls == [0, 1, 2, 3]

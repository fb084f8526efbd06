"""The two scripts between a rocprofv3 counter pass and bench.py's `roofline.traffic`: tools/rocpd_summary.py (per-kernel counter sums,
and the same over a kernel's FULL-SIZE dispatches) and tools/make_traffic_json.py (bytes per launch; full-size figure for the
streaming ICP kernels, whose launches in bench.py's timed region all cover the whole scans)."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, rows, counter):
    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer)")
    db.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    t = 0
    for name, value in rows:
        db.execute("insert into kernels values (?, ?, ?)", (name, t, t + 1000)); t += 2000
        db.execute("insert into counters_collection values (?, ?, ?)", (name, counter, value))
    db.commit(); db.close()


def test_full_size_launches_are_what_the_streaming_kernels_report(tmp_path):
    lm = "void e3d::k_lm_pass<1>(HIP_vector_type<float, 4u> const*, int)"
    nb = "void e3d::k_nn_bounded_half<8>(HIP_vector_type<float, 4u> const*, unsigned int)"
    # a run that ramps up: launches over 22, 37, 73, 100, 100 M correspondences; KiB read (the counter reports half of it)
    fetch = [(lm, 22e6 * 48 / 2048), (lm, 37e6 * 48 / 2048), (lm, 73e6 * 48 / 2048), (lm, 100e6 * 48 / 2048), (lm, 100e6 * 48 / 2048),
             (nb, 1.0e6), (nb, 0.2e6)]
    write = [(lm, 8.0), (lm, 8.0), (lm, 8.0), (lm, 8.0), (lm, 8.0), (nb, 1.0e5), (nb, 0.2e5)]
    src = tmp_path / "prof"; src.mkdir()
    for tag, rows, ctr in (("icp_fetch", fetch, "FETCH_SIZE"), ("icp_write", write, "WRITE_SIZE")):
        _db(str(tmp_path / (tag + ".db")), rows, ctr)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), str(tmp_path / (tag + ".db")),
                               str(src / (tag + ".txt")), "e3d"], stdout=subprocess.DEVNULL)
    txt = open(src / "icp_fetch.txt").read()
    assert ", FETCH_SIZE, " in txt and ", FETCH_SIZE_FULL, " in txt
    dst = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), str(src), str(dst)], stdout=subprocess.DEVNULL)
    k = json.load(open(dst))["kernels"]
    full = k["k_lm_pass<1>"]
    assert full["full_size_launches"] == 2 and full["launches"] == 5
    assert abs(full["fetch_bytes_per_launch"] - 100e6 * 48) <= 1e-6 * 100e6 * 48          # the two 100 M launches, not the mean of five
    assert abs(full["all_launches_hbm_bytes_per_launch"] - (332e6 * 48 / 5 + 8 * 1024)) <= 1e-6 * 332e6 * 48
    b = k["k_nn_bounded_half<8>"]                                                        # list lengths vary: the plain mean
    assert "full_size_launches" not in b and abs(b["fetch_bytes_per_launch"] - 0.6e6 * 2048) <= 1.0

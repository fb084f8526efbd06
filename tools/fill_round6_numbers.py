"""Fills the @@NAME@@ placeholders of DESIGN.md / README.md from a bench_detail.json (the closing run of round 6).
    python tools/fill_round6_numbers.py profiles/round6_bench_detail.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
ap, sm = d["allpairs"], d["allpairs"]["scale_model"]
g, nl = d["image_registrator"], d["normal_estimation"]
f1 = lambda v: "%.1f" % v
f2 = lambda v: "%.2f" % v
vals = {
    "HEAD_MS": f2(d["ms_per_step"]), "HEAD_G": f2(d["value"] / 1e9), "HEAD_EACH": ", ".join(f1(v) for v in d["ms_per_step_each"]),
    "HEAD_STEADY": f1(d["ms_per_step_steady"]),
    "REGR_MS": f2(d["regression"]["ms_per_step"]), "SCAN_MS": f2(d["scanner_sampled"]["ms_per_step"]),
    "SCAN_RATIO": f2(d["scanner_sampled"]["ms_per_step"] / d["ms_per_step"]), "PART_MS": f2(d["partial_overlap"]["ms_per_step"]),
    "AP_MS": "%.0f" % ap["ms_per_iter"], "AP_G": f1(ap["value"] / 1e9), "AP_STEADY": "%.0f" % ap["ms_per_iter_steady"],
    "AP_EACH": ", ".join("%.0f" % v for v in ap["ms_per_iter_each"]),
    "SM_T1": "%.0f" % sm["ms_per_iter_n1"], "SM_T8": f1(sm["ms_per_iter_as_rank0_of_world"]), "SM_X": f2(sm["modelled_speedup"]),
    "SM_XS": f2(sm["steady"]["modelled_speedup"]), "SM_S1": "%.0f" % sm["steady"]["ms_per_iter_n1"], "SM_S8": f1(sm["steady"]["ms_per_iter_as_rank0_of_world"]),
    "SM_ND": f1(sm["non_dividing_ms_per_iter"]),
    "REG_G": f1(g["value"] / 1e9), "REG_ACC": f1(g["accumulate_ms_all_images"]), "REG_MS": f1(g["ms_per_run_iteration"]),
    "N32_MS": f2(nl["k32"]["ms_per_call"]), "N8_MS": f2(nl["k8"]["ms_per_call"]),
    "NS32_MS": f1(nl["scanner_sampled"]["k32"]["ms_per_call"]), "NS8_MS": f1(nl["scanner_sampled"]["k8"]["ms_per_call"]),
}
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    if name == "DESIGN.md":
        vals["DESIGN_KB"] = "%.0f" % (len(s.encode()) / 1024.0)
    for k, v in vals.items():
        s = s.replace("@@%s@@" % k, v)
    left = [w for w in s.split("@@")[1::2] if w.isupper() or "_" in w]
    open(p, "w").write(s)
    print(name, "placeholders left:", sorted(set(left))[:10])

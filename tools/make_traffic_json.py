"""profiles/round<N>_traffic.json from the rocprofv3 PMC passes of tools/prof_round6.sh (gpurun_out/r6prof/*_fetch.txt, *_write.txt:
lines `kernel signature, COUNTER, value summed over the launches, launches`).

    python tools/make_traffic_json.py [gpurun_out/r6prof] [profiles/round6_traffic.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOTE = ("HBM traffic per launch from rocprofv3 PMC, separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (tools/prof_round6.sh): the ICP "
        "kernels from `bench.py --no-cpu-baseline --no-reg --no-normals --no-allpairs --no-partial --no-whole-run --steps 6 --warmup 10` (the headline leg alone, 2 x 50 M points, iterations 0 .. 15: k_lm_pass<1> etc. from the full-size launches only), the "
        "ImageRegistrator kernels (k_reg_*, k_obs_*, k_splat_*, k_min_filter_*, k_color_*) from `bench.py --only reg --no-cpu-baseline --reg-images 4` (6048 x 4032 "
        "THIN_PRISM_FISHEYE, 10 M points: observation refreshes, accumulate passes and RunOnCurrentScale iterations), the normals "
        "entries k_knn_normals_k32 / _k8 = ALL kernels of one e3d_normals_knn call on 20 M points (`tools/bench_normals.py --repeat 1`: "
        "two calls per run, sums halved). Units: the counters are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read "
        "(MI355X_MICROARCH.md, HBM section) -> bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE); calibrated in round 1 on k_transform_bbox "
        "(16 B read + 16 B written per point x 50 M = 800 MB each).")


FULL_SIZE = ("k_lm_pass", "k_lm_cost_multi", "k_compact_corr", "k_corr_update", "k_nn_certify", "k_transform_bbox", "k_match_block_counts")


def parse(path, counter):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if ", %s, " % counter not in line:
            continue
        head, rest = line.rsplit(", %s, " % counter, 1)
        name = head.split("(")[0].replace("void ", "").replace("e3d::", "").strip()
        val, launches = rest.strip().split(", ")
        v, n = out.get(name, (0.0, 0))
        out[name] = (v + float(val), n + int(launches))
    return out


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6prof")
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "round6_traffic.json")
    kernels = {}
    for tag in ("icp", "reg"):
        f, w = parse(os.path.join(src, tag + "_fetch.txt"), "FETCH_SIZE"), parse(os.path.join(src, tag + "_write.txt"), "WRITE_SIZE")
        ff, wf = parse(os.path.join(src, tag + "_fetch.txt"), "FETCH_SIZE_FULL"), parse(os.path.join(src, tag + "_write.txt"), "WRITE_SIZE_FULL")
        for name in sorted(set(f) & set(w)):
            if name in kernels:
                continue                     # a helper both paths launch (k_match_block_counts ...): the ICP run's figure stands
            fb, wb = 2048.0 * f[name][0] / f[name][1], 1024.0 * w[name][0] / w[name][1]
            kernels[name] = {"launches": f[name][1], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
            if tag == "icp" and name.startswith(FULL_SIZE) and name in ff and name in wf:
                # the ICP run ramps up (9 .. 100 M correspondences over its sixteen iterations): for the kernels whose launches in bench.py's timed region all cover the whole 2 x 50 M scans, the
                # figure to compare is that of the full-size launches -- each pass's dispatches with >= 0.8 of its largest counter value
                fb, wb = 2048.0 * ff[name][0] / ff[name][1], 1024.0 * wf[name][0] / max(wf[name][1], 1)
                kernels[name].update({"all_launches_hbm_bytes_per_launch": kernels[name]["hbm_bytes_per_launch"], "full_size_launches": ff[name][1],
                                      "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb})
    for k in (32, 8):
        f = parse(os.path.join(src, "normals_k%d_fetch.txt" % k), "FETCH_SIZE"); w = parse(os.path.join(src, "normals_k%d_write.txt" % k), "WRITE_SIZE")
        ours = [n for n in f if n.startswith("k_") or "rocprim" in n or "rocclr" in n]       # the library's kernels, sorts, fills and copies
        if not ours:
            continue
        calls = 2.0                                                                          # bench_normals.py --repeat 1: warm-up + one timed call
        fb = sum(2048.0 * f[n][0] for n in ours) / calls; wb = sum(1024.0 * w[n][0] for n in ours if n in w) / calls
        kernels["k_knn_normals_k%d" % k] = {"launches": 1, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                                             "kernels_summed": sorted(ours)}
    json.dump({"_note": NOTE, "kernels": kernels}, open(dst, "w"), indent=1)
    print(len(kernels), "kernels")


if __name__ == "__main__":
    main()

"""profiles/round1_traffic.json from the two rocprofv3 PMC passes of tools/prof_final.sh (gpurun_out/pmc_fetch.txt, pmc_write.txt:
lines `kernel signature, COUNTER, value summed over the launches, launches`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOTE = ("HBM traffic per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py --no-cpu-baseline "
        "--steps 2`: 2 x 50 M points for the ICP kernels, 4 images 3840x2160 + 4 M points for the k_reg_* kernels; tools/prof_final.sh). "
        "Units: counters are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> "
        "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE). Calibrated on k_transform_bbox: 16 B read + 16 B written per point x 50 M = 800 MB each.")


def parse(path, counter):
    out = {}
    for line in open(path):
        if ", %s, " % counter not in line:
            continue
        head, rest = line.rsplit(", %s, " % counter, 1)
        name = head.split("(")[0].replace("void ", "").replace("e3d::", "").strip()
        val, launches = rest.strip().split(", ")
        out[name] = (float(val), int(launches))
    return out


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
    f, w = parse(os.path.join(src, "pmc_fetch.txt"), "FETCH_SIZE"), parse(os.path.join(src, "pmc_write.txt"), "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) & set(w)):
        fb, wb = 2048.0 * f[name][0] / f[name][1], 1024.0 * w[name][0] / w[name][1]
        kernels[name] = {"launches": f[name][1], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump({"_note": NOTE, "kernels": kernels}, open(os.path.join(ROOT, "profiles", "round1_traffic.json"), "w"), indent=1)
    print(len(kernels), "kernels")


if __name__ == "__main__":
    main()

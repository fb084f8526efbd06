"""Reads the bench lines of a 1 / 2 / 4 / 8-GPU run of bench.py (the LAST stdout line of each run: the compact JSON object) and prints
what the first multi-GPU run has to be read against: weak-scaling efficiency of the headline (2 scans, N x the points on N GPUs),
strong-scaling speed-up of the all-pairs job next to the single-GPU run's `scale_model` (rank 0 of a world of 8 measured on one
GPU: an upper bound without collectives), and the time the all-reduces took.

    python tools/check_scale8.py N1.txt N2.txt N4.txt N8.txt      (files holding bench.py's stdout; or a driver SCALE_rNN.json)
Needs no GPU."""
import json
import sys


def last_json_line(path):
    txt = open(path).read().strip()
    try:
        d = json.loads(txt)
        if isinstance(d, dict) and "metric" in d:
            return d
        if isinstance(d, dict):                       # a driver record: look for parsed lines inside
            found = []
            def walk(o):
                if isinstance(o, dict):
                    if "metric" in o and "n_gpus" in o:
                        found.append(o)
                    for v in o.values():
                        walk(v)
                elif isinstance(o, list):
                    for v in o:
                        walk(v)
            walk(d)
            return found
    except json.JSONDecodeError:
        pass
    for line in reversed(txt.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise SystemExit("%s: no JSON line" % path)


def main():
    runs = {}
    for p in sys.argv[1:]:
        d = last_json_line(p)
        for o in (d if isinstance(d, list) else [d]):
            runs[int(o["n_gpus"])] = o
    if 1 not in runs:
        raise SystemExit("need the N = 1 line")
    base = runs[1]
    ap1 = base.get("legs", {}).get("allpairs")
    print("%3s  %14s %9s %8s   %14s %9s %8s   %s" % ("N", "headline corr/s", "ms/step", "weak eff", "all-pairs corr/s", "ms/iter", "speed-up", "all-reduce ms/iter (max / min over ranks)"))
    for n in sorted(runs):
        o = runs[n]
        ap = o.get("legs", {}).get("allpairs")
        eff = o["value"] / (n * base["value"])
        sp = (ap["value"] / ap1["value"]) if (ap and ap1) else float("nan")
        cm = (ap or {}).get("comm", {})
        print("%3d  %14.4g %9.3f %8.3f   %14.4g %9.1f %8.2f   %s" % (
            n, o["value"], o["ms_per_step"], eff, ap["value"] if ap else float("nan"), ap["ms_per_iter"] if ap else float("nan"), sp,
            "%.2f / %.2f" % (cm.get("allreduce_ms_per_iter_max_over_ranks", float("nan")), cm.get("allreduce_ms_per_iter_min_over_ranks", float("nan"))) if cm else "-"))
    sm = (ap1 or {}).get("scale_model")
    if sm:
        w = sm["world"]
        print("\nscale model of the N = 1 run (rank 0 of a world of %d, measured on one GPU, no collectives): %.1f ms per iteration -> speed-up <= %.2f "
              "(steady iterations: <= %.2f); %.1f ms per iteration do not divide by N" % (w, sm["ms_per_iter_as_rank0_of_world"], sm["modelled_speedup"],
                                                                                           sm["steady_modelled_speedup"], sm["non_dividing_ms_per_iter"]))
        if w in runs and runs[w].get("legs", {}).get("allpairs"):
            got = runs[w]["legs"]["allpairs"]["ms_per_iter"]
            print("measured at N = %d: %.1f ms per iteration = model + %.1f ms (collectives, skew between ranks)" % (w, got, got - sm["ms_per_iter_as_rank0_of_world"]))


if __name__ == "__main__":
    main()

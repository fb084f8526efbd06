"""Prints the summary of `bench.py --only allpairs` output files (the last JSON line of each): N = 1 and replayed-rank iteration times,
the modelled 8-GPU speed-up.    python tools/show_allpairs.py file.json ..."""
import json
import sys

for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    d = d.get("allpairs", d)
    sm = d.get("scale_model")
    print(p, "N=1 %.1f ms/iter" % d["ms_per_iter"], [round(x) for x in d["ms_per_iter_each"]])
    if sm:
        print("   rank 0 of %d: %.2f ms/iter" % (sm["world"], sm["ms_per_iter_as_rank0_of_world"]), [round(x, 1) for x in sm["ms_per_iter_each_as_rank0"]],
              "speed-up <= %.3f (steady %.3f), non-dividing %.2f ms" % (sm["modelled_speedup"], sm["steady"]["modelled_speedup"], sm["non_dividing_ms_per_iter"]))

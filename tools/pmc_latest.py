"""profiles/pmc_spmv_*_latest.json (what bench.py reads `traffic` from) out of the per-workload summaries of a
tools/prof_round3.sh run merged into gpurun_out/TAG/:   python tools/pmc_latest.py TAG [p50 s50 p100 s100]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
which = sys.argv[2:] or ["p50", "s50", "p100", "s100"]
names = {"p50": "pmc_spmv_latest.json", "s50": "pmc_spmv_streaming_latest.json", "p100": "pmc_spmv_large_palette_latest.json",
         "s100": "pmc_spmv_large_streaming_latest.json"}
for w in which:
    d = json.load(open(os.path.join(ROOT, "gpurun_out", tag, f"pmc_{w}.json")))
    k = [n for n in d if "k_spmv_tile" in n]
    best = [n for n in k if "double, 0>" in n and "<true, true" in n][0]     # the fp64 masked + dot K*p of the PCG
    f, wr = d[best]["FETCH_SIZE_KB_median"], d[best]["WRITE_SIZE_KB_median"]
    kernel = best.split("<")[0].split("::")[-1].split()[-1]
    print(w, best, "fetch KB", f, "write KB", wr, "traffic MB", (2 * f + wr) / 1024)
    json.dump({"spmv_kernel": kernel, "record_palette": 1 if w[0] == "p" else 0, "fetch_kb": f, "write_kb": wr,
               "build": "round 3, " + tag,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/prof_round3.sh), median over the "
                         "dispatches of " + best + "; traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of "
                         "16-B/lane streaming reads, profiles/README.md)"},
              open(os.path.join(ROOT, "profiles", names[w]), "w"), indent=1)
    json.dump(d, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{w}.json"), "w"), indent=1)

# Round-3 profile set: bench line, kernel stats of the same command under rocprofv3, and separate FETCH_SIZE / WRITE_SIZE
# passes (PMC) for K*p on four workloads - 50^3 palette (headline), 50^3 streaming (palette off), 100^3 palette and
# 100^3 streaming (roofline_large: working set far beyond the 256 MiB Infinity Cache).  Writes gpurun_out/$TAG/ and the
# pmc_spmv_*_latest.json files bench.py reads `traffic` from.
#   usage (on the GPU box):  bash tools/prof_round3.sh r03_c
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r03_c}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python bench.py > $O/bench.json 2> $O/bench.log
python tools/bench_line.py $TAG < $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 bench.py --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 > $O/bench_profiled.json 2> $O/bench_profiled.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
head -12 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf $O/stats
for cfg in "p50:--cells 50 --palette 1" "s50:--cells 50 --palette 0" "p100:--cells 100 --palette 1" "s100:--cells 100 --palette 0"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  echo "== pmc $tag"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$tag -o f --output-format csv -- python3 tools/profile_kernels.py $args --reps 4 > $O/prof_$tag.json 2> $O/fetch_$tag.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$tag -o w --output-format csv -- python3 tools/profile_kernels.py $args --reps 4 > /dev/null 2> $O/write_$tag.log
  python3 tools/pmc_summary.py $(find $O/fetch_$tag -name "*counter_collection.csv" | head -1) $(find $O/write_$tag -name "*counter_collection.csv" | head -1) $O/pmc_$tag.json
  rm -rf $O/fetch_$tag $O/write_$tag
  python3 - "$O/pmc_$tag.json" "$tag" "$TAG" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = [n for n in d if "k_spmv_tile" in n]
best = [n for n in k if "double, 0>" in n and "<true, true" in n][0]   # the fp64 masked + dot K*p of the PCG
f, w = d[best]["FETCH_SIZE_KB_median"], d[best]["WRITE_SIZE_KB_median"]
print(sys.argv[2], best, "fetch KB", f, "write KB", w, "traffic MB", (2 * f + w) / 1024)
name = {"p50": "pmc_spmv_latest.json", "s50": "pmc_spmv_streaming_latest.json", "p100": "pmc_spmv_large_palette_latest.json",
        "s100": "pmc_spmv_large_streaming_latest.json"}[sys.argv[2]]
json.dump({"spmv_kernel": best.split("<")[0].split("::")[-1].split()[-1], "record_palette": 1 if sys.argv[2][0] == "p" else 0, "fetch_kb": f, "write_kb": w,
           "build": "round 3, " + sys.argv[3],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/prof_round3.sh), median over the "
                     "dispatches of " + best + "; traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of "
                     "16-B/lane streaming reads, profiles/README.md)"},
          open("profiles/" + name, "w"), indent=1)
PY
done

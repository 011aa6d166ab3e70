# Round-4 counter passes (separate rocprofv3 --pmc runs over tools/profile_kernels.py, never together with other traces):
#   FETCH_SIZE / WRITE_SIZE of K*p on 50^3 palette (headline), 50^3 streaming, 100^3 palette, 100^3 streaming
#     -> profiles/pmc_spmv_*_latest.json (what bench.py reads `traffic` from)
#   SQ counters of the palette K*p at 50^3 -> profiles/sq_spmv_latest.json (what bench.py prices `roofline.frac` with)
#   usage (on the GPU box):  bash tools/prof_round4.sh r04_v
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r04_v}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
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
           "build": "round 4, " + sys.argv[3],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/prof_round4.sh), median over the "
                     "dispatches of " + best + "; traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of "
                     "16-B/lane streaming reads, profiles/README.md)"},
          open("profiles/" + name, "w"), indent=1)
PY
done
echo "== SQ counters"
KERNEL="k_spmv_tile_lds<true, true, double, 0>"
PL_ROWS=0 bash tools/prof_kp_sq.sh $TAG/sq "$KERNEL" > $O/sq.log 2>&1
python3 - "$O/sq/kp_counters.json" "$TAG" "$KERNEL" <<'PY'
import json, sys
c = json.load(open(sys.argv[1]))
keep = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CU_CYCLES",
        "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES"]
missing = [k for k in keep if k not in c]
if missing:
    raise SystemExit("SQ pass incomplete, profiles/sq_spmv_latest.json left alone: " + ", ".join(missing))
json.dump({"spmv_kernel": "k_spmv_tile_lds", "record_palette": 1, "build": "round 4, " + sys.argv[2],
           "counters": {k: c[k] for k in keep},
           "source": "rocprofv3 --pmc (three separate passes, tools/prof_kp_sq.sh), median over the dispatches of void pl::" +
                     sys.argv[3] + " on the 50^3 Octet bench lattice; counters are sums over the 256 CUs / 1024 SIMDs of one dispatch"},
          open("profiles/sq_spmv_latest.json", "w"), indent=1)
print(json.dumps({k: c[k] for k in keep}))
PY
cp profiles/pmc_spmv_*latest.json profiles/sq_spmv_latest.json $O/

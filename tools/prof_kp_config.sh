# SQ counters (+ FETCH / WRITE) of ONE K*p application of a BASELINE configuration, as the solve applies it:
#   bash tools/prof_kp_config.sh CONFIG PRECISION OUTNAME      e.g.  2 0 sq_spmv_config2_fp64
# Separate --pmc passes (never combined with other trace domains) over tools/profile_config.py; the summary is written to
# gpurun_out/OUTNAME.json, stamped with the K*p source hash of the library (pl_version) - copy it into profiles/.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; CFG=${1:-1}; PREC=${2:-0}; NAME=${3:-sq_spmv_config${CFG}}
O=$R/gpurun_out/prof_$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 420 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o c --output-format csv -- python3 $R/tools/profile_config.py --config $CFG --precision $PREC --reps 5 > $O/p$i.out 2> $O/p$i.log || echo "pass $i failed"
  echo "pass $i done"
done
python3 - $O $CFG $PREC $R/gpurun_out/$NAME.json <<'PY'
import csv, glob, json, statistics, sys
O, cfg, prec, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
per = {}          # kernel name -> counter -> values
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_spmv_tile" not in r["Kernel_Name"] and "k_spmv_rows" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
names = sorted(per)
ndisp = {k: max(len(v) for v in per[k].values()) for k in names}
keep = [k for k in names if ndisp[k] >= 0.5 * max(ndisp.values())]      # the operator's launches (not a stray lifting product)
tot = {}
for k in keep:
    for c, v in per[k].items():
        tot[c] = tot.get(c, 0.0) + statistics.median(v)
info = json.loads(open(O + "/p1.out").read().strip().splitlines()[-1])
ver = info["version"]
out = {"spmv_kernel": "k_spmv_tile_lds" if all("k_spmv_tile_lds<" in k for k in keep) else keep[0].split("<")[0].split("::")[-1],
       "record_palette": 1, "config": cfg, "precision": prec, "kp_hash": ver.split("kp=")[-1] if "kp=" in ver else "unknown",
       "build": "round 5, " + ver, "kernels": [{"name": k, "dispatches": ndisp[k]} for k in keep],
       "counters": {c: v for c, v in tot.items() if c.startswith("SQ_")},
       "fetch_kb": tot.get("FETCH_SIZE"), "write_kb": tot.get("WRITE_SIZE"), "operator_ms_under_profiler": info["operator_ms"],
       "struts": info["struts"],
       "source": "rocprofv3 --pmc, separate passes (tools/prof_kp_config.sh) over tools/profile_config.py: per kernel the median "
                 "over its dispatches, summed over the launches of ONE operator application (two under node elimination); counters "
                 "are sums over the 256 CUs / 1024 SIMDs"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/p[0-9] $O/p[0-9][0-9]

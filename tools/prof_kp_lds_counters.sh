# SQ counters of the LDS-resident K*p (two separate --pmc passes, tools/profile_kernels.py); TA_* passes hung in round 2: not collected.
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-kpl}; mkdir -p $O; cd $R
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o c --output-format csv -- python3 tools/profile_kernels.py --reps 5 > /dev/null 2> $O/p$i.log || echo "pass $i failed"
done
python3 - $O <<'PY'
import csv, glob, sys, statistics, json
out = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_spmv_tile_lds<true, true, double, 0>" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        out[k] = statistics.median(v)
json.dump(out, open(sys.argv[1] + "/kp_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/p1 $O/p2

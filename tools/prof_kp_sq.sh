# SQ counters of the K*p kernel of the 50^3 Octet bench lattice (separate --pmc passes over tools/profile_kernels.py):
#   bash tools/prof_kp_sq.sh TAG "k_spmv_rows<true, true, double, 0>"      (PL_ROWS=0 in the environment for the tile kernel)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-kp_sq}; KERNEL=${2:-"k_spmv_rows<true, true, double, 0>"}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES_EQ_64 SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o c --output-format csv -- python3 $R/tools/profile_kernels.py --reps 5 > /dev/null 2> $O/p$i.log || echo "pass $i failed"
done
python3 - $O "$KERNEL" <<'PY'
import csv, glob, sys, statistics, json
out = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        out[k] = statistics.median(v)
out["kernel"] = sys.argv[2]
json.dump(out, open(sys.argv[1] + "/kp_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/p1 $O/p2 $O/p3

# Round 4: DDM tests, then the 32^3-cell operator under rocprofv3 --stats with the matrix-pipe cell product and with the
# register GEMV of rounds 2-3 (PL_DDM_MFMA=0).   bash tools/prof_ddm4.sh [TAG]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-r04_ddm}; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ddm.py tests/test_gpu_opti.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  PL_DDM_MFMA=$v rocprofv3 --kernel-trace --stats -d $O/ddm$v -o ddm --output-format csv -- python3 $R/tools/profile_ddm.py 32 > $O/ddm32_mfma$v.json 2> $O/ddm32_mfma$v.log
  cat $O/ddm32_mfma$v.json
  f=$(find $O/ddm$v -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-160; cp $f $O/ddm32_mfma${v}_kernel_stats.csv
  rm -rf $O/ddm$v
done

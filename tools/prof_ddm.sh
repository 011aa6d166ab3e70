set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02_j; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ddm.py tests/test_gpu_opti.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
rocprofv3 --kernel-trace --stats -d $O/ddm -o ddm --output-format csv -- python3 tools/profile_ddm.py 32 > $O/ddm32.json 2> $O/ddm32.log
cat $O/ddm32.json
head -8 $(find $O/ddm -name "*kernel_stats.csv" | head -1) | cut -c1-200
find $O/ddm -name "*kernel_trace.csv" -delete

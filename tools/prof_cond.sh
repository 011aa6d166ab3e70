set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02_n; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/st -o c --output-format csv -- python3 tools/ab_condense.py BCC 100 0.05 > $O/ab.log 2>&1
cat $O/ab.log | tail -1
head -12 $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-130,380-480
find $O/st -name "*kernel_trace.csv" -delete

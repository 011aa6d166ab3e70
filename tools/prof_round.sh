# Round profile: full GPU tests, default bench line, kernel stats under rocprofv3, FETCH/WRITE PMC passes.
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r02_k}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || true
tail -3 $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.log
cat $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 bench.py --cpu-cells 0 --no-e2e --no-streaming > $O/bench_profiled.json 2> $O/bench_profiled.log
head -14 $(find $O/stats -name "*kernel_stats.csv" | head -1) | cut -c1-220
find $O/stats -name "*kernel_trace.csv" -delete
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- python3 tools/profile_kernels.py --reps 5 > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- python3 tools/profile_kernels.py --reps 5 > /dev/null 2> $O/write.log
python3 tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_fetch_write.json
rm -rf $O/fetch $O/write

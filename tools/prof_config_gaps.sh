# Kernel stats AND the idle gaps inside one solve of a bench.py configuration:  bash tools/prof_config_gaps.sh TAG <bench.py args>
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python3 bench.py "$@" --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 > $O/bench.json 2> $O/bench.log
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 bench.py "$@" --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 > $O/bench_profiled.json 2> $O/bench_profiled.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 tools/solve_gaps.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/solve_gaps.txt
rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv 16 || head -16 $O/kernel_stats.csv | cut -c1-160
cat $O/solve_gaps.txt
python3 tools/bench_line.py $TAG < $O/bench.json

# Kernel stats of one bench.py configuration under rocprofv3:  bash tools/prof_config.sh TAG <bench.py args>
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 bench.py "$@" --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 > $O/bench_profiled.json 2> $O/bench_profiled.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv 14 || head -14 $O/kernel_stats.csv | cut -c1-160
python3 tools/bench_line.py $TAG < $O/bench_profiled.json

# rocprofv3 kernel stats of one bench.py invocation.  usage: bash tools/prof_bench_stats.sh TAG <bench.py args...>
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 bench.py --cpu-cells 0 --no-e2e --no-streaming "$@" > $O/bench.json 2> $O/bench.log
find $O/stats -name "*kernel_trace.csv" -delete
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s} %")
PY

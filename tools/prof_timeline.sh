# Assembly timeline of a bench step from a rocprofv3 kernel trace.  usage: bash tools/prof_timeline.sh TAG [bench.py args]
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rocprofv3 --kernel-trace -d $O/trace -o b --output-format csv -- python3 bench.py --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 --steps 4 --warmup 1 "$@" > $O/bench.json 2> $O/bench.log
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/assembly_timeline.py $F 2 > $O/assembly_timeline.txt
rm -rf $O/trace
tail -70 $O/assembly_timeline.txt

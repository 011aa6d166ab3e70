# One bench.py line per BASELINE.json configuration on one GPU (JSON per configuration + a digest):  bash tools/config_lines.sh TAG
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1; TAG=${1:-configs}; O=gpurun_out/$TAG; mkdir -p $O
B="python3 bench.py --cpu-cells 0 --no-e2e --no-streaming --large-cells 0"
$B --config 1 --steps 20 --warmup 5 > $O/config1.json 2> $O/config1.log; python3 tools/bench_line.py "configs[1]" < $O/config1.json
$B --config 2 --steps 3 --warmup 1 > $O/config2.json 2> $O/config2.log; python3 tools/bench_line.py "configs[2]" < $O/config2.json
$B --config 4 --steps 3 --warmup 1 > $O/config4.json 2> $O/config4.log; python3 tools/bench_line.py "configs[4]" < $O/config4.json
$B --config 3 > $O/config3.json 2> $O/config3.log; python3 -c "import json; d=json.load(open('$O/config3.json')); print('configs[3]', round(d['value']/1e6,2), 'M beams/s', round(d['ms_per_step'],2), 'ms per design iteration', d['config']['pcg_iterations_first_last'])"

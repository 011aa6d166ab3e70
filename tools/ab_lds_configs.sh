# bench lines of the four configurations with the LDS-resident K*p (default) and with the gather kernel (PL_TILE_LDS=0)
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1
B="python3 bench.py --cpu-cells 0 --no-e2e --no-streaming --large-cells 0"
for cfg in "1 --steps 10" "2 --steps 2 --warmup 1" "4 --steps 2 --warmup 1"; do
  for v in 0 1; do
    PL_TILE_LDS=$v $B --config $cfg | python3 tools/bench_line.py "config $cfg lds=$v"
  done
done
for v in 0 1; do PL_TILE_LDS=$v $B --config 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 3 lds=$v', round(d['ms_per_step'],2), 'ms per design iteration')"; done

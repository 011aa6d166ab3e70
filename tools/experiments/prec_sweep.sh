# precision = 1 with and without the continued Krylov process (PL_MP_FLY), against fp64
set -e
mkdir -p gpurun_out/r04_l; : > gpurun_out/r04_l/prec.txt
run() { python bench.py --steps 2 --warmup 1 "$@" --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); c=d['config']
print(' '.join(sys.argv[1:]), 'fly', os.environ.get('PL_MP_FLY','1'), round(d['value']/1e6,1), 'M', round(d['ms_per_step'],2), 'ms', c.get('pcg_iterations'), 'its inner', c.get('inner_solves'))
" "$@" >> gpurun_out/r04_l/prec.txt; }
for c in "--config 1" "--cells 100 100 100" "--config 2" "--config 4"; do
  [ "$c" != "--config 4" ] && run $c --precision 0
  PL_MP_FLY=0 run $c --precision 1
  PL_MP_FLY=1 run $c --precision 1
done
cat gpurun_out/r04_l/prec.txt

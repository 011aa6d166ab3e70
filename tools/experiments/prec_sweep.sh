# fp64 against precision = 1 (fp32-stored PCG vectors, fp64 refinement) by lattice size: where the host's rule
# "precision = 1 from 2 M nodes" (LatticeSim.device_model) comes from.  Writes gpurun_out/r04_l/prec.txt
set -e
mkdir -p gpurun_out/r04_l; : > gpurun_out/r04_l/prec.txt
run() { python bench.py --steps 2 --warmup 1 "$@" --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(' '.join(sys.argv[1:]), round(d['value']/1e6,1), 'M', round(d['ms_per_step'],2), 'ms', c.get('pcg_iterations'), 'its inner', c.get('inner_solves'))
" "$@" >> gpurun_out/r04_l/prec.txt; }
for c in "--config 1" "--cells 100 100 100" "--config 2" "--config 4"; do
  [ "$c" != "--config 4" ] && run $c --precision 0
  run $c --precision 1
done
cat gpurun_out/r04_l/prec.txt

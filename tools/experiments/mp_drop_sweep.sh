# precision = 1: depth of the fp32 inner solves.  PL_MP_DROP = fixed drop of ||r||^2 per stage; PL_MP_STAGE = equal stages of at
# most that many decades (the default rule).  usage: [DROPS="1e-8 1e-6"] [STAGES="5 6 7"] bash tools/experiments/mp_drop_sweep.sh
run() { python bench.py --steps 2 --warmup 1 --precision 1 "$@" --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); c=d['config']
print('drop', os.environ.get('PL_MP_DROP','-'), 'stage', os.environ.get('PL_MP_STAGE','-'), ' '.join(sys.argv[1:]), round(d['value']/1e6,1), 'M', round(d['ms_per_step'],2), 'ms', c.get('pcg_iterations'), 'its inner', c.get('inner_solves'), 'rel', c.get('rel_residual'))" "$@"; }
for d in ${DROPS:-}; do for c in "--config 1" "--cells 100 100 100" "--config 2" "--config 4"; do PL_MP_DROP=$d run $c; done; done
for s in ${STAGES:-6}; do for c in "--config 1" "--cells 100 100 100" "--config 2" "--config 4"; do PL_MP_STAGE=$s run $c; done; done

# assembly_ms_last of the default bench; usage: [ENV=..] bash tools/experiments/ab_assembly.sh [runs] [bench args]
n=${1:-3}; shift || true
for i in $(seq 1 $n); do python bench.py --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('cumask', os.environ.get('PL_BSR_CUMASK','-'), 'trtri_rows', os.environ.get('PL_TRTRI_ROWS','-'), round(d['value']/1e6,1),'M', round(d['ms_per_step'],2),'ms | assembly',round(k['assembly_ms_last'],3),'solve',round(k['solve_ms_last'],2), d['config']['pcg_iterations'])"; done

# weak-scaling workload as eight loopback slabs: iterations of the ordinary and of the single-reduction form
mkdir -p gpurun_out/r04_t
for f in 0 1; do python bench.py --config 1 --loopback 8 --cg-form $f --steps 1 --warmup 1 --cpu-cells 0 --no-e2e --no-streaming --large-cells 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('cg_form', c.get('cg_form'), 'iterations', c['pcg_iterations'], 'converged', c['converged'], 'rel', c['rel_residual'], 'ranks', c['ranks'], 'ms/step (loopback, not an 8-GPU time)', round(d['ms_per_step'],1), c.get('preconditioner'))
" >> gpurun_out/r04_t/loop8_cgform.txt; done; cat gpurun_out/r04_t/loop8_cgform.txt

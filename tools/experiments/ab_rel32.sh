set -o pipefail
python -m pytest tests/test_gpu_parity.py -q -x -k "float_lever or assembly_schedules or strain_mode or lds_resident" > gpurun_out/r05_r_pytest.log 2>&1; tail -3 gpurun_out/r05_r_pytest.log
for rep in 1 2; do
PL_NO_REL32=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other --no-e2e --no-streaming > gpurun_out/r05_r_off$rep.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-other --no-e2e --no-streaming > gpurun_out/r05_r_on$rep.json 2>/dev/null
done
PL_NO_REL32=1 python bench.py --config 2 --steps 3 --warmup 1 --no-other --no-e2e --no-streaming > gpurun_out/r05_r_c2off.json 2>/dev/null
python bench.py --config 2 --steps 3 --warmup 1 --no-other --no-e2e --no-streaming > gpurun_out/r05_r_c2on.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_r_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"]/1e6,2), round(d["ms_per_step"],3), d["kernels_ms"].get("pcg_iteration"), d["kernels_ms"].get("spmv"))
    except Exception as e:
        print(f, "failed", e)
PY

# A/B of the LDS-resident K*p builds (build_exp/, tools/build_exp.sh) against the gather kernel (PL_TILE_LDS=0)
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1
run() { python3 tools/profile_kernels.py --reps 20 $ARGS | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'K*p %.1f us  iteration %.1f us' % (d['spmv_ms']*1e3, d['pcg_iter_ms']*1e3))"; }
for i in 1 2; do
PL_TILE_LDS=0 run old
run lds_512
for v in ${VARIANTS:-b256 b384 recg}; do PYLATTICE_HIP_LIB=$R/build_exp/lib_$v.so run lds_$v; done
done
if [ -f build_exp/lib_stamps.so ]; then
PYLATTICE_HIP_LIB=$R/build_exp/lib_stamps.so python3 tools/experiments/tile_stamps.py 50 1
fi

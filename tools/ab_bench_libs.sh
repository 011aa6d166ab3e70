set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1
for i in $(seq 1 ${ROUNDS:-3}); do
  for lib in ${LIBS:-build_exp/lib_A.so default}; do
    if [ "$lib" = default ]; then unset PYLATTICE_HIP_LIB; else export PYLATTICE_HIP_LIB=$R/$lib; fi
    python3 bench.py --no-e2e --cpu-cells 0 --no-streaming | python3 tools/bench_line.py $lib
  done
done

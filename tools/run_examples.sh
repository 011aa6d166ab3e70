# Every script of examples/ on the GPU box, non-interactive (MPLBACKEND=Agg); prints status, wall time and the last line.
# (optimization_DDM_surrogate_chain.py first: it builds the reduced basis domain_decomposition_example.py reads.)
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1; export MPLBACKEND=Agg
for f in examples/optimization/optimization_DDM_surrogate_chain.py examples/simulation/*.py examples/optimization/Simple_optimization_case.py examples/optimization/graded_bcc_adjoint.py; do
  log=gpurun_out/example_$(basename $f .py).log
  t0=$(date +%s)
  if timeout -k 10 300 python3 $f > $log 2>&1; then s=ok; else s="FAILED($?)"; fi
  echo "$f $s $(( $(date +%s) - t0 )) s | $(tail -1 $log | cut -c1-100)"
done

# tools/profile_kernels.py (HIP-event timings of K*p, a whole PCG iteration, record build, BSR fill) for several builds
# of the library.  usage: bash tools/exp_libs.sh "extra args of profile_kernels.py" lib1.so lib2.so ...
set -e
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1; ARGS=$1; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset PYLATTICE_HIP_LIB; else export PYLATTICE_HIP_LIB=$R/$lib; fi
  echo "== $lib"; python3 tools/profile_kernels.py $ARGS | tail -1
done

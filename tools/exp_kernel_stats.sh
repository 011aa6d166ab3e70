# Kernel stats of whole PCG iterations (pl_time_kernel case 3) for several builds of the library.
# usage: bash tools/exp_kernel_stats.sh TAG lib1.so lib2.so ...   ("default" = the in-tree build)
set -e
cd /tmp && export TMPDIR=/tmp
set -uo pipefail; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset PYLATTICE_HIP_LIB; else export PYLATTICE_HIP_LIB=$R/$lib; fi
  rocprofv3 --kernel-trace --stats -d $O/$name -o s --output-format csv -- python3 tools/profile_kernels.py --reps 20 $PK_ARGS > $O/$name.json 2> $O/$name.log
  python3 - $O/$name <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"{sys.argv[1].split('/')[-1]:10s} {r['Name'][:58]:58s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.2f} us")
PY
  find $O/$name -name "*kernel_trace.csv" -delete
done

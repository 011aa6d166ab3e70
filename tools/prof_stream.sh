set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02_l}
mkdir -p $O
cd $R
for cfg in "p0:--palette 0" "gr:--graded 1"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d $O/stats_$tag -o s --output-format csv -- python3 tools/profile_kernels.py $args > $O/stats_$tag.json 2> $O/stats_$tag.log
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$tag -o f --output-format csv -- python3 tools/profile_kernels.py $args --reps 5 > /dev/null 2> $O/fetch_$tag.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$tag -o w --output-format csv -- python3 tools/profile_kernels.py $args --reps 5 > /dev/null 2> $O/write_$tag.log
  F=$(find $O/fetch_$tag -name "*counter_collection.csv" | head -1); W=$(find $O/write_$tag -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py $F $W $O/pmc_$tag.json
  cat $O/stats_$tag.json
  S=$(find $O/stats_$tag -name "*kernel_stats.csv" | head -1); head -12 $S
  rm -rf $O/fetch_$tag $O/write_$tag
  find $O/stats_$tag -name "*kernel_trace.csv" -delete
done

R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r06h
mkdir -p $O
cd /tmp
for v in base dq2x; do
  if [ "$v" = "base" ]; then lib=$R/convdr_amd/libconvdr_hip.so; else lib=$R/convdr_amd/libconvdr_hip_$v.so; fi
  export CONVDR_HIP_LIB=$lib
  rocprofv3 --kernel-trace --stats -d $O/prof_$v -o kd -- python3 $R/bench.py --workload train_kd --steps 5 --warmup 2 > $O/prof_$v.log 2>&1
  DB=$(find $O/prof_$v -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py $DB > $O/$v.kernel_stats.txt
  python3 $R/tools/train_timeline.py $DB > $O/$v.timeline.txt
  python3 $R/tools/train_timeline.py $DB --dispatches > $O/$v.dispatches.txt
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cd $R
head -30 $O/base.timeline.txt; head -30 $O/dq2x.timeline.txt

set -x
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof_kd -o kd -- python3 $R/bench.py --workload train_kd --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG/prof_kd.log 2>&1
cd $R
DB=$(find gpurun_out/$TAG/prof_kd -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/$TAG/train_kd.kernel_stats.txt
python tools/train_timeline.py $DB > gpurun_out/$TAG/train_kd.timeline.txt
python tools/train_timeline.py $DB --dispatches > gpurun_out/$TAG/train_kd.dispatches.txt
find gpurun_out/$TAG -name "*.db" -delete
cat gpurun_out/$TAG/train_kd.timeline.txt

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/dbg/ab_train.sh "base frag2" 3 > gpurun_out/ab_frag2_train.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite2.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite2.log )
N=48 python tools/dbg/stream_outlier_hunt.py > gpurun_out/stream_hunt48.log 2>&1
cat gpurun_out/ab_frag2_train.log; tail -6 gpurun_out/gpu_suite2.log; cut -c1-14 gpurun_out/stream_hunt48.log | sort | uniq -c | sort -k2 -n | tail -50; grep -c "keeping set 1\|moving to" gpurun_out/stream_hunt48.log

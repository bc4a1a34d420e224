cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/dbg/ab_opt.sh "CONVDR_DGRAD_FFN1_256=0 CONVDR_DGRAD_FFN1_256=1 CONVDR_DGRAD_AO_256=1" 3 > gpurun_out/ab_dgrad_256b.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite3.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite3.log )
cat gpurun_out/ab_dgrad_256b.log; tail -4 gpurun_out/gpu_suite3.log

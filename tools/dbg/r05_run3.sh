cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.log )
bash tools/dbg/ab_opt.sh "CONVDR_OPT_GELU_GP=1 CONVDR_OPT_GELU_GP=0" 2 > gpurun_out/ab_gelu_gp2.log 2>&1
tail -15 gpurun_out/gpu_suite.log; cat gpurun_out/ab_gelu_gp2.log

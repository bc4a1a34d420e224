cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.log )
bash tools/dbg/ab_opt.sh "CONVDR_OPT_GELU_GP=1 CONVDR_OPT_GELU_GP=0" 3 > gpurun_out/ab_gelu_gp.log 2>&1
for rep in 1 2; do
  python tools/enc_kernels.py 64 144 2>/dev/null | grep total >> gpurun_out/ln_fuse_9k.log
  CONVDR_OPTIONS=fused_ln_min_rows=0 python tools/enc_kernels.py 64 144 2>/dev/null | grep total >> gpurun_out/ln_fuse_9k.log
done
tail -5 gpurun_out/gpu_suite.log; cat gpurun_out/ab_gelu_gp.log gpurun_out/ln_fuse_9k.log

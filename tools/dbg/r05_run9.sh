cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/dbg/ab_opt.sh "CONVDR_DGRAD_FFN1_256=0 CONVDR_DGRAD_FFN1_256=1 CONVDR_DGRAD_QKV_256=1 CONVDR_DGRAD_FFN1_256=1,CONVDR_DGRAD_QKV_256=1" 3 > gpurun_out/ab_dgrad_256.log 2>&1
( timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -k "encoder_backward_matches" > gpurun_out/gpu_bwd.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_bwd.log )
cat gpurun_out/ab_dgrad_256.log; tail -3 gpurun_out/gpu_bwd.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "wgrad or backward_at_256 or encoder_backward or flat_arena or layer_completion" > gpurun_out/gpu_tn2.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_tn2.log )
bash tools/dbg/ab_train.sh "base tnblk" 3 > gpurun_out/ab_tn_thread.log 2>&1
bash tools/dbg/ab_opt.sh "CONVDR_WGRAD_AFTER_LN=0 CONVDR_WGRAD_AFTER_LN=1" 3 > gpurun_out/ab_wgrad_after_ln.log 2>&1
tail -3 gpurun_out/gpu_tn2.log; cat gpurun_out/ab_tn_thread.log gpurun_out/ab_wgrad_after_ln.log

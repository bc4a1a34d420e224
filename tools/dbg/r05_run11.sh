cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "wgrad or backward_at_256 or encoder_backward or flat_arena or configs2" > gpurun_out/gpu_tn.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_tn.log )
bash tools/dbg/ab_train.sh "base tn2" 3 > gpurun_out/ab_tn_r3.log 2>&1
tail -3 gpurun_out/gpu_tn.log; cat gpurun_out/ab_tn_r3.log

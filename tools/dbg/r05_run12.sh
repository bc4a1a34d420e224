cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/dbg/ab_libs.sh "base lnjit" 3 > gpurun_out/ab_ln_rolling.log 2>&1
( timeout 1200 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x > gpurun_out/gpu_enc.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_enc.log )
cat gpurun_out/ab_ln_rolling.log; tail -3 gpurun_out/gpu_enc.log

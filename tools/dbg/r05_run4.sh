cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -k "256_tile or configs2 or teacher_embedding or watchdog or trained_model or dropout_forward" > gpurun_out/gpu_sel.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_sel.log )
bash tools/dbg/ab_opt.sh "CONVDR_OPT_GELU_GP=1 CONVDR_OPT_GELU_GP=0" 3 > gpurun_out/ab_gelu_gp3.log 2>&1
./tools/proto/bin/w16_proto 768 > gpurun_out/w16_modes_768.log 2>&1
N=16 python tools/dbg/stream_outlier_hunt.py > gpurun_out/stream_hunt16.log 2>&1
tail -8 gpurun_out/gpu_sel.log; cat gpurun_out/ab_gelu_gp3.log; tail -11 gpurun_out/w16_modes_768.log; cat gpurun_out/stream_hunt16.log

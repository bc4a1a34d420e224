mkdir -p gpurun_out/r06d
CONVDR_HIP_LIB=$GRAFT_REPO_ROOT/convdr_amd/libconvdr_hip_trace.so python tools/dbg/attn_bwd_trace.py > gpurun_out/r06d/attn_bwd_trace.txt 2>&1
python -m pytest tests/test_train_gpu.py -q -x -k "attention or dropout or encoder_backward or deterministic or reproducible" 2>&1 | tail -15 > gpurun_out/r06d/t.txt
cp .ab/r05/convdr_amd/libconvdr_hip.so convdr_amd/libconvdr_hip_r05.so
bash tools/dbg/ab_train.sh "r05 base dq2" 3 > gpurun_out/r06d/ab_spans.txt 2>&1
cat gpurun_out/r06d/attn_bwd_trace.txt gpurun_out/r06d/t.txt gpurun_out/r06d/ab_spans.txt

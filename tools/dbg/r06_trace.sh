mkdir -p gpurun_out/r06c
CONVDR_HIP_LIB=$GRAFT_REPO_ROOT/convdr_amd/libconvdr_hip_trace.so python tools/dbg/attn_bwd_trace.py > gpurun_out/r06c/attn_bwd_trace.txt 2>&1
cp convdr_amd/../.ab/r05/convdr_amd/libconvdr_hip.so convdr_amd/libconvdr_hip_r05.so
bash tools/dbg/ab_train.sh "r05 base" 2 > gpurun_out/r06c/ab_spans.txt 2>&1
python -m pytest tests/test_train_gpu.py -q -x -k "deterministic or configs2 or replay or reference_run or dpr_checkpoint" 2>&1 | tail -15 > gpurun_out/r06c/t.txt
cat gpurun_out/r06c/attn_bwd_trace.txt gpurun_out/r06c/ab_spans.txt gpurun_out/r06c/t.txt

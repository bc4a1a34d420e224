mkdir -p gpurun_out/r06m
python -m pytest tests/test_train_gpu.py -q -x -k "attention or dropout or encoder_backward or watchdog" 2>&1 | tail -4 > gpurun_out/r06m/t.txt
python tools/dbg/attn_fused_vs_split.py 2>&1 | grep -v "^Using\|amdgpu.ids" > gpurun_out/r06m/fused_vs_split.txt
bash tools/dbg/ab_train.sh "pre base" 4 > gpurun_out/r06m/ab_tail.txt 2>&1
cat gpurun_out/r06m/t.txt gpurun_out/r06m/fused_vs_split.txt gpurun_out/r06m/ab_tail.txt

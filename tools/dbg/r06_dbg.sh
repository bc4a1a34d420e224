mkdir -p gpurun_out/r06q
bash tools/dbg/ab_env.sh "CONVDR_EVENT_SYSTEM_FENCE=1 CONVDR_EVENT_SYSTEM_FENCE=0" 4 > gpurun_out/r06q/ab_fence.txt 2>&1
python -m pytest tests/test_train_gpu.py tests/test_parallel_gpu.py -q -x -k "layer_completion or two_ranks or configs2 or sumsq or gradient_norm" 2>&1 | tail -4 >> gpurun_out/r06q/ab_fence.txt
cat gpurun_out/r06q/ab_fence.txt

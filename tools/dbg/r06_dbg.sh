mkdir -p gpurun_out/r06r
python -m pytest tests/test_parallel_gpu.py tests/test_train_gpu.py -q -x -k "two_ranks or layer_completion or gradient_norm_summed or sparse" 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/r06r/t.txt
cat gpurun_out/r06r/t.txt

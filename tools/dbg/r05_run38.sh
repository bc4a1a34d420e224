mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_parallel_gpu.py -q -m gpu > gpurun_out/r38_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r38_pytest.log

mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -k "straight_line" > gpurun_out/r23_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r23_pytest.log

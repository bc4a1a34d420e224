mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_encoder_gpu.py -q -m gpu -k "straight_line or split_contraction" > gpurun_out/r33_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r33_pytest.log

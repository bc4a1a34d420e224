mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -k "watchdog or teacher_embedding or configs2" > gpurun_out/r41_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r41_pytest.log
N=32 timeout 1500 python tools/dbg/stream_outlier_hunt.py > gpurun_out/stream_outlier_hunt_by_queue.log 2>&1

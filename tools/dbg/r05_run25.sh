mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_train_gpu.py -q -m gpu -k "straight_line" > gpurun_out/r25_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r25_pytest.log
bash tools/dbg/ab_opt.sh "CONVDR_EXP_ASYNC_ADAMW=0 CONVDR_EXP_ASYNC_ADAMW=1" 4 > gpurun_out/ab_exp_async_adamw.log 2>&1

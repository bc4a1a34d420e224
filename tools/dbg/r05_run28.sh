mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_encoder_gpu.py -q -m gpu -x > gpurun_out/r28_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r28_pytest.log
bash tools/dbg/ab_opt.sh "CONVDR_OPT_LN_ROWS=0 CONVDR_OPT_LN_ROWS=1" 4 > gpurun_out/ab_ln_rows_fwd.log 2>&1

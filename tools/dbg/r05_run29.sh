mkdir -p gpurun_out
timeout 2000 python -m pytest tests/test_train_gpu.py tests/test_encoder_gpu.py tests/test_parallel_gpu.py -q -m gpu > gpurun_out/r29_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r29_pytest.log
bash tools/dbg/ab_opt.sh "CONVDR_EMB_SUMSQ_MAIN=1 CONVDR_EMB_SUMSQ_MAIN=0" 4 > gpurun_out/ab_emb_sumsq.log 2>&1

mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_parallel_gpu.py -q -m gpu -x > gpurun_out/r27_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r27_pytest.log
bash tools/dbg/ab_opt.sh "CONVDR_HEAD_WGRAD_INLINE=1,CONVDR_KD_DIRECT_BACKWARD=0 CONVDR_HEAD_WGRAD_INLINE=0,CONVDR_KD_DIRECT_BACKWARD=0 CONVDR_HEAD_WGRAD_INLINE=0,CONVDR_KD_DIRECT_BACKWARD=1" 4 > gpurun_out/ab_head_tail.log 2>&1

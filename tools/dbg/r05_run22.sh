mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "lookahead or teacher_embedding or configs2 or layernorm" > gpurun_out/r22_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r22_pytest.log
bash tools/dbg/ab_opt.sh "CONVDR_BENCH_LOOKAHEAD=0 CONVDR_BENCH_LOOKAHEAD=1" 4 > gpurun_out/ab_lookahead.log 2>&1

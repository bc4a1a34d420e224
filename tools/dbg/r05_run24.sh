mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -k "straight_line or 256_tile or configs2" > gpurun_out/r24_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r24_pytest.log
bash tools/dbg/ab_opt.sh "CONVDR_DGRAD_GRID_HINT=0 CONVDR_DGRAD_GRID_HINT=1" 4 > gpurun_out/ab_dgrad_grid_hint.log 2>&1

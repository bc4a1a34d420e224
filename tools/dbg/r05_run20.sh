# LayerNorm backward: straight-line kernel vs the general one, grids (one box, alternating)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -x > gpurun_out/r20_pytest_train.log 2>&1; echo "rc=$?" >> gpurun_out/r20_pytest_train.log
bash tools/dbg/ab_opt.sh "CONVDR_LN_BWD_ROWS=0 CONVDR_LN_BWD_ROWS=1 CONVDR_LN_BWD_ROWS=2 CONVDR_LN_BWD_ROWS=1,CONVDR_LN_BWD_GRID=562 CONVDR_LN_BWD_ROWS=1,CONVDR_LN_BWD_GRID=592 CONVDR_LN_BWD_ROWS=2,CONVDR_LN_BWD_GRID=444" 3 > gpurun_out/ab_ln_bwd_rows.log 2>&1

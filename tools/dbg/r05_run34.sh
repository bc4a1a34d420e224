mkdir -p gpurun_out
timeout 600 python tools/dbg/train_determinism_cfg2.py > gpurun_out/train_determinism_cfg2.log 2>&1

mkdir -p gpurun_out
N=32 timeout 1500 python tools/dbg/stream_outlier_hunt.py > gpurun_out/stream_outlier_hunt_final.log 2>&1

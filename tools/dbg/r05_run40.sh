mkdir -p gpurun_out
N=14 timeout 1500 python tools/dbg/stream_queue_clusters.py > gpurun_out/stream_queue_clusters.log 2>&1

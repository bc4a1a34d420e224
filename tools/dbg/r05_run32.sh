mkdir -p gpurun_out
timeout 600 python tools/dbg/small_batch_graph.py > gpurun_out/small_batch_graph.log 2>&1

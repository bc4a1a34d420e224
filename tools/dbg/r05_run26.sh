mkdir -p gpurun_out
python tools/dbg/kd_dropout_spans.py > gpurun_out/kd_dropout_spans.log 2>&1

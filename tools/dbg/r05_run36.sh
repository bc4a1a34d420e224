mkdir -p gpurun_out
timeout 300 python tools/dbg/power_probe.py kd > gpurun_out/power_probe_kd.log 2>&1

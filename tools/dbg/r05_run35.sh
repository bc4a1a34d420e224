mkdir -p gpurun_out
which rocm-smi amd-smi > gpurun_out/power_probe.log 2>&1
timeout 300 python tools/dbg/power_probe.py >> gpurun_out/power_probe.log 2>&1

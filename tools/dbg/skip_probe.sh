# Sensitivity of the KD step to classes of kernels (timing only: the `make TRACE=1` library drops launches by
# CONVDR_DBG_SKIP bit mask: 1 gelu', 2 LayerNorm backward, 4 forward attention, 8 weight gradients, 16 attention backward,
# 32 bias column sums, 64 forward LayerNorm, 128 gradient-norm partials, 256 transposed-weight refresh).  A class whose removal
# returns MORE than its kernels' busy time is interfering with something (round 6: the refresh, 0.48 ms for 0.17 ms of kernel).
# The loss is printed: a NaN run measures the operands, not the code (profiles/r06_kd_power_probe.txt).
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ${SKIPS:-0 1 2 4 8 16 32 64 128 256}; do
CONVDR_DBG_SKIP=$v CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_trace.so python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[skip=$v] step %.3f ms  loss %s' % (d['ms_per_step'], d.get('final_loss')))"
done; done

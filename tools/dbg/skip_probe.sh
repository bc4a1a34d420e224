# Sensitivity of the KD step to classes of kernels (timing only: the `make TRACE=1` library drops launches by
# CONVDR_DBG_SKIP bit mask: 1 gelu', 2 LayerNorm backward, 4 forward attention, 8 weight gradients, 16 attention backward, 64 forward LayerNorm)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1 2 4 8 16 64; do
CONVDR_DBG_SKIP=$v CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_trace.so python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[skip=$v] step %.3f ms' % d['ms_per_step'])"
done; done

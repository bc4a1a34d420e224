# timing-only switches of k_attention_bwd_fused (debug variant library): attention_bwd ms per step (12 layers, the last one on the old kernels)
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 8 9; do
CONVDR_DBG_ATTF=$v CONVDR_NO_WGRAD_FORK=1 CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_dbgf.so python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[dbg=$v] step %.3f ms | attention_bwd %.3f' % (d['ms_per_step'], k['attention_bwd']['ms_per_step']))"
done

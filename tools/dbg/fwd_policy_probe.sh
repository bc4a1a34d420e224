R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 0 3 2; do
CONVDR_DBG_FWD_POLICY=$v CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_fwdp.so python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[student forward tile policy $v] step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in ('gemm_qkv','gemm_attn_out','gemm_ffn1','gemm_ffn2') if n in k))"
done; done

# A/B of library options on the KD training step (one box, alternating): bash tools/dbg/ab_opt.sh "CONVDR_OPT_ATTN_BWD_FUSED=1 CONVDR_OPT_ATTN_BWD_FUSED=0" [reps]
for rep in $(seq 1 ${2:-3}); do
  for v in $1; do
    envs=$(echo $v | tr ',' ' ')
    env $envs python tools/dbg/opt_bench.py --workload train_kd --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[$v] step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in sorted(k)))"
  done
done

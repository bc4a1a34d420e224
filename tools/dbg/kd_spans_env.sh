# KD step + per-class kernel spans under environment switches: bash tools/dbg/kd_spans_env.sh "A=1 A=0" [reps]
for rep in $(seq 1 ${2:-2}); do
  for v in $1; do
    envs=$(echo $v | tr ',' ' ')
    env $envs python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[$v] step %.3f ms  loss %.5f | ' % (d['ms_per_step'], d.get('final_loss', float('nan'))) + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in k))"
  done
done

for rep in 1 2; do
for p in 0.1 0.0; do
python bench.py --workload train_kd --steps 30 --warmup 8 --no-cpu-baseline --train-dropout $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[p=$p] step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in k))"
done; done

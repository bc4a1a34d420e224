# A/B of library variants on the KD training step (one box): usage: bash tools/dbg/ab_train.sh "base respre" [reps]
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
  for v in $1; do
    if [ "$v" = "base" ]; then lib=$R/convdr_amd/libconvdr_hip.so; else lib=$R/convdr_amd/libconvdr_hip_$v.so; fi
    CONVDR_HIP_LIB=$lib python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
import math
bad = '' if math.isfinite(d.get('final_loss', float('nan'))) else '  !!! final_loss %s: THIS LIBRARY COMPUTES GARBAGE (NaN operands run faster: profiles/r06_kd_power_probe.txt) !!!' % d.get('final_loss')
print('[$v] step %.3f ms  %.0f samples/s  loss %.5f%s | ' % (d['ms_per_step'], d['value'], d.get('final_loss', float('nan')), bad) + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in k))"
  done
done

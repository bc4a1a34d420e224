# A/B of environment switches on the KD training step (one box, alternating): bash tools/dbg/ab_env.sh "A=1,B=0 A=0,B=0" [reps]
for rep in $(seq 1 ${2:-3}); do
  for v in $1; do
    envs=$(echo $v | tr ',' ' ')
    env $envs python bench.py --workload train_kd --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v] step %.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"
  done
done

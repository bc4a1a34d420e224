mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.0f  ms_per_step %.3f  encode %.3f ms  search %.3f ms  clock %.0f MHz  frac_hipevent %.4f' % (d['value'], d['ms_per_step'], d['encode']['ms_per_batch'], d['ip_search']['ms_per_search_incl_fold'], d['roofline'].get('clock_mhz_delivered') or 0, d['roofline'].get('frac_hipevent') or d['roofline']['frac']))"
done > gpurun_out/bench_default_10_processes.log 2>&1

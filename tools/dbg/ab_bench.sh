# A/B of library variants on the default bench line (one box): prints ms_per_step, encode ms, search ms and the scan kernel
# usage: bash tools/dbg/ab_bench.sh "base emit2" [reps]
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
  for v in $1; do
    if [ "$v" = "base" ]; then lib=$R/convdr_amd/libconvdr_hip.so; else lib=$R/convdr_amd/libconvdr_hip_$v.so; fi
    CONVDR_HIP_LIB=$lib python bench.py --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('[$v] step %.2f ms  encode %.2f  search %.3f  scan_emit %.3f  rescore %.3f  pairs/s %.1fG' % (d['ms_per_step'], d['encode']['ms_per_batch'], d['ip_search']['ms_per_search_incl_fold'], k['ip_scan_emit']['avg_ms'], k['ip_rescore']['avg_ms'], d['ip_search']['pairs_per_s_per_gpu'] / 1e9))"
  done
done

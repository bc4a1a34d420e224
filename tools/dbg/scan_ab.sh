# A/B of the scan variants inside ONE box (box-to-box spread is +-3 %)
for rep in 1 2; do
for lib in "" r3v0 r3v1 r3v2; do
  for ne in "X=1" "CONVDR_DBG_SCAN_NOEMIT=1"; do
  if [ -z "$lib" ]; then e="$ne"; else e="$ne CONVDR_DBG_SCAN_R3=1 CONVDR_HIP_LIB=$PWD/convdr_amd/libconvdr_hip_$lib.so"; fi
  env $e python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('${lib:-base} $ne', round(j['kernels']['ip_scan_emit']['avg_ms'],4), j['ip_search']['uncertified_queries'])"
done; done; done

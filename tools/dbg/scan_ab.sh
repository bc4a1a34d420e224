# A/B of the scan's K step inside ONE box (box-to-box spread is +-3 %): R3 (default) vs the two-stage loop,
# emitting and with thresholds of +inf.
for rep in 1 2; do
for e in "X=1" "CONVDR_DBG_SCAN_NO_R3=1" "CONVDR_DBG_SCAN_NOEMIT=1" "CONVDR_DBG_SCAN_NO_R3=1 CONVDR_DBG_SCAN_NOEMIT=1"; do
  env $e python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('$e', round(j['kernels']['ip_scan_emit']['avg_ms'],4), j['ip_search']['uncertified_queries'])"
done; done

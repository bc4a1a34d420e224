cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_ip_search_gpu.py tests/test_parallel_gpu.py tests/test_capi_host_gpu.py -m gpu -q -x > gpurun_out/gpu_search.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_search.log )
python tools/dbg/search_nq_sweep.py > gpurun_out/search_nq_sweep_r05.txt 2>gpurun_out/search_nq_sweep_r05.err
for rep in 1 2 3; do
  for v in "CONVDR_PACK_STREAM=0" "CONVDR_PACK_STREAM=2"; do
    env $v python bench.py --workload train_kd --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v] step %.3f ms' % d['ms_per_step'])" >> gpurun_out/ab_pack_stream.log
  done
done
tail -4 gpurun_out/gpu_search.log; cat gpurun_out/search_nq_sweep_r05.txt; tail -3 gpurun_out/search_nq_sweep_r05.err; cat gpurun_out/ab_pack_stream.log

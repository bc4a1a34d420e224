# round 6, first lease: the new tests, the --gpus 2 launcher rehearsal, today's baseline of the KD step
mkdir -p gpurun_out/r06a
python -m pytest tests/test_ip_search_gpu.py -q -x -k "top_n_above or bit_exact" 2>&1 | tail -5 > gpurun_out/r06a/t_search.txt
python -m pytest tests/test_train_gpu.py -q -x -k "dpr_checkpoint or watchdog or round_trips" 2>&1 | tail -5 > gpurun_out/r06a/t_train.txt
CONVDR_BENCH_SHARE_GPU=1 CONVDR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r06a/rehearsal_gpus2_encode_search.json 2> gpurun_out/r06a/rehearsal_gpus2_encode_search.err
CONVDR_BENCH_SHARE_GPU=1 CONVDR_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload train_kd --steps 5 --warmup 2 > gpurun_out/r06a/rehearsal_gpus2_train_kd.json 2> gpurun_out/r06a/rehearsal_gpus2_train_kd.err
WORLD_SIZE=1 python bench.py --gpus 2 --steps 1 > gpurun_out/r06a/mismatch.out 2>&1; echo "mismatch rc=$?" >> gpurun_out/r06a/mismatch.out
for i in 1 2; do python bench.py --workload train_kd --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[HEAD] step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in k))"; done > gpurun_out/r06a/kd_baseline.txt
cat gpurun_out/r06a/t_search.txt gpurun_out/r06a/t_train.txt gpurun_out/r06a/mismatch.out gpurun_out/r06a/kd_baseline.txt
tail -c 600 gpurun_out/r06a/rehearsal_gpus2_encode_search.json; tail -c 400 gpurun_out/r06a/rehearsal_gpus2_encode_search.err

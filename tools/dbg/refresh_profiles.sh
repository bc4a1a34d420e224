set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r01f/pytest_gpu.log 2>&1; tail -3 gpurun_out/r01f/pytest_gpu.log
python bench.py 2>gpurun_out/r01f/bench.err | tail -1 > gpurun_out/r01f/bench_default.json
python bench.py --workload train_kd 2>gpurun_out/r01f/bench_kd.err | tail -1 > gpurun_out/r01f/bench_train_kd.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/r01f/bench_tr.err | tail -1 > gpurun_out/r01f/bench_torchrun1.json
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01f/prof -o r01f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01f/prof_kd -o r01f_kd -- python3 $R/bench.py --workload train_kd --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r01f/prof_kd.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01f/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01f/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/pmc_write.log 2>&1
cd $R
find gpurun_out/r01f -name "*.db" | head; du -sh gpurun_out/r01f
python tools/rocpd_summary.py $(find gpurun_out/r01f/prof -name "*.db" | head -1) > gpurun_out/r01f/bench_default.kernel_stats.txt
python tools/rocpd_summary.py $(find gpurun_out/r01f/prof_kd -name "*.db" | head -1) > gpurun_out/r01f/train_kd.kernel_stats.txt
python tools/pmc_summary.py gpurun_out/r01f/pmc_fetch gpurun_out/r01f/pmc_write > gpurun_out/r01f/pmc_hbm_traffic.json
# keep the merge under the size cap: drop raw traces
find gpurun_out/r01f -name "*.db" -delete; find gpurun_out/r01f -name "*counter_collection.csv" -delete
du -sh gpurun_out/r01f

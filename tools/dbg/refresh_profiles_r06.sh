# Round-6 evidence run on the GPU box: full GPU test suite, smoke, the default bench line (with its live rocprofv3 passes), the
# train_kd line, rocprofv3 kernel stats of both workloads, PMC passes (HBM traffic; MFMA / LDS utilisation), the training
# timeline, the N > 1 rehearsals with their `comm` blocks, the small-batch search sweep.  Results under gpurun_out/$TAG.
set -x
TAG=${1:-r06f}
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
cp gpurun_out/margins.json $O/margins.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json
python bench.py --workload train_kd 2>$O/bench_kd.err | tail -1 > $O/bench_train_kd.json
CONVDR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>$O/bench_tr.err | tail -1 > $O/bench_torchrun1_forced_dist.json
CONVDR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 6 --warmup 2 --workload train_kd 2>$O/bench_tr_kd.err | tail -1 > $O/bench_torchrun1_forced_dist_train_kd.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python3 $R/bench.py --steps 8 --warmup 4 --passages 65536 --queries 64 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_kd -o kd -- python3 $R/bench.py --workload train_kd --steps 5 --warmup 2 > $O/prof_kd.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc_mfma_kd -- python3 $R/bench.py --workload train_kd --steps 2 --warmup 1 > $O/pmc_mfma_kd.log 2>&1
cd $R
# (the same command, step count and warm-up exclusion as bench.py's live_kernel_trace: the committed file reproduces roofline.frac)
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) --skip-fraction 0.3333 > $O/bench_default.kernel_stats.txt
python tools/rocpd_summary.py $(find $O/prof_kd -name "*.db" | head -1) > $O/train_kd.kernel_stats.txt
python tools/train_timeline.py $(find $O/prof_kd -name "*.db" | head -1) > $O/train_kd.timeline.txt
python tools/train_timeline.py $(find $O/prof_kd -name "*.db" | head -1) --dispatches > $O/train_kd.dispatches.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_traffic.json
python tools/pmc_summary.py $O/pmc_mfma $O/pmc_lds > $O/pmc_mfma_lds.json
python tools/pmc_summary.py $O/pmc_mfma_kd > $O/pmc_mfma_train_kd.json
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.csv" -size +1M -delete
du -sh $O
python tools/pmc_table.py $O/pmc_mfma_lds.json > $O/pmc_mfma_lds.summary.txt
python tools/pmc_table.py $O/pmc_mfma_train_kd.json > $O/pmc_mfma_train_kd.summary.txt
( export CONVDR_BENCH_SHARE_GPU=1 CONVDR_BENCH_BACKEND=gloo
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 4 --warmup 1 --passages 300000 --no-cpu-baseline 2>$O/reh2_es.err | tail -1 > $O/rehearsal_2ranks_1gpu_gloo_encode_search.json
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --steps 4 --warmup 1 --workload train_kd 2>$O/reh2_kd.err | tail -1 > $O/rehearsal_2ranks_1gpu_gloo_train_kd.json
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29573 bench.py --gpus 8 --steps 3 --warmup 1 --passages 200000 --no-cpu-baseline 2>$O/reh8_es.err | tail -1 > $O/rehearsal_8ranks_1gpu_gloo_encode_search.json
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29574 bench.py --gpus 8 --steps 3 --warmup 1 --workload train_kd 2>$O/reh8_kd.err | tail -1 > $O/rehearsal_8ranks_1gpu_gloo_train_kd.json )
( export CONVDR_BENCH_SHARE_GPU=1 CONVDR_BENCH_BACKEND=gloo
  timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --passages 300000 --no-cpu-baseline --no-extras 2>$O/launch2_es.err | tail -1 > $O/rehearsal_gpus2_launcher_encode_search.json
  timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --workload train_kd 2>$O/launch2_kd.err | tail -1 > $O/rehearsal_gpus2_launcher_train_kd.json )
NQS=50,100,250,479,1000 python tools/dbg/search_nq_sweep.py 2>/dev/null > $O/search_nq_sweep.txt
python tools/dbg/block_load_sweep.py 2>/dev/null > $O/block_load_sweep.txt
ls -la $O | head -60

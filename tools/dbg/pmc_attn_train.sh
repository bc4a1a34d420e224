# PMC passes over the training step, summarised for the attention kernels (where does a wave's time go?)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/bench.py --workload train_kd --steps 2 --warmup 1 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/b -- python3 $R/bench.py --workload train_kd --steps 2 --warmup 1 > $O/b.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/c -- python3 $R/bench.py --workload train_kd --steps 2 --warmup 1 > $O/c.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/d -- python3 $R/bench.py --workload train_kd --steps 2 --warmup 1 > $O/d.log 2>&1
cd $R
python tools/pmc_summary.py $O/a $O/b $O/c $O/d > $O/pmc_attn.json
find $O -name "*counter_collection.csv" -delete
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_attn/pmc_attn.json"))
for k,v in d.items():
    if "attention" in k or "layernorm_bwd" in k or "k_gemm<0, convdr::TileCfg<2, 2" in k:
        print(k[:60], {a:(round(b) if isinstance(b,float) else b) for a,b in v.items()})
PY

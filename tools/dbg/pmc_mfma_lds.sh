set -x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_lds -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_lds.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_mfma gpurun_out/pmc_lds > gpurun_out/pmc_mfma_lds.json
find gpurun_out/pmc_mfma gpurun_out/pmc_lds -name "*counter_collection.csv" -delete
tail -3 gpurun_out/pmc_mfma.log | cut -c1-300

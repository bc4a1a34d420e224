mkdir -p gpurun_out
bash tools/dbg/ab_opt.sh "CONVDR_EXP_STALE_PACKT=0 CONVDR_EXP_STALE_PACKT=1" 4 > gpurun_out/ab_exp_stale_packt.log 2>&1

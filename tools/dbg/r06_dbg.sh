mkdir -p gpurun_out/r06p
python tools/train_ab.py --reps 3 f1:attn_bwd_fused=1 kp:attn_bwd_fused=2 2>&1 | grep -v "^Using" > gpurun_out/r06p/ab_kp.txt
tail -9 gpurun_out/r06p/ab_kp.txt

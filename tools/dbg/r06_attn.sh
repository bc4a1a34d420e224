# round 6: attention diet (mask bits from the forward, real ragged branch, scale folded): parity + A/B against the r05 tree
mkdir -p gpurun_out/r06b
python -m pytest tests/test_train_gpu.py -q -x -k "watchdog" 2>&1 | tail -40 > gpurun_out/r06b/t_watchdog.txt
python -m pytest tests/test_train_gpu.py -q -x -k "attention or dropout or encoder_backward or configs2 or replay or reference_run" 2>&1 | tail -15 > gpurun_out/r06b/t_attn.txt
bash tools/ab_worktrees.sh "r05 HEAD" 3 train > gpurun_out/r06b/ab_train.txt 2>&1
cat gpurun_out/r06b/t_watchdog.txt gpurun_out/r06b/t_attn.txt gpurun_out/r06b/ab_train.txt

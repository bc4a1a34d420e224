# The driver's round-end check, N times in a row on one box (fresh process each): python -m pytest tests -x -q -m gpu
mkdir -p gpurun_out/soak
for i in $(seq 1 ${1:-3}); do
  python -m pytest tests -x -q -m gpu > gpurun_out/soak/run$i.log 2>&1
  grep -E "passed|failed" gpurun_out/soak/run$i.log | tail -1 | sed "s/^/[run $i] /"
  if grep -q "failed" gpurun_out/soak/run$i.log; then grep -B 40 "short test summary" gpurun_out/soak/run$i.log | tail -60; fi
done | tee gpurun_out/soak/soak.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/soak/soak.txt

# The driver's round-end check, N times in a row on one box (fresh process each): python -m pytest tests -x -q -m gpu
mkdir -p gpurun_out/soak
for i in $(seq 1 ${1:-3}); do
  python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | sed "s/^/[run $i] /"
done | tee gpurun_out/soak/soak.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/soak/soak.txt

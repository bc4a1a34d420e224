mkdir -p gpurun_out
python bench.py 2>gpurun_out/r37_bench.err | tail -1 > gpurun_out/r37_bench_default.json

# Upper bound of what folding the bias column sums / the gradient-norm partials into other kernels could return to the KD step
# (review r05 1c): the `make TRACE=1` library replaces the launches by memsets of their outputs (CONVDR_DBG_SKIP 32 / 128; the
# gradients of the biases are then zero and the clip sees a partial norm: finite, so no data-dependent power artefact).
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 0 32 128 160; do
CONVDR_DBG_SKIP=$v CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_trace.so python bench.py --workload train_kd --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[skip=$v] step %.3f ms  loss %s' % (d['ms_per_step'], d.get('final_loss', d.get('config', {}).get('final_loss'))))"
done; done

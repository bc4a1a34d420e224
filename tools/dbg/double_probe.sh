# What each class of launches costs the KD step (marginal, with the product's own operands): the `make TRACE=1` library launches a class
# of IDEMPOTENT kernels twice (CONVDR_DBG_DOUBLE bit mask: 1 forward GEMMs (student + teacher), 2 data-gradient GEMMs, 4 weight
# gradients, 8 forward attention, 16 attention backward, 32 forward LayerNorm, 64 LayerNorm backward, 128 bias column sums + reductions,
# 256 gradient-norm partials; CONVDR_DBG_SKIP=512: the transposed-weight refresh twice).  The loss must not change.
R=$GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_trace.so python bench.py --workload train_kd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('[$tag] step %.3f ms  loss %.5f | ' % (d['ms_per_step'], d.get('final_loss', float('nan'))) + ' '.join('%s %.2f' % (n.replace('gemm_', ''), k[n]['ms_per_step']) for n in k))"; }
for rep in 1 2; do
  run base X=0
  for v in 1 2 4 8 16 32 64 128 256; do run double=$v CONVDR_DBG_DOUBLE=$v; done
  run double=refresh CONVDR_DBG_SKIP=512
done

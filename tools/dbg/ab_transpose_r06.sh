# The transposed-weight refresh (one 20,880-workgroup launch per step on a side stream, under the forward):
#   ldstr = the round 2-5 kernel (a 16 KB LDS tile per workgroup), base = the LDS-free kernel,
#   skip  = without the launch (TRACE library, CONVDR_DBG_SKIP=256: stale transposed weights, timing bound only)
# (how the variant was built: `git stash; make -C convdr_amd/csrc VARIANT=ldstr; git stash pop` with the LDS-free kernel in the
#  working tree -- i.e. libconvdr_hip_ldstr.so is the library of commit 3c9f617)
R=$GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --workload train_kd --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$tag] step %.3f ms  loss %.5f' % (d['ms_per_step'], d.get('final_loss', float('nan'))))"; }
for rep in 1 2 3 4; do
  run ldstr CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_ldstr.so
  run base CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip.so
  run skip CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_trace.so CONVDR_DBG_SKIP=256
done

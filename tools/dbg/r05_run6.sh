cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/proto/bin/w16_proto 768 > gpurun_out/w16_ilv2_768.log 2>&1
bash tools/dbg/ab_libs.sh "base fragblock" 3 > gpurun_out/ab_fragilv_fwd.log 2>&1
bash tools/dbg/ab_train.sh "base fragblock" 2 > gpurun_out/ab_fragilv_train.log 2>&1
for v in base fragblock; do
  if [ "$v" = "base" ]; then lib=$GRAFT_REPO_ROOT/convdr_amd/libconvdr_hip.so; else lib=$GRAFT_REPO_ROOT/convdr_amd/libconvdr_hip_$v.so; fi
  CONVDR_HIP_LIB=$lib NQS=100,1000 python tools/dbg/search_nq_sweep.py 2>/dev/null | sed "s|^|[$v] |" >> gpurun_out/ab_fragilv_scan.log
done
( timeout 1500 python -m pytest tests -m gpu -q -x -k "not 38m and not one_million and not configs4" > gpurun_out/gpu_most.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_most.log )
tail -17 gpurun_out/w16_ilv2_768.log; cat gpurun_out/ab_fragilv_fwd.log gpurun_out/ab_fragilv_train.log gpurun_out/ab_fragilv_scan.log; tail -4 gpurun_out/gpu_most.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/proto/bin/w16_proto 768 > gpurun_out/w16_ilv_768.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q -k "configs2 and trained" > gpurun_out/gpu_cfg2t.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_cfg2t.log )
bash tools/dbg/ab_opt.sh "CONVDR_WGRAD_TILE128=0 CONVDR_WGRAD_TILE128=1" 3 > gpurun_out/ab_wgrad128.log 2>&1
bash tools/dbg/r05_raster_ab.sh > gpurun_out/raster_ab.log 2>&1
tail -14 gpurun_out/w16_ilv_768.log; grep -E "1-cos: hip|passed|failed|Error" gpurun_out/gpu_cfg2t.log | cut -c1-200; cat gpurun_out/ab_wgrad128.log; cat gpurun_out/raster/summary.txt

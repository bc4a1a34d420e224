mkdir -p gpurun_out/r06j
R=$GRAFT_REPO_ROOT
run() { CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip$1.so CONVDR_OPTS=$2 python tools/dbg/kd_power.py 250 2>&1 | grep "^\["; }
{
run "" ""
run _w1 ""
run _w2 ""
run _w3 ""
run _w4 ""
run "" ""
run _w2 ""
run _w3 ""
} > gpurun_out/r06j/kd_power.txt
for v in w2 w3; do echo "=== $v"; CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_$v.so python tools/dbg/attn_fused_vs_split.py 2>&1 | grep -v "^Using\|amdgpu.ids"; done > gpurun_out/r06j/fused_vs_split.txt
cat gpurun_out/r06j/kd_power.txt gpurun_out/r06j/fused_vs_split.txt

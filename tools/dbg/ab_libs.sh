# A/B of library variants inside ONE box: tools/enc_kernels.py per variant, interleaved, REPS rounds
# usage: bash tools/dbg/ab_libs.sh "base prio sprio" [reps]
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
  for v in $1; do
    if [ "$v" = "base" ]; then lib=$R/convdr_amd/libconvdr_hip.so; else lib=$R/convdr_amd/libconvdr_hip_$v.so; fi
    CONVDR_HIP_LIB=$lib python tools/enc_kernels.py 2>/dev/null | grep total | sed "s|^\[[^]]*\]|[$v]|"
  done
done

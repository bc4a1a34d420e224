# What bounds the K = 768 forward GEMMs and the scan: timing-only runs of the TRACE library (results are garbage) with
#   (i)   the epilogue skipped             CONVDR_DBG_SKIP_EPI=1 / CONVDR_DBG_SCAN_NOEMIT=1
#   (ii)  the tile prologue "pre-landed"   CONVDR_DBG_PRELANDED=1   (a tile's first K chunks are issued but never waited for)
#   (iii) both
# against the same library without knobs, interleaved inside one box.  Activation buffers are pre-filled with random finite
# data (CONVDR_FILL_WS=1) so that skipped stores do not leave zeros behind (data-dependent clock).
#   usage: bash tools/dbg/ceiling.sh [reps]      (needs make -C convdr_amd/csrc TRACE=1)
R=$GRAFT_REPO_ROOT
export CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_trace.so
export CONVDR_FILL_WS=1
for rep in $(seq 1 ${1:-2}); do
  for e in "X=1" "CONVDR_DBG_SKIP_EPI=1" "CONVDR_DBG_PRELANDED=1" "CONVDR_DBG_SKIP_EPI=1 CONVDR_DBG_PRELANDED=1" "CONVDR_DBG_SAME_TILE=1" "CONVDR_DBG_SAME_TILE=1 CONVDR_DBG_SKIP_EPI=1 CONVDR_DBG_PRELANDED=1"; do
    env $e python tools/enc_kernels.py 2>/dev/null | grep total | sed "s|^\[[^]]*\]|[$e]|"
  done
  for e in "X=1" "CONVDR_DBG_SCAN_NOEMIT=1" "CONVDR_DBG_PRELANDED=1" "CONVDR_DBG_SCAN_NOEMIT=1 CONVDR_DBG_PRELANDED=1"; do
    env $e python tools/dbg/scan_ceiling.py 2>/dev/null | tail -1
  done
done

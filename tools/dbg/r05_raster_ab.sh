# Round 5, item 2a: weight-set-aware raster (panel walk of the feature tiles) A/B on the headline GEMMs: time (interleaved) and
# FETCH_SIZE / WRITE_SIZE / L2 hit rate per variant.  Libraries: make VARIANT=panelN EXTRA=-DCONVDR_GEMM_PANEL=N
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/raster
mkdir -p $O
cd $R
bash tools/dbg/ab_libs.sh "base panel6 panel4 panel3" 3 > $O/time.txt 2>&1
for v in base panel6 panel4; do
  if [ "$v" = "base" ]; then export CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip.so; else export CONVDR_HIP_LIB=$R/convdr_amd/libconvdr_hip_$v.so; fi
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- python3 $R/tools/enc_kernels.py > $O/fetch_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$v -- python3 $R/tools/enc_kernels.py > $O/write_$v.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2_$v -- python3 $R/tools/enc_kernels.py > $O/l2_$v.log 2>&1
  cd $R
  python tools/pmc_summary.py $O/fetch_$v $O/write_$v $O/l2_$v > $O/pmc_$v.json
  rm -rf $O/fetch_$v $O/write_$v $O/l2_$v
done
unset CONVDR_HIP_LIB
python - <<'PY' > $O/summary.txt
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "raster")
print(open(os.path.join(O, "time.txt")).read())
for v in ("base", "panel6", "panel4"):
    d = json.load(open(os.path.join(O, "pmc_%s.json" % v)))
    for k, c in d.items():
        if "k_gemm<8" in k or "k_gemm<3" in k or "k_gemm_resid_ln" in k:
            f, w = c.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9
            h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
            print("%-7s %-70s FETCH (x2) %.2f GB  WRITE %.2f GB  L2 hit %.3f  (%d dispatches)" % (v, k[:70], f, w, h / max(1.0, h + m), c["dispatches"]))
PY
cat $O/summary.txt

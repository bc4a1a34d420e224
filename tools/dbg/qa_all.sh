# Fuzz + stress pass over the product on the GPU box (tails only): bash tools/dbg/qa_all.sh > gpurun_out/qa.txt
cd $GRAFT_REPO_ROOT
for t in train_fuzz search_fuzz search_fuzz2 edge_probe model_surface_fuzz forward_stress train_stress search_stress train_determinism train_determinism_cfg2; do
  echo "=== $t"
  timeout 900 python tools/dbg/$t.py 2>&1 | grep -v "Using mean" | tail -6
done

# Same-box, interleaved A/B of whole bench.py runs across source trees (git worktrees built under .ab/, which travels to the
# GPU box but is git-ignored):   bash tools/ab_worktrees.sh "r02 r03 HEAD" [reps] [train|encode|both]
# r02 / r03 = .ab/r02, .ab/r03 (git worktree add .ab/r02 be9507d; make -C .ab/r02/convdr_amd/csrc), HEAD = the working tree.
R=$GRAFT_REPO_ROOT
what=${3:-both}
for rep in $(seq 1 ${2:-3}); do
  for t in $1; do
    if [ "$t" = "HEAD" ]; then dir=$R; else dir=$R/.ab/$t; fi
    if [ "$what" != "encode" ]; then
      (cd $dir && python bench.py --workload train_kd --steps 20 --warmup 5 2>/dev/null) | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$t] train_kd  %.3f ms/step  %.0f samples/s' % (d['ms_per_step'], d['value']))"
    fi
    if [ "$what" != "train" ]; then
      (cd $dir && python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null) | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('[$t] encode+search  %.3f ms/step  %.0f passages/s | encode %.2f ms  search %.3f ms | ' % (d['ms_per_step'], d['value'], d['encode']['ms_per_batch'], d['ip_search']['ms_per_search_incl_fold']) + ' '.join('%s %.3f' % (n.replace('gemm_', ''), k[n]['avg_ms']) for n in ('gemm_qkv', 'gemm_attn_out', 'gemm_ffn1', 'gemm_ffn2', 'attention', 'ip_scan_emit') if n in k))"
    fi
  done
done

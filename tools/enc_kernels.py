"""Per-kernel timing of the encoder forward (2048 x 128-token passages, roberta-base shape): prints one line per
profiled span.  Experiment driver for the GEMM knobs (CONVDR_DBG_* environment variables); not part of the product."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from convdr_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    steps = 6
    model = bench.random_rdot_model(0).cuda().eval()
    for kv in os.environ.get("CONVDR_OPTIONS", "").split(","):   # e.g. CONVDR_OPTIONS=ln_stagger=2
        if "=" in kv:
            k, v = kv.split("=")
            _lib.check(_lib.lib().convdr_set_option(k.encode(), int(v)), "convdr_set_option")
    import numpy as np
    ids = bench.synthetic_tokens(B, L, 0, "cuda")
    lens = np.full(B, L, np.int32)
    tower, head = model.roberta, (model.embeddingHead, model.norm)
    run = lambda: tower.embed(ids, None, head=head, seq_lens=lens)
    with torch.no_grad():
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        if os.environ.get("CONVDR_FILL_WS"):
            # timing-only builds that skip epilogues leave activation buffers unwritten: give every buffer random finite
            # contents (stale zeros toggle fewer wires and raise the power-managed clock: DESIGN.md, "measurement trap")
            ws = tower._ws
            ws[: ws.numel() // 2 * 2].view(torch.bfloat16).normal_(0.0, 1.0)
            torch.cuda.synchronize()
        _lib.lib().convdr_prof_enable(1)
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(steps):
            run()
        t1.record()
        torch.cuda.synchronize()
    spans = {}
    for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn1", "gemm_ffn2", "attention", "layernorm"):
        ms, cnt = _lib.prof_collect(k)
        if cnt:
            spans[k] = (ms, cnt)
    tot = t0.elapsed_time(t1) / steps
    tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("CONVDR_"))
    out = ["total %.2f ms (%.0f passages/s)" % (tot, B / tot * 1e3)]
    for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn1", "gemm_ffn2", "attention", "layernorm"):
        if k in spans:
            out.append("%s %.2f" % (k, spans[k][0] / steps))
    print("[%s] %s" % (tag, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()

"""Calibration only (not product, not a dependency): what the vendor library's bf16 GEMM reaches on the encoder's
shapes on this GPU, for judging how far k_gemm is from the practical ceiling."""
import torch
torch.manual_seed(0)
shapes = [("ffn1", 262144, 3072, 768), ("qkv", 262144, 2304, 768), ("ffn2", 262144, 768, 3072), ("attn_out", 262144, 768, 768),
          ("scan", 1000448, 1024, 768)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        C = A @ W.T
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        C = A @ W.T
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    print("%-9s M=%d N=%d K=%d  %.3f ms  %.0f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del A, W, C

// The encoder and the training step through the C ABI from a host with no Python and no torch in it: plain HIP runtime
// allocations + include/convdr_hip.h.  What a C / C++ (or cgo / JNI) host would do in place of
//   /root/reference/model/models.py:140-148            (RobertaDot_NLL_LN.query_emb / body_emb)
//   /root/reference/drivers/run_convdr_train.py:109-191 (forward, MSE, backward, clip, AdamW)
// It exercises the whole weight-packing contract of the header from the outside: one flat fp32 parameter arena in the order
// the reference's state_dict names give (q / k / v adjacent), ONE cast to the bf16 copy, K-slice-major copies
// (convdr_pack_kslice), the batched transposed copies (convdr_pack_transposed), HOST arrays of device pointers
// (convdr_layer_weights / _t / _grads), workspace sizing, the status word, cu_seqlens / seq_lens.
//
//   encoder_host <in.bin> <out.bin>
// in.bin  (little endian; written by tests/test_capi_host_gpu.py): "CVDRHOST", int32 hidden heads layers intermediate vocab
//         max_pos pad_idx out_dim B L, float ln_eps head_ln_eps lr beta1 beta2 adam_eps weight_decay max_grad_norm,
//         fp32 parameters (arena order, below), int64 ids [B, L], int64 mask [B, L], fp32 teacher embeddings [B, out_dim]
// out.bin: fp32 inference embeddings [B, E], fp32 training-forward embeddings [B, E], fp32 loss, fp32 grad norm,
//         int32 status words (inference, training), fp32 gradients [n] (after the clip), fp32 parameters after the step [n]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/convdr_hip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
  } while (0)
#define CV(x)                                                                   \
  do {                                                                          \
    if ((x) != 0) { std::printf("convdr error: %s (%s:%d)\n", convdr_last_error(), __FILE__, __LINE__); return 3; } \
  } while (0)

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 3) { std::printf("usage: encoder_host in.bin out.bin\n"); return 64; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::printf("cannot open %s\n", argv[1]); return 65; }
  char magic[8];
  int32_t hd[10];
  float hp[8];
  if (!rd(f, magic, 8) || std::memcmp(magic, "CVDRHOST", 8) || !rd(f, hd, 10) || !rd(f, hp, 8)) { std::printf("bad header\n"); return 66; }
  const int H = hd[0], heads = hd[1], NL = hd[2], I = hd[3], V = hd[4], P = hd[5], pad = hd[6], E = hd[7], B = hd[8], L = hd[9];
  const float ln_eps = hp[0], head_eps = hp[1], lr = hp[2], b1 = hp[3], b2 = hp[4], adam_eps = hp[5], wd = hp[6], max_norm = hp[7];

  // ---- the arena: embeddings (5 tensors), 16 tensors per layer with q, k, v adjacent, the head (4) ----
  struct Off { int64_t word, pos, type, eg, eb, w0, head_w, head_b, head_g, head_bb, n; } o{};
  struct LOff { int64_t wqkv, bqkv, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2; };
  std::vector<LOff> lo(NL);
  int64_t at = 0;
  auto take = [&](int64_t n) { int64_t a = at; at += n; return a; };
  o.word = take((int64_t)V * H); o.pos = take((int64_t)P * H); o.type = take(H); o.eg = take(H); o.eb = take(H);
  o.w0 = at;
  for (int l = 0; l < NL; ++l) {
    lo[l].wqkv = take(3ll * H * H); lo[l].bqkv = take(3 * H); lo[l].wo = take((int64_t)H * H); lo[l].bo = take(H);
    lo[l].g1 = take(H); lo[l].be1 = take(H); lo[l].w1 = take((int64_t)I * H); lo[l].b1 = take(I);
    lo[l].w2 = take((int64_t)H * I); lo[l].b2 = take(H); lo[l].g2 = take(H); lo[l].be2 = take(H);
  }
  o.head_w = take((int64_t)E * H); o.head_b = take(E); o.head_g = take(E); o.head_bb = take(E);
  o.n = at;
  if (o.w0 % 4 || (o.n - o.w0) % 4) { std::printf("arena not 4-aligned\n"); return 67; }
  std::vector<float> hostP(o.n), teacher((size_t)B * E);
  std::vector<int64_t> ids((size_t)B * L), mask((size_t)B * L);
  if (!rd(f, hostP.data(), hostP.size()) || !rd(f, ids.data(), ids.size()) || !rd(f, mask.data(), mask.size()) ||
      !rd(f, teacher.data(), teacher.size())) { std::printf("short file\n"); return 68; }
  std::fclose(f);

  // host side of the packed-row contract: lens, cu (multiples of 8)
  std::vector<int32_t> lens(B), cu(B + 1, 0);
  int max_len = 0;
  for (int b = 0; b < B; ++b) {
    int n = 0;
    for (int l = 0; l < L; ++l) n += (int)mask[(size_t)b * L + l];
    lens[b] = n;
    cu[b + 1] = cu[b] + (n + 7) / 8 * 8;
    if (n > max_len) max_len = n;
  }
  const int64_t rows = cu[B];

  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *dP, *dG, *dM, *dV, *dT, *dOut1, *dOut2, *dLoss, *dDs, *dScratch, *dNorm;
  void *dPb, *dPt, *dKs;
  int64_t *dIds, *dMask;
  int32_t *dLens, *dCu;
  CK(hipMalloc(&dP, o.n * 4)); CK(hipMalloc(&dG, o.n * 4)); CK(hipMalloc(&dM, o.n * 4)); CK(hipMalloc(&dV, o.n * 4));
  CK(hipMalloc(&dPb, (o.n - o.w0) * 2));
  CK(hipMalloc(&dIds, ids.size() * 8)); CK(hipMalloc(&dMask, mask.size() * 8)); CK(hipMalloc(&dLens, B * 4)); CK(hipMalloc(&dCu, (B + 1) * 4));
  CK(hipMalloc(&dT, teacher.size() * 4)); CK(hipMalloc(&dOut1, (size_t)B * E * 4)); CK(hipMalloc(&dOut2, (size_t)B * E * 4));
  CK(hipMalloc(&dLoss, 4)); CK(hipMalloc(&dDs, (size_t)B * E * 4)); CK(hipMalloc(&dScratch, 1024 * 4)); CK(hipMalloc(&dNorm, 8));
  CK(hipMemcpy(dP, hostP.data(), o.n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dIds, ids.data(), ids.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dMask, mask.data(), mask.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dLens, lens.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dCu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dT, teacher.data(), teacher.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dG, 0, o.n * 4)); CK(hipMemset(dM, 0, o.n * 4)); CK(hipMemset(dV, 0, o.n * 4));

  // ---- packing: one cast for every matrix, K-slice-major copies, transposed copies ----
  CV(convdr_cast_f32_bf16(dP + o.w0, dPb, o.n - o.w0, st));
  auto f32 = [&](int64_t off) { return (const float*)(dP + off); };
  auto b16 = [&](int64_t off) { return (const void*)((char*)dPb + 2 * (off - o.w0)); };
  CK(hipMalloc(&dKs, (size_t)NL * ((size_t)H * H + (size_t)H * I) * 2));
  std::vector<convdr_layer_weights> lw(NL);
  size_t ks = 0;
  for (int l = 0; l < NL; ++l) {
    lw[l] = convdr_layer_weights{b16(lo[l].wqkv), f32(lo[l].bqkv), b16(lo[l].wo), f32(lo[l].bo), f32(lo[l].g1), f32(lo[l].be1),
                                 b16(lo[l].w1), f32(lo[l].b1), b16(lo[l].w2), f32(lo[l].b2), f32(lo[l].g2), f32(lo[l].be2), nullptr, nullptr};
    if (H % 32 == 0 && I % 32 == 0) {   // (read only by the fused projection + LayerNorm kernel: H == 768, >= 24576 rows; packed here all the same)
      void* a = (char*)dKs + ks * 2; ks += (size_t)H * H;
      void* b = (char*)dKs + ks * 2; ks += (size_t)H * I;
      CV(convdr_pack_kslice(lw[l].wo, H, H, a, st));
      CV(convdr_pack_kslice(lw[l].w2, H, I, b, st));
      lw[l].wo_ks = a; lw[l].w2_ks = b;
    }
  }
  convdr_encoder_weights w{f32(o.word), f32(o.pos), f32(o.type), f32(o.eg), f32(o.eb), lw.data(), b16(o.head_w), f32(o.head_b),
                           f32(o.head_g), f32(o.head_bb)};
  convdr_encoder_config cfg{0, H, heads, NL, I, V, P, pad, E, ln_eps, head_eps, 0};
  const int nm = 4 * NL + 1;
  std::vector<int64_t> src(nm), dst(nm);
  std::vector<int32_t> tn(nm), tk(nm);
  int64_t tat = 0;
  for (int l = 0; l < NL; ++l) {
    const int64_t so[4] = {lo[l].wqkv, lo[l].wo, lo[l].w1, lo[l].w2};
    const int nn[4] = {3 * H, H, I, H}, kk[4] = {H, H, H, I};
    for (int j = 0; j < 4; ++j) { src[4 * l + j] = so[j]; tn[4 * l + j] = nn[j]; tk[4 * l + j] = kk[j]; dst[4 * l + j] = tat; tat += (int64_t)nn[j] * kk[j]; }
  }
  src[nm - 1] = o.head_w; tn[nm - 1] = E; tk[nm - 1] = H; dst[nm - 1] = tat; tat += (int64_t)E * H;
  CK(hipMalloc(&dPt, tat * 2));
  CV(convdr_pack_transposed(dP, nm, src.data(), tn.data(), tk.data(), dst.data(), dPt, st));
  std::vector<convdr_layer_weights_t> lt(NL);
  for (int l = 0; l < NL; ++l)
    lt[l] = convdr_layer_weights_t{(char*)dPt + 2 * dst[4 * l], (char*)dPt + 2 * dst[4 * l + 1], (char*)dPt + 2 * dst[4 * l + 2], (char*)dPt + 2 * dst[4 * l + 3]};
  const void* head_t = (char*)dPt + 2 * dst[nm - 1];

  // ---- inference forward ----
  const size_t ws1 = convdr_encoder_workspace_bytes(&cfg, rows, B), ws2 = convdr_encoder_train_workspace_bytes(&cfg, rows, B);
  void *dWs1, *dWs2;
  CK(hipMalloc(&dWs1, ws1)); CK(hipMalloc(&dWs2, ws2));
  CV(convdr_encoder_forward(&cfg, &w, dIds, 0, dMask, B, L, dCu, dLens, rows, max_len, dWs1, ws1, dOut1, st));
  // ---- one KD training step: forward, MSE, backward, clip, AdamW ----
  CV(convdr_encoder_train_forward(&cfg, &w, dIds, 0, dMask, B, L, dCu, dLens, rows, max_len, dWs2, ws2, dOut2, nullptr, st));
  CV(convdr_mse_fwd_bwd(dOut2, dT, (int64_t)B * E, 1.0f, dLoss, dDs, st));
  std::vector<convdr_layer_grads> lg(NL);
  auto g = [&](int64_t off) { return dG + off; };
  for (int l = 0; l < NL; ++l)
    lg[l] = convdr_layer_grads{g(lo[l].wqkv), g(lo[l].bqkv), g(lo[l].wo), g(lo[l].bo), g(lo[l].g1), g(lo[l].be1), g(lo[l].w1), g(lo[l].b1),
                               g(lo[l].w2), g(lo[l].b2), g(lo[l].g2), g(lo[l].be2)};
  convdr_encoder_grads gr{g(o.word), g(o.pos), g(o.type), g(o.eg), g(o.eb), lg.data(), g(o.head_w), g(o.head_b), g(o.head_g), g(o.head_bb)};
  CV(convdr_encoder_backward(&cfg, &w, lt.data(), dCu, dLens, head_t, B, rows, max_len, dWs2, ws2, dDs, &gr, nullptr, st));
  // ---- convdr_encoder_backward_fresh on buffers nobody has written: everything behind the embedding prefix [0, w0) holds NaN on
  // entry and must come out bit-identical to the accumulating call on zeros (the tables in front: fp32 atomics, order may differ) ----
  {
    std::vector<float> Gacc(o.n), Gfresh(o.n);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(Gacc.data(), dG, o.n * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dG, 0, o.w0 * 4));
    CK(hipMemset(dG + o.w0, 0xff, (o.n - o.w0) * 4));
    CV(convdr_encoder_train_forward(&cfg, &w, dIds, 0, dMask, B, L, dCu, dLens, rows, max_len, dWs2, ws2, dOut2, nullptr, st));
    CV(convdr_mse_fwd_bwd(dOut2, dT, (int64_t)B * E, 1.0f, dLoss, dDs, st));
    CV(convdr_encoder_backward_fresh(&cfg, &w, lt.data(), dCu, dLens, head_t, B, rows, max_len, dWs2, ws2, dDs, &gr, nullptr, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(Gfresh.data(), dG, o.n * 4, hipMemcpyDeviceToHost));
    const bool same = std::memcmp(Gacc.data() + o.w0, Gfresh.data() + o.w0, (size_t)(o.n - o.w0) * 4) == 0;
    double worst = 0.0;
    for (int64_t i = 0; i < o.w0; ++i) {
      const double d = std::fabs((double)Gacc[i] - (double)Gfresh[i]), tol = 1e-6 + 1e-4 * std::fabs((double)Gacc[i]);
      if (d / tol > worst) worst = d / tol;
    }
    if (!same || !(worst <= 1.0)) { std::printf("backward_fresh: MISMATCH (stored gradients identical %d, tables %.3g of tolerance)\n", (int)same, worst); return 5; }
    std::printf("backward_fresh ok: %lld stored gradients bit-identical to the accumulating call, NaN-poisoned on entry\n", (long long)(o.n - o.w0));
    CK(hipMemcpy(dG, Gacc.data(), o.n * 4, hipMemcpyHostToDevice));   // the step continues with the first call's gradients
  }
  CV(convdr_grad_norm_clip(dG, o.n, max_norm, 1.0f, dScratch, dNorm, 1, st));
  CV(convdr_adamw_step(dP, dG, dM, dV, o.n, lr, b1, b2, adam_eps, wd, 1, 1, nullptr, st));
  CK(hipStreamSynchronize(st));

  // ---- the exchange steps of the ABI from a host without torch: a one-rank RCCL communicator (in-place all-reduce and all-gather are
  // the identity at one rank; what is exercised is the binding: library lookup, unique id, init, ranks, collectives on `st`, destroy) ----
  {
    char id[CONVDR_COMM_ID_BYTES];
    if (convdr_comm_unique_id(id) != 0) {
      std::printf("convdr_comm: skipped (%s)\n", convdr_last_error());
    } else {
      convdr_comm_t comm = nullptr;
      int nr = -1, rk = -1;
      CV(convdr_comm_init(&comm, 1, 0, id));
      CV(convdr_comm_ranks(comm, &nr, &rk));
      float *dA, *dB;
      CK(hipMalloc(&dA, 4096 * 4)); CK(hipMalloc(&dB, 4096 * 4));
      CK(hipMemcpyAsync(dA, dG, 4096 * 4, hipMemcpyDeviceToDevice, st));
      CV(convdr_comm_allreduce_f32(comm, dA, dA, 4096, st));
      CV(convdr_comm_allgather(comm, dA, dB, 4096 * 4, st));
      CK(hipStreamSynchronize(st));
      std::vector<float> a(4096), bb(4096), g0(4096);
      CK(hipMemcpy(a.data(), dA, 4096 * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(bb.data(), dB, 4096 * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(g0.data(), dG, 4096 * 4, hipMemcpyDeviceToHost));
      const bool same = std::memcmp(a.data(), g0.data(), 4096 * 4) == 0 && std::memcmp(bb.data(), g0.data(), 4096 * 4) == 0;
      CV(convdr_comm_destroy(comm));
      if (nr != 1 || rk != 0 || !same) { std::printf("convdr_comm: MISMATCH (ranks %d/%d, identity %d)\n", rk, nr, (int)same); return 4; }
      std::printf("convdr_comm ok: 1-rank communicator, all-reduce + all-gather on the host's own stream\n");
    }
  }

  std::vector<float> out1((size_t)B * E), out2((size_t)B * E), newP(o.n), G(o.n);
  float loss = 0, norm[2] = {0, 0};
  int32_t st1 = 0, st2 = 0;
  CK(hipMemcpy(out1.data(), dOut1, out1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(out2.data(), dOut2, out2.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&loss, dLoss, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(norm, dNorm, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&st1, dWs1, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&st2, dWs2, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(G.data(), dG, o.n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(newP.data(), dP, o.n * 4, hipMemcpyDeviceToHost));
  FILE* fo = std::fopen(argv[2], "wb");
  if (!fo) { std::printf("cannot write %s\n", argv[2]); return 69; }
  std::fwrite(out1.data(), 4, out1.size(), fo);
  std::fwrite(out2.data(), 4, out2.size(), fo);
  std::fwrite(&loss, 4, 1, fo);
  std::fwrite(norm, 4, 1, fo);
  std::fwrite(&st1, 4, 1, fo);
  std::fwrite(&st2, 4, 1, fo);
  std::fwrite(G.data(), 4, G.size(), fo);
  std::fwrite(newP.data(), 4, newP.size(), fo);
  std::fclose(fo);
  std::printf("encoder host ok: %d sequences, %lld packed rows, %lld parameters, loss %.6g, grad norm %.6g (ABI version %d)\n", B,
              (long long)rows, (long long)o.n, loss, norm[0], convdr_version());
  return 0;
}

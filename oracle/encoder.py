"""Oracle: HF-free fp32 restatement of the dual-encoder forward (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, line by line:
  * /root/reference/model/models.py:129-148  RobertaDot_NLL_LN.query_emb/body_emb
      roberta(ids, mask) -> CLS (models.py:43, use_mean=False for every registered
      config: models.py:295,300,307) -> embeddingHead Linear(H,768) -> LayerNorm(768)
  * /root/reference/model/models.py:191-262  HFBertEncoder / BiEncoder
      two BERT towers, embedding = raw last-layer CLS (models.py:210), no head
  * /root/reference/model/models.py:32-35    masked_mean (use_mean=True branch)
  * /root/reference/model/models.py:66-75    NLL.forward triple branch
The encoder arithmetic itself is third-party (transformers==2.3.0,
requirements.txt:1, not vendored).  Restated from its published architecture
(modeling_bert.py / modeling_roberta.py of that release):
  embeddings   = LayerNorm(word[ids] + pos[position_ids] + type[0])
  RoBERTa pos  = cumsum(ids != pad_idx) * (ids != pad_idx) + pad_idx   (pad_idx = 1)
  BERT pos     = 0..L-1
  self-attn    = softmax(Q K^T / sqrt(d) + (1 - mask) * -10000) V
  attn output  = LayerNorm(dense(ctx) + x)
  ffn          = LayerNorm(dense2(gelu_erf(dense1(x))) + x)
Dropout is identity here (eval mode / p = 0); train-mode dropout RNG cannot be
matched (SURVEY.md §7 hard part 4).

Weights are addressed by the reference's state_dict names so the same dict
feeds the reference module, this oracle and ``convdr_amd.model.models``.
"""
import math

import torch
import torch.nn.functional as F


def roberta_position_ids(input_ids, padding_idx=1):
    """HF create_position_ids_from_input_ids: pads (id == padding_idx) get
    padding_idx, every other token counts up from padding_idx + 1.  Note the
    reference pads with id 0 (utils/util.py:146-185), which is NOT the RoBERTa
    pad id, so right-pad positions keep counting (SURVEY.md §7 hard part 7)."""
    m = input_ids.ne(padding_idx).to(torch.int64)
    return torch.cumsum(m, dim=1) * m + padding_idx


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu_erf(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bf16_round(t):
    """Round to bf16 (nearest even) and back, with a straight-through gradient: the value every kernel of the HIP path
    stores when an activation / weight becomes an MFMA operand."""
    return t + (t.detach().to(torch.bfloat16).float() - t.detach())


def gelu_tail_fit(x):
    """The GELU the FFN1 epilogue evaluates (csrc/encoder_kernels.hpp:gelu_tail): x Phi(x) = max(x, 0) - t Q(t), t = min(|x|, 9),
    Q(t) = 0.5 exp2(-(c1 t + .. + c4 t^4)); max |error| vs the erf form 8.8e-6 (tests/test_gelu_fit_cpu.py).  Used by the
    bf16-emulating mode so that the value rounded to bf16 is the one the kernel rounds."""
    t = x.abs().clamp(max=9.0)
    p = t * 0.0041585 - 0.04571999
    p = p * t - 0.46495319
    p = p * t - 1.14955714
    q = torch.exp2(p * t - 1.0)
    return torch.relu(x) - t * q


def _attention_bf16(q, k, v, lens, scale, drop_mask):
    """The attention kernels' arithmetic (csrc/encoder_kernels.hpp:k_attention_fwd, attention_train.hpp): keys in tiles of 64
    from the start of the sequence, online softmax in fp32 (base 2), the UNNORMALISED probabilities of a tile rounded to
    bf16 (after dropout) for the P V product, fp32 accumulators rescaled per tile, one division at the end.
    q, k, v [B, h, L, d] (bf16-valued); lens [B]; drop_mask [B, h, L, L] or None."""
    B, nh, L, d = q.shape
    c = scale * 1.44269504088896341
    m = torch.full((B, nh, L, 1), float("-inf"))
    l = torch.zeros((B, nh, L, 1))
    o = torch.zeros((B, nh, L, d))
    key_ok = torch.arange(L)[None, :] < torch.as_tensor(lens)[:, None]          # [B, L]
    for k0 in range(0, L, 64):
        k1 = min(L, k0 + 64)
        live = key_ok[:, k0:k1].any(dim=1)                                         # sequences that reach this tile
        if not bool(live.any()):
            break
        s = q @ k[:, :, k0:k1].transpose(-1, -2)
        s = s.masked_fill(~key_ok[:, None, None, k0:k1], float("-inf"))
        mx = s.max(dim=-1, keepdim=True).values
        mn = torch.maximum(m, mx)
        mn_safe = torch.where(torch.isinf(mn), torch.zeros_like(mn), mn)          # (a finished sequence's rows: no-op tile)
        alpha = torch.exp2((m - mn_safe) * c)
        alpha = torch.where(torch.isinf(m) & torch.isinf(mn), torch.ones_like(alpha), alpha)
        pr = torch.exp2(s * c - mn_safe * c)
        l = l * alpha + pr.sum(dim=-1, keepdim=True)
        if drop_mask is not None:
            pr = pr * drop_mask[:, :, :, k0:k1]
        o = o * alpha + bf16_round(pr) @ v[:, :, k0:k1]
        m = mn
    return o * (1.0 / l)


def encoder_hidden(sd, prefix, input_ids, attention_mask, *, kind, num_layers,
                   num_heads, eps, return_all=False, dropout=None, emulate_bf16=False, stats=None):
    """Last-layer hidden states [B, L, H] of the BERT/RoBERTa tower stored under
    ``prefix`` (e.g. 'roberta.' or 'question_model.') in state dict ``sd``.
    dropout = (p_hidden, p_attention, seed): train-mode forward with the counter-based masks of the HIP kernels
    (oracle/dropout.py) at the four sites HF applies dropout (embeddings output, attention probabilities,
    attention-output dense, FFN-output dense); None: eval mode.
    emulate_bf16: the SAME forward with every value rounded to bf16 where the HIP path rounds it (weights, the LayerNorm
    outputs that feed GEMMs and residuals, the Q / K / V, context and FFN activations, the attention probabilities of a
    64-key tile) and fp32 everywhere it keeps fp32 (accumulators, pre-LayerNorm sums, softmax statistics): what is left
    between this mode and the kernels is accumulation order.  CPU only, test infrastructure: it exists to show that the
    distance between the kernels and the fp32 oracle IS that rounding (tests/test_train_gpu.py).
    stats: optional dict that receives what the activations looked like (fp32 mode): 'max_abs_x' = the largest magnitude in
    any LayerNorm output of an unmasked token, 'softmax_peak_mean' = the mean over (layer, head, unmasked query) of the
    largest attention probability -- the trained-model-statistics parity test checks that its pathologies are real."""
    rb = bf16_round if emulate_bf16 else (lambda t: t)
    ids = input_ids.long()
    B, L = ids.shape
    g = lambda n: sd[prefix + n].float()
    if kind == "roberta":
        pos = roberta_position_ids(ids, 1)
    elif kind == "bert":
        pos = torch.arange(L).unsqueeze(0).expand(B, L)
    else:
        raise KeyError(kind)
    x = (g("embeddings.word_embeddings.weight")[ids]
         + g("embeddings.position_embeddings.weight")[pos]
         + g("embeddings.token_type_embeddings.weight")[0])
    x = _ln(x, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    H = x.shape[-1]
    if dropout is not None:
        from . import dropout as OD
        p_h, p_a, seed = dropout
        lens = attention_mask.sum(1).numpy()
        hid = lambda site, layer: torch.from_numpy(OD.hidden_mask(seed, site, layer, p_h, lens, L, H))
        x = x * hid(OD.SITE_EMB, 0)
    x = rb(x)
    live = attention_mask.bool()
    peaks = []

    def note(t):
        if stats is not None:
            stats["max_abs_x"] = max(stats.get("max_abs_x", 0.0), float(t.detach()[live].abs().max()))
    note(x)
    d = H // num_heads
    add_mask = (1.0 - attention_mask.float())[:, None, None, :] * -10000.0
    hs = [x]
    for i in range(num_layers):
        p = "encoder.layer.%d." % i
        lin = lambda t, n: F.linear(t, rb(g(p + n + ".weight")), g(p + n + ".bias"))
        q = rb(lin(x, "attention.self.query")).view(B, L, num_heads, d).transpose(1, 2)
        k = rb(lin(x, "attention.self.key")).view(B, L, num_heads, d).transpose(1, 2)
        v = rb(lin(x, "attention.self.value")).view(B, L, num_heads, d).transpose(1, 2)
        if emulate_bf16:
            dm = torch.from_numpy(OD.attention_mask(seed, i, p_a, lens, L, num_heads)) if dropout is not None else None
            ctx = _attention_bf16(q, k, v, attention_mask.sum(1), 1.0 / math.sqrt(d), dm)
            ctx = rb(ctx.transpose(1, 2).reshape(B, L, H))
        else:
            s = q @ k.transpose(-1, -2) / math.sqrt(d) + add_mask
            probs = torch.softmax(s, dim=-1)
            if stats is not None:
                peaks.append(float(probs.detach().max(dim=-1).values.transpose(1, 2)[live].mean()))
            if dropout is not None:
                probs = probs * torch.from_numpy(OD.attention_mask(seed, i, p_a, lens, L, num_heads))
            ctx = (probs @ v).transpose(1, 2).reshape(B, L, H)
        ao = lin(ctx, "attention.output.dense")
        if dropout is not None:
            ao = ao * hid(OD.SITE_ATTN_OUT, i)
        x = rb(_ln(ao + x,
                   g(p + "attention.output.LayerNorm.weight"),
                   g(p + "attention.output.LayerNorm.bias"), eps))
        note(x)
        h = rb(gelu_tail_fit(lin(x, "intermediate.dense"))) if emulate_bf16 else gelu_erf(lin(x, "intermediate.dense"))
        fo = lin(h, "output.dense")
        if dropout is not None:
            fo = fo * hid(OD.SITE_FFN_OUT, i)
        x = rb(_ln(fo + x, g(p + "output.LayerNorm.weight"),
                   g(p + "output.LayerNorm.bias"), eps))
        note(x)
        hs.append(x)
    if stats is not None and peaks:
        stats["softmax_peak_mean"] = sum(peaks) / len(peaks)
    return hs if return_all else x


def masked_mean(t, mask):
    """models.py:32-35."""
    s = torch.sum(t * mask.unsqueeze(-1).float(), dim=1)
    return s / mask.sum(dim=1, keepdim=True).float()


def rdot_nll_emb(sd, input_ids, attention_mask, *, num_layers, num_heads,
                 eps=1e-5, use_mean=False, dropout=None, emulate_bf16=False, stats=None):
    """RobertaDot_NLL_LN.query_emb == body_emb (models.py:140-148)."""
    h = encoder_hidden(sd, "roberta.", input_ids, attention_mask, kind="roberta",
                       num_layers=num_layers, num_heads=num_heads, eps=eps, dropout=dropout, emulate_bf16=emulate_bf16,
                       stats=stats)
    full = masked_mean(h, attention_mask) if use_mean else h[:, 0]     # (already bf16-valued in the emulating mode)
    hw = sd["embeddingHead.weight"].float()
    y = F.linear(full, bf16_round(hw) if emulate_bf16 else hw, sd["embeddingHead.bias"].float())
    return _ln(y, sd["norm.weight"].float(), sd["norm.bias"].float(), 1e-5)  # nn.LayerNorm(768) default eps


def dpr_emb(sd, input_ids, attention_mask, *, tower, num_layers, num_heads, eps=1e-12):
    """BiEncoder.query_emb (tower='question_model') / body_emb ('ctx_model'),
    models.py:227-235: raw CLS of the last layer (models.py:210)."""
    h = encoder_hidden(sd, tower + ".", input_ids, attention_mask, kind="bert",
                       num_layers=num_layers, num_heads=num_heads, eps=eps)
    return h[:, 0]


def pairwise_nll(q, a, b):
    """NLL.forward triple branch models.py:66-75 == BiEncoder.forward :254-262."""
    logits = torch.stack([(q * a).sum(-1), (q * b).sum(-1)], dim=1)
    return (-F.log_softmax(logits, dim=1)[:, 0]).mean()


def rdot_multi_chunk_body_emb(sd, input_ids, attention_mask, *, num_layers, num_heads, base_len=512):
    """RobertaDot_CLF_ANN_NLL_MultiChunk.body_emb (models.py:164-188): [B, n * 512] -> [B, n, 768]."""
    B, full = input_ids.shape
    n = full // base_len
    e = rdot_nll_emb(sd, input_ids.reshape(B * n, base_len), attention_mask.reshape(B * n, base_len),
                     num_layers=num_layers, num_heads=num_heads)
    return e.reshape(B, n, -1)


def multi_chunk_nll(q, a, b, mask_a, mask_b, base_len=512):
    """NLL_MultiChunk.forward triple branch (models.py:92-126): MaxP over chunks, padding chunks biased by -9999."""
    def maxp(embs, mask):
        first = mask.reshape(mask.shape[0], -1, base_len)[:, :, 0]
        s = torch.matmul(q.unsqueeze(1), embs.transpose(1, 2))[:, 0, :] + ((1 - first) * (-9999)).float()
        return s.max(dim=-1).values
    logits = torch.stack([maxp(a, mask_a), maxp(b, mask_b)], dim=1)
    return (-F.log_softmax(logits, dim=1)[:, 0]).mean()

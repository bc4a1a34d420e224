"""Oracle: the counter-based dropout masks of the training kernels (CPU, numpy uint32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference trains with torch's dropout (hidden_dropout_prob = attention_probs_dropout_prob = 0.1 inside HF
BertModel / RobertaModel, switched on by ``model.train()`` at /root/reference/drivers/run_convdr_train.py:107), whose RNG
stream no other implementation can reproduce.  The HIP kernels therefore DEFINE their masks as a pure function of
(seed, site, layer, element) -- convdr_amd/csrc/dropout.hpp -- and this module restates that function bit for bit, so
that the oracle forward / autograd backward can replay exactly the masks a GPU step used ("parity at p > 0 is parity
under the same mask"; the statistical properties -- keep rate, scaling, independence across sites -- are tested too).

Sites: 0 embeddings output (after LayerNorm), 1 attention-output dense, 2 FFN-output dense, 3 attention probabilities
-- the four places HF transformers 2.3.0 applies dropout on this path (modeling_bert.py BertEmbeddings, BertSelfAttention,
BertSelfOutput, BertOutput).
"""
import numpy as np

SITE_EMB, SITE_ATTN_OUT, SITE_FFN_OUT, SITE_ATT_PROBS = 0, 1, 2, 3


def mix32(a):
    """Bob Jenkins' 6-line integer hash on uint32 arrays (wrap-around arithmetic)."""
    a = np.asarray(a, dtype=np.uint32)
    with np.errstate(over="ignore"):
        a = (a + np.uint32(0x7ed55d16)) + (a << np.uint32(12))
        a = (a ^ np.uint32(0xc761c23c)) ^ (a >> np.uint32(19))
        a = (a + np.uint32(0x165667b1)) + (a << np.uint32(5))
        a = (a + np.uint32(0xd3a2646c)) ^ (a << np.uint32(9))
        a = (a + np.uint32(0xfd7046c5)) + (a << np.uint32(3))
        a = (a ^ np.uint32(0xb55a4f09)) ^ (a >> np.uint32(16))
    return a


def site_params(seed, site, layer, p):
    """(key, thresh16, scale) exactly as drop_site() computes them."""
    thresh = min(65535, int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))) if p > 0 else 0
    with np.errstate(over="ignore"):
        k = np.uint32(seed) ^ (np.uint32(site) * np.uint32(0x9E3779B9) + np.uint32(layer) * np.uint32(0x85EBCA6B))
    key = mix32(k)
    scale = np.float32(65536.0) / np.float32(65536 - thresh) if thresh else np.float32(1.0)
    return np.uint32(key), thresh, np.float32(scale)


def _pair_multipliers(pair_index, key, thresh, scale):
    h = mix32(np.asarray(pair_index, np.uint32) ^ key)
    even = np.where((h & np.uint32(0xffff)) >= thresh, scale, np.float32(0)).astype(np.float32)
    odd = np.where((h >> np.uint32(16)) >= thresh, scale, np.float32(0)).astype(np.float32)
    return even, odd


def packed_rows(lens):
    """cu[b]: first packed row of sequence b (every sequence is padded to a multiple of 8 rows), as the kernels lay
    the tokens out (convdr_amd/csrc/encoder_kernels.hpp)."""
    lens = np.asarray(lens, np.int64)
    cu = np.zeros(len(lens) + 1, np.int64)
    np.cumsum((lens + 7) // 8 * 8, out=cu[1:])
    return cu


def hidden_mask(seed, site, layer, p, lens, L, H):
    """float32 [B, L, H] multipliers (0 or 1 / keep) for a hidden-state dropout site; positions l >= len: 1 (unused)."""
    key, thresh, scale = site_params(seed, site, layer, p)
    B = len(lens)
    out = np.ones((B, L, H), np.float32)
    if not thresh:
        return out
    cu = packed_rows(lens)
    cols = np.arange(H // 2, dtype=np.uint32)
    for b in range(B):
        n = int(lens[b])
        rows = (cu[b] + np.arange(n)).astype(np.uint32)
        with np.errstate(over="ignore"):
            pair = rows[:, None] * np.uint32(H // 2) + cols[None, :]
        even, odd = _pair_multipliers(pair, key, thresh, scale)
        out[b, :n, 0::2] = even
        out[b, :n, 1::2] = odd
    return out


def attention_mask(seed, layer, p, lens, L, heads):
    """float32 [B, heads, L, L] multipliers for the attention probabilities (query, key); padding: 1 (unused)."""
    key, thresh, scale = site_params(seed, SITE_ATT_PROBS, layer, p)
    B = len(lens)
    out = np.ones((B, heads, L, L), np.float32)
    if not thresh:
        return out
    cu = packed_rows(lens)
    for b in range(B):
        n = int(lens[b])
        rows = (cu[b] + np.arange(n)).astype(np.uint32)
        kp = np.arange((n + 1) // 2, dtype=np.uint32)
        for h in range(heads):
            with np.errstate(over="ignore"):
                base = (rows * np.uint32(heads) + np.uint32(h)) << np.uint32(9)
                pair = base[:, None] + kp[None, :]
            even, odd = _pair_multipliers(pair, key, thresh, scale)
            m = np.empty((n, 2 * len(kp)), np.float32)
            m[:, 0::2], m[:, 1::2] = even, odd
            out[b, h, :n, :n] = m[:, :n]
    return out

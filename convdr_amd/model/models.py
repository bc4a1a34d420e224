"""Operator surface of the reference's ``model/models.py`` on MI355X.

Same names, same call contract, same parameter (state_dict) names as
/root/reference/model/models.py, so the reference drivers' call sites
(`MSMarcoConfigDict[name].model_class.from_pretrained(...)`, `model(ids, mask)`,
`model(ids, mask, is_query=False)`, `.query_emb`, `.body_emb`, `.named_parameters()`,
`.save_pretrained`, `.resize_token_embeddings`) work unchanged -- but every forward runs the
hand-written gfx950 kernels of libconvdr_hip.so (csrc/encoder*.hip) instead of HuggingFace
``transformers`` modules.  torch.nn here only names and owns the fp32 master parameters.

  EmbeddingMixin            models.py:13-49
  NLL                       models.py:52-75
  RobertaDot_NLL_LN         models.py:129-148   RoBERTa -> CLS -> Linear(H,768) -> LayerNorm(768)
  RobertaDot_NLL_LN_Inference models.py:151-156
  HFBertEncoder / BiEncoder models.py:191-262   two BERT towers, raw CLS
  MSMarcoConfig / MSMarcoConfigDict models.py:275-311
"""
import ctypes as C
import json
import os

import numpy as np
import torch
from torch import nn

from .. import _lib


# --------------------------------------------------------------------------------------------
# configuration (stand-in for transformers.RobertaConfig / BertConfig: only the fields the path reads)
# --------------------------------------------------------------------------------------------
class EncoderConfig:
    model_type = "roberta"
    _defaults = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                     layer_norm_eps=1e-5, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                     hidden_act="gelu", num_labels=2, finetuning_task=None)

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            setattr(self, k, kw.pop(k, v))
        self.extra = kw
        if self.hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' (HF's erf form x * Phi(x); the kernels evaluate a fitted normal "
                                      "tail, max error 8.8e-6) is implemented, got hidden_act=%r" % self.hidden_act)

    @classmethod
    def from_pretrained(cls, path, **kw):
        kw.pop("cache_dir", None)
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.update(kw)
        return cls(**d)

    def to_dict(self):
        d = {k: getattr(self, k) for k in self._defaults}
        d.update(self.extra)
        d["model_type"] = self.model_type
        return d

    def save_pretrained(self, path):
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True, default=str)


class RobertaConfig(EncoderConfig):
    model_type = "roberta"


class BertConfig(EncoderConfig):
    model_type = "bert"
    _defaults = dict(EncoderConfig._defaults, vocab_size=30522, max_position_embeddings=512, type_vocab_size=2,
                     pad_token_id=0, layer_norm_eps=1e-12)


class _TokenizerPlaceholder:
    """The tokenizers stay HuggingFace's (host-side text processing is out of scope);
    resolved lazily so that importing this module never needs vocab files."""

    def __init__(self, hf_name):
        self.hf_name = hf_name

    def from_pretrained(self, *a, **kw):
        import transformers
        return getattr(transformers, self.hf_name).from_pretrained(*a, **kw)


RobertaTokenizer = _TokenizerPlaceholder("RobertaTokenizer")
BertTokenizer = _TokenizerPlaceholder("BertTokenizer")


# --------------------------------------------------------------------------------------------
# parameter containers with HuggingFace's names
# --------------------------------------------------------------------------------------------
class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _SelfAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)


class _DenseLN(nn.Module):
    def __init__(self, n_in, n_out, eps):
        super().__init__()
        self.dense = nn.Linear(n_in, n_out)
        self.LayerNorm = nn.LayerNorm(n_out, eps=eps)


class _Dense(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.dense = nn.Linear(n_in, n_out)


class _Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _SelfAttention(cfg)
        self.output = _DenseLN(cfg.hidden_size, cfg.hidden_size, cfg.layer_norm_eps)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _Attention(cfg)
        self.intermediate = _Dense(cfg.hidden_size, cfg.intermediate_size)
        self.output = _DenseLN(cfg.intermediate_size, cfg.hidden_size, cfg.layer_norm_eps)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


# packed rows from which the inference forward runs the fused projection + LayerNorm kernel (the library's
# "fused_ln_min_rows" option; tests lower both to reach that kernel with small inputs)
KSLICE_MIN_ROWS = 24576


class EncoderTower(nn.Module):
    """Parameters of one BERT/RoBERTa tower under HF's names (embeddings.*, encoder.layer.N.*, pooler.dense.*)
    plus the packed bf16 device copies the kernels read."""

    def __init__(self, cfg, kind, out_dim=0):
        super().__init__()
        self.config = cfg
        self.kind = kind  # "roberta" | "bert"
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.pooler = _Dense(cfg.hidden_size, cfg.hidden_size)  # present in transformers==2.3.0 checkpoints; unused
        self._packed = None
        self._packed_key = None
        self._ws = None

    # ---- packed weights ---------------------------------------------------------------------
    def _version_key(self, extra):
        # (walking the module tree costs 0.2 ms per call at 200 parameters; the flat list is kept and re-validated against
        # the modules' own `_parameters` dicts -- an identity check per entry, no generator machinery -- so replacing ANY
        # Parameter object (resize_token_embeddings, `layer.dense.weight = nn.Parameter(...)`, pruning / LoRA-style
        # re-parametrisation) rebuilds it: ADVICE r2)
        cache = self.__dict__.get("_plist")
        if cache is not None:
            ps, owners = cache
            # (non-None entries only: nn.Linear(bias=False) or register_parameter(name, None) leave None entries behind, which
            #  the list never contained -- counting them rebuilt the list on every call.  A sub-MODULE added after the list was
            #  built is not seen here -- re-walking the tree per forward is the 0.2 ms this cache exists to save; code that
            #  grafts modules onto a tower calls invalidate_packed(), like resize_token_embeddings does)
            live = sum(1 for m in self.__dict__["_pmods"] for p in m._parameters.values() if p is not None)
            if not all(d.get(k) is p for p, (d, k) in zip(ps, owners)) or live != len(ps):
                cache = None
        if cache is None:
            ps, owners, mods = [], [], []
            for mod in self.modules():
                mods.append(mod)
                for k, p in mod._parameters.items():
                    if p is not None:
                        ps.append(p)
                        owners.append((mod._parameters, k))
            self.__dict__["_plist"] = (ps, owners)
            self.__dict__["_pmods"] = mods
        return tuple((p.data_ptr(), p._version) for p in ps) + tuple((p.data_ptr(), p._version) for p in extra)

    def invalidate_packed(self):
        """Forget the packed bf16 copies.  Needed after writes that bypass the version counters (``p.data`` writes such
        as a broadcast into ``p.data``)."""
        self._packed = None
        self._packed_key = None
        self.__dict__.pop("_packed_t", None)
        self.__dict__.pop("_plist", None)

    def fused_update_target(self, P):
        """(bf16 copy, first arena element it covers) when the packed weights of this tower are views of the flat arena
        `P` -- then an optimizer that updates all of P may refresh the copy itself (convdr_adamw_step_packed) and call
        adopt_fused_update() instead of leaving a cast pass to the next forward -- else None."""
        flat = self.__dict__.get("_flat")
        if flat is None or self._packed is None or "_packed_extra" not in self.__dict__ or flat.get("Pb") is None:
            return None
        if flat["P"].data_ptr() != P.data_ptr() or flat["P"].numel() != P.numel():
            return None
        return flat["Pb"], flat["w0"]

    def adopt_fused_update(self):
        """The optimizer has rewritten the whole bf16 copy next to the fp32 weights: the packed structs (pointers into the
        two arenas) are valid for the parameters' new versions; the transposed copies for the backward are not."""
        self._packed_key = self._version_key(self.__dict__["_packed_extra"])
        self.__dict__.pop("_packed_t", None)

    def check_positions(self, max_len):
        """The reference's position-embedding lookup raises IndexError when a sequence needs a position past the table
        (RoBERTa: position = pad_idx + running count of non-pad tokens; BERT: the column index)."""
        need = max_len + (self.config.pad_token_id + 1 if self.kind == "roberta" else 0)
        if need > self.config.max_position_embeddings:
            raise IndexError("a sequence of %d tokens needs position %d of a %d-row position-embedding table"
                             % (max_len, need - 1, self.config.max_position_embeddings))

    @staticmethod
    def _bf16(t):
        t = t.detach().float().contiguous()
        out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        _lib.check(_lib.lib().convdr_cast_f32_bf16(_lib.ptr(t), _lib.ptr(out), t.numel(), _lib.stream_ptr()),
                   "convdr_cast_f32_bf16")
        return out

    def packed(self, head=None):
        """(cfg struct, weights struct, keepalive) for the C ABI; rebuilt when any parameter changed."""
        extra = [] if head is None else [head[0].weight, head[0].bias, head[1].weight, head[1].bias]
        key = self._version_key(extra)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        cfg = self.config
        dev = self.embeddings.word_embeddings.weight.device
        if dev.type != "cuda":
            raise _lib.ConvdrError("the encoder runs on the GPU only (parameters are on %s); call .to('cuda')" % dev)
        flat = getattr(self, "_flat", None)
        if flat is not None and flat["head"] is (None if head is None else head[0]):
            self._packed, self._packed_key = self._packed_from_flat(flat, head), key
            self.__dict__["_packed_extra"] = extra
            return self._packed
        self.__dict__.pop("_packed_extra", None)
        keep = []

        def f32(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()

        def b16(t):
            o = self._bf16(t)
            keep.append(o)
            return o.data_ptr()

        with torch.cuda.device(dev):
            layers = (_lib.LayerWeights * cfg.num_hidden_layers)()
            for i, ly in enumerate(self.encoder.layer):
                s = ly.attention.self
                lw = layers[i]
                lw.wqkv = b16(torch.cat([s.query.weight, s.key.weight, s.value.weight], 0))
                lw.bqkv = f32(torch.cat([s.query.bias, s.key.bias, s.value.bias], 0))
                lw.wo, lw.bo = b16(ly.attention.output.dense.weight), f32(ly.attention.output.dense.bias)
                lw.ln1_g, lw.ln1_b = f32(ly.attention.output.LayerNorm.weight), f32(ly.attention.output.LayerNorm.bias)
                lw.w1, lw.b1 = b16(ly.intermediate.dense.weight), f32(ly.intermediate.dense.bias)
                lw.w2, lw.b2 = b16(ly.output.dense.weight), f32(ly.output.dense.bias)
                lw.ln2_g, lw.ln2_b = f32(ly.output.LayerNorm.weight), f32(ly.output.LayerNorm.bias)
            w = _lib.EncoderWeights()
            e = self.embeddings
            w.word_emb, w.pos_emb = f32(e.word_embeddings.weight), f32(e.position_embeddings.weight)
            w.type_emb = f32(e.token_type_embeddings.weight)
            w.emb_ln_g, w.emb_ln_b = f32(e.LayerNorm.weight), f32(e.LayerNorm.bias)
            w.layers = C.cast(layers, C.POINTER(_lib.LayerWeights))
            out_dim, head_eps = 0, 1e-5
            if head is not None:
                lin, ln = head
                w.head_w, w.head_b = b16(lin.weight), f32(lin.bias)
                w.head_ln_g, w.head_ln_b = f32(ln.weight), f32(ln.bias)
                out_dim, head_eps = lin.out_features, ln.eps
        c = _lib.EncoderConfig(kind=0 if self.kind == "roberta" else 1, hidden=cfg.hidden_size,
                               heads=cfg.num_attention_heads, layers=cfg.num_hidden_layers,
                               intermediate=cfg.intermediate_size, vocab=e.word_embeddings.num_embeddings,
                               max_pos=cfg.max_position_embeddings, pad_idx=cfg.pad_token_id if self.kind == "roberta" else 0,
                               out_dim=out_dim, ln_eps=cfg.layer_norm_eps, head_ln_eps=head_eps,
                               pool_mean=int(bool(getattr(self, "pool_mean", False))))
        keep.append(layers)
        self._packed, self._packed_key = (c, w, keep), key
        return self._packed

    def _packed_from_flat(self, flat, head):
        """Training fast path (train.flatten_parameters): every parameter is a view of ONE fp32 arena laid out so that
        q/k/v weights (and biases) are adjacent, so the packed bf16 copies are a single cast of the arena's weight
        range and every pointer is arena base + offset -- no torch.cat, no per-tensor kernels."""
        cfg = self.config
        P, off = flat["P"], flat["off"]
        w0 = flat["w0"]                     # first element of the per-layer + head range
        Pb = flat.get("Pb")
        if Pb is None:
            Pb = flat["Pb"] = torch.empty(P.numel() - w0, dtype=torch.bfloat16, device=P.device)
        with torch.cuda.device(P.device):
            _lib.check(_lib.lib().convdr_cast_f32_bf16(C.c_void_p(P.data_ptr() + 4 * w0), _lib.ptr(Pb), P.numel() - w0,
                                                       _lib.stream_ptr()), "convdr_cast_f32_bf16")
        f32 = lambda p: P.data_ptr() + 4 * off[id(p)]
        b16 = lambda p: Pb.data_ptr() + 2 * (off[id(p)] - w0)
        layers = (_lib.LayerWeights * cfg.num_hidden_layers)()
        for i, ly in enumerate(self.encoder.layer):
            s = ly.attention.self
            lw = layers[i]
            lw.wqkv, lw.bqkv = b16(s.query.weight), f32(s.query.bias)
            lw.wo, lw.bo = b16(ly.attention.output.dense.weight), f32(ly.attention.output.dense.bias)
            lw.ln1_g, lw.ln1_b = f32(ly.attention.output.LayerNorm.weight), f32(ly.attention.output.LayerNorm.bias)
            lw.w1, lw.b1 = b16(ly.intermediate.dense.weight), f32(ly.intermediate.dense.bias)
            lw.w2, lw.b2 = b16(ly.output.dense.weight), f32(ly.output.dense.bias)
            lw.ln2_g, lw.ln2_b = f32(ly.output.LayerNorm.weight), f32(ly.output.LayerNorm.bias)
        w = _lib.EncoderWeights()
        e = self.embeddings
        w.word_emb, w.pos_emb, w.type_emb = f32(e.word_embeddings.weight), f32(e.position_embeddings.weight), \
            f32(e.token_type_embeddings.weight)
        w.emb_ln_g, w.emb_ln_b = f32(e.LayerNorm.weight), f32(e.LayerNorm.bias)
        w.layers = C.cast(layers, C.POINTER(_lib.LayerWeights))
        out_dim, head_eps = 0, 1e-5
        if head is not None:
            lin, ln = head
            w.head_w, w.head_b = b16(lin.weight), f32(lin.bias)
            w.head_ln_g, w.head_ln_b = f32(ln.weight), f32(ln.bias)
            out_dim, head_eps = lin.out_features, ln.eps
        c = _lib.EncoderConfig(kind=0 if self.kind == "roberta" else 1, hidden=cfg.hidden_size,
                               heads=cfg.num_attention_heads, layers=cfg.num_hidden_layers,
                               intermediate=cfg.intermediate_size, vocab=e.word_embeddings.num_embeddings,
                               max_pos=cfg.max_position_embeddings, pad_idx=cfg.pad_token_id if self.kind == "roberta" else 0,
                               out_dim=out_dim, ln_eps=cfg.layer_norm_eps, head_ln_eps=head_eps,
                               pool_mean=int(bool(getattr(self, "pool_mean", False))))
        return (c, w, [layers, P, Pb])

    def _ensure_kslice(self, c, w, keep, rows, dev):
        """K-slice-major copies of wo / w2 for the fused projection + LayerNorm kernel (convdr_layer_weights.wo_ks):
        built lazily, once per packing, and only for the shapes that kernel serves (the training forward never
        reads them, so a training step does not pay for them)."""
        if c.hidden != 768 or rows < KSLICE_MIN_ROWS or c.layers == 0 or w.layers[0].wo_ks:
            return
        L_ = _lib.lib()
        H, I = c.hidden, c.intermediate
        buf = torch.empty(c.layers * (H * H + H * I), dtype=torch.bfloat16, device=dev)
        o = 0
        for i in range(c.layers):
            lw = w.layers[i]
            for src, k, field in ((lw.wo, H, "wo_ks"), (lw.w2, I, "w2_ks")):
                dst = buf.data_ptr() + 2 * o
                _lib.check(L_.convdr_pack_kslice(C.c_void_p(src), H, k, C.c_void_p(dst), _lib.stream_ptr()), "convdr_pack_kslice")
                setattr(lw, field, dst)
                o += H * k
        keep.append(buf)

    # ---- forward ------------------------------------------------------------------------------
    def embed(self, input_ids, attention_mask, head=None, seq_lens=None):
        """-> fp32 [B, out_dim or H] embeddings (CLS pooling, models.py:43).
        Fast path for the corpus loop: int32 ids, attention_mask=None and host `seq_lens` (right padding)."""
        L_ = _lib.lib()
        if input_ids.device.type != "cuda":
            raise _lib.ConvdrError("encoder inputs must be CUDA tensors (no CPU fallback)")
        ids32 = input_ids.dtype == torch.int32
        ids = input_ids.contiguous() if ids32 else input_ids.long().contiguous()
        mask = None if attention_mask is None else attention_mask.long().contiguous()
        if mask is None and seq_lens is None:
            raise ValueError("attention_mask=None needs seq_lens")
        B, L = ids.shape
        dev = ids.device
        from ..train import _lens_and_check, _pinned_upload, _status_poll, _status_post
        _status_poll(self)          # (a batch flagged by an earlier forward raises here at the latest)
        # one small D2H round trip per batch unless the caller knows the lengths; it also validates the token ids
        seq_lens, lens_host = _lens_and_check(ids, mask, self.embeddings.word_embeddings.num_embeddings, seq_lens)
        if lens_host.min() < 1:
            raise ValueError("every sequence needs at least one unmasked token")
        self.check_positions(int(lens_host.max()))
        cu_host = np.zeros(B + 1, np.int32)
        np.cumsum((lens_host + 7) // 8 * 8, out=cu_host[1:])
        rows, max_len = int(cu_host[-1]), int(lens_host.max())
        cu = _pinned_upload(cu_host, dev)
        with torch.cuda.device(dev):
            c, w, _keep = self.packed(head)
            self._ensure_kslice(c, w, _keep, rows, dev)
            out = torch.empty((B, c.out_dim or c.hidden), dtype=torch.float32, device=dev)
            need = L_.convdr_encoder_workspace_bytes(C.byref(c), rows, B)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
            _lib.check(L_.convdr_encoder_forward(C.byref(c), C.byref(w), _lib.ptr(ids), int(ids32), _lib.ptr(mask), B, L,
                                                 _lib.ptr(cu), _lib.ptr(seq_lens), rows, max_len, _lib.ptr(self._ws),
                                                 self._ws.numel(), _lib.ptr(out), _lib.stream_ptr()),
                       "convdr_encoder_forward")
            _status_post(self, self._ws)
        return out


# --------------------------------------------------------------------------------------------
# checkpoint I/O shared by the model classes (HF directory layout: config.json + pytorch_model.bin)
# --------------------------------------------------------------------------------------------
def _load_state_dict_file(path):
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    return torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")


class _PretrainedMixin:
    @classmethod
    def from_pretrained(cls, path, config=None, from_tf=False, cache_dir=None, **kw):
        if from_tf:
            raise NotImplementedError("TensorFlow checkpoints are not supported")
        if config is None:
            config = cls.config_class.from_pretrained(path)
        model = cls(config, **kw)
        sd = _load_state_dict_file(path)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        real_missing = [k for k in missing if "pooler" not in k and not k.startswith("classifier")]
        if real_missing:
            raise KeyError("checkpoint %s lacks parameters: %s" % (path, real_missing[:8]))
        return model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        self.config.save_pretrained(path)
        torch.save({k: v.detach().cpu() for k, v in self.state_dict().items()}, os.path.join(path, "pytorch_model.bin"))


# --------------------------------------------------------------------------------------------
# the reference's classes
# --------------------------------------------------------------------------------------------
class EmbeddingMixin:
    """models.py:13-49."""

    def __init__(self, model_argobj):
        if model_argobj is None:
            self.use_mean = False
        else:
            self.use_mean = model_argobj.use_mean
        print("Using mean:", self.use_mean)

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding, nn.Conv1d)):
            module.weight.data.normal_(mean=0.0, std=0.02)

    def masked_mean(self, t, mask):
        s = torch.sum(t * mask.unsqueeze(-1).float(), axis=1)
        d = mask.sum(axis=1, keepdim=True).float()
        return s / d

    def masked_mean_or_first(self, emb_all, mask):
        assert isinstance(emb_all, tuple)
        if self.use_mean:
            return self.masked_mean(emb_all[0], mask)
        return emb_all[0][:, 0]

    def query_emb(self, input_ids, attention_mask):
        raise NotImplementedError("Please Implement this method")

    def body_emb(self, input_ids, attention_mask):
        raise NotImplementedError("Please Implement this method")


def _wants_autograd(module):
    """The activation-saving (differentiable) forward runs in train() mode with gradients enabled -- what the reference's
    training loop does (run_convdr_train.py:107).  An eval()-mode call takes the inference kernels even when the caller
    forgot torch.no_grad(): the training forward keeps ~12 layers of activations, which at corpus-encode batch sizes is an
    out-of-memory, not a convenience (set ``module.autograd_in_eval = True`` to differentiate through an eval() model)."""
    if not torch.is_grad_enabled():
        return False
    if not (module.training or getattr(module, "autograd_in_eval", False)):
        return False
    return any(p.requires_grad for p in module.parameters())


def _pairwise_nll(q_embs, a_embs, b_embs):
    """models.py:66-75: logits = [<q, a>, <q, b>], loss = mean(-log_softmax(logits)[:, 0]); one HIP launch for the loss and
    the gradients of all three embeddings (convdr_pair_nll_fwd_bwd)."""
    from ..train import pairwise_nll
    return pairwise_nll(q_embs, a_embs, b_embs)


class NLL(EmbeddingMixin):
    """models.py:52-75."""

    def forward(self, query_ids, attention_mask_q, input_ids_a=None, attention_mask_a=None, input_ids_b=None,
                attention_mask_b=None, is_query=True, seq_lens=None):
        kw = {} if seq_lens is None else {"seq_lens": seq_lens}
        if input_ids_b is None and is_query:
            return self.query_emb(query_ids, attention_mask_q, **kw)
        elif input_ids_b is None:
            return self.body_emb(query_ids, attention_mask_q, **kw)
        q_embs = self.query_emb(query_ids, attention_mask_q)
        a_embs = self.body_emb(input_ids_a, attention_mask_a)
        b_embs = self.body_emb(input_ids_b, attention_mask_b)
        return (_pairwise_nll(q_embs, a_embs, b_embs), )


class _Classifier(nn.Module):
    """RobertaForSequenceClassification's head: present in checkpoints, never used on this path
    (hence find_unused_parameters=True at gen_passage_embeddings.py:68)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.out_proj = nn.Linear(cfg.hidden_size, cfg.num_labels)


class RobertaDot_NLL_LN(NLL, _PretrainedMixin, nn.Module):
    """models.py:129-148: RoBERTa -> CLS -> embeddingHead Linear(H, 768) -> norm LayerNorm(768)."""
    config_class = RobertaConfig

    def __init__(self, config, model_argobj=None):
        nn.Module.__init__(self)
        NLL.__init__(self, model_argobj)
        self.config = config
        self.roberta = EncoderTower(config, "roberta")
        self.classifier = _Classifier(config)
        self.embeddingHead = nn.Linear(config.hidden_size, 768)
        self.norm = nn.LayerNorm(768)
        self.apply(self._init_weights)

    def query_emb(self, input_ids, attention_mask, seq_lens=None):
        """seq_lens (extension): optional host int array of token counts; saves the forward's device -> host round trip.
        use_mean (models.py:40-41): masked mean over the tokens instead of the CLS row -- the pooling runs inside the
        library (k_masked_mean), so ``self.roberta`` carries the flag."""
        if bool(self.use_mean) != bool(getattr(self.roberta, "pool_mean", False)):
            self.roberta.pool_mean = bool(self.use_mean)
            self.roberta.invalidate_packed()
        if _wants_autograd(self):
            from ..train import encoder_autograd
            return encoder_autograd(self, self.roberta, (self.embeddingHead, self.norm), input_ids, attention_mask, seq_lens)
        return self.roberta.embed(input_ids, attention_mask, head=(self.embeddingHead, self.norm), seq_lens=seq_lens)

    def body_emb(self, input_ids, attention_mask, seq_lens=None):
        return self.query_emb(input_ids, attention_mask, seq_lens)

    def resize_token_embeddings(self, new_num_tokens):
        """run_convdr_train.py:474: grow the word-embedding table, new rows N(0, 0.02) like HF's _init_weights."""
        old = self.roberta.embeddings.word_embeddings
        if new_num_tokens is None or new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim).to(old.weight.device)
        new.weight.data.normal_(mean=0.0, std=0.02)
        n = min(old.num_embeddings, new_num_tokens)
        new.weight.data[:n] = old.weight.data[:n]
        self.roberta.embeddings.word_embeddings = new
        self.roberta.__dict__.pop("_plist", None)
        self.roberta.__dict__.pop("_flat", None)     # the word table left the flat arena: re-run flatten_parameters
        self.roberta.invalidate_packed()
        self.config.vocab_size = new_num_tokens
        return new


class RobertaDot_NLL_LN_Inference(RobertaDot_NLL_LN):
    """models.py:151-156."""

    def forward(self, input_ids, attention_mask):
        return self.query_emb(input_ids, attention_mask)


class NLL_MultiChunk(EmbeddingMixin):
    """models.py:78-126: MaxP over 512-token chunks of the documents."""

    def forward(self, query_ids, attention_mask_q, input_ids_a=None, attention_mask_a=None, input_ids_b=None,
                attention_mask_b=None, is_query=True):
        if input_ids_b is None and is_query:
            return self.query_emb(query_ids, attention_mask_q)
        elif input_ids_b is None:
            return self.body_emb(query_ids, attention_mask_q)
        q_embs = self.query_emb(query_ids, attention_mask_q)
        a_embs = self.body_emb(input_ids_a, attention_mask_a)
        b_embs = self.body_emb(input_ids_b, attention_mask_b)
        batchS, full_length = input_ids_a.size()
        chunk_factor = full_length // self.base_len

        def bias(mask):   # chunks whose first position is masked are pure padding: -9999 keeps them out of the max (:100-107)
            first = mask.reshape(batchS, chunk_factor, -1)[:, :, 0]
            return ((1 - first) * (-9999)).float()
        from ..train import pairwise_nll
        return (pairwise_nll(q_embs, a_embs, b_embs, bias(attention_mask_a), bias(attention_mask_b)), )


class RobertaDot_CLF_ANN_NLL_MultiChunk(NLL_MultiChunk, RobertaDot_NLL_LN):
    """models.py:159-188: body_emb = [B, n_chunks, 768], one embedding per 512-token chunk.
    Chunks whose first position is masked (pure padding) get a zero embedding here; the reference feeds them through
    the encoder and then excludes them from the max with a -9999 bias (models.py:100-107), so their value is never
    observable unless every chunk of a document is padding."""

    def __init__(self, config, model_argobj=None):
        RobertaDot_NLL_LN.__init__(self, config, model_argobj)
        self.base_len = 512

    def body_emb(self, input_ids, attention_mask):
        batchS, full_length = input_ids.size()
        chunk_factor = full_length // self.base_len
        ids = input_ids.reshape(batchS * chunk_factor, full_length // chunk_factor)
        mask = attention_mask.reshape(batchS * chunk_factor, full_length // chunk_factor)
        live = mask[:, 0] != 0
        out = torch.zeros((batchS * chunk_factor, 768), dtype=torch.float32, device=input_ids.device)
        if bool(live.any()):
            out[live] = RobertaDot_NLL_LN.query_emb(self, ids[live], mask[live])
        return out.reshape(batchS, chunk_factor, 768)


class HFBertEncoder(EncoderTower):
    """models.py:191-216: a BERT tower whose forward returns (sequence_output, pooled = CLS, None).
    Only the CLS row is materialised on this path (nothing in the reference reads the rest)."""

    def __init__(self, config):
        super().__init__(config, "bert")
        assert config.hidden_size > 0, "Encoder hidden_size can't be zero"

    @classmethod
    def init_encoder(cls, args, dropout: float = 0.1):
        cfg = getattr(args, "bert_config", None) or BertConfig()
        if dropout != 0:
            cfg.attention_probs_dropout_prob = dropout
            cfg.hidden_dropout_prob = dropout
        enc = cls(cfg)
        path = getattr(args, "bert_path", None)     # the reference downloads "bert-base-uncased"; no network here
        if path:
            enc.load_state_dict({k[5:] if k.startswith("bert.") else k: v
                                 for k, v in _load_state_dict_file(path).items()}, strict=False)
        return enc

    def forward(self, input_ids, attention_mask):
        if _wants_autograd(self):
            from ..train import encoder_autograd
            pooled = encoder_autograd(self, self, None, input_ids, attention_mask)
        else:
            pooled = self.embed(input_ids, attention_mask)
        return None, pooled, None

    def get_out_size(self):
        return self.config.hidden_size


class BiEncoder(nn.Module):
    """models.py:219-262."""

    def __init__(self, args):
        super().__init__()
        self.question_model = HFBertEncoder.init_encoder(args)
        self.ctx_model = HFBertEncoder.init_encoder(args)

    def query_emb(self, input_ids, attention_mask):
        return self.question_model(input_ids, attention_mask)[1]

    def body_emb(self, input_ids, attention_mask):
        return self.ctx_model(input_ids, attention_mask)[1]

    def forward(self, query_ids, attention_mask_q, input_ids_a=None, attention_mask_a=None, input_ids_b=None,
                attention_mask_b=None, is_query=True):
        if input_ids_b is None:
            if input_ids_a is None:
                return self.query_emb(query_ids, attention_mask_q) if is_query else self.body_emb(
                    query_ids, attention_mask_q)
            return (self.query_emb(query_ids, attention_mask_q), self.body_emb(input_ids_a, attention_mask_a))
        q_embs = self.query_emb(query_ids, attention_mask_q)
        a_embs = self.body_emb(input_ids_a, attention_mask_a)
        b_embs = self.body_emb(input_ids_b, attention_mask_b)
        return (_pairwise_nll(q_embs, a_embs, b_embs), )


# --------------------------------------------------------------------------------------------
ALL_MODELS = ()
default_process_fn = None   # data/process_fn.py is dead code on this path


class MSMarcoConfig:
    """models.py:275-288."""

    def __init__(self, name, model, process_fn=default_process_fn, use_mean=True, tokenizer_class=RobertaTokenizer,
                 config_class=RobertaConfig):
        self.name = name
        self.process_fn = process_fn
        self.model_class = model
        self.use_mean = use_mean
        self.tokenizer_class = tokenizer_class
        self.config_class = config_class


configs = [
    MSMarcoConfig(name="rdot_nll", model=RobertaDot_NLL_LN, use_mean=False),
    MSMarcoConfig(name="rdot_nll_multi_chunk", model=RobertaDot_CLF_ANN_NLL_MultiChunk, use_mean=False),
    MSMarcoConfig(name="dpr", model=BiEncoder, tokenizer_class=BertTokenizer, config_class=BertConfig, use_mean=False),
]

MSMarcoConfigDict = {cfg.name: cfg for cfg in configs}

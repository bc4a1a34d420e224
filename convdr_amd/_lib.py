"""ctypes binding of libconvdr_hip.so (the C ABI declared in include/convdr_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CONVDR_HIP_LIB: load another build of the same library (kernel A/B experiments); default is the in-tree build
LIB_PATH = os.environ.get("CONVDR_HIP_LIB") or os.path.join(_HERE, "libconvdr_hip.so")

_lib = None

_p = C.c_void_p
_SIGNATURES = {
    "convdr_version": (C.c_int, []),
    "convdr_last_error": (C.c_char_p, []),
    "convdr_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "convdr_prof_enable": (C.c_int, [C.c_int]),
    "convdr_prof_collect": (C.c_int, [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "convdr_ip_column_mean": (C.c_int, [_p, C.c_int64, C.c_int, _p, _p, _p]),
    "convdr_ip_prepare_block": (C.c_int, [_p, C.c_int64, C.c_int, _p, _p, _p, _p, _p]),
    "convdr_ip_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "convdr_ip_search": (C.c_int, [_p, C.c_int, _p, _p, _p, C.c_int64, C.c_int, C.c_int, _p, _p, C.c_int, C.c_int,
                                   _p, C.c_size_t, _p, _p, _p, _p, _p]),
    "convdr_ip_prepare_block_f16": (C.c_int, [_p, C.c_int64, C.c_int, _p, C.c_float, _p, _p, _p, _p]),
    "convdr_ip_f16_scale": (C.c_float, [C.c_float]),
    "convdr_ip_search_f16": (C.c_int, [_p, C.c_int, _p, _p, _p, C.c_float, C.c_int64, C.c_int, C.c_int, _p, _p, C.c_int,
                                       C.c_int, _p, C.c_size_t, _p, _p, _p, _p, _p]),
    "convdr_ip_debug_counts": (_p, [_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "convdr_ip_debug_band": (_p, [_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]),
}


class ConvdrError(RuntimeError):
    pass


def exported_symbols():
    """Names declared in include/convdr_hip.h that the library must export."""
    return sorted(_SIGNATURES)


def register(name, restype, argtypes):
    _SIGNATURES[name] = (restype, argtypes)
    if _lib is not None:
        f = getattr(_lib, name)
        f.restype, f.argtypes = restype, argtypes


def lib():
    """Load the shared library (once).  Raises if it has not been built: the product
    path never falls back to a CPU implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ConvdrError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C convdr_amd/csrc`" % LIB_PATH)
        # torch first: libconvdr_hip.so must bind to the HIP runtime (libamdhip64) that torch has loaded, not a
        # second copy -- two runtimes in one process do not share devices, streams or allocations
        # (CONVDR_LIB_NO_TORCH=1: tests/capi/san_driver.py only -- the host-sanitizer build is driven without torch in the process)
        if not os.environ.get("CONVDR_LIB_NO_TORCH"):
            import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            f = getattr(_lib, name)
            f.restype, f.argtypes = restype, argtypes
    return _lib


def check(rc, what):
    if rc != 0:
        raise ConvdrError("%s failed (%d): %s" % (what, rc, lib().convdr_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def prof_collect(name):
    """(total_ms, launches) of the spans recorded under `name` since convdr_prof_enable(1)."""
    ms, cnt = C.c_float(0), C.c_int(0)
    check(lib().convdr_prof_collect(name.encode(), C.byref(ms), C.byref(cnt)), "convdr_prof_collect")
    return ms.value, cnt.value


# ---- structs of include/convdr_hip.h -------------------------------------------------------
class EncoderConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("layers", C.c_int32),
                ("intermediate", C.c_int32), ("vocab", C.c_int32), ("max_pos", C.c_int32), ("pad_idx", C.c_int32),
                ("out_dim", C.c_int32), ("ln_eps", C.c_float), ("head_ln_eps", C.c_float), ("pool_mean", C.c_int32)]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2",
                                          "ln2_g", "ln2_b", "wo_ks", "w2_ks")]


class EncoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")] + \
               [("layers", C.POINTER(LayerWeights))] + \
               [(n, C.c_void_p) for n in ("head_w", "head_b", "head_ln_g", "head_ln_b")]


register("convdr_cast_f32_bf16", C.c_int, [_p, _p, C.c_int64, _p])
register("convdr_pack_kslice", C.c_int, [_p, C.c_int, C.c_int, _p, _p])
register("convdr_encoder_workspace_bytes", C.c_size_t, [C.POINTER(EncoderConfig), C.c_int64, C.c_int])
register("convdr_encoder_forward", C.c_int, [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), _p, C.c_int, _p, C.c_int,
                                             C.c_int, _p, _p, C.c_int64, C.c_int, _p, C.c_size_t, _p, _p])
register("convdr_encoder_debug_layout", C.c_int, [C.POINTER(EncoderConfig), C.c_int64, C.c_int, C.POINTER(C.c_int64)])


class LayerWeightsT(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv_t", "wo_t", "w1_t", "w2_t")]


class LayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2",
                                          "ln2_g", "ln2_b")]


class EncoderGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")] + \
               [("layers", C.POINTER(LayerGrads))] + \
               [(n, C.c_void_p) for n in ("head_w", "head_b", "head_ln_g", "head_ln_b")]


class Dropout(C.Structure):
    _fields_ = [("p_hidden", C.c_float), ("p_attention", C.c_float), ("seed", C.c_uint32)]


register("convdr_encoder_train_workspace_bytes", C.c_size_t, [C.POINTER(EncoderConfig), C.c_int64, C.c_int])
register("convdr_encoder_train_forward", C.c_int, [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), _p, C.c_int, _p,
                                                   C.c_int, C.c_int, _p, _p, C.c_int64, C.c_int, _p, C.c_size_t, _p,
                                                   C.POINTER(Dropout), _p])
register("convdr_encoder_backward", C.c_int, [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), C.POINTER(LayerWeightsT),
                                              _p, _p, _p, C.c_int, C.c_int64, C.c_int, _p, C.c_size_t, _p,
                                              C.POINTER(EncoderGrads), C.POINTER(Dropout), _p])
register("convdr_encoder_backward_fresh", C.c_int, [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), C.POINTER(LayerWeightsT),
                                                    _p, _p, _p, C.c_int, C.c_int64, C.c_int, _p, C.c_size_t, _p,
                                                    C.POINTER(EncoderGrads), C.POINTER(Dropout), _p])
register("convdr_backward_wait_layer", C.c_int, [C.c_int, _p])
register("convdr_wgrad", C.c_int, [_p, C.c_int, C.c_int64, _p, C.c_int, C.c_int64, C.c_int64, _p, C.c_size_t, _p, _p])
register("convdr_transpose_f32_bf16", C.c_int, [_p, C.c_int, C.c_int, _p, _p])
register("convdr_mse_fwd_bwd", C.c_int, [_p, _p, C.c_int64, C.c_float, _p, _p, _p])
register("convdr_rank_ce_fwd_bwd", C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, C.c_int, _p])
register("convdr_inbatch_ce_fwd_bwd", C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p, C.c_float, _p, _p, C.c_int, _p])
register("convdr_grad_norm_clip", C.c_int, [_p, C.c_int64, C.c_float, C.c_float, _p, _p, C.c_int, _p])
register("convdr_pair_nll_fwd_bwd", C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, _p, _p, _p])
register("convdr_train_set_side_stream", C.c_int, [_p])
# RCCL-backed collectives for torch-free hosts (csrc/comm.hip); convdr_amd/parallel.py itself stays on torch.distributed
register("convdr_comm_unique_id", C.c_int, [_p])
register("convdr_comm_init", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, _p])
register("convdr_comm_ranks", C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)])
register("convdr_comm_allgather", C.c_int, [_p, _p, _p, C.c_size_t, _p])
register("convdr_comm_allreduce_f32", C.c_int, [_p, _p, _p, C.c_size_t, _p])
register("convdr_comm_destroy", C.c_int, [_p])
register("convdr_grad_sumsq", C.c_int, [_p, C.c_int64, _p, C.c_int, _p])
register("convdr_grad_norm_finish", C.c_int, [_p, C.c_int, C.c_float, C.c_float, _p, _p])
register("convdr_scale_f32", C.c_int, [_p, C.c_int64, _p, _p])
register("convdr_adamw_step", C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_int, C.c_int, _p, _p])
register("convdr_adamw_step_packed", C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                               C.c_double, C.c_int, C.c_int, _p, _p, C.c_int64, _p])
register("convdr_set_option", C.c_int, [C.c_char_p, C.c_int64])
register("convdr_topk_merge", C.c_int, [_p, _p, C.c_int, C.c_int64, _p, _p, C.c_int, C.c_int64, C.c_int, C.c_int, _p, _p,
                                        C.c_int64, _p])
register("convdr_pack_transposed", C.c_int, [_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int64), _p, _p])
register("convdr_pack_transposed_bf16", C.c_int, [_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int64), _p, _p])

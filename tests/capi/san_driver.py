"""Driven by tests/test_capi_san_cpu.py in a subprocess with clang's ASan runtime preloaded and CONVDR_HIP_LIB pointing at the
host-sanitizer build (make -C convdr_amd/csrc SAN=1): every C-ABI entry point's argument validation, the workspace planners over
a sweep of sizes, the option table and the host-side helpers -- the code that runs on the host BEFORE any launch -- under
AddressSanitizer + UndefinedBehaviorSanitizer.  No GPU: a call that gets past validation fails at its first HIP call, which is
the expected (and checked) outcome.  Prints "SAN_DRIVER_OK <calls>" when nothing was reported."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["CONVDR_LIB_NO_TORCH"] = "1"
from convdr_amd import _lib  # noqa: E402

L = _lib.lib()
calls = 0


def bad(rc, needle=None):
    global calls
    calls += 1
    assert rc != 0, "expected a rejection"
    msg = L.convdr_last_error()
    assert msg, "no error text"
    if needle is not None:
        assert needle.encode() in msg, (needle, msg)


assert L.convdr_version() >= 100
# ---- search: planners over a sweep, validation of both ABIs, the merge ----
for nq in (1, 7, 128, 129, 1000, 4096):
    for n in (0, 1, 4095, 4096, 65537, 1_000_000, 38_000_000):
        for k, cap in ((1, 1024), (100, 4096), (2048, 4096), (4096, 8192)):
            assert L.convdr_ip_workspace_bytes(nq, n, 768, k, cap) > 0
            calls += 1
bad(L.convdr_ip_prepare_block(None, 10, 70, None, None, None, None, None), "d % 64")
bad(L.convdr_ip_prepare_block_f16(None, 10, 70, None, 1.0, None, None, None, None))
bad(L.convdr_ip_search(None, 0, None, None, None, 10, 768, 10, None, None, 4096, 0, None, 0, None, None, None, None, None))
bad(L.convdr_ip_search(None, 4, None, None, None, 10, 768, 10, None, None, 1000, 0, None, 0, None, None, None, None, None), "cap")
bad(L.convdr_ip_search(None, 4, None, None, None, 10, 768, 3000, None, None, 4096, 0, None, 0, None, None, None, None, None), "too large")
bad(L.convdr_ip_search(None, 4, None, None, None, 10, 768, 10, None, None, 4096, 0, None, 16, None, None, None, None, None), "workspace")
bad(L.convdr_ip_search_f16(None, 4, None, None, None, 1.0, 1 << 31, 768, 10, None, None, 4096, 0, None, 0, None, None, None, None, None))
bad(L.convdr_topk_merge(None, None, 5000, 5000, None, None, 10, 10, 4, 100, None, None, 100, None))
assert L.convdr_ip_f16_scale(1.0) > 0 and L.convdr_ip_f16_scale(0.0) > 0 and L.convdr_ip_f16_scale(1e30) > 0
calls += 3
# ---- encoder: planners for every shape the tests use, configs that must be refused ----
Cfg = _lib.EncoderConfig
shapes = [(768, 12, 12, 3072, 768), (128, 2, 2, 256, 128), (128, 2, 3, 320, 0), (1024, 16, 24, 4096, 1024)]
for hidden, heads, layers, inter, out_dim in shapes:
    cfg = Cfg(0, hidden, heads, layers, inter, 50265, 514, 1, out_dim, 1e-5, 1e-5, 0)
    for rows, B in ((8, 1), (4096, 64), (9216, 64), (262144, 2048), (327680, 640)):
        assert L.convdr_encoder_workspace_bytes(C.byref(cfg), rows, B) > 0
        assert L.convdr_encoder_train_workspace_bytes(C.byref(cfg), rows, B) > 0
        lay = (C.c_int64 * 16)()
        L.convdr_encoder_debug_layout(C.byref(cfg), rows, B, lay)
        calls += 3
cfg = Cfg(0, 768, 12, 12, 3072, 50265, 514, 1, 768, 1e-5, 1e-5, 0)
w = _lib.EncoderWeights()
lw = (_lib.LayerWeights * 12)()
w.layers = C.cast(lw, C.POINTER(_lib.LayerWeights))
bad(L.convdr_encoder_forward(C.byref(cfg), C.byref(w), None, 0, None, 0, 128, None, None, 0, 0, None, 0, None, None))
bad(L.convdr_encoder_train_forward(C.byref(cfg), C.byref(w), None, 0, None, 4, 128, None, None, 7, 128, None, 0, None, None, None), "bad sizes")
bad(L.convdr_encoder_train_forward(C.byref(cfg), C.byref(w), None, 0, None, 4, 128, None, None, 512, 128, None, 16, None, None, None), "workspace")
dr = _lib.Dropout(1.5, 0.1, 1)
bad(L.convdr_encoder_train_forward(C.byref(cfg), C.byref(w), None, 0, None, 4, 128, None, None, 512, 128, None, 0, None, C.byref(dr), None), "dropout")
badcfg = Cfg(0, 100, 12, 12, 3072, 50265, 514, 1, 768, 1e-5, 1e-5, 0)
bad(L.convdr_encoder_train_forward(C.byref(badcfg), C.byref(w), None, 0, None, 4, 128, None, None, 512, 128, None, 0, None, None, None), "hidden")
toomany = Cfg(0, 768, 12, 64, 3072, 50265, 514, 1, 768, 1e-5, 1e-5, 0)
assert L.convdr_encoder_train_workspace_bytes(C.byref(toomany), 512, 4) == 0
gr = _lib.EncoderGrads()
lg = (_lib.LayerGrads * 12)()
gr.layers = C.cast(lg, C.POINTER(_lib.LayerGrads))
wt = (_lib.LayerWeightsT * 12)()
bad(L.convdr_encoder_backward(C.byref(cfg), C.byref(w), wt, None, None, None, 4, 512, 128, None, 16, None, C.byref(gr), None, None), "workspace")
bad(L.convdr_pack_kslice(None, 10, 30, None, None), "bad shape")
bad(L.convdr_wgrad(None, 0, 0, None, 0, 0, 10, None, 0, None, None), "bad sizes")
bad(L.convdr_train_set_side_stream(None), "null stream")
bad(L.convdr_backward_wait_layer(99, None))
# ---- options: every documented name is accepted, an unknown one is refused ----
for name in ("fused_ln_min_rows", "hm_blocked", "gemm_tile_policy", "attn_bwd_fused", "ip_fused_finish", "gelu_gp", "ffn2_splitk", "ln_rows",
             "ln_bwd_rows", "embed_bwd_deterministic"):
    assert L.convdr_set_option(name.encode(), 1) == 0, name
    calls += 1
for name, v in (("fused_ln_min_rows", 24576), ("gemm_tile_policy", 0), ("ln_bwd_rows", 2), ("embed_bwd_deterministic", 0)):
    assert L.convdr_set_option(name.encode(), v) == 0
bad(L.convdr_set_option(b"no_such_option", 1))
# ---- profiling spans and the collectives' argument checks ----
L.convdr_prof_enable(1)
ms, cnt = _lib.prof_collect("gemm_ffn1")
assert cnt == 0
L.convdr_prof_enable(0)
comm = C.c_void_p()
bad(L.convdr_comm_init(C.byref(comm), 0, 0, None))
bad(L.convdr_comm_ranks(None, None, None))
bad(L.convdr_comm_allgather(None, None, None, 16, None))
bad(L.convdr_comm_allreduce_f32(None, None, None, 16, None))
assert L.convdr_comm_destroy(None) == 0
buf = C.create_string_buffer(64)
L.convdr_device_pci_bus_id(0, buf, 64)        # (no GPU: an error code, not a crash)
L.convdr_device_pci_bus_id(0, buf, 1)
print("SAN_DRIVER_OK %d" % calls)

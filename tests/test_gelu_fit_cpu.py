"""The one-transcendental GELU of the FFN1 epilogue (convdr_amd/csrc/encoder_kernels.hpp: gelu_tail) against the exact erf
form the reference uses (transformers 2.3.0 modeling_bert.gelu, reached from /root/reference/model/models.py:141).
Same constants, same operation order in fp32 (fused multiply-adds emulated in fp64), all of the activation range."""
import numpy as np
from scipy.special import erf

F = np.float32


def _fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.float64(c)).astype(F)


def gelu_tail_fp32(x):
    x = x.astype(F)
    t = np.minimum(np.abs(x), F(9.0))
    p = _fma(t, F(0.0041585), F(-0.04571999))
    p = _fma(p, t, F(-0.46495319))
    p = _fma(p, t, F(-1.14955714))
    q = np.exp2(_fma(p, t, F(-1.0)).astype(np.float64)).astype(F)
    return (-(t.astype(np.float64)) * q + np.maximum(x, F(0.0))).astype(F)


def test_gelu_fit_is_within_1e5_of_exact_erf_gelu():
    x = np.concatenate([np.linspace(-12, 12, 960001), np.array([-1e4, -100.0, -30.0, 30.0, 100.0, 1e4, 0.0])])
    exact = 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))
    got = gelu_tail_fp32(x).astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - exact).max() < 1e-5
    # saturation: identity for large positive inputs, nothing left (|t Q(t)| < 1e-15) for large negative ones
    assert got[-2] == 1e4 and abs(got[-7]) < 1e-15 and got[-1] == 0.0


def gelu_grad_fp32(x):
    """encoder_kernels.hpp: gelu_grad (same constants and operation order)."""
    x = x.astype(F)
    t = np.minimum(np.abs(x), F(9.0))
    p = _fma(t, F(0.0041585), F(-0.04571999))
    p = _fma(p, t, F(-0.46495319))
    p = _fma(p, t, F(-1.14955714))
    q = np.exp2(_fma(p, t, F(-1.0)).astype(np.float64)).astype(F)
    cdf = np.where(x >= 0, F(1.0) - q, q).astype(F)
    e = np.exp2((F(-0.7213475204444817) * x * x).astype(np.float64)).astype(F)
    return _fma(x * F(0.3989422804014327), e, cdf)


def test_gelu_grad_fit_is_within_1e4_of_exact_derivative():
    x = np.concatenate([np.linspace(-12, 12, 960001), np.array([-1e4, -100.0, -30.0, 30.0, 100.0, 1e4, 0.0])])
    exact = 0.5 * (1.0 + erf(x / np.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)
    got = gelu_grad_fp32(x).astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - exact).max() < 1e-4          # bf16 gradients resolve 4e-3
    assert got[-2] == 1.0 and abs(got[-7]) < 1e-15 and got[-1] == 0.5


def gelu_and_grad_shared_tail_fp32(x):
    """encoder_kernels.hpp: gelu_tail2_gp (EPI_GELU_GP, round 5) -- gelu(x) and gelu'(x) from ONE evaluation of Q(|x|):
    gelu' = 0.5 + copysign(0.5 + q m(t), x), m = quartic fit of t phi(t) / Q(t) - 1 against the fitted q."""
    x = x.astype(F)
    t = np.minimum(np.abs(x), F(9.0))
    p = _fma(t, F(0.0041585), F(-0.04571999))
    p = _fma(p, t, F(-0.46495319))
    p = _fma(p, t, F(-1.14955714))
    q = np.exp2(_fma(p, t, F(-1.0)).astype(np.float64)).astype(F)
    m = _fma(t, F(-0.0117551), F(0.09555683))
    m = _fma(m, t, F(0.64461331))
    m = _fma(m, t, F(0.79644568))
    m = _fma(m, t, F(-0.99992798))
    hr = _fma(q, m, F(0.5))
    grad = (F(0.5) + np.copysign(hr, x)).astype(F)
    gelu = (-(t.astype(np.float64)) * q + np.maximum(x, F(0.0))).astype(F)
    return gelu, grad


def test_shared_tail_gelu_derivative_is_within_1e4_of_exact_derivative():
    x = np.concatenate([np.linspace(-12, 12, 960001), np.array([-1e4, -100.0, -30.0, 30.0, 100.0, 1e4, 0.0])])
    exact = 0.5 * (1.0 + erf(x / np.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)
    gelu, got = gelu_and_grad_shared_tail_fp32(x)
    assert np.isfinite(got).all()
    assert np.abs(got.astype(np.float64) - exact).max() < 1e-4      # measured 3.7e-5; the stored bf16 derivative resolves 4e-3
    assert got[-2] == 1.0 and abs(got[-7]) < 1e-15 and abs(got[-1] - 0.5) < 1e-4
    assert np.array_equal(gelu, gelu_tail_fp32(x))                    # the forward value is the unchanged fit

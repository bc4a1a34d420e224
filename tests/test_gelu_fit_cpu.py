"""The logistic-polynomial GELU used by the FFN1 epilogue (convdr_amd/csrc/encoder_kernels.hpp: gelu_sig) against the
exact erf form the reference uses (transformers 2.3.0 modeling_bert.gelu, reached from /root/reference/model/models.py:141).
Same constants, fp32 arithmetic, all of the representable activation range."""
import numpy as np
from scipy.special import erf


def gelu_sig_fp32(x):
    x = x.astype(np.float32)
    u = np.minimum(x * x, np.float32(64.0))
    t = u * np.float32(0.001023812276) + np.float32(-0.106834618)
    t = t * u + np.float32(-2.30105646)
    with np.errstate(over="ignore"):
        e = np.exp2(t * x)
    return x * (np.float32(1.0) / (np.float32(1.0) + e))


def test_gelu_fit_is_within_3e5_of_exact_erf_gelu():
    x = np.concatenate([np.linspace(-12, 12, 960001), np.array([-1e4, -100.0, -30.0, 30.0, 100.0, 1e4, 0.0])])
    exact = 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))
    got = gelu_sig_fp32(x).astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - exact).max() < 3e-5
    # saturation: identity for large positive, (signed) zero for large negative inputs
    assert got[-2] == 1e4 and got[-7] == 0.0

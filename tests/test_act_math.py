"""CPU: the erf() minimax polynomials of cambrian_amd/csrc/common.h (cmb_erf), re-evaluated in float32 numpy with the
coefficients PARSED FROM THE HEADER, against math.erf on a dense grid — the GELU of the reference is the exact-erf
nn.GELU() (vision_sampler.py:241, cambrian_arch.py:56), so the device approximation must stay within fp32 round-off."""
import math
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coeffs():
    src = open(os.path.join(ROOT, "cambrian_amd", "csrc", "common.h")).read()
    body = src[src.index("float cmb_erf(float a)"):]
    body = body[:body.index("return t >")]
    nums = [float(x.rstrip("f")) for x in re.findall(r"-?\d\.\d+e-\d+f", body)]
    assert len(nums) == 13, nums
    thr = float(re.search(r"return t > (\d\.\d+)f", src).group(1))
    return nums, thr


def test_cmb_erf_polynomials_within_one_ulp():
    c, thr = _coeffs()
    f = np.float32

    def fma(a, b, cc):
        return (np.float64(a) * np.float64(b) + np.float64(cc)).astype(np.float32)

    x = np.concatenate([np.linspace(-6, 6, 400001), np.random.default_rng(0).normal(size=200000) * 2]).astype(f)
    t, s = np.abs(x), (x * x).astype(f)
    r = fma(np.full_like(x, f(c[0])), t, f(c[1]))
    u = fma(np.full_like(x, f(c[2])), t, f(c[3]))
    r = fma(r, s, u)
    for k in (c[4], c[5], c[6]):
        r = fma(r, t, f(k))
    r = fma(r, t, -t)
    big = np.copysign((1.0 - np.exp(r.astype(np.float64))).astype(f), x)
    q = np.full_like(x, f(c[7]))
    for k in c[8:13]:
        q = fma(q, s, f(k))
    small = fma(q, x, x)
    got = np.where(t > f(thr), big, small).astype(np.float64)
    ref = np.array([math.erf(float(v)) for v in x])
    err = np.abs(got - ref)
    assert err.max() < 1.2e-7, err.max()
    assert (err / np.maximum(np.abs(ref), 1e-30)).max() < 2e-7


def test_cmb_gelu_erf_bf16_within_bf16_rounding():
    """The bf16 GEMM epilogues' GELU (common.h::cmb_gelu_erf_bf16: relu(x) - t 2^P(t), t = min(|x|, 7)), coefficients parsed from the
    header and evaluated in float32: absolute error <= 8e-6, relative error <= 5e-5 wherever |GELU| > 1e-3 (half a bf16 ulp
    is 2e-3), against the
    exact-erf GELU of the reference (nn.GELU(), vision_sampler.py:241); P stays decreasing up to the clamp, so large inputs
    cannot turn the 2^P term back on; infinities stay finite / signed as GELU's limits."""
    src = open(os.path.join(ROOT, "cambrian_amd", "csrc", "common.h")).read()
    body = src[src.index("float cmb_gelu_erf_bf16(float x)"):]
    body = body[:body.index("return fmaf(-t")]
    clamp = float(re.search(r"fminf\(fabsf\(x\), (\d+\.\d+)f\)", body).group(1))
    nums = [float(x.rstrip("f")) for x in re.findall(r"-?\d\.\d+(?:e-\d+)?f", body.split("float p =")[1])]
    assert len(nums) == 6 and clamp == 7.0
    # the pair form the GEMM epilogue calls carries the same numbers
    pair = src[src.index("void cmb_gelu_erf_bf16_pair"):]
    pair = pair[:pair.index("const f32x2_t h =")]
    for c in nums:
        assert pair.count(repr(c)) >= 2 or pair.count(("%.16g" % c)) >= 2, c
    f = np.float32
    x = np.concatenate([np.linspace(-20, 20, 800001), np.random.default_rng(2).normal(size=200000) * 2]).astype(f)
    t = np.minimum(np.abs(x), f(clamp)).astype(f)
    p = np.full_like(x, f(nums[0]))
    for k in nums[1:]:
        p = (p.astype(np.float64) * t + np.float64(f(k))).astype(f)          # fmaf
    h = np.exp2(p.astype(np.float64)).astype(f)
    got = (np.maximum(x, f(0)).astype(np.float64) - t.astype(np.float64) * h).astype(f).astype(np.float64)
    ref = np.array([0.5 * float(v) * (1.0 + math.erf(float(v) / math.sqrt(2.0))) for v in x])
    err = np.abs(got - ref)
    assert err.max() < 8e-6, err.max()
    assert (err / np.maximum(np.abs(ref), 1e-3)).max() < 5e-5
    tt = np.linspace(0, clamp, 200001)
    P = np.polyval(nums, tt)
    assert (np.diff(P) < 0).all() and P[-1] < -39 and clamp * 2.0 ** P[-1] < 1e-11

"""The two split arithmetics of the matrix-core kernels, restated in numpy (no GPU): what the operand splits represent and what the
dropped products cost, against an fp32 matmul of the same data.  The device kernels are checked against fp64 in tests/test_kernels_gpu.py
(test_conv_split_engine_is_fp32_equivalent); this file pins the claims DESIGN.md 4.0 makes about the number formats themselves.
"""
import numpy as np


def bf16_parts(x, n=3):
    x = x.astype(np.float32)
    out = []
    for _ in range(n):
        u = x.view(np.uint32)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)      # round to nearest even, as v_cvt_pk_bf16_f32
        out.append(r.astype(np.float64))
        x = x - r
    return out


def f16_parts(x, scale=1.0):
    x = x.astype(np.float32) * np.float32(scale)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return [hi.astype(np.float64) / scale, lo.astype(np.float64) / scale]


def f16_act_parts(x):
    """The shipped activation split (round 5): hi = rne(x), lo_s = rne((x - hi) * 2^11); represents hi + 2^-11 lo_s."""
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo_s = ((x - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi.astype(np.float64), lo_s.astype(np.float64)


def f16x2_contract(x, w):
    """out = 2^-8 (hi Whi + hi Wlo + lo_s (2^-11 Whi)) with W = 2^8 w -- what conv_wino.hip / conv_pw.hip / dcn.hip issue (products exact, fp64 sums)."""
    hi, lo_s = f16_act_parts(x)
    W = w.astype(np.float32) * np.float32(256)
    whi = W.astype(np.float16)
    wlo = (W - whi.astype(np.float32)).astype(np.float16)
    whs = (whi.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)          # v_pk_mul_f16: one rounding, exact while normal
    f = lambda a: a.astype(np.float64)
    return (hi @ f(whi) + hi @ f(wlo) + lo_s @ f(whs)) / 256.0


def test_scaled_low_part_is_normal_whenever_the_high_part_is_and_never_overflows():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(400000) * np.exp(rng.uniform(-12, 10, 400000))).astype(np.float32)
    x = x[(np.abs(x) >= 2.0 ** -14) & (np.abs(x) < 65504)]
    hi, lo_s = f16_act_parts(x)
    assert np.all(np.isfinite(lo_s)) and np.all(np.abs(lo_s) <= np.abs(x) * (1 + 2.0 ** -10)), "|x - hi| <= 2^-11 |x|: the scaled low part never exceeds |x|"
    assert np.all(np.abs(x.astype(np.float64) - hi - lo_s / 2048) <= 2.0 ** -22 * np.abs(x)), "hi + 2^-11 lo_s holds x to 2^-22 for EVERY |x| >= 2^-14"
    tiny = (rng.uniform(-1, 1, 100000) * 2.0 ** -15).astype(np.float32)
    hi, lo_s = f16_act_parts(tiny)
    assert np.all(np.abs(tiny.astype(np.float64) - hi - lo_s / 2048) <= 2.0 ** -36), "below fp16's normal range: an absolute 2^-36"


def test_shipped_two_part_contraction_is_fp32_equivalent_from_1e_minus_4_to_1e4():
    """activation scales 1 .. 1e-4 (and up to 1e4), weight scales 1/24 and 1e-2, the bound of the device tests."""
    rng = np.random.default_rng(0)
    K, M, N = 576, 192, 192
    for xs in (1.0, 1e-2, 1e-3, 1e-4, 30.0, 1e4):
        for ws in (1 / 24, 1e-2):
            x = ((rng.random((M, K)) * 2 - 1) * xs).astype(np.float32)
            w = ((rng.random((K, N)) * 2 - 1) * ws).astype(np.float32)
            ref = x.astype(np.float64) @ w.astype(np.float64)
            e32 = np.abs((x @ w).astype(np.float64) - ref)
            e3 = np.abs(f16x2_contract(x, w) - ref)
            rms = lambda e: float(np.sqrt((e ** 2).mean()))
            assert rms(e3) < rms(e32) and e3.max() <= 1.25 * e32.max(), (xs, ws, rms(e32), rms(e3), e32.max(), e3.max())
            # the round-4 form (plain low part) for contrast: it leaves the bound as soon as the activations are small
            a, b = f16_parts(x), f16_parts(w, 256.0)
            e_old = np.abs(a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0] - ref)
            if xs <= 1e-2:
                assert rms(e_old) > 5 * rms(e32), "the test must be able to see the defect it guards against"


def test_two_fp16_parts_hold_22_bits_and_three_bf16_parts_hold_all_24():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-2, 8, 200000))).astype(np.float32)      # |x| from ~0.1 to a few thousand
    x = x[np.abs(x) > 0.25]
    hi, lo = f16_parts(x)
    assert np.all(np.abs(x.astype(np.float64) - hi - lo) <= 2.0 ** -23 * np.abs(x)), "hi + lo must hold x to 2^-23 while both parts are normal"
    p = bf16_parts(x)
    assert np.all(x.astype(np.float64) == p[0] + p[1] + p[2]), "three bf16 parts are an exact split of an fp32 number"
    # below fp16's normal range the low part is a subnormal: an ABSOLUTE error of at most half its quantum 2^-24
    t = (rng.uniform(-1, 1, 100000) * 1e-3).astype(np.float32)
    hi, lo = f16_parts(t)
    assert np.all(np.abs(t.astype(np.float64) - hi - lo) <= 2.0 ** -25)
    # the packing scale of the weights (2^8) lifts everyday weights (1e-3 .. 1) into the relative regime
    w = (rng.uniform(-1, 1, 100000) * 0.05).astype(np.float32)
    w = w[np.abs(w) > 1e-3]
    hi, lo = f16_parts(w, 256.0)
    assert np.all(np.abs(w.astype(np.float64) - hi - lo) <= 2.0 ** -23 * np.abs(w))


def test_out_of_range_values_fail_loudly():
    big = np.array([7.0e4, -1.0e5], np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi, lo = f16_parts(big)
        assert np.all(np.isinf(hi)) and np.all(np.isnan(hi + lo)), "beyond fp16's range the form yields inf / NaN, never a clamped number"


def test_contraction_error_is_below_an_fp32_matmul():
    rng = np.random.default_rng(0)
    K, M, N = 576, 192, 192
    for xs, ws in ((1.0, 1 / 24), (1.0, 0.01), (30.0, 0.02)):
        x = ((rng.random((M, K)) * 2 - 1) * xs).astype(np.float32)
        w = ((rng.random((K, N)) * 2 - 1) * ws).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64)
        e32 = np.abs((x @ w).astype(np.float64) - ref)
        a, b = bf16_parts(x), bf16_parts(w)
        e6 = np.abs(sum(a[i] @ b[j] for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0))) - ref)
        a, b = f16_parts(x), f16_parts(w, 256.0)
        e3 = np.abs(a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0] - ref)
        rms = lambda e: float(np.sqrt((e ** 2).mean()))
        assert rms(e6) < rms(e32) and rms(e3) < rms(e32), (xs, ws, rms(e32), rms(e6), rms(e3))
        assert e3.max() < 1.5 * e32.max(), (xs, ws, e32.max(), e3.max())

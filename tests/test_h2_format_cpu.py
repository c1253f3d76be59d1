"""The two-f16-plane operand format of the InfoNCE kernels (csrc/gcr_infonce.hip, EngH2), emulated in numpy: the bounds
DESIGN §4.2b states for it.  x = hi + lo with hi = f16(x), lo = f16(x - hi) (round-to-nearest-even, gradual underflow —
numpy's float16 conversion does both, as v_cvt_pk_f16_f32 does); a product keeps hi*hi + hi*lo + lo*hi."""
import numpy as np


def split2(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def test_residual_of_the_two_planes():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * np.exp(rng.uniform(-3, 5, 200_000))).astype(np.float32)
    x = x[np.abs(x) < 6.0e4]                                    # inside the f16 range, as the kernels' pre-scales ensure
    hi, lo = split2(x)
    res = np.abs(x.astype(np.float64) - hi - lo)
    normal = np.abs(x) >= 2.0 ** -3                             # lo = (x - hi) ~ 2^-11 |x| stays a normal f16 (>= 2^-14)
    assert (res[normal] <= 2.0 ** -22 * np.abs(x[normal])).all()
    assert (res[~normal] <= 2.0 ** -25).all()                   # below: the f16 sub-normal spacing 2^-24, half of it


def test_three_term_dot_product_of_unit_rows():
    """Scores of unit-norm rows with the pre-scales of rounds 2-3 (stationary x 2^4 after the 1/tau log2 e factor,
    streamed x 2^8): |three-term product - exact| <= 3 * 2^-22 * sum |x_k y_k|, typically ~10x less."""
    rng = np.random.default_rng(1)
    d, n = 64, 4000
    a = rng.standard_normal((n, d)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.standard_normal((n, d)); b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[:100] = a[:100] + 1e-3 * rng.standard_normal((100, d))   # near-duplicates: logits at the maximum
    b[:100] /= np.linalg.norm(b[:100], axis=1, keepdims=True)
    scale2 = 10.0 * 1.4426950408889634                         # 1/tau = 10 in the log2 domain
    xs = (a * scale2 * 16.0).astype(np.float32)
    ys = (b * 256.0).astype(np.float32)
    xh, xl = split2(xs)
    yh, yl = split2(ys)
    got = ((xh * yh).sum(1) + (xh * yl).sum(1) + (xl * yh).sum(1)) / 4096.0
    ref = (xs.astype(np.float64) * ys.astype(np.float64)).sum(1) / 4096.0
    bound = 3 * 2.0 ** -22 * (np.abs(xs.astype(np.float64)) * np.abs(ys.astype(np.float64))).sum(1) / 4096.0
    err = np.abs(got - ref)
    assert (err <= bound).all()
    assert np.median(err / bound) < 0.2
    assert err.max() < 1.5e-6 * scale2                          # a log2-domain logit is good to ~1e-6 of its 14.4 range


def test_probability_plane_range():
    """P of the flash forward: <= 2 relative to the lagging reference, scaled by 2^14: every value down to 2^-28 of the
    reference keeps 11 bits in hi, smaller ones go sub-normal with absolute error <= 2^-25 (scaled) = 2^-39."""
    p = np.exp2(np.linspace(-40, 1, 4000)).astype(np.float32)
    hi, lo = split2(p * np.float32(2.0 ** 14))
    err = np.abs(p.astype(np.float64) * 2.0 ** 14 - hi - lo) / 2.0 ** 14
    big = p >= 2.0 ** -17                                       # scaled value >= 2^-3: both planes normal
    assert (err[big] <= 2.0 ** -22 * p[big]).all()
    assert (err[~big] <= 2.0 ** -39).all()
    assert np.isfinite(hi).all() and hi.max() <= 32768.0


def test_unit_product_prescales_cost_no_accuracy():
    """Round 4: stationary x 2^-2, streamed x 2^2 — product 1, so the accumulator is the log2-domain score itself (EngH2::kSX /
    kSY / kSInv in csrc/gcr_infonce.hip).  Elements under 2^-3 after scaling have a sub-normal residual plane (absolute error
    <= 2^-25), which a dot product of unit rows tolerates: the worst score error over dense rows, near-duplicate pairs and
    rows with a few large and many tiny elements stays within 10 % of the old pre-scales' and under 3e-6 at 1/tau = 20."""
    rng = np.random.default_rng(2)
    n = 6000

    def unit(x):
        return x / np.linalg.norm(x, axis=1, keepdims=True)

    for d in (64, 32):
        a, b = unit(rng.standard_normal((n, d))), unit(rng.standard_normal((n, d)))
        b[:600] = unit(a[:600] + 1e-3 * rng.standard_normal((600, d)))
        c = unit(rng.standard_normal((n, d)) * np.exp(rng.uniform(-6, 0, (n, d))))
        for inv_tau in (2.0, 10.0, 20.0):
            scale2 = inv_tau * 1.4426950408889634
            worst = {}
            for name, (sx, sy) in (("old", (16.0, 256.0)), ("new", (0.25, 4.0))):
                errs = []
                for x, y in ((a, b), (c, c[::-1]), (a, c)):
                    xs, ys = (x * scale2 * sx).astype(np.float32), (y * sy).astype(np.float32)
                    xh, xl = split2(xs)
                    yh, yl = split2(ys)
                    got = ((xh * yh).sum(1) + (xh * yl).sum(1) + (xl * yh).sum(1)) / (sx * sy)
                    ref = (xs.astype(np.float64) * ys.astype(np.float64)).sum(1) / (sx * sy)
                    errs.append(np.abs(got - ref).max())
                worst[name] = max(errs)
            assert worst["new"] <= 1.1 * worst["old"] + 2e-7, (d, inv_tau, worst)
            assert worst["new"] < 3e-6 * inv_tau / 10.0 + 3e-7

"""The error bound of the SDAV similarity's int8 arg-min filter (csrc/gram_i8.hip, dlc_sim_window in csrc/gemm_internal.h),
restated in exact integer arithmetic on the CPU: for column-centred values v in [-0.498, 0.498], q = rint(v 2^24) as three
signed 8-bit digits, acc = C2 + ((C3 + (C4 >> 8)) >> 8) from the six kept digit products, the kernel's
d2 = rint(|v_b|^2 2^15) - acc (units of 2^-15) must lie within the bound Ed of the true |v_b|^2 - 2 v_a . v_b -- the bound
the kernel's window (2 Ed + 1e-8) is built on.  Worst-case-leaning data included (all digits at their extremes)."""
import numpy as np
import pytest


def digits(q):
    """q = s1 2^16 + s2 2^8 + s3 with s_i in [-128, 127] (sim_rows_kernel: the signed low byte, then (q + 128) >> 8)."""
    s3 = ((q + 128) & 255) - 128
    q1 = (q - s3) >> 8
    s2 = ((q1 + 128) & 255) - 128
    s1 = (q1 - s2) >> 8
    assert np.all(s1 * 65536 + s2 * 256 + s3 == q) and s1.min() >= -128 and s1.max() <= 127
    return s1, s2, s3


def bound_ed(sv, h):
    return 2.0 ** -23 * sv + h * (2.0 ** -24 + 2.0 ** -33 + 2.0 ** -46) + 1.004 * 2.0 ** -15 + 2.0 ** -16


@pytest.mark.parametrize("h,kind", [(2500, "uniform"), (2500, "extreme"), (64, "uniform"), (8192, "small"), (2500, "sign")])
def test_filter_error_bound_holds(h, kind):
    rng = np.random.RandomState(h + len(kind))
    n = 48
    if kind == "uniform":
        v = rng.uniform(-0.498, 0.498, size=(n, h))
    elif kind == "extreme":                      # digits at their extremes: q = +-(127 * 65536 + 127 * 256 + 127) and around
        v = rng.choice([-0.498, 0.498, 0.4980392, -0.4980392, 0.00390625 - 2.0 ** -25, -0.00390625], size=(n, h))
    elif kind == "small":
        v = rng.uniform(-1e-3, 1e-3, size=(n, h))
    else:
        v = 0.498 * np.sign(rng.standard_normal((n, h))) * rng.uniform(0.99, 1.0, size=(n, h))
    v = np.clip(v, -0.498, 0.498)
    q = np.rint(v * 2.0 ** 24).astype(np.int64)
    s1, s2, s3 = digits(q)
    sv = np.abs(v).sum(1).max()
    ed = bound_ed(sv, h)
    nb = np.rint((v * v).sum(1) * 2.0 ** 15).astype(np.int64)         # |v_b|^2 in units of 2^-15
    worst = 0.0
    for a in range(n):
        c2 = s1 @ s1[a]
        c3 = s2 @ s1[a] + s1 @ s2[a]
        c4 = s3 @ s1[a] + s2 @ s2[a] + s1 @ s3[a]
        assert max(np.abs(c2).max(), np.abs(c3).max(), np.abs(c4).max()) < 2 ** 31      # the int32 accumulators cannot overflow
        acc = c2 + ((c3 + (c4 >> 8)) >> 8)                              # arithmetic shifts: floors, as the epilogue's
        d2_kernel = (nb - acc) * 2.0 ** -15
        d2_true = (v * v).sum(1) - 2.0 * (v @ v[a])
        worst = max(worst, np.abs(d2_kernel - d2_true).max())
    assert worst <= ed, (worst, ed)
    assert worst >= 1e-3 * ed or kind == "small"                       # (the bound is not vacuous: within three decades of it)


def test_int32_accumulators_at_the_largest_width():
    """H = 32768 (the filter's limit): the class sums stay below 2^31 even with every digit at -128."""
    h = 32768
    worst_c4 = 3 * h * 128 * 128
    assert worst_c4 < 2 ** 31 and 2 * h * 128 * 128 < 2 ** 31
    nb_max = int(np.rint(0.498 ** 2 * h * 2.0 ** 15))
    assert nb_max + h * 128 * 128 + 2 ** 16 < 2 ** 31                   # d2 = nb - acc fits an int32 too

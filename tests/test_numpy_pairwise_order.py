"""The summation order the similarity's tie-breaking follows (csrc/gram_i8.hip sim_pairwise_program_kernel,
csrc/match_ref.hip stage 2 of the direct evaluation) is NumPy's: np.linalg.norm(x, axis=1) of the reference
(SimilarityCalculator.py:34) = sqrt(np.add.reduce(x * x, axis=1)) with pairwise summation.  This restates that
algorithm in plain Python exactly as the GPU runs it -- leaves of at most 128 elements in eight strided accumulators,
((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), the n % 8 last elements one by one, fewer than 8 elements one by one,
a longer range split at n / 2 rounded down to a multiple of 8 -- and holds it bit-equal to the NumPy installed here, so
that a NumPy that sums differently is noticed before a GPU near-tie test is."""
import numpy as np


def pairwise(a):
    n = len(a)
    if n < 8:
        r = 0.0
        for v in a:
            r = r + v
        return r
    if n <= 128:
        r = [a[i] for i in range(8)]
        m = n - (n % 8)
        for i in range(8, m, 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(m, n):
            res = res + a[i]
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise(a[:n2]) + pairwise(a[n2:])


def program(h):
    """The postfix program sim_pairwise_program_kernel builds for rows of h elements."""
    prog, stack = [], [(0, h, 0)]
    while stack:
        start, n, phase = stack.pop()
        if n <= 128:
            prog.append((start, n))
            continue
        n2 = n // 2
        n2 -= n2 % 8
        if phase == 0:
            stack.append((start, n, 1)); stack.append((start, n2, 0))
        elif phase == 1:
            stack.append((start, n, 2)); stack.append((start + n2, n - n2, 0))
        else:
            prog.append((-1, 0))
    return prog


def run_program(prog, a):
    st = []
    for start, n in prog:
        if start < 0:
            rhs, lhs = st.pop(), st.pop()
            st.append(lhs + rhs)
        else:
            st.append(pairwise(a[start:start + n]))
    assert len(st) == 1
    return st[0]


def test_pairwise_restatement_equals_numpy():
    rng = np.random.RandomState(0)
    for h in (1, 2, 7, 8, 9, 64, 78, 127, 128, 129, 250, 256, 300, 1000, 1031, 2500, 4096, 5000):
        x = rng.rand(6, h) * rng.choice([1.0, 1e-8, 1e6], size=(6, 1))
        s = x * x
        ref = np.add.reduce(s, axis=1)
        nrm = np.linalg.norm(x, axis=1)
        prog = program(h)
        assert len(prog) <= 1023 and max(n for _, n in prog) <= 128
        for r in range(6):
            row = [float(v) for v in s[r]]
            assert pairwise(row) == ref[r], h
            assert run_program(prog, row) == ref[r], h
            assert np.sqrt(ref[r]) == nrm[r], h


def test_program_depth_fits_the_kernel_stack():
    for h in (129, 2500, 32768):
        depth, top = 0, 0
        for start, n in program(h):
            depth = depth - 1 if start < 0 else depth + 1
            top = max(top, depth)
        assert top <= 16, (h, top)          # PF_STACK_DEPTH in csrc/match_ref.hip

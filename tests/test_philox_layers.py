"""CPU-only pins of the layers that exist in the counter-based back-ends only (the MT back-end, which is what ties the
race logic to the reference bit for bit, draws per lap as the reference does and has none of them):

  * the once-per-race retirement draw (reference src/simulation.py:190-197: a fresh `random.random() < p` per running
    car and lap; here ONE word per driver and race and a chain of integer survival thresholds) -- the law it
    implements is checked EXACTLY, with integer arithmetic, for every lap up to 1000, and the three statements of it
    (oracle, kernel header through the host build, this file) are compared word for word;
  * the inverse-normal transforms (reference :302,330 np.random.normal): every cell of the 448-row binary32 table and
    of the 784-row binary64 table, several points per cell, both signs, against scipy's ndtri, in the oracle's and the
    kernel header's statement;
  * the table generators assert the stated bounds themselves and regenerate the committed files.
"""
import ctypes as C
import subprocess
import sys
from fractions import Fraction

import numpy as np
import pytest

import kernel_host_build as K
import oracle_py as O

P_VALUES = [1e-6, 8.3e-4, 0.05 / 60, 0.05, 0.5, 1 - 2.0 ** -32, 1.0, 1.5]


def _oracle():
    L = O.lib()
    L.orc_retirement_lap.restype = C.c_int
    L.orc_retirement_lap.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_int]
    return L


def _emu():
    L = K.lib()
    L.emu_draw_retirement_lap.restype = C.c_uint32
    L.emu_draw_retirement_lap.argtypes = [C.c_uint32, C.c_double, C.c_int]
    for f in ('emu_threshold', 'emu_threshold53', 'emu_survival64'):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_double]
    L.emu_normal_from_u32.restype = C.c_float
    L.emu_normal_from_u32.argtypes = [C.c_uint32]
    L.emu_normal53.restype = C.c_double
    L.emu_normal53.argtypes = [C.c_uint32, C.c_uint32]
    return L


def _chain(p, bits, laps):
    """S_2 .. S_laps of the survival chain in `bits`-bit words, as Python integers: t = ceil(p 2^bits) (p 2^bits is exact
    in binary64), q = 2^bits - t, S_2 = q, S_{k+1} = floor(S_k q / 2^bits)."""
    one = 1 << bits
    t = min(one, -((-Fraction(p) * one).__floor__()))        # ceil of an exact rational
    q = one - t
    S, out = q, []
    for _ in range(2, laps + 1):
        out.append(S)
        S = (S * q) >> bits
    return t, out


@pytest.mark.parametrize('bits', [32, 64])
@pytest.mark.parametrize('p', P_VALUES)
def test_retirement_chain_is_the_per_lap_probability(p, bits):
    """A car is running at the start of lap k iff its word is below S_{k-1} (S_1 = 2^bits: every word) and retires on
    it iff the word is not below S_k: P(retire on lap k | running) = 1 - S_k / S_{k-1}, which must be the reference's
    per-lap probability p to within the resolution of a `bits`-bit draw: |.. - p| <= 2^-bits + 1 / S_{k-1}, for EVERY
    lap up to 1000 (exact rational arithmetic; the second term is the flooring of the chain, relative to the words
    still running)."""
    one = 1 << bits
    t, S = _chain(p, bits, 1000)
    E = _emu()
    if bits == 32:
        assert t == E.emu_threshold(p)                       # the threshold the parameter block carries
    elif p < 1.0:
        assert (one - t) % one == E.emu_survival64(p)        # ... and the 64-bit survival factor
    pe = min(Fraction(p), Fraction(1))
    prev = one
    for k, s in enumerate(S, start=2):
        if prev == 0:
            break                                            # nobody is running any more
        cond = 1 - Fraction(s, prev)
        assert abs(cond - pe) <= Fraction(1, one) + Fraction(1, prev), (p, bits, k)
        assert s <= prev
        prev = s
    if p >= 1.0:
        assert S[0] == 0                                     # a certain retirement: out on lap 2


@pytest.mark.parametrize('p', P_VALUES)
def test_retirement_lap_oracle_header_and_law_agree_word_for_word(p):
    """oracle/mcgp_oracle.c: retirement_lap_of_words, csrc/race_common.hip.h: draw_retirement_lap (compiled for the host,
    with the threshold csrc/params_build.h computes) and the law stated above, on 10^5 random words plus every word at
    a threshold S_k and just below it; 60- and 1000-lap races."""
    Lo, E = _oracle(), _emu()
    _, S = _chain(p, 32, 1000)
    rng = np.random.default_rng(int(p * 1e9) % 1000003)
    edge = [s for s in S[:300] if s < (1 << 32)] + [s - 1 for s in S[:300] if s >= 1] + [0, 1, (1 << 32) - 1]
    words = np.concatenate([rng.integers(0, 1 << 32, 100_000, dtype=np.uint64), np.array(edge, dtype=np.uint64)])
    Sa = np.array(S, dtype=np.float64)                       # (for the vectorised law: exact below 2^53)
    for laps in (60, 1000):
        # first k in 2..laps with w >= S_k, i.e. 2 + the number of thresholds above the word (S is non-increasing)
        survived = (words[:, None].astype(np.float64) < Sa[None, :laps - 1]).sum(axis=1) if laps == 60 else None
        for i, w in enumerate(words):
            w = int(w)
            a = Lo.orc_retirement_lap(w, 0, 0, p, laps)
            b = E.emu_draw_retirement_lap(w, p, laps)
            assert a == b, (p, w, laps, a, b)
            if survived is not None:
                want = 2 + int(survived[i])
                assert a == (want if want <= laps else 0), (p, w, a, want)
    # the reference's guard: p <= 0 (and NaN) never retires
    for bad in (0.0, -0.1, float('nan')):
        assert Lo.orc_retirement_lap(123, 0, 0, bad, 60) == 0 and E.emu_draw_retirement_lap(123, bad, 60) == 0


@pytest.mark.parametrize('p', P_VALUES)
def test_reference_width_retirement_lap_follows_the_64_bit_chain(p):
    """PHILOX53: the word refined to 53 bits, left-aligned in 64, against S_2 = q, S_{k+1} = floor(S_k q / 2^64)."""
    Lo = _oracle()
    _, S = _chain(p, 64, 200)
    rng = np.random.default_rng(5)
    ws = rng.integers(0, 1 << 32, 4000, dtype=np.uint64)
    xs = rng.integers(0, 1 << 21, 4000, dtype=np.uint64)
    # words that sit on a threshold of the chain, and just beside it
    for s in S[:60]:
        for d in (-1, 0, 1):
            v = min(max(s + d * (1 << 11), 0), (1 << 64) - 1)
            ws = np.append(ws, np.uint64(v >> 32))
            xs = np.append(xs, np.uint64((v >> 11) & 0x1FFFFF))
    for w, x in zip(ws, xs):
        w, x = int(w), int(x)
        Q = (w << 32) | (x << 11)
        want = next((k for k, s in enumerate(S[:59], start=2) if not Q < s), 0)
        assert Lo.orc_retirement_lap(w, x, 1, p, 60) == want, (p, w, x)


def _cell_words32(points=9):
    out = []
    for e in range(4, 32):
        for k in range(16):
            lo, width = (16 + k) << (e - 4), 1 << (e - 4)
            for j in range(points):
                m = lo + min(width - 1, (width * j) // (points - 1)) - 16
                out += [m, m | 0x80000000]
    return sorted(set(out))


def test_binary32_inverse_normal_over_every_cell_of_its_table():
    """Every one of the 448 rows, nine points per cell (first, last, between), both signs: oracle == kernel header
    bit for bit, and both within 4.8e-7 of scipy's ndtri at the draw's tail probability (m + 0.5) / 2^32."""
    from scipy.special import ndtri
    Lo, E = O.lib(), _emu()
    ws = _cell_words32()
    assert len(ws) > 448 * 2 * 5
    worst = 0.0
    for w in ws:
        a, b = Lo.orc_normal_from_u32(w), E.emu_normal_from_u32(w)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), hex(w)
        m = w & 0x7fffffff
        ex = float(ndtri((m + 0.5) / 2.0 ** 32)) * (-1.0 if w >> 31 else 1.0)
        worst = max(worst, abs(a - ex))
    assert worst <= 4.8e-7, worst


def test_binary64_inverse_normal_over_every_cell_of_its_table():
    """Every one of the 784 rows of the reference-width table (the 16 single-index rows, and first / middle / last
    index of every log-spaced cell), both signs: the oracle's normal53_tail == the kernel header's normal53 bit for
    bit, within 4e-15 (2 ulp at |z| = 8) of scipy's ndtri and of the independent erfc iteration."""
    from scipy.special import ndtri
    Lo, E = O.lib(), _emu()
    qs = list(range(16))
    for sh in range(48):
        for k in range(16):
            lo, width = (16 + k) << sh, 1 << sh
            qs += [lo, lo + width // 2, lo + width - 1]
    worst = 0.0
    for q in sorted(set(qs)):
        z = Lo.orc_normal53_tail(q)
        w, x = q >> 21, (q & 0x1FFFFF) << 11
        for sign in (0, 1):
            got = E.emu_normal53(w | (sign << 31), x | 0x3FF)          # (the companion's low 11 bits are not used)
            assert got == (-z if sign else z), (q, sign)
        ex = float(ndtri((q + 0.5) / 2.0 ** 53))
        worst = max(worst, abs(z - ex))
        assert abs(z - Lo.orc_phi_inverse_tail(q, Lo.orc_normal_from_u32(q >> 21))) <= 4e-15 * max(1.0, abs(ex)), q
    assert worst <= 4e-15, worst


@pytest.mark.parametrize('script', ['gen_normal_table.py', 'gen_normal53_table.py'])
def test_table_generators_assert_their_bounds_and_reproduce_the_committed_tables(script):
    """tools/gen_normal*_table.py --check: regenerates the table, ASSERTS the stated error bound over random words,
    special words and every cell, and compares the words with both committed copies (nothing is written)."""
    r = subprocess.run([sys.executable, O.ROOT + '/tools/' + script, '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'every cell' in r.stderr

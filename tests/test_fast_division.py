"""The identity behind K1's division by a wave-uniform divisor (cvgpuspeedup_amd/csrc/k_taps.hpp, div_by_uniform):
q0 = x*r, two FMA correction steps, r = RN(1/d)  ==  the IEEE quotient RN(x/d), bit for bit.

Theory: after the first correction q1 is within half an ulp (+2^-45) of x/d, so e1 = x - d*q1 is exact and the second
step is Markstein's correction (1990), which yields the correctly rounded quotient when r is within half an ulp of 1/d
(all-ones divisor significands excepted -- the product's host guard sends those to the real division).
This test is the empirical side: EVERY divisor significand (2^23 of them, exponents swept over the guarded range) against
dividends just above / below powers of two, random ones and near-tie ones, through the oracle's checker (fmaf)."""
import numpy as np


def test_every_divisor_significand(oracle):
    lib = oracle.load_oracle()
    bad = lib.oracle_fastdiv_mismatches(0, 1 << 23, 12, 0xC0FFEE, 2)
    assert bad == 0


def test_more_dividends_on_a_sample_of_divisors(oracle):
    lib = oracle.load_oracle()
    rng = np.random.default_rng(5)
    for start in rng.integers(0, (1 << 23) - 4096, size=24):
        assert lib.oracle_fastdiv_mismatches(int(start), int(start) + 4096, 512, int(start) * 7 + 1, 2) == 0
    # the K1 test chain's own divisors (3.2, 0.6, 11.8, 33): many dividends each
    for d in (3.2, 0.6, 11.8, 33.0):
        sig = int(np.float32(d).view(np.uint32)) & 0x7FFFFF
        assert lib.oracle_fastdiv_mismatches(sig, sig + 1, 4_000_000, sig, 2) == 0

"""The multiply-high division of the GEMM operand loaders (csrc/gemm.h: fastdiv_magic / fdiv): m = ceil(2^32 / d),
q = (n * m) >> 32 (+ n when d == 1, whose magic does not fit 32 bits) is the exact quotient while n * d < 2^32 -- the
bound the launchers' index ranges stay inside (rows x row length of every conv geometry used here)."""
import random


def magic(d):
    return 0 if d <= 1 else ((1 << 32) + d - 1) // d


def fdiv(n, d, m):
    one = 0xFFFFFFFF if m == 0 else 0
    return (((n * m) >> 32) + (n & one)) & 0xFFFFFFFF


def test_exact_below_the_bound():
    rng = random.Random(0)
    for d in list(range(1, 70)) + [81, 100, 256, 400, 512, 576, 3136, 7056, 28224]:
        m = magic(d)
        assert m < (1 << 32)
        top = ((1 << 32) - 1) // d
        probes = {0, 1, d - 1, d, d + 1, top, max(top - 1, 0)} | {rng.randrange(0, top + 1) for _ in range(2000)}
        probes |= {k * d + r for k in (1, 7, top // d if d > 1 else 3) for r in (-1, 0, 1) if 0 <= k * d + r <= top}
        for n in probes:
            assert fdiv(n, d, m) == n // d, (n, d)


def test_the_geometries_of_the_learner_fit_the_bound():
    # (largest index, divisor) pairs of the loaders at batch 4096: im2col rows / output pixels, reduction / channels
    B = 4096
    for n, d in [(B * 400, 400), (400, 20), (B * 81, 81), (81, 9), (B * 49, 49), (49, 7), (512, 128), (576, 192),
                 (B * 100, 100), (100, 10), (512 + 3 * 64, 64), (16, 2), (16, 3), (B * 400 + 64, 400)]:
        assert n * d < (1 << 32), (n, d)

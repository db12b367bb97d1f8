"""The division the kernels' refill uses for launch-uniform divisors (kernels.hip div_uniform, reciprocals from
api_launch.cpp enqueue): q = mulhi(x, floor(2^32 / d)), one correction step - restated in numpy and checked against integer
division over the divisors a launch can have (image widths, tile counts, frames per launch) and over random ones."""
import numpy as np


def rcp32(d):
    return 0xffffffff if d <= 1 else (1 << 32) // d


def div_uniform(x, d):
    m = np.uint64(rcp32(d))
    q = (x.astype(np.uint64) * m) >> np.uint64(32)
    r = x.astype(np.uint64) - q * np.uint64(d)
    return np.where(r >= d, q + 1, q).astype(np.uint64)


def test_exact_for_every_kind_of_divisor():
    rng = np.random.default_rng(5)
    edge = np.array([0, 1, 2, 3, 63, 64, 65, 2**16 - 1, 2**16, 2**31 - 1, 2**31, 2**32 - 2, 2**32 - 1], dtype=np.uint64)
    divisors = [1, 2, 3, 5, 7, 8, 60, 64, 135, 240, 480, 1080, 1920, 2160, 3840, 7680, 32400, 129600, 2073600, 2**16 + 1, 2**31 - 1, 2**31,
                2**32 - 1] + [int(v) for v in rng.integers(1, 2**32, 200)]
    for d in divisors:
        x = np.concatenate([edge, rng.integers(0, 2**32, 4000, dtype=np.uint64),
                            (np.arange(-3, 4) + np.uint64(d) * rng.integers(0, max(2**32 // d, 1), 1, dtype=np.uint64)).astype(np.int64).clip(0, 2**32 - 1).astype(np.uint64)])
        assert np.array_equal(div_uniform(x, d), x // np.uint64(d)), d

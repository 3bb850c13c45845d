"""include/introsort_emul.h must reproduce GCC libstdc++ std::sort's permutation
(duplicates included) -- compared against the real std::sort inside liboracle.so."""
import numpy as np
import pytest


def _median3_killer(n):
    # Musser's median-of-3 killer: drives introsort into its heapsort fallback
    k = n // 2
    a = np.zeros(n, np.uint64)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def test_emulation_equals_std_sort(built):
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    total = 0
    for n in list(range(0, 70)) + [100, 127, 128, 129, 255, 1000, 4097, 20000, 100000]:
        for hi in (2, 5, 50, 1 << 20, 1 << 60):
            keys = rng.integers(0, hi, size=n, dtype=np.uint64)
            assert O.introsort_mismatches(keys) == 0, (n, hi)
            total += 1
    for n in (17, 64, 1000, 5000):
        for arr in (np.arange(n), np.arange(n)[::-1], np.zeros(n), np.arange(n) % 3, np.arange(n) // 7,
                    np.concatenate([np.arange(n // 2), np.arange(n - n // 2)])):
            assert O.introsort_mismatches(arr.astype(np.uint64)) == 0
    assert total > 300


@pytest.mark.parametrize("n", [64, 1000, 20000, 200000])
def test_heapsort_fallback_path(built, n):
    from oracle import oracle as O
    a = _median3_killer(n)
    assert O.introsort_mismatches(a) == 0
    # the same with heavy duplication
    assert O.introsort_mismatches(a // 3) == 0

"""CPU restatement of wave_sort64 (cuvs_amd/csrc/ivf_common.hpp): the 64-lane bitonic network a scan kernel's wave runs over the
(distance, row) pairs of its first tile instead of 64 serial insertions. Lane by lane the device code's compare-exchange rule
(partner = lane ^ j2, keep the smaller pair in the lower lane of an ascending block) - checked against a plain sort for random
distances, many equal distances (ties break by row), +inf padding in any number of lanes, and already sorted / reversed input."""
import numpy as np
import pytest

pytestmark = pytest.mark.filterwarnings("ignore:overflow encountered")


def wave_sort64(d, i):
    d, i = d.copy(), i.copy()
    lane = np.arange(64)
    k2 = 2
    while k2 <= 64:
        j2 = k2 >> 1
        while j2 > 0:
            od, oi = d[lane ^ j2], i[lane ^ j2]           # __shfl_xor
            other_less = (od < d) | ((od == d) & (oi < i))
            same = (od == d) & (oi == i)
            take_min = ((lane & j2) == 0) == ((lane & k2) == 0)
            swap = np.where(take_min, other_less, ~other_less & ~same)
            d, i = np.where(swap, od, d), np.where(swap, oi, i)
            j2 >>= 1
        k2 <<= 1
    return d, i


def reference(d, i):
    order = np.lexsort((i, d))  # by distance, then by row
    return d[order], i[order]


@pytest.mark.parametrize("seed", range(40))
def test_network_equals_a_sort(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 5
    if kind == 0:
        d = rng.standard_normal(64).astype(np.float32)
    elif kind == 1:
        d = rng.integers(0, 4, 64).astype(np.float32)          # integer data: many exact ties
    elif kind == 2:
        d = np.sort(rng.standard_normal(64).astype(np.float32))[::-1].copy()
    elif kind == 3:
        d = np.full(64, 7.0, np.float32)                        # all equal: pure row order
    else:
        d = np.abs(rng.standard_normal(64).astype(np.float32)) * np.float32(1e30) * np.float32(1e10)  # overflows to +inf in places
    i = rng.permutation(1 << 20)[:64].astype(np.uint32)
    pad = rng.random(64) < (0.0, 0.1, 0.5, 0.9, 0.3)[kind]      # lanes without a candidate: (+inf, 0xffffffff)
    d[pad], i[pad] = np.inf, 0xFFFFFFFF
    gd, gi = wave_sort64(d, i)
    wd, wi = reference(d, i)
    assert (gd == wd).all() and (gi == wi).all()
    assert sorted(zip(gd.tolist(), gi.tolist())) == sorted(zip(d.tolist(), i.tolist()))  # a permutation: nothing lost or doubled

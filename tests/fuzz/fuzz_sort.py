#!/usr/bin/env python3
"""Differential fuzz of the exact std::sort (order.hip) against libstdc++'s own through the oracle: random lengths from 1 to
90,000 keys (most below 6,000, some around the 4,096-key workgroup threshold, the 19,114-key list limit and the 28,672 keys LDS
holds), score distributions from all-equal over a handful of values (97 % ties, as on real frames) to all-distinct, sorted,
reversed, organ-pipe, block-constant and sawtooth inputs, occasionally McIlroy's adversary (the depth-limit heapsort) -- every
case through the one-launch form and through the two-launch form with a random number of piece workgroups (round 5:
sort_split_kernel + sort_pieces_kernel, the chunk regimes of the partitions).  Test infrastructure.
usage: tests/fuzz/fuzz_sort.py [cases] [seed]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding


def scores(rng, oracle, n):
    kind = int(rng.integers(0, 10))
    i = np.arange(n)
    if kind == 0:
        return np.full(n, int(rng.integers(0, 4096)))
    if kind == 1:
        return rng.integers(0, int(rng.integers(2, 12)), n)             # a handful of values
    if kind == 2:
        return rng.integers(0, int(rng.choice([40, 300, 1600, 4096])), n)
    if kind == 3:
        return np.minimum(i, n - 1 - i) % 4096                              # organ pipe
    if kind == 4:
        return (i // int(rng.integers(1, 200))) % 4096                      # block-constant, ascending
    if kind == 5:
        return (n - i) % 4096                                               # descending (already sorted for `greater`)
    if kind == 6:
        return (i * 2654435761 >> int(rng.integers(3, 12))) % int(rng.integers(2, 4096))
    if kind == 7:
        base = rng.integers(0, 4096, n)
        base[rng.random(n) < 0.97] = int(rng.integers(0, 4096))             # one dominant value
        return base
    if kind == 8 and n <= 6000:
        v = oracle.antiqsort(n)                                             # drives introsort into its heapsort fallback
        return np.minimum((n - 1) - v if rng.integers(0, 2) else v, 4095)
    return np.sort(rng.integers(0, 4096, n))[:: int(rng.choice([1, -1]))]   # sorted either way


def main(cases=None, seed=None):
    """cases / seed: from the command line when None (tests/test_fuzz_smoke_gpu.py passes fixed ones)."""
    if cases is None:
        cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    slamx = importlib.import_module("slam-cin0051_amd")
    oracle = oracle_binding.load()
    ctx = slamx.Context(0)
    keys = 0
    for case in range(cases):
        rng = np.random.default_rng([seed, case])
        pick = int(rng.integers(0, 20))
        if pick < 12:
            n = int(rng.integers(1, 6000))
        elif pick < 15:
            n = int(rng.choice([4096, 19114, 28672]) + rng.integers(-3, 4))
        elif pick < 19:
            n = int(rng.integers(6000, 30000))
        else:
            n = int(rng.integers(30000, 90000))
        s = np.ascontiguousarray(scores(rng, oracle, n), np.uint32)
        ref = oracle.sort_desc(s)
        for groups in (1, int(rng.choice([2, 3, 5, 8, 16]))):
            got = ctx.debug_sort_desc(s, groups)
            if not np.array_equal(got, ref):
                bad = int(np.nonzero(got != ref)[0][0])
                raise AssertionError("case %d (seed %d): n = %d, %d group(s): permutation differs first at %d" % (case, seed, n, groups, bad))
        keys += n
    print("fuzz_sort: %d cases (%d keys) identical to std::sort in both forms (seed %d)" % (cases, keys, seed))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Experiment (GPU box): xf_sort_key_pos alone — 10^7 hashed keys, with the bench stream's hot
field, Zipf(1.1), 1.25e6 and 3e7 keys — beside the library's radix sort (xf_tune key_build = 1),
checked against numpy's stable argsort.
    python tools/r6/sort_probe.py [repeat]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xflow_amd import capi  # noqa: E402

repeat = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calibrate = "--pmc-calibrate" in sys.argv       # (tools/pmc2.sh: known-traffic patterns, same run)
hand_only = "--by-hand" in sys.argv             # (no library half: the counters' file stays small)
rng = np.random.RandomState(1)
cases = ((10_000_000, 0, 0), (10_000_000, 32, 0), (10_000_000, 0, 1.1), (1_250_000, 0, 0),
         (30_000_000, 0, 0))
if "--first" in sys.argv:                       # (the 1e7 hashed keys alone: per-launch counters)
    cases = cases[:1]
if "--zipf" in sys.argv:
    cases = cases[2:3]
for n, hot, zipf in cases:
    pool = rng.randint(0, 2**63, size=n).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    if zipf:
        keys = pool[np.minimum(rng.zipf(zipf, size=n), n) - 1]
    else:
        keys = pool[rng.randint(0, int(n * 0.8), size=n)]
    if hot:
        hk = rng.randint(0, 2**63, size=hot).astype(np.uint64) * np.uint64(2)
        keys[rng.randint(0, n, size=50000)] = hk[rng.randint(0, hot, size=50000)]
    sk, sp, h, ms = capi.sort_key_pos(keys, repeat=repeat)
    order = np.argsort(keys, kind="stable")
    ok = np.array_equal(sp, order.astype(np.uint32)) and np.array_equal(sk, keys[order])
    ms2 = float("nan")
    if not hand_only:
        capi.tune("key_build", 1)
        _, _, h2, ms2 = capi.sort_key_pos(keys, repeat=repeat)
        capi.tune("key_build", 0)
    print("n %d hot %d zipf %s: by hand %s %.3f ms (equals numpy's stable argsort: %s); "
          "library %.3f ms" % (n, hot, zipf, h, ms, ok, ms2), flush=True)
if calibrate:
    for kind in range(10):
        capi.check(capi.lib().xf_calib_stream(kind, 1 << 30, 3))

#!/usr/bin/env python3
"""What one shard owner does per step at N source ranks, measured on one GPU: N sorted key
lists (a random 11.7 % of the shard's 1e7 keys each, as at N = 8 in bench.py) resolved + read
and then updated (a) one pass per source, (b) merged by key in one pass.
Evidence for DESIGN.md section 6."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xflow_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K, PER = 10_000_000, 1_170_000
torch.cuda.set_device(0)
L = capi.lib()
s = torch.cuda.current_stream().cuda_stream
keytab = np.sort(capi.hash_decimal_range(0, K))
t = capi.Table(capi.OPT_FTRL, 1, capacity=2 * K + 1024)
allk = torch.from_numpy(keytab.view(np.int64)).cuda()
rows = torch.empty(K, dtype=torch.int32, device="cuda")
t.resolve_dev(allk.data_ptr(), K, rows.data_ptr())
t.check()
t.defrag()
rng = np.random.RandomState(0)
lists = [np.sort(rng.choice(K, PER, replace=False)) for _ in range(N)]
cat = torch.from_numpy(np.concatenate([keytab[i] for i in lists]).view(np.int64)).cuda()
n = cat.numel()
flip = torch.iinfo(torch.int64).min
srt, order = torch.sort(cat ^ flip, stable=True)
ksorted, order = srt ^ flip, order.to(torch.int32)
slots = torch.empty(n, dtype=torch.int32, device="cuda")
vals = torch.empty(n, dtype=torch.float32, device="cuda")
g = torch.randn(n, device="cuda") * 1e-4


def timed(name, fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    print("%-44s %8.1f us" % (name, e0.elapsed_time(e1) / reps * 1e3), flush=True)


def pull_concat():
    capi.check(L.xf_table_pull_dev(t.h, cat.data_ptr(), n, slots.data_ptr(), vals.data_ptr(), s))


def pull_merged():
    capi.check(L.xf_table_pull_ordered_dev(t.h, ksorted.data_ptr(), order.data_ptr(), n,
                                           slots.data_ptr(), vals.data_ptr(), s))


def update_per_source():
    for r in range(N):
        o = r * PER
        capi.check(L.xf_table_update_dev(t.h, slots.data_ptr() + 4 * o, PER,
                                         g.data_ptr() + 4 * o, s))


def update_merged():
    capi.check(L.xf_table_update_merged_dev(t.h, ksorted.data_ptr(), order.data_ptr(), n,
                                            slots.data_ptr(), g.data_ptr(), s))


print("%d source lists x %d keys, shard of %d keys (LR + FTRL)" % (N, PER, K))
timed("Pull, lists one after the other", pull_concat)
timed("Pull, merged by key", pull_merged)
timed("Push, one pass per source (rank order)", update_per_source)
timed("Push, merged (a key's sources in rank order)", update_merged)

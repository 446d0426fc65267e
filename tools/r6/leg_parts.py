"""Experiment: where a steady-state minibatch of the first-epoch leg spends its host time
(update call / free / the wait for the table size)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from xflow_amd import capi  # noqa: E402
from xflow_amd.single import SingleGpuTrainer  # noqa: E402

L = capi.lib()
R, nnz, nkeys = 50000, 200, 10_000_000
keytab = torch.from_numpy(bench.make_key_table(nkeys).view(np.int64)).cuda()
g = torch.Generator(device="cuda")
g.manual_seed(5)
rp = torch.arange(0, (R + 1) * nnz, nnz, dtype=torch.int32, device="cuda")
raw = [(keytab[torch.randint(0, nkeys, (R * nnz,), generator=g, device="cuda")],
        torch.randint(0, 2, (R,), generator=g, device="cuda", dtype=torch.int32)) for _ in range(30)]
tr = SingleGpuTrainer(model="lr", optimizer="ftrl", capacity=2 * nkeys)
capi.check(L.xf_scratch_reserve(R * nnz * 40 + (64 << 20)))
prev = None
parts = []
for i, (k, lb) in enumerate(raw):
    t0 = time.perf_counter()
    h = capi.vp()
    capi.check(L.xf_lr_update_dev(C.byref(h), tr.w.h, k.data_ptr(), rp.data_ptr(), lb.data_ptr(),
                                  R, R * nnz, 0, tr.ws.h, None))
    t1 = time.perf_counter()
    if prev is not None:
        L.xf_batch_free(prev)
    prev = h
    t2 = time.perf_counter()
    n = len(tr.w)
    t3 = time.perf_counter()
    if i in (0, 1, 2, 4, 8):
        tr.defrag()
        L.xf_batch_free(prev)
        prev = None
    parts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n - tr.w.settled))
for p in parts:
    print("update %.3f  free %.3f  size %.3f   unsettled keys %d" % p)

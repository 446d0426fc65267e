import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xflow_amd import capi
rows, nnz, K = 100000, 200, 10_000_000
d = tempfile.mkdtemp(); path = os.path.join(d, "t-00000")
rng = np.random.RandomState(0)
fid = rng.randint(0, K, size=(rows, nnz)); lab = rng.randint(0, 2, size=rows)
with open(path, "w") as f:
    for r in range(rows):
        f.write("%d\t" % lab[r] + " ".join("%d:%d:1" % (j & 31, v) for j, v in enumerate(fid[r])) + "\n")
mb = os.path.getsize(path) / 1e6
L = capi.lib(); C = capi.C
for nt in (1, 8, 16, 32, 64, 128):
    capi.tune("parse_threads", nt)
    best = 1e9
    for rep in range(3):
        h = capi.vp(); capi.check(L.xf_reader_open(C.byref(h), path.encode(), 64 << 20))
        t0 = time.perf_counter(); tot = 0
        while True:
            rws, nz = C.c_size_t(0), C.c_size_t(0)
            capi.check(L.xf_reader_next(h, C.byref(rws), C.byref(nz), None, None, None, None))
            if rws.value == 0: break
            tot += rws.value
        best = min(best, time.perf_counter() - t0)
        L.xf_reader_close(h)
    print("threads %3d: %.3f s  %.0f MB/s  %.0f rows/s" % (nt, best, mb / best, tot / best), flush=True)

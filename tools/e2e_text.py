#!/usr/bin/env python3
"""End-to-end from libsvm-style text through the xflow_lr CLI: synthetic file -> parse (host,
multi-threaded) -> key build (GPU) -> steps (GPU) -> predict/AUC.  Tuning/measurement aid."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows, nnz, K = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 200, 10_000_000
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
t0 = time.time()
for name, n in (("train-00000", rows), ("test-00000", rows // 10)):
    fid = rng.randint(0, K, size=(n, nnz))
    lab = rng.randint(0, 2, size=n)
    with open(os.path.join(d, name), "w") as f:
        for r in range(n):
            f.write("%d\t" % lab[r] + " ".join("%d:%d:1" % (j & 31, v) for j, v in enumerate(fid[r])) + "\n")
print("generated %d rows (%.0f MB) in %.1f s" % (rows, os.path.getsize(os.path.join(d, "train-00000")) / 1e6, time.time() - t0), flush=True)
# (first run: cold page cache / first HIP start; block_cache=1 twice: build the cache, use it)
for epochs, extra in ((1, []), (1, []), (4, []), (1, ["block_cache=1"]), (1, ["block_cache=1"])):
    t0 = time.time()
    out = subprocess.run([os.path.join(ROOT, "xflow_amd/lib/xflow_lr"), os.path.join(d, "train"), os.path.join(d, "test"),
                          "0", str(epochs), "block_size_mb=64", "capacity=30000000", "pred_path=" + os.path.join(d, "pred.txt")] + extra,
                         capture_output=True, text=True)
    print("epochs=%d %s wall %.2f s :: %s" % (epochs, extra, time.time() - t0, " | ".join(out.stdout.strip().splitlines()[-3:])), out.stderr[-300:], flush=True)

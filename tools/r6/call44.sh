#!/bin/bash
# round 6, call 44: xf_sort_key_pos with the heavy ranges' merge sort (k_sp_parts / k_sp_merge):
# its tests, the builds that use it, times on the uniform, the hot-field and the Zipf stream
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -m xflow_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 900 python -m pytest tests/test_gpu_keybuild.py -m gpu -x -q -k "sort_key_pos" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_key_build or device_built" 2>&1 | tail -5
timeout 600 python - <<'PY'
import numpy as np
from xflow_amd import capi
rng = np.random.RandomState(1)
for n, hot, zipf in ((10_000_000, 0, 0), (10_000_000, 32, 0), (1_250_000, 0, 0), (10_000_000, 0, 1.1), (30_000_000, 0, 0)):
    pool = rng.randint(0, 2**63, size=n).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    if zipf:
        keys = pool[np.minimum(rng.zipf(zipf, size=n), n) - 1]
    else:
        keys = pool[rng.randint(0, int(n * 0.8), size=n)]
    if hot:
        hk = rng.randint(0, 2**63, size=hot).astype(np.uint64) * np.uint64(2)
        at = rng.randint(0, n, size=50000)
        keys[at] = hk[rng.randint(0, hot, size=50000)]
    sk, sp, h, ms = capi.sort_key_pos(keys, repeat=20)
    order = np.argsort(keys, kind="stable")
    ok = np.array_equal(sp, order.astype(np.uint32)) and np.array_equal(sk, keys[order])
    capi.tune("key_build", 1)
    _, _, h2, ms2 = capi.sort_key_pos(keys, repeat=20)
    capi.tune("key_build", 0)
    print("n %d hot %d zipf %s: by hand %s %.3f ms (ok %s); library %.3f ms" % (n, hot, zipf, h, ms, ok, ms2))
PY

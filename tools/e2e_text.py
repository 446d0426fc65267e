#!/usr/bin/env python3
"""End to end through the `xflow_lr` binary on a synthetic libsvm-style file (GPU box):
text -> multi-threaded block parser (host) -> key build (GPU) -> steps (GPU) -> predict / AUC,
first epoch from the text, first epoch from the binarized block cache, and the average over 4
epochs (later epochs replay the compiled minibatches from HBM).  Writes the numbers as JSON
(profiles/e2e_latest.json is what bench.py quotes as `end_to_end`).
    python tools/e2e_text.py [rows] [out.json]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "e2e_latest.json")
nnz, K = 200, 10_000_000
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
t0 = time.time()
CHUNK = 100000   # the text is generated once for this many rows and written out repeatedly
for name, n in (("train-00000", rows), ("test-00000", min(rows // 10, 40000))):
    m = min(n, CHUNK)
    fid = rng.randint(0, K, size=(m, nnz))
    lab = rng.randint(0, 2, size=m)
    lines = ["%d\t" % lab[r] + " ".join("%d:%d:1" % (j & 31, v) for j, v in enumerate(fid[r])) + "\n"
             for r in range(m)]
    with open(os.path.join(d, name), "w") as f:
        left = n
        while left > 0:
            f.write("".join(lines[:min(left, m)]))
            left -= m
size_mb = os.path.getsize(os.path.join(d, "train-00000")) / 1e6
print("generated %d rows (%.0f MB) in %.1f s" % (rows, size_mb, time.time() - t0), flush=True)


def run(epochs, extra):
    t0 = time.time()
    out = subprocess.run([os.path.join(ROOT, "xflow_amd/lib/xflow_lr"), os.path.join(d, "train"),
                          os.path.join(d, "test"), "0", str(epochs), "block_size_mb=64",
                          "capacity=30000000", "pred_path=" + os.path.join(d, "pred.txt")] + extra,
                         capture_output=True, text=True,
                         env=dict(os.environ, XF_TRACE_WORKER="1"))
    wall = time.time() - t0
    m = re.search(r"examples/sec \(train loop\): ([0-9.e+]+)", out.stdout)
    eps = float(m.group(1)) if m else None
    # per-block timeline (XF_TRACE_WORKER): rows / (wait for the parser + key build + step)
    blk = re.findall(r"block: (\d+) rows  waited for the parser ([0-9.]+) ms  key build "
                     r"([0-9.]+) ms  step\+check ([0-9.]+) ms", out.stderr)
    steady = None
    if len(blk) > 4:
        per = sorted(float(w) + float(b) + float(s_) for _, w, b, s_ in blk[2:-1])
        steady = float(blk[2][0]) / (per[len(per) // 2] * 1e-3)
    run.steady = steady
    print("epochs=%d %s wall %.2f s train-loop %s ex/s, median block %s ex/s\n%s" % (
        epochs, extra, wall, eps, steady,
        out.stderr if os.environ.get("E2E_TRACE") == "2" else
        out.stderr[-1200:] if os.environ.get("E2E_TRACE") else ""),
        flush=True)
    return eps, wall


run(1, [])                                   # cold page cache / first HIP start: not recorded
text1, w_text = run(1, [])
text_steady = run.steady
text4, _ = run(4, [])
gpu1, w_gpu = run(1, ["ingest=gpu"])         # the text tokenised and hashed on the GPU
gpu_steady = run.steady
gpu4, _ = run(4, ["ingest=gpu"])
run(1, ["block_cache=1"])                    # builds the cache
cache1, w_cache = run(1, ["block_cache=1"])  # uses it
cache_steady = run.steady
res = {"what": "xflow_lr on a synthetic libsvm-style file: %d rows x %d tokens, %.0f MB (a "
               "%d-row random chunk written out repeatedly), 64 MiB blocks, LR + FTRL, key space "
               "1e7; examples/sec of the train loop (parse + key build + steps), predict "
               "excluded" % (rows, nnz, size_mb, min(rows, CHUNK)),
       "host": {"nproc": os.cpu_count()},
       "first_epoch_from_text": text1, "first_epoch_from_block_cache": cache1,
       "first_epoch_from_text_gpu_tokeniser": gpu1,
       "median_block_rate_from_text_gpu_tokeniser": gpu_steady,
       "average_over_4_epochs_from_text_gpu_tokeniser": gpu4,
       "median_block_rate_from_text": text_steady, "median_block_rate_from_block_cache": cache_steady,
       "note": "first_epoch_* = rows / wall time of the training loop of a fresh process that "
               "runs ONE epoch (its first block still pins the block buffers and sizes the build "
               "scratch; code loading happens at start-up, before the loop); "
               "median_block_rate_* = rows of a block / (wait for the parser + key build incl. "
               "upload + step) for the median block of that epoch",
       "average_over_4_epochs_from_text": text4,
       "wall_seconds_incl_predict": {"text_1_epoch": w_text, "cache_1_epoch": w_cache,
                                     "text_1_epoch_gpu_tokeniser": w_gpu},
       "git_head": (subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                                   capture_output=True, text=True).stdout.strip() or
                    (open(os.path.join(ROOT, ".git_head")).read().strip()
                     if os.path.exists(os.path.join(ROOT, ".git_head")) else None)),
       "measured_by": "tools/e2e_text.py"}
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res))

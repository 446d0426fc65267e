#!/usr/bin/env python3
"""Regenerate the reference-derived golden vectors in tests/golden/.

Runs ONLY in the build container (needs /root/reference and oracle/_ref, the subset of
the real reference that compiles from its own sources: src/io + src/base).  The outputs
are data (inputs + expected outputs); no reference source text is stored.

  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root
from oracle import pyoracle as O  # noqa: E402

REF_DATA = "/root/reference/data"


def main():
    assert O.ref_available(), "build oracle/_ref first: make -C oracle"
    R = O.ref()
    out = {}
    # (1) key-hash known answers: SURVEY §8c KATs + more strings through the real
    #     std::hash<std::string> used at src/io/io.h:53
    strs = ["0", "1163", "185", "8672", "7755", "123456789", "1234567890123", "", "7",
            "12345678", "99999999", "100000000", "abcdefghijklmnopqrstuvwxyz",
            "9999999", "4294967296"]
    out["hash"] = {s: "%016x" % R.ref_hash(s.encode(), len(s)) for s in strs}
    # (2) sigmoid table through the real Base::sigmoid (src/base/base.h:54-63)
    xs = [-31.0, -30.0, -29.99, -10.0, -1.0, -1e-3, 0.0, 1e-3, 0.5, 1.0, 10.0, 29.99,
          30.0, 30.01, 31.0]
    out["sigmoid"] = [[x, float(R.ref_sigmoid(np.float32(x))).hex()] for x in xs]
    # (3) AUC / logloss line through the real Base::calculate_auc (base.h:84-110)
    rng = np.random.RandomState(20260926)
    lab = (rng.rand(500) < 0.3).astype(np.int32)
    p = np.clip(rng.rand(500) * 0.8 + 0.1 * lab, 1e-3, 0.999).astype(np.float32)
    p[::7] = p[3]  # ties exercise std::sort's order
    ll, line = O.ref_auc(lab, p)
    out["auc"] = {"labels": lab.tolist(), "pctr_hex": [float(v).hex() for v in p],
                  "logloss_hex": float(ll).hex(), "line": line}
    with open(os.path.join(HERE, "ref_kats.json"), "w") as f:
        json.dump(out, f, indent=1)
    # (4) parser output of the real load_minibatch_hash_data_fread for the sample files,
    #     at the reference's block sizes and at a tiny block (boundary / carry cases)
    for name in ("small_train-00000", "small_test-00000"):
        path = os.path.join(REF_DATA, name)
        for cap in (2 << 20, 4096, 1000):
            blocks = list(O.ref_read_blocks(path, cap))
            rows = np.array([len(b[3]) for b in blocks], dtype=np.int64)
            np.savez_compressed(
                os.path.join(HERE, "ref_parse_%s_cap%d.npz" % (name, cap)),
                block_rows=rows,
                rowptr=np.concatenate([[0]] + [np.diff(b[0]) for b in blocks]).cumsum()
                .astype(np.uint64),
                keys=np.concatenate([b[1] for b in blocks]),
                fgid=np.concatenate([b[2] for b in blocks]),
                labels=np.concatenate([b[3] for b in blocks]))
    # (5) a synthetic input of our own that holds the parser's less obvious behaviours — empty
    #     tokens (consecutive blanks, a blank before the block terminator) duplicating the
    #     previous token, labels around the 1e-7 threshold, alphanumeric fids, float fgids —
    #     with the real parser's output at several block sizes.  Same generator as the fuzz in
    #     tests/test_capi_cpu.py, fixed seed.
    from tests.test_capi_cpu import _random_libsvm_text
    rng = np.random.RandomState(424242)
    text = _random_libsvm_text(rng, 120)
    lines = text.split("\n")
    lines[3] = lines[3].rstrip(" ") + "  " + lines[4].split("\t")[1].split(" ")[0]  # two blanks
    lines[10] = "0.5\t3.7:abc:1 4:0:0"                                               # fgid 3.7
    text = "\n".join(lines)
    with open(os.path.join(HERE, "quirks-00000"), "w") as f:
        f.write(text)
    longest = max(len(l) for l in lines) + 2
    for cap in (longest + 1, 777, 1 << 20):
        blocks = list(O.ref_read_blocks(os.path.join(HERE, "quirks-00000"), cap))
        np.savez_compressed(
            os.path.join(HERE, "ref_parse_quirks_cap%d.npz" % cap),
            block_rows=np.array([len(b[3]) for b in blocks], dtype=np.int64),
            rowptr=np.concatenate([[0]] + [np.diff(b[0]) for b in blocks]).cumsum()
            .astype(np.uint64),
            keys=np.concatenate([b[1] for b in blocks]),
            fgid=np.concatenate([b[2] for b in blocks]),
            labels=np.concatenate([b[3] for b in blocks]))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

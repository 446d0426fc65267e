"""CPU-side checks of the product library: it loads, exports every symbol the header
declares, and its host logic (hash, block reader, key build, metrics) matches the golden
vectors and the oracle.  No compute kernels run here (there is no GPU)."""
import json
import os
import re

import numpy as np
from ctypes import c_float as C_float
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    from xflow_amd import build
    build.build(verbose=False)


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "xflow_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(xf_[a-z0-9_]+|XF[A-Z][A-Za-z]+)\s*\(", hdr))
    assert len(names) >= 45
    L = capi.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert names == set(capi.SIGNATURES), names ^ set(capi.SIGNATURES)


def test_hash_kats(golden_dir):
    kats = json.load(open(os.path.join(golden_dir, "ref_kats.json")))
    for s, hx in kats["hash"].items():
        assert capi.hash_str(s) == int(hx, 16), s
    rng = np.random.RandomState(5)
    for _ in range(500):
        b = bytes(rng.randint(1, 255, size=rng.randint(0, 33)).astype(np.uint8))
        assert capi.lib().xf_hash_bytes(b, len(b)) == O.lib().xo_hash_bytes(b, len(b))


def test_shard_rule_matches_oracle():
    rng = np.random.RandomState(6)
    keys = [0, 2**64 - 1, 0x799107141a3182b9] + [int(x) for x in
                                                   rng.randint(0, 2**63, size=200) * 2]
    for n in (1, 2, 3, 8):
        for k in keys:
            assert capi.lib().xf_shard_of(k, n) == O.lib().xo_shard_of(k, n)
    assert capi.lib().xf_shard_of(0x799107141a3182b9, 8) == 3


@pytest.mark.parametrize("name", ["small_train-00000", "small_test-00000"])
@pytest.mark.parametrize("cap", [2 << 20, 4096, 1000])
def test_reader_golden(golden_dir, name, cap):
    g = np.load(os.path.join(golden_dir, "ref_parse_%s_cap%d.npz" % (name, cap)))
    blocks = list(capi.read_blocks(os.path.join(golden_dir, name), cap))
    assert [len(b[3]) for b in blocks] == g["block_rows"].tolist()
    assert np.array_equal(np.concatenate([b[1] for b in blocks]), g["keys"])
    assert np.array_equal(np.concatenate([b[2] for b in blocks]), g["fgid"])
    assert np.array_equal(np.concatenate([b[3] for b in blocks]), g["labels"])


def test_reader_edges_vs_oracle(tmp_path):
    line = "1\t0:12345:1 1:678:1\n"
    cases = {"nonl": "1\t0:1:1 2:22:0.5\n0\t3:333:1",
             "blank": "0.5\t0:1:1 2:22:0.5 \n0.00000001\t3:333:1\n",
             "neg": "-1\t0:1:1\n1e-7\t1:2:1\n2e-7\t1:2:1\n", "empty": "", "exact": line * 10,
             # empty tokens duplicate the previous token, as in the reference (oracle test)
             "dblank": "1\t0:1:1  2:2:2 \n0\t3:3:3   4:4:4\n",
             "blank_eof": "1\t0:1:1 2:22:0.5 \n0\t3:333:1 ",
             "blank_cut": ("1\t0:12345:1 1:67:1 \n") * 6 + "0\t5:5:5\n"}
    for nm, txt in cases.items():
        p = tmp_path / nm
        p.write_text(txt)
        for cap in (41, 64, 1 << 16):
            mine = list(capi.read_blocks(str(p), cap))
            theirs = list(O.read_blocks(str(p), cap))
            assert len(mine) == len(theirs)
            for a, b in zip(mine, theirs):
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (nm, cap)
    for txt in ["1 0:1:1\n", "1\t0:1\n", "1\t 0:1:1\n", "1\t0:1:1\n0\t"]:
        p = tmp_path / "bad"
        p.write_text(txt)
        with pytest.raises(capi.XFError):
            list(capi.read_blocks(str(p), 1 << 16))
    with pytest.raises(capi.XFError):
        list(capi.read_blocks(str(tmp_path / "missing"), 1 << 16))


def _random_csr(rng, R, max_len, nkeys):
    lens = rng.randint(0, max_len + 1, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keys = rng.randint(0, nkeys, size=int(lens.sum())).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    return rowptr, keys, labels


@pytest.mark.parametrize("R,max_len,nkeys", [(1, 3, 5), (50, 8, 40), (3000, 40, 500),
                                             (20000, 12, 100000)])
def test_batch_compile_vs_oracle(R, max_len, nkeys):
    rng = np.random.RandomState(R)
    rowptr, keys, labels = _random_csr(rng, R, max_len, nkeys)
    a, b0 = R // 5, R - R // 7
    mine = capi.Batch(rowptr, keys, labels, a, b0)
    ref = O.Batch(rowptr, keys, labels, a, b0)
    h = mine.host()
    assert (mine.R, mine.NNZ, mine.U) == (ref.R, ref.NNZ, ref.U)
    assert np.array_equal(h["ukeys"], ref.ukeys)        # Pull/Push key list: bit-exact
    assert np.array_equal(h["rowptr"], ref.rowptr)
    assert np.array_equal(h["uidx"], ref.uidx)
    assert np.array_equal(h["segptr"], ref.segptr)
    assert np.array_equal(h["labels"], ref.labels)
    # within a key the reference's order is std::sort's (unspecified): compare as multisets
    for u in range(0, mine.U, max(1, mine.U // 300)):
        s, e = ref.segptr[u], ref.segptr[u + 1]
        assert np.array_equal(np.sort(h["coo_row"][s:e]), np.sort(ref.coo_row[s:e]))
    seglen = np.diff(h["segptr"])
    assert np.array_equal(h["heavy"], np.nonzero(seglen > capi.HEAVY_SEG)[0])


def test_batch_compile_empty_and_ragged():
    rowptr = np.array([0, 0, 2, 2, 5, 5], dtype=np.uint64)
    keys = np.array([7, 7, 3, 9, 3], dtype=np.uint64)
    labels = np.array([0, 1, 0, 1, 1], dtype=np.int32)
    b = capi.Batch(rowptr, keys, labels)
    h = b.host()
    assert (b.R, b.NNZ, b.U) == (5, 5, 3)
    assert h["ukeys"].tolist() == [3, 7, 9]
    assert h["uidx"].tolist() == [1, 1, 0, 2, 0]
    assert h["segptr"].tolist() == [0, 2, 4, 5]
    assert h["coo_row"].tolist() == [3, 3, 1, 1, 3]
    e = capi.Batch(np.array([0], dtype=np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.int32))
    assert (e.R, e.NNZ, e.U) == (0, 0, 0)


def test_auc_logloss_golden(golden_dir):
    a = json.load(open(os.path.join(golden_dir, "ref_kats.json")))["auc"]
    lab = np.array(a["labels"], dtype=np.int32)
    p = np.array([float.fromhex(h) for h in a["pctr_hex"]], dtype=np.float32)
    ll, auc, tp, fp, nat = capi.auc_logloss(lab, p)
    assert ll == float.fromhex(a["logloss_hex"])
    assert O.format_auc_line(ll, auc, tp, fp) == a["line"]
    assert np.isclose(nat, -np.mean(np.where(lab == 1, np.log(p.astype(np.float64)),
                                             np.log1p(-p.astype(np.float64)))), rtol=1e-12)


def test_product_fails_loudly_without_gpu():
    n = capi.C.c_int(0)
    capi.check(capi.lib().xf_device_count(capi.C.byref(n)))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.XFError, match="no HIP device"):
        capi.Table(capacity=1024)
    with pytest.raises(capi.XFError):
        capi.XFlow("/nonexistent/train", "/nonexistent/test").train()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "xflow_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "xflow_oracle" not in txt and \
                    "liboracle" not in txt, os.path.join(dp, f)


def test_forward_panels_partition_the_csr():
    """The panel-major view holds exactly the CSR's nonzeros, each in the panel its uidx
    falls in, CSR order kept inside a (panel,row) cell."""
    rng = np.random.RandomState(12)
    rowptr, keys, labels = _random_csr(rng, 400, 30, 3000)
    capi.tune("min_panel_nnz", 0)
    capi.tune("panel_slice_bytes", 1024)      # force several panels on a tiny batch
    try:
        b = capi.Batch(rowptr, keys, labels)
    finally:
        capi.tune("min_panel_nnz", 4e6)
        capi.tune("panel_slice_bytes", 1.5 * 1024 * 1024)
    h = b.host()
    P, pptr, pidx = b.panels()
    assert P >= 8 and P % 8 == 0 and len(pidx) == b.NNZ
    pptr = pptr.reshape(P, b.R + 1)
    assert pptr[0, 0] == 0 and pptr[-1, -1] == b.NNZ
    for r in range(0, b.R, 7):
        row = h["uidx"][h["rowptr"][r]:h["rowptr"][r + 1]]
        got = []
        for p in range(P):
            cell = pidx[pptr[p, r]:pptr[p, r + 1]]
            assert np.all((cell.astype(np.uint64) * P) // b.U == p)
            assert np.array_equal(cell, row[(row.astype(np.uint64) * P) // b.U == p])
            got.append(cell)
        assert sorted(np.concatenate(got).tolist()) == sorted(row.tolist())
    small = capi.Batch(rowptr, keys, labels)
    assert small.panels()[0] == 0               # default: small batches keep the plain CSR


def test_gradient_tiles_cover_the_keys():
    rng = np.random.RandomState(21)
    R = 6000
    lens = rng.randint(0, 30, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    fid = np.minimum(rng.zipf(1.3, size=int(lens.sum())), 5000)   # heavy heads
    keys = fid.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    b = capi.Batch(rowptr, keys, rng.randint(0, 2, size=R).astype(np.int32))
    h, tp = b.host(), b.tiles()
    assert tp[0] == 0 and tp[-1] == b.U and np.all(np.diff(tp) > 0)
    seg = np.diff(h["segptr"])
    assert b.H > 0
    for a, e in zip(tp[:-1], tp[1:]):
        nnz = int(h["segptr"][e] - h["segptr"][a])
        if e - a == 1 and seg[a] > capi.HEAVY_SEG:
            continue                       # heavy key: a tile of its own
        assert nnz <= 2048 and e - a <= 2048 and np.all(seg[a:e] <= capi.HEAVY_SEG)


def test_forward_tiles_cover_every_cell_once():
    rng = np.random.RandomState(13)
    rowptr, keys, labels = _random_csr(rng, 500, 40, 4000)
    capi.tune("min_panel_nnz", 0)
    capi.tune("panel_slice_bytes", 2048)
    try:
        b = capi.Batch(rowptr, keys, labels)
    finally:
        capi.tune("min_panel_nnz", 4e6)
        capi.tune("panel_slice_bytes", 1.5 * 1024 * 1024)
    P, pptr, pidx = b.panels()
    tp, pf, grid = b.fwd_tiles()
    nt = len(tp) - 1
    assert tp[0] == 0 and tp[-1] == (P - 1) * (b.R + 1) + b.R and np.all(np.diff(tp) > 0)
    assert pf[0] == 0 and pf[-1] == nt and len(pf) == P + 1
    seen = np.zeros(P * (b.R + 1), dtype=np.int32)
    for t, (a, e) in enumerate(zip(tp[:-1], tp[1:])):
        p = a // (b.R + 1)
        assert p == (e - 1) // (b.R + 1) and pf[p] <= t < pf[p + 1]   # one panel per tile
        assert pptr[e] - pptr[a] <= 2048 or e - a == 1
        assert e - a <= 2049
        seen[a:e] += 1
    cells = seen.reshape(P, b.R + 1)
    assert np.all(cells[:, :b.R] == 1)
    # the workgroup -> tile map of k_lr_forward_tiled reaches every tile exactly once
    hit = np.zeros(nt, dtype=np.int32)
    assert grid % 8 == 0
    for w in range(grid):
        q = w // 8
        for p in range(w % 8, P, 8):
            cnt = pf[p + 1] - pf[p]
            if q < cnt:
                hit[pf[p] + q] += 1
                break
            q -= cnt
    assert np.all(hit == 1)


def test_tiling_rules_bounds_with_heavy_and_ragged_rows():
    """xf_tiling.h: gradient tiles < 2048 occurrences / keys, heavy keys alone; forward tiles
    < 2048 nonzeros / <= 2048 cells even with many empty cells and one huge row."""
    rng = np.random.RandomState(14)
    R = 9000
    lens = np.where(rng.rand(R) < 0.7, 0, rng.randint(1, 12, size=R))
    lens[1234] = 40000                                 # one oversized row: cells > 2048
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    fid = np.minimum(rng.zipf(1.25, size=int(lens.sum())), 3000)
    keys = fid.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    capi.tune("min_panel_nnz", 0)
    capi.tune("panel_slice_bytes", 4096)
    try:
        b = capi.Batch(rowptr, keys, rng.randint(0, 2, size=R).astype(np.int32))
    finally:
        capi.tune("min_panel_nnz", 4e6)
        capi.tune("panel_slice_bytes", 1.5 * 1024 * 1024)
    h, tp = b.host(), b.tiles()
    seg = np.diff(h["segptr"])
    for a, e in zip(tp[:-1], tp[1:]):
        if e - a == 1 and seg[a] > capi.HEAVY_SEG:
            continue
        assert h["segptr"][e] - h["segptr"][a] < 2048 and e - a <= 2048
        assert np.all(seg[a:e] <= capi.HEAVY_SEG)
    P, pptr, _ = b.panels()
    ftp = b.fwd_tiles()[0]
    big = 0
    for a, e in zip(ftp[:-1], ftp[1:]):
        n = int(pptr[e] - pptr[a])
        if n >= 2048:
            assert e - a <= 2 and n > 256          # a single oversized cell (+ the panel gap)
            big += 1
        assert e - a <= 2049
    assert big >= 1


def test_block_cache_replays_the_parse_bit_for_bit(tmp_path, sample_prefixes):
    """the binarized block cache.  A first pass parses the text and writes the
    cache, later passes are served from it and must return exactly the same blocks; the cache
    is tied to the block size and to the source file (size, mtime), and a pass that stops
    before end of file leaves none."""
    import shutil
    src = str(tmp_path / "train-00000")
    shutil.copy(sample_prefixes[0] + "-00000", src)
    cache = str(tmp_path / "train.xfcsr")
    cap = 6000                                     # several blocks
    plain = list(capi.read_blocks(src, cap))
    assert len(plain) > 3

    def same_blocks(got):
        assert len(got) == len(plain)
        for a, b in zip(got, plain):
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and np.array_equal(x, y)

    info = {}
    it = capi.read_blocks(src, cap, cache, info)    # abandoned after one block: no cache file
    next(it)
    it.close()
    assert not info["from_cache"] and not os.path.exists(cache)
    assert [f for f in os.listdir(str(tmp_path)) if ".tmp." in f] == []
    same_blocks(list(capi.read_blocks(src, cap, cache, info)))
    assert not info["from_cache"] and os.path.exists(cache)
    same_blocks(list(capi.read_blocks(src, cap, cache, info)))
    assert info["from_cache"]
    # another block size does not match the cache: parsed again, cache rebuilt for that size
    other = list(capi.read_blocks(src, 3 * cap, cache, info))
    assert not info["from_cache"] and len(other) < len(plain)
    list(capi.read_blocks(src, cap, cache, info))
    assert not info["from_cache"]                  # (it was overwritten for 3*cap)
    list(capi.read_blocks(src, cap, cache, info))
    assert info["from_cache"]
    # the source changes: stale cache is ignored
    with open(src, "a") as f:
        f.write("1\t0:424242:1\n")
    changed = list(capi.read_blocks(src, cap, cache, info))
    assert not info["from_cache"]
    assert sum(len(b[3]) for b in changed) == sum(len(b[3]) for b in plain) + 1
    # a truncated cache file is reported, not silently shortened
    list(capi.read_blocks(src, cap, cache, info))
    assert info["from_cache"]
    with open(cache, "r+b") as f:
        f.truncate(os.path.getsize(cache) - 64)
    with pytest.raises(capi.XFError):
        list(capi.read_blocks(src, cap, cache, info))


def test_binding_demo_builds_against_the_c_abi_only():
    """examples/kv_demo.cc + examples/ps_gpu.h compile with g++ and nothing but
    include/xflow_amd.h; without a GPU the program reports the library's error and exits 1."""
    import subprocess
    import torch
    from xflow_amd import build
    exe = os.path.join(build.LIBDIR, "kv_demo")
    assert os.path.exists(exe), "python -m xflow_amd.build builds it"
    if torch.cuda.is_available():
        pytest.skip("GPU present: the demo's result is checked in the gpu suite")
    out = subprocess.run([exe, "8", "1"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "no HIP device" in out.stderr


def _random_libsvm_text(rng, rows):
    """label\\tfg:fid:val tokens the way real files look: integer and float labels around the
    1e-7 threshold, decimal and alphanumeric fids of 1..14 bytes, int/float vals, an occasional
    trailing blank, the last line with or without newline."""
    labels = ["0", "1", "0.0", "1.0", "0.5", "1e-7", "1.1e-7", "-1", "3", "0.00000009", "2e-7"]
    alnum = "0123456789abcdefXYZ_"
    out = []
    for _ in range(rows):
        toks = []
        for _ in range(rng.randint(1, 12)):
            n = rng.randint(1, 15)
            fid = "".join(alnum[i] for i in rng.randint(0, 10 if rng.rand() < 0.7 else len(alnum), n))
            val = ["1", "0", "0.3651", "12", "1e-3"][rng.randint(0, 5)]
            toks.append("%d:%s:%s" % (rng.randint(0, 40), fid, val))
        out.append(labels[rng.randint(0, len(labels))] + "\t" + " ".join(toks) +
                   (" " if rng.rand() < 0.15 else ""))
    text = "\n".join(out) + ("\n" if rng.rand() < 0.7 else "")
    # not generated: a blank before the file's final newline.  The reference is left pointing
    # at that newline and scans past its terminator for the next tab (:127-128) — undefined.
    return text[:-2] + "\n" if text.endswith(" \n") else text


@pytest.mark.parametrize("seed", range(40))
def test_reader_fuzz_three_way(tmp_path, seed):
    """Random well-formed files x random block sizes: the product's parser, the oracle's
    restatement and — when oracle/_ref is built (this container) — the real reference parser
    must return identical blocks (row partition, keys, fgid, labels)."""
    rng = np.random.RandomState(100 + seed)
    txt = _random_libsvm_text(rng, rng.randint(1, 400))
    p = tmp_path / "f"
    p.write_text(txt)
    longest = max(len(l) for l in txt.split("\n")) + 2
    for cap in sorted({longest + 1, longest + int(rng.randint(1, 200)), 4096, 1 << 20}):
        mine = list(capi.read_blocks(str(p), cap))
        theirs = list(O.read_blocks(str(p), cap))
        sets = [("oracle", theirs)]
        if O.ref_available():
            sets.append(("reference", list(O.ref_read_blocks(str(p), cap))))
        for name, other in sets:
            assert len(mine) == len(other), (name, cap)
            for a, b in zip(mine, other):
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (name, cap)
        assert sum(len(b[3]) for b in mine) == txt.count("\n") + (0 if txt.endswith("\n") else 1)


@pytest.mark.parametrize("seed", range(12))
def test_auc_logloss_fuzz_three_way(seed):
    """Random score vectors with many exact ties (rows that only hit unseen keys all score
    0.5), the sigmoid clamps (1e-6 and exactly 1.0 -> log2(0)), one-class label sets and sizes
    beyond 2^24 / n where the fp32 `area` accumulator starts rounding: xf_auc_logloss = the
    oracle = Base::calculate_auc of the real reference (when oracle/_ref is built), including
    the printed line."""
    rng = np.random.RandomState(500 + seed)
    n = int(rng.choice([1, 2, 7, 200, 5000, 60000]))
    pool = np.concatenate([rng.rand(max(2, n // 20)).astype(np.float32),
                           np.array([0.5, 0.5, 1e-6, 1.0, 0.999999], dtype=np.float32)])
    p = pool[rng.randint(0, len(pool), size=n)]
    lab = rng.randint(0, 2, size=n).astype(np.int32)
    if seed % 5 == 0:
        lab[:] = seed % 2                      # one class only: the reference prints tp_n alone
    if seed % 3 == 0:
        lab[p == 1.0] = 1                      # keep -inf out of this case
    ll, auc, tp, fp, _ = capi.auc_logloss(lab, p)
    oll, oauc, otp, ofp = O.auc_logloss(lab, p)
    assert (tp, fp) == (otp, ofp)
    assert np.array_equal(np.float32([ll, auc]), np.float32([oll, oauc]), equal_nan=True)
    if O.ref_available():
        rll, line = O.ref_auc(lab, p)
        assert np.array_equal(np.float32([ll]), np.float32([rll]), equal_nan=True)
        if tp and fp:
            assert line == O.format_auc_line(oll, oauc, otp, ofp)


def test_sigmoid_fuzz_vs_live_ref():
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.RandomState(9)
    xs = np.concatenate([rng.uniform(-40, 40, 4000), rng.normal(0, 1e-3, 500),
                         [-30.0, 30.0, -30.000002, 30.000002, 0.0, -0.0, 88.0, -104.0]])
    for x in xs.astype(np.float32):
        assert O.sigmoid(x) == O.ref().ref_sigmoid(C_float(x)), x


@pytest.mark.parametrize("cap", [202, 777, 1 << 20])
def test_reader_quirks_golden(golden_dir, cap):
    """tests/golden/quirks-00000 (our own synthetic input) against the real parser's output:
    empty tokens duplicating the previous token — how many depends on where the blocks are cut
    —, labels around 1e-7, alphanumeric fids, a fractional fgid."""
    g = np.load(os.path.join(golden_dir, "ref_parse_quirks_cap%d.npz" % cap))
    blocks = list(capi.read_blocks(os.path.join(golden_dir, "quirks-00000"), cap))
    assert [len(b[3]) for b in blocks] == g["block_rows"].tolist()
    assert np.array_equal(np.concatenate([b[1] for b in blocks]), g["keys"])
    assert np.array_equal(np.concatenate([b[2] for b in blocks]), g["fgid"])
    assert np.array_equal(np.concatenate([b[3] for b in blocks]), g["labels"])
    rp = np.concatenate([[0]] + [np.diff(b[0]) for b in blocks]).cumsum()
    assert np.array_equal(rp.astype(np.uint64), g["rowptr"])


def test_fp32_division_by_row_count_equals_the_reference_double_division():
    """The FM gradient kernel computes `(float)acc / (float)R` where the reference does
    `float /= 1.0 * line_num` (double division, rounded back to float; fm_worker.cc:150-156).
    For fp32 x and integer R < 2^24 the two are the same number (xf_model.hip: div_by_rows);
    checked here on 2e7 random (x, R) with IEEE arithmetic on the host."""
    rng = np.random.RandomState(3)
    for it in range(4):
        n = 5_000_000
        m = rng.randint(1, 1 << 24, size=n).astype(np.float32)
        x = np.ldexp(m, rng.randint(-70, 20, size=n)).astype(np.float32)
        x *= np.where(rng.rand(n) < 0.5, -1, 1).astype(np.float32)
        R = rng.randint(1, 1 << 24, size=n) if it % 2 else \
            rng.choice(np.array([3, 7, 200, 49999, 50000, 65537, (1 << 24) - 1]), size=n)
        want = (x.astype(np.float64) / R.astype(np.float64)).astype(np.float32)
        assert np.array_equal(want, x / R.astype(np.float32))


def test_reader_next_into_keeps_blocks_alive(sample_prefixes):
    """xf_reader_next_into moves a block's arrays into a caller-owned xf_block: two blocks stay
    readable while the reader moves on (what the worker's prefetch thread relies on)."""
    import ctypes as C
    L = capi.lib()
    path = sample_prefixes[0] + "-00000"
    want = list(capi.read_blocks(path, 6000))
    rd = capi.vp()
    capi.check(L.xf_reader_open(C.byref(rd), path.encode(), 6000))
    blks = [capi.vp(), capi.vp()]
    for b in blks:
        capi.check(L.xf_block_create(C.byref(b)))
    held = [None, None]
    seen = 0
    try:
        for i in range(len(want) + 1):
            rows, nnz = C.c_size_t(0), C.c_size_t(0)
            rp, ks, fg, lb = capi.u64p(), capi.u64p(), capi.i32p(), capi.i32p()
            capi.check(L.xf_reader_next_into(rd, blks[i % 2], C.byref(rows), C.byref(nnz),
                                             C.byref(rp), C.byref(ks), C.byref(fg), C.byref(lb)))
            if rows.value == 0:
                break
            held[i % 2] = (i, rows.value, nnz.value, rp, ks, lb)
            for h in held:                       # the previous block is still intact
                if h is None:
                    continue
                j, r, n, hrp, hks, hlb = h
                assert np.array_equal(np.ctypeslib.as_array(hrp, (r + 1,)), want[j][0])
                assert np.array_equal(np.ctypeslib.as_array(hks, (n,)), want[j][1])
                assert np.array_equal(np.ctypeslib.as_array(hlb, (r,)), want[j][3])
            seen += 1
        assert seen == len(want) > 3
    finally:
        for b in blks:
            L.xf_block_destroy(b)
        L.xf_reader_close(rd)


def test_bench_byte_model_matches_survey_8d():
    """bench.py's whole-step figure is SURVEY 8(d)'s formula, the per-kernel figures of the
    fused LR step (what `roofline.frac` divides by the launch time) are 8(d)'s rows and add up to
    it exactly, and the implementation's own byte count is reported beside them, not as them."""
    import bench
    R, NNZ, U = 50_000, 10_000_000, 6_320_289
    per, survey = bench.bytes_model("lr", 0, R, NNZ, U, "ftrl", fused=True)
    assert survey == 12 * NNZ + 8 * R + 32 * U
    assert set(per) == {"forward", "gradient"}
    assert per["forward"] == NNZ * 12 + R * 8           # NNZ x (8 key + 4 w) + R x (label + loss)
    assert per["gradient"] == U * (4 + 28)              # write g; g + (w,n,z) read + written
    assert sum(per.values()) == survey
    info = {"G": 85, "nwin": 3, "W": 16667}
    impl = bench.impl_bytes_cells(R, NNZ, U, "ftrl", info, 10**7)
    assert impl["forward"] == NNZ * 8 + 2 * 85 * 3 * 16667 * 8 + R * 8
    assert impl["gradient"] == NNZ * 8 + U * 24
    _, survey = bench.bytes_model("lr", 0, R, NNZ, U, "sgd", fused=True)
    assert survey == 12 * NNZ + 8 * R + 16 * U
    for opt, state in (("sgd", 16), ("ftrl", 32)):
        _, survey = bench.bytes_model("fm", 16, R, NNZ, U, opt)
        assert survey == NNZ * (12 + 4 * 16) + 8 * R + state * U * 17
    per, _ = bench.bytes_model("fm", 16, R, NNZ, U, "sgd", fused_fm=True)
    assert per["forward"] == NNZ * 36 + R * 12 + 4        # one 32-byte record per nonzero


def test_reader_unusual_tokens_take_the_general_path(tmp_path):
    """The parser's one-pass route only takes `digits:fid:val`; every other token shape goes
    through the general route.  Both must give what the oracle (and the real reference parser)
    give: float / signed / zero-padded / ten-digit fgids, empty fid, empty val, extra colons in
    the val, an empty token (duplicates the previous one), fids longer than 8 bytes, a fid that
    ends the file (the 8-byte load of the short-fid hash stops at the buffer's end)."""
    lines = [
        "1\t1.5:abc:1 -3:77:1 0007:5:2 1234567890:9:1",
        "0\t7::1 7:5: 3:a:b:c 12:123456789012345:0.5",
        "1\t4:44:1  5:55:1 6:6:1",                 # two blanks: an empty token
        "0.3\t000000001:x:1 999999999:yy:2 1e1:z:3 +2:q:1",
        "1\t9:7",                                  # filled below: the last token ends the file
    ]
    lines[-1] = "1\t31:8:1 2:3:4"
    txt = "\n".join(lines)                         # no trailing newline: "4" is the last byte
    p = tmp_path / "odd"
    p.write_text(txt)
    for cap in (len(max(lines, key=len)) + 3, 4096):
        mine = list(capi.read_blocks(str(p), cap))
        sets = [("oracle", list(O.read_blocks(str(p), cap)))]
        if O.ref_available():
            sets.append(("reference", list(O.ref_read_blocks(str(p), cap))))
        for name, other in sets:
            assert len(mine) == len(other), (name, cap)
            for a, b in zip(mine, other):
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (name, cap, x, y)
    rows = sum(len(b[3]) for b in mine)
    assert rows == len(lines)


def test_library_names_the_sources_it_was_built_from():
    """xf_source_hash: a prebuilt libxflow_amd.so that does not match the sources next to it is
    refused on load (file times do not survive the snapshot to the GPU box)."""
    from xflow_amd import build
    assert capi.lib().xf_source_hash().decode() == build.source_hash()
    assert len(build.source_hash()) == 32


def test_build_reads_the_built_hash_without_loading_the_library():
    """build.py decides "rebuild or not" from a sidecar file written at link time, never by
    dlopen-ing the library: a probe left mapped would make glibc hand the OLD handle to the
    load after a rebuild (same path), and the binding would then refuse it for the rest of the
    process.  The sidecar names the library by content, so a stale one counts as unknown."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from xflow_amd import build\n"
            "assert build._built_hash() == build.source_hash(), 'sidecar does not match'\n"
            "assert 'libxflow_amd' not in open('/proc/self/maps').read(), 'library was mapped'\n"
            "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
    from xflow_amd import build
    src, libsha = open(build.HASH_FILE).read().split()
    assert libsha == build._file_sha(build.LIB)


def test_worker_parameters_are_validated_without_a_gpu():
    """XFCreate / XFSetParam (the worker's host side: main.cc's argv + the extra knobs): values
    are checked when they are set, by name, with the reason in xf_last_error; training is what
    needs the GPU (XFStartTrain fails loudly without one: test below)."""
    import ctypes as C
    L = capi.lib()
    h = capi.vp()
    assert L.XFCreate(C.byref(h), b"/nonexistent/train", b"/nonexistent/test") == 0
    try:
        good = [("model", "1"), ("k", "16"), ("optimizer", "sgd"), ("optimizer", "ftrl"),
                ("epochs", "3"), ("schedule", "sequential"), ("schedule", "stale1"),
                ("schedule", "owner"), ("update", "rank_ordered"), ("update", "sum_then_step"),
                ("parity", "exact"), ("parity", "reference_order"), ("transport", "host")]
        for n, v in good:
            assert L.XFSetParam(h, n.encode(), v.encode()) == 0, (n, v, L.xf_last_error())
        bad = [("schedule", "bogus", "sequential, stale1 or owner"),
               ("update", "x", "rank_ordered or sum_then_step"),
               ("optimizer", "adam", "ftrl or sgd"), ("model", "2", "0 (LR) or 1 (FM)"),
               ("nope", "1", "unknown parameter")]
        for n, v, why in bad:
            assert L.XFSetParam(h, n.encode(), v.encode()) != 0, (n, v)
            assert why in L.xf_last_error().decode()
    finally:
        L.XFDestroy(h)


def test_a_library_built_with_experiment_flags_never_passes_for_the_plain_one():
    """XF_EXTRA_FLAGS (tools/grad_timeline.py: -DXF_GRAD_TIMELINE) is part of the source hash:
    the binding refuses a library built with other flags than the process asks for"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from xflow_amd import build\n"
            "print(build.source_hash())\n" % root)
    env = dict(os.environ, XF_EXTRA_FLAGS="-DXF_GRAD_TIMELINE")
    a = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    env.pop("XF_EXTRA_FLAGS")
    b = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout.strip() != b.stdout.strip()
    assert b.stdout.strip() == capi.lib().xf_source_hash().decode()


def test_the_update_entry_points_fail_loudly_without_a_gpu():
    """xf_lr_update_dev / xf_batch_compile_fm / xf_batch_compile_fm_dev: argument checks first
    (null tables: XF_EINVAL with the reason), never a silent host path"""
    n = capi.C.c_int(0)
    capi.check(capi.lib().xf_device_count(capi.C.byref(n)))
    if n.value > 0:
        pytest.skip("a GPU is present")
    C, L = capi.C, capi.lib()
    h = capi.vp()
    assert L.xf_lr_update_dev(C.byref(h), None, None, None, None, 0, 0, 0, None, None) != 0
    assert b"null argument" in L.xf_last_error()
    rp = np.zeros(2, np.uint64)
    lb = np.zeros(1, np.int32)
    assert L.xf_batch_compile_fm(C.byref(h), None, None, rp.ctypes.data, None, lb.ctypes.data, 0, 1,
                                 None, None) != 0
    assert b"bad argument" in L.xf_last_error()
    assert L.xf_batch_compile_fm_dev(C.byref(h), None, None, None, None, None, 0, 0, None,
                                     None) != 0
    assert L.xf_sbatch_fm_keyed(None) < 0


def test_tune_takes_the_named_switches_and_no_experiment_knob():
    """xf_tune: the code paths a test may pin have names (xf_common.h); the experiments' numeric
    knob exists only in a library built with -DXF_EXPERIMENTS, and values out of range are refused."""
    for name, hi in (("key_build", 3), ("old_weight", 2), ("lr_gradient", 3), ("owner_pass", 4)):
        for v in range(hi + 1):
            capi.tune(name, v)
        with pytest.raises(capi.XFError):
            capi.tune(name, hi + 1)
        capi.tune(name, 0)
    with pytest.raises(capi.XFError):
        capi.tune("exp_knob", 77)
    with pytest.raises(capi.XFError):
        capi.tune("no_such_switch", 1)


def test_ftrl_tables_refuse_an_alpha_that_is_not_positive():
    """the step divides by alpha twice (ftrl.h:63,70) and the table keeps a verdict in the sign of
    1 / alpha: zero, negative and non-finite alphas are refused when the table is created"""
    import ctypes as C
    for bad in (0.0, -0.05, float("inf"), float("nan")):
        c = capi.TableConfig()
        capi.lib().xf_table_config_default(C.byref(c))
        c.alpha = bad
        h = capi.vp()
        rc = capi.lib().xf_table_create(C.byref(h), C.byref(c))
        assert rc != 0 and b"alpha" in capi.lib().xf_last_error()

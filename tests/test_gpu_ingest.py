"""The GPU tokeniser (xf_ingest.hip) on a real MI355X: a block of libsvm-style text ->
(rowptr, keys, labels) on the device, for blocks of the common shape; everything else is handed
back (ok = 0) and stays the host parser's.

* accepted blocks: bit for bit the host parser's arrays (whose parse is pinned to the real
  reference's: tests/golden/ref_parse_*.npz) — the reference's sample files at several block
  sizes, generated blocks with tokens, fids and lines of every length around the kernels' tile
  (4 KiB), thread (16 B) and workgroup-span boundaries;
* every single defect the shape excludes (other labels, empty tokens, a blank before the line's
  end, NUL, a token without two colons, an over-long fid ...) is REJECTED — for several of
  them the reference's own result differs from what a naive tokeniser would emit (an empty token
  duplicates the previous one), which is why they are not the GPU's;
* the worker end to end with ingest = gpu: the tables and the metric line of ingest = host."""
import os

import numpy as np
import pytest

from xflow_amd import capi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def gpu():
    capi.require_gpu()


@pytest.fixture(scope="module")
def ing():
    return capi.Ingest(1 << 24)


def same_as_host(ing, text):
    ok, rp, ks, lb = ing.block(text)
    assert ok, "a block of the common shape was handed back"
    hrp, hks, _, hlb = capi.parse_text_block(text)
    assert np.array_equal(rp.astype(np.uint64), hrp)
    assert np.array_equal(ks, hks)
    assert np.array_equal(lb, hlb)
    return len(lb), len(ks)


@pytest.mark.parametrize("name,cap", [("small_train-00000", 1000), ("small_train-00000", 4096),
                                      ("small_train-00000", 2097152), ("small_test-00000", 4096)])
def test_sample_files_block_for_block(ing, name, cap):
    rows = 0
    for t in capi.read_text_blocks(os.path.join(GOLD, name), cap):
        rows += same_as_host(ing, t)[0]
    assert rows == 200


def gen_block(rng, rows, max_tok=12, fid_len=(1, 32), f0_len=(0, 16), rest_len=(0, 9)):
    """well-formed text: every byte value the shape allows in its fields"""
    fid_alpha = np.array([c for c in range(0x21, 0x7F) if chr(c) not in ":"] +
                         list(range(0x80, 0x100)), np.uint8)       # no ' ', ':', control bytes
    rest_alpha = np.array([c for c in range(0x21, 0x100)], np.uint8)   # ':' allowed in `rest`
    out = []
    for _ in range(rows):
        toks = []
        for _ in range(rng.randint(1, max_tok + 1)):
            f0 = bytes(rng.choice(fid_alpha, rng.randint(f0_len[0], f0_len[1] + 1)))
            fid = bytes(rng.choice(fid_alpha, rng.randint(fid_len[0], fid_len[1] + 1)))
            rest = bytes(rng.choice(rest_alpha, rng.randint(rest_len[0], rest_len[1] + 1)))
            toks.append(f0 + b":" + fid + b":" + rest)
        out.append((b"1" if rng.rand() < 0.3 else b"0") + b"\t" + b" ".join(toks) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("seed", range(6))
def test_generated_blocks_of_the_common_shape(ing, seed):
    rng = np.random.RandomState(seed)
    # short rows / long rows / many tiny tokens / the longest fields: tokens and lines straddle
    # every 16-byte, 4 KiB and span boundary somewhere in a few MB
    shapes = [dict(rows=30000, max_tok=3), dict(rows=3000, max_tok=60),
              dict(rows=8000, max_tok=20, fid_len=(1, 2), f0_len=(0, 1), rest_len=(0, 0)),
              dict(rows=4000, max_tok=8, fid_len=(30, 32), f0_len=(15, 16), rest_len=(0, 40))]
    text = gen_block(rng, **shapes[seed % len(shapes)])
    R, N = same_as_host(ing, text)
    assert R == text.count(b"\n") and N > R
    # the same without the final newline (a block cut at its last newline comes like that), and
    # prefixes that end at other line ends (other lengths modulo 16 and 4096)
    same_as_host(ing, text[:-1])
    ends = [i for i in range(len(text)) if text[i:i + 1] == b"\n"]
    for e in (ends[0], ends[len(ends) // 3], ends[len(ends) // 2]):
        same_as_host(ing, text[:e + 1])


def test_the_sample_shape_at_full_block_size():
    """a 64 MiB block of the bench's row shape (200 tokens "fg:fid:1"): rows, tokens and the
    key array against the host parser"""
    ing = capi.Ingest(1 << 26)
    rng = np.random.RandomState(5)
    R = 26000
    fid = rng.randint(0, 10_000_000, size=(R, 200))
    lab = rng.randint(0, 2, size=R)
    text = "".join("%d\t" % lab[r] + " ".join("%d:%d:1" % (j & 31, v) for j, v in enumerate(fid[r]))
                   + "\n" for r in range(R)).encode()
    assert 50 << 20 < len(text) < 64 << 20
    ok, rp, ks, lb = ing.block(text)
    assert ok and len(lb) == R and len(ks) == R * 200
    assert np.array_equal(ks, capi.hash_decimal_ids(fid.ravel().astype(np.uint64)))
    assert np.array_equal(lb, lab) and np.array_equal(rp, np.arange(R + 1, dtype=np.uint32) * 200)


def test_cr_lf_and_other_ordinary_bytes(ing):
    """the reference's sample files are CR LF: the CR is a byte of the last token's third field
    (never read) to the reference's parser and to this one; so is every other control byte but
    NUL, tab and newline"""
    same_as_host(ing, b"0\t1:22:0.5 3:4:1\r\n1\t7:abc:x\r\n")
    same_as_host(ing, b"1\t1:a\x01b:\x02 2:\x7f\x1f:\x0b\n")


GOOD = b"0\t1:22:0.5 3:4:1\n1\t7:abc:x\n"
DEFECTS = [
    ("a label that is not one digit", b"0.5\t1:22:0.5\n"),
    ("a label 2", b"2\t1:22:0.5\n"),
    ("an empty token (two blanks): the reference duplicates the previous token",
     b"0\t1:22:0.5  3:4:1\n"),
    ("a blank before the end of the line", b"0\t1:22:0.5 \n"),
    ("a blank right after the tab", b"0\t 1:22:0.5\n"),
    ("a row without tokens", b"0\t\n"),
    ("a line without a tab", b"0 1:22:0.5\n"),
    ("an empty line", b"0\t1:2:3\n\n1\t1:2:3\n"),
    ("a NUL byte", b"0\t1:22:0.5\x00\n"),
    ("a second tab", b"0\t1:22:0.5\t3:4:1\n"),
    ("a token with one colon", b"0\t1:22\n"),
    ("a token without colons", b"0\t122 3:4:1\n"),
    ("a fid of 33 bytes", b"0\t1:" + b"a" * 33 + b":1\n"),
    ("a first field of 17 bytes", b"0\t" + b"1" * 17 + b":2:3\n"),
    ("a blank as the block's first byte", b" 0\t1:2:3\n"),
]


@pytest.mark.parametrize("what,bad", DEFECTS, ids=[d[0].split(":")[0] for d in DEFECTS])
def test_blocks_outside_the_common_shape_are_handed_back(ing, what, bad):
    assert ing.block(GOOD)[0]
    # the defect alone, at the start, in the middle and at the end of a larger block — and past a
    # tile boundary
    filler = gen_block(np.random.RandomState(1), 400, max_tok=10)
    for text in (bad, bad + GOOD, GOOD + bad, filler + bad + filler, GOOD + bad[:-1]):
        if text.endswith(b"\t") or not text:
            continue
        ok = ing.block(text)[0]
        assert not ok, what
    assert ing.block(GOOD)[0]    # (the object is fine afterwards)


def test_empty_text_and_a_block_beyond_the_buffer(ing):
    ok, rp, ks, lb = ing.block(b"")
    assert ok and len(lb) == 0 and len(ks) == 0 and list(rp) == [0]
    small = capi.Ingest(4096)
    with pytest.raises(capi.XFError, match="room for"):
        small.block(b"0\t1:2:3\n" * 1000)
    # more tokens than the arrays are sized for (4 bytes of text per token; "::" with its blank
    # is 3): handed back — the host parser takes empty fids as they come
    dense = b"0\t" + b" ".join([b"::"] * 1300) + b"\n"
    assert len(dense) < 4096 and not small.block(dense)[0]
    assert len(capi.parse_text_block(dense)[1]) == 1300
    ok, rp, ks, lb = small.block(b"0\t" + b" ".join([b"::"] * 900) + b"\n")
    assert ok and len(ks) == 900 and len(set(ks.tolist())) == 1


def _worker_run(tmp_path, ingest, model, data):
    x = capi.XFlow(data["train"], data["test"], model=model, epochs=3, block_size_mb=1,
                   capacity=1 << 16, ingest=ingest, pred_path=str(tmp_path / ("p_" + ingest)))
    x.train()
    out = {m: x.metric(m) for m in ("logloss_ref", "auc", "rows_trained", "keys", "blocks_gpu",
                                    "blocks_host")}
    wh, vh = x.tables()
    tabs = [capi.Table.from_handle(wh, 1).export()]
    if model == 1:
        tabs.append(capi.Table.from_handle(vh, 10).export())
    return out, tabs


@pytest.mark.parametrize("model", [0, 1])
def test_worker_with_gpu_ingest_trains_the_same_model(tmp_path, model):
    """three epochs over a text file whose SECOND block holds quirks (a label 0.00000009, an empty
    token): with ingest = gpu the clean blocks are tokenised on the GPU, that one is handed back
    to the host parser — tables and metric line of ingest = host, bit for bit"""
    rng = np.random.RandomState(11)
    clean = gen_block(rng, 30000, max_tok=14, fid_len=(1, 6), f0_len=(1, 2), rest_len=(1, 4))
    quirky = b"0.00000009\t1:22:0.5  3:4:1\n2e-7\t5:6:7\n"
    lines = clean.split(b"\n")[:-1]
    cut = len(lines) // 2
    text = b"\n".join(lines[:cut]) + b"\n" + quirky + b"\n".join(lines[cut:]) + b"\n"
    assert len(text) > 2 << 20            # three or more 1 MiB blocks
    (tmp_path / "train-00000").write_bytes(text)
    (tmp_path / "test-00000").write_bytes(gen_block(rng, 500, max_tok=14, fid_len=(1, 6)))
    data = {"train": str(tmp_path / "train"), "test": str(tmp_path / "test")}
    host, htabs = _worker_run(tmp_path, "host", model, data)
    gpu, gtabs = _worker_run(tmp_path, "gpu", model, data)
    assert host["blocks_gpu"] == 0 and gpu["blocks_gpu"] >= 2 and gpu["blocks_host"] == 1
    for m in ("logloss_ref", "auc", "rows_trained", "keys"):
        assert host[m] == gpu[m], (m, host[m], gpu[m])
    for a, b in zip(htabs, gtabs):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert open(str(tmp_path / "p_host")).read() == open(str(tmp_path / "p_gpu")).read()


def test_worker_with_gpu_ingest_and_an_empty_training_file(tmp_path):
    """ingest = gpu with a training shard the reader cannot map (an empty file): the epoch falls
    back to the host reader, which yields zero rows — the same run as ingest = host (the init push's
    key 0 and whatever the predict pass pulls), no error."""
    rng = np.random.RandomState(5)
    (tmp_path / "train-00000").write_bytes(b"")
    (tmp_path / "test-00000").write_bytes(gen_block(rng, 300, max_tok=10, fid_len=(1, 5)))
    data = {"train": str(tmp_path / "train"), "test": str(tmp_path / "test")}
    host, htabs = _worker_run(tmp_path, "host", 0, data)
    gpu, gtabs = _worker_run(tmp_path, "gpu", 0, data)
    assert host["rows_trained"] == 0 and gpu["rows_trained"] == 0 and gpu["blocks_gpu"] == 0
    for m in ("logloss_ref", "auc", "keys"):
        assert host[m] == gpu[m], (m, host[m], gpu[m])
    for x, y in zip(htabs[0], gtabs[0]):
        assert np.array_equal(x, y)

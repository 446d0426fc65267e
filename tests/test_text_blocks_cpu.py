"""The block-as-text API of the reader (xf_reader_peek_text / _skip_text / _copy_text /
_parse_text) — what feeds the GPU tokeniser and takes back the blocks it rejects: the blocks'
text by the reference's block rule (load_data_from_disk.cc:104-121), parsed from memory, must be
block for block what xf_reader_next yields from the file (whose parse is pinned to the real
reference's, tests/golden/ref_parse_*.npz).  No GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from xflow_amd import capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,cap", [("small_train-00000", 1000), ("small_train-00000", 4096),
                                      ("small_test-00000", 2097152), ("quirks-00000", 202),
                                      ("quirks-00000", 777), ("quirks-00000", 1048576)])
def test_text_blocks_parse_to_the_readers_blocks(name, cap):
    path = os.path.join(GOLD, name)
    want = list(capi.read_blocks(path, cap))
    texts = list(capi.read_text_blocks(path, cap))
    assert len(texts) == len(want) > 0
    assert sum(len(t) for t in texts) <= os.path.getsize(path)
    for t, (rp, ks, fg, lb) in zip(texts, want):
        assert b"\n" not in t[-1:] or cap > len(t) + 1    # a cut block comes without its newline
        grp, gks, gfg, glb = capi.parse_text_block(t)
        assert np.array_equal(grp, rp) and np.array_equal(gks, ks)
        assert np.array_equal(gfg, fg) and np.array_equal(glb, lb)


def test_copy_text_is_the_peeked_text_and_next_parses_the_same_block():
    path = os.path.join(GOLD, "small_train-00000")
    L = capi.lib()
    r = capi.vp()
    capi.check(L.xf_reader_open(C.byref(r), path.encode(), 4096))
    try:
        t, n = capi.vp(), C.c_size_t()
        capi.check(L.xf_reader_peek_text(r, C.byref(t), C.byref(n)))
        first = C.string_at(t.value, n.value)
        buf = C.create_string_buffer(8192)
        m = C.c_size_t()
        capi.check(L.xf_reader_copy_text(r, buf, 8192, C.byref(m), 3))
        assert m.value == n.value and buf.raw[:m.value] == first
        with pytest.raises(capi.XFError, match="room for"):
            capi.check(L.xf_reader_copy_text(r, buf, 16, C.byref(m), 1))
        # nothing has moved: xf_reader_next parses this very block
        rows, nnz = C.c_size_t(), C.c_size_t()
        rp, ks, fg, lb = capi.u64p(), capi.u64p(), capi.i32p(), capi.i32p()
        capi.check(L.xf_reader_next(r, C.byref(rows), C.byref(nnz), C.byref(rp), C.byref(ks),
                                    C.byref(fg), C.byref(lb)))
        got = np.ctypeslib.as_array(ks, (nnz.value,)).copy()
        assert np.array_equal(got, capi.parse_text_block(first)[1])
        # ... and has moved on: the next peek is the second block
        capi.check(L.xf_reader_peek_text(r, C.byref(t), C.byref(n)))
        assert C.string_at(t.value, n.value) == list(capi.read_text_blocks(path, 4096))[1]
    finally:
        L.xf_reader_close(r)


def test_a_reader_with_a_block_cache_has_no_text_to_peek(tmp_path):
    path = os.path.join(GOLD, "small_train-00000")
    L = capi.lib()
    r = capi.vp()
    hit = C.c_int()
    capi.check(L.xf_reader_open_cached(C.byref(r), path.encode(), 4096,
                                       str(tmp_path / "c").encode(), C.byref(hit)))
    try:
        t, n = capi.vp(), C.c_size_t()
        with pytest.raises(capi.XFError, match="without a block cache"):
            capi.check(L.xf_reader_peek_text(r, C.byref(t), C.byref(n)))
    finally:
        L.xf_reader_close(r)

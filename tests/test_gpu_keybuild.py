"""The range-partitioned key build (xf_keybuild.hip) through the C ABI on a real MI355X: raw
keys -> cells against a table with a settled tier, keys found where the tier's keys sit in LDS,
first-touch keys as holes + a second segment of cells over the arrival rows.  Every case is
checked three ways: against the oracle's exact-sum mode (bit for bit), against the general
build of the same minibatches (table probe per nonzero + radix sort: tune key_build = 1), and
through the shape the build reports (segments)."""
import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

from .test_gpu_parity import same, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu():
    capi.require_gpu()


def general_path(on):
    capi.tune("key_build", 1 if on else 0)


def steps_vs_oracle(t, s, raws, ws, steps, retain=True, defrag_at=None):
    obs = [O.Batch(*x) for x in raws]
    segs = []
    bs = []
    for i in range(steps):
        ob = obs[i % len(obs)]
        if retain:
            if i < len(raws):
                bs.append(capi.LocalBatch(t, *raws[i]))
            b = bs[i % len(bs)]
        else:
            b = capi.LocalBatch(t, *raws[i % len(raws)], retain_keys=False)
        with O.sum_mode(1):
            loss_ex, _ = ob.lr_loss(s.pull(ob.ukeys))
            O.lr_update(s, ob)
        capi.lr_step(t, b, ws)
        segs.append(b.cells_info()["segments"])
        same(ws.fetch_loss(b.R), loss_ex)
        if defrag_at is not None and i == defrag_at:
            t.defrag()
    t.check()
    for a, e in zip(t.export(), s.export()):
        same(a, e)
    return segs


@pytest.mark.parametrize("R,nnz,nkeys,zipf,ragged", [
    (3000, 40, 30000, None, False),      # one window, 15 chunks, 4 super-chunks
    (40000, 25, 300000, None, True),     # three windows, ragged / empty rows, 37 super-chunks
    (30000, 40, 100000, 1.15, True),     # power-law heads: split chunks, a heavy super-chunk
    (20000, 3, 50000, None, False),      # short rows: many rows per scatter tile
])
def test_settled_keys_new_keys_and_the_general_build_agree(R, nnz, nkeys, zipf, ragged):
    """... and the TWO-LEVEL build (key_build = 2 takes it on these small tables: groups of two
    super-chunks, k_kb_hist_groups / k_kb_regroup) gives the same cells' results again"""
    rng = np.random.RandomState(R + nnz)
    ws = capi.Workspace()
    raws = [synth(rng, R, nnz, nkeys, zipf, ragged) for _ in range(4)]
    tabs = []
    for knob in (0, 2, 1):
        capi.tune("key_build", knob)
        try:
            t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 20)
            s = O.Store(O.OPT_FTRL, 1)
            t.push(np.array([0], np.uint64), np.zeros(1, np.float32))   # lr_worker.cc:180-182
            s.push(np.array([0], np.uint64), np.zeros(1, np.float32))
            # minibatch 0 goes in through the general path (no settled tier yet) ...
            segs = steps_vs_oracle(t, s, raws[:1], ws, 1, defrag_at=0)
            assert segs == [1]
            # ... minibatches 1-3 meet a settled tier that holds some of their keys: holes +
            # an arrival segment; after the second defrag everything is settled
            segs = steps_vs_oracle(t, s, raws[1:], ws, 7, defrag_at=3)
            if knob != 1:
                assert segs[0] == 2 and segs[-1] == 1, segs
            tabs.append(t.export())
        finally:
            general_path(False)
    for other in tabs[1:]:
        for a, b in zip(tabs[0], other):
            same(a, b)


def test_one_shot_minibatches_on_a_growing_table():
    """retain_keys = 0 (no key-sorted copy, cells built once), a table that must grow while the
    arrival segment goes in, keys that are not hashes (a run of integers: every key-range guess
    is wrong, the boundary search and the directory must not care)"""
    rng = np.random.RandomState(9)
    t = capi.Table(capi.OPT_SGD, 1, capacity=4096)
    s = O.Store(O.OPT_SGD, 1)
    ws = capi.Workspace()
    R, nnz = 2000, 30
    rowptr = (np.arange(R + 1) * nnz).astype(np.uint64)
    for step in range(5):
        hi = 9000 * (step + 1)
        keys = rng.randint(0, hi, size=R * nnz).astype(np.uint64) + np.uint64(1 << 40)
        if step == 3:
            keys[::97] = np.uint64(2**64 - 1)           # the reserved key value
            keys[5::101] = np.uint64(3)                 # below every settled key
        labels = rng.randint(0, 2, size=R).astype(np.int32)
        b = capi.LocalBatch(t, rowptr, keys, labels, retain_keys=False)
        ob = O.Batch(rowptr, keys, labels)
        with O.sum_mode(1):
            loss_ex, _ = ob.lr_loss(s.pull(ob.ukeys))
            O.lr_update(s, ob)
        capi.lr_step(t, b, ws)
        same(ws.fetch_loss(R), loss_ex)
        assert b.cells_info()["segments"] == (1 if step == 0 else 2)
        t.defrag()
    t.check()
    for a, e in zip(t.export(), s.export()):
        same(a, e)
    assert t.capacity > 4096


def test_predict_and_replay_after_a_renumbering():
    """a retained minibatch is rebuilt by the same path when the table renumbers its rows"""
    rng = np.random.RandomState(2)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 18)
    s = O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    raws = [synth(rng, 5000, 20, 60000) for _ in range(2)]
    obs = [O.Batch(*x) for x in raws]
    bs = [capi.LocalBatch(t, *x) for x in raws]
    for ob in obs:
        s.pull(ob.ukeys)
    for i in range(6):
        with O.sum_mode(1):
            O.lr_update(s, obs[i % 2])
        capi.lr_step(t, bs[i % 2], ws)
        if i in (1, 3):
            t.defrag()
    assert bs[0].cells_info()["segments"] == 1
    with O.sum_mode(1):
        same(capi.lr_predict(t, bs[1], ws), obs[1].lr_loss(s.pull(obs[1].ukeys))[1])
    for a, e in zip(t.export(), s.export()):
        same(a, e)


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_keyed_and_general_builds_give_the_same_table(seed):
    """random minibatch shapes (rows, row lengths, key space, skew, empty rows, windows), a
    random share of the keys settled before: the range-partitioned build and the general build
    end in the same table and the same predictions, bit for bit"""
    rng = np.random.RandomState(1000 + seed)
    R = int(rng.choice([37, 600, 5000, 18000, 36000]))
    nnz = int(rng.choice([1, 3, 17, 60]))
    nkeys = int(rng.choice([2500, 9000, 70000, 400000]))
    zipf = [None, 1.1, 1.6][rng.randint(3)]
    ragged = bool(rng.randint(2))
    opt = [capi.OPT_FTRL, capi.OPT_SGD][rng.randint(2)]
    raws = [synth(rng, R, nnz, nkeys, zipf, ragged) for _ in range(3)]
    # the keys settled beforehand: a random share of the key space (plus none / all)
    share = [0.0, 0.3, 0.9, 1.0][rng.randint(4)]
    pre = np.array([O.hash_str(str(i)) for i in range(nkeys)], np.uint64)
    pre = np.sort(pre[rng.rand(nkeys) < share]) if share < 1.0 else np.sort(pre)
    outs = []
    for general in (False, True):
        general_path(general)
        try:
            t = capi.Table(opt, 1, capacity=2 * len(pre) + 4096)   # (the compiles grow it)
            ws = capi.Workspace()
            if len(pre):
                t.pull(pre)                    # inserts
                t.defrag()
            bs = [capi.LocalBatch(t, *x) for x in raws]   # kept: rebuilt after the defrag
            for i in range(4):
                capi.lr_step(t, bs[i % 3] if i % 3 != 0 else capi.LocalBatch(t, *raws[0],
                                                                            retain_keys=False), ws)
                if i == 1:
                    t.defrag()
            t.check()
            outs.append((t.export(), capi.lr_predict(t, bs[1], ws)))
        finally:
            general_path(False)
    for a, b in zip(outs[0][0], outs[1][0]):
        same(a, b)
    same(outs[0][1], outs[1][1])


def test_a_table_beyond_the_full_tile_limit_takes_half_tiles():
    """2e7 settled keys = 2442 super-chunks: their per-tile arrays do not fit the LDS next to a
    full 8192-record stage, the scatter runs with 4096-record tiles; same losses and
    predictions as the general build"""
    n = 20_000_000
    keys = capi.hash_decimal_range(0, n)
    rng = np.random.RandomState(4)
    R, nnz = 20000, 50
    rowptr = (np.arange(R + 1) * nnz).astype(np.uint64)
    fid = rng.randint(0, n + 5000, size=R * nnz)           # a few keys beyond the settled ones
    extra = capi.hash_decimal_range(n, 5000)
    bk = np.where(fid < n, keys[np.minimum(fid, n - 1)], extra[np.maximum(fid - n, 0)])
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    outs = []
    for general in (False, True):
        general_path(general)
        try:
            t = capi.Table(capi.OPT_FTRL, 1, capacity=2 * n + 4096)
            ws = capi.Workspace()
            srt = np.sort(keys)
            for i in range(0, n, 4_000_000):
                t.pull(srt[i:i + 4_000_000])               # inserts
            t.defrag()
            b = capi.LocalBatch(t, rowptr, bk, labels)
            if not general:
                assert b.cells_info()["segments"] == 2
            capi.lr_step(t, b, ws)
            capi.lr_step(t, b, ws)
            outs.append((ws.fetch_loss(R), capi.lr_predict(t, b, ws)))
            t.check()
            del b, t
        finally:
            general_path(False)
    same(outs[0][0], outs[1][0])
    same(outs[0][1], outs[1][1])


def test_table_growth_is_sized_by_distinct_new_keys():
    """a minibatch of 10^6 nonzeros over 2 * 10^4 keys meets a table with room for them: the
    growth decision counts the distinct new keys, not the nonzeros (which would have taken a
    65 536-position table to 4 M positions) — untiered (first minibatch) and against a settled
    tier alike; a table that really is too small still grows"""
    rng = np.random.RandomState(5)
    ws = capi.Workspace()
    for settled in (False, True):
        t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 16)
        if settled:
            t.pull(capi.hash_decimal_range(100000, 3000))
            t.defrag()
        raw = synth(rng, 20000, 50, 20000)
        b = capi.LocalBatch(t, *raw)
        assert t.capacity == 1 << 16, t.capacity
        capi.lr_step(t, b, ws)
        t.check()
        assert len(t) == len(np.unique(raw[1])) + (3000 if settled else 0)
        raw = synth(rng, 20000, 50, 200000)          # ~2e5 distinct keys: now it must grow
        b2 = capi.LocalBatch(t, *raw)
        assert t.capacity > 1 << 16
        capi.lr_step(t, b2, ws)
        t.check()


def test_update_in_one_call_is_the_key_build_and_the_step():
    """xf_lr_update_dev (the build's host wait taken under the forward) against the two calls
    and the oracle: settled minibatches, minibatches that bring new keys (holes in the first
    forward, a second segment, the forward run again), a defrag in between, a replay of the
    minibatches it returns"""
    R, nnz, nkeys = 20000, 30, 120000
    rng = np.random.RandomState(17)
    ta, tb = (capi.Table(capi.OPT_FTRL, 1, capacity=1 << 19) for _ in range(2))
    so = O.Store(O.OPT_FTRL, 1)
    wa, wb = capi.Workspace(), capi.Workspace()
    kept = []
    segs = []
    for step in range(8):
        # steps 0-1 before any settled tier (the general build: nothing deferred), 2-3 settled
        # keys only, 4-5 half the keys new, 6-7 settled again after the second defrag
        lo, hi = (0, nkeys // 2) if step < 4 else (0, nkeys)
        raw = synth(rng, R, nnz, hi - lo, 1.15 if step % 2 else None, True)
        ob = O.Batch(*raw)
        with O.sum_mode(1):
            loss_ex, _ = ob.lr_loss(so.pull(ob.ukeys))
            O.lr_update(so, ob)
        b = capi.LocalBatch.update(ta, wa, *raw)
        same(wa.fetch_loss(R), loss_ex)
        capi.lr_step(tb, capi.LocalBatch(tb, *raw), wb)
        same(wb.fetch_loss(R), loss_ex)
        segs.append(b.cells_info()["segments"])
        kept.append((b, ob))
        if step in (1, 5):                       # every key of the range so far, then settle
            allk = capi.hash_decimal_range(0, hi)
            for t in (ta, tb, so):
                t.pull(allk)
            ta.defrag()
            tb.defrag()
    assert segs[2:] == [1, 1, 2, 2, 1, 1], segs
    for b, ob in kept[-2:]:                      # replays of what the one call returned
        with O.sum_mode(1):
            loss_ex, _ = ob.lr_loss(so.pull(ob.ukeys))
            O.lr_update(so, ob)
        capi.lr_step(ta, b, wa)
        same(wa.fetch_loss(R), loss_ex)
    ta.check()
    for a, e in zip(ta.export(), so.export()):
        same(a, e)


def test_two_level_build_on_a_table_beyond_the_one_level_limit():
    """3.6e7 settled keys (the one-level partition stops at 3.4e7: the scatter's per-super-chunk
    arrays no longer fit the LDS): the minibatch takes the two-level build by itself — against
    the sort-based build (key_build = 1) on a second table and against the oracle, on the keys the
    minibatch touches (every other row of the tables is zero), with keys the tier does not hold
    among them (holes + an arrival segment)."""
    nkeys = 36_000_000
    keytab = capi.hash_decimal_range(0, nkeys + 4000)
    rng = np.random.RandomState(3)
    R, nnz = 30000, 40
    rowptr = (np.arange(R + 1) * nnz).astype(np.uint64)
    fid = rng.randint(0, nkeys, size=R * nnz)
    fid[::997] = nkeys + rng.randint(0, 4000, size=len(fid[::997]))     # not in the table yet
    keys = keytab[fid]
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    ob = O.Batch(rowptr, keys, labels)
    s = O.Store(O.OPT_FTRL, 1)
    with O.sum_mode(1):
        loss_ex, _ = ob.lr_loss(s.pull(ob.ukeys))
        O.lr_update(s, ob)
    want = s.pull(ob.ukeys)
    ws = capi.Workspace()
    got = []
    for knob in (0, 1):
        t = capi.Table(capi.OPT_FTRL, 1, capacity=2 * nkeys + 65536)
        for lo in range(0, nkeys, 9_000_000):     # every key once: the table holds them all
            kk = keytab[lo:min(lo + 9_000_000, nkeys)]
            rows = len(kk) // 200
            rp = (np.arange(rows + 1, dtype=np.uint64) * np.uint64(200))
            rp[-1] = len(kk)
            capi.LocalBatch(t, rp, kk, np.zeros(rows, np.int32), retain_keys=False)
        t.defrag()
        assert len(t) == nkeys
        capi.tune("key_build", knob)
        try:
            b = capi.LocalBatch(t, rowptr, keys, labels, retain_keys=False)
            info = b.cells_info()
            capi.lr_step(t, b, ws)
        finally:
            capi.tune("key_build", 0)
        t.check()
        assert info["segments"] == (2 if knob == 0 else 1), (knob, info)   # (the general build
        same(ws.fetch_loss(R), loss_ex)                                    #  makes one segment)
        got.append(t.pull(ob.ukeys))
        del b, t
    same(got[0], want)
    same(got[1], want)


@pytest.mark.parametrize("cap", [256, 1024, 5000])
def test_defrag_without_a_sort_a_cluster_that_wraps_around_the_end_of_the_index(cap):
    """xf_table_defrag reads the order-preserving arrival index front to back (k_df_count / _sort /
    _merge): keys at the very top of the key space pile up at the index's last position and run on
    into its first ones, where the smallest keys live — one cluster with keys of both ends.  Twice
    (the second defrag merges with a settled tier), against the radix-sort defrag (key_build = 1)
    and the oracle's store."""
    rng = np.random.RandomState(cap)
    top = (np.uint64(2**64 - 2) - np.arange(40, dtype=np.uint64))
    low = np.arange(1, 41, dtype=np.uint64)
    mid = capi.hash_decimal_range(0, cap // 4)
    waves = [np.unique(np.concatenate([top[:20], low[:20], mid[: len(mid) // 2]])),
             np.unique(np.concatenate([top, low, mid, capi.hash_decimal_range(10**6, cap // 16)]))]
    grads = [rng.randn(len(k)).astype(np.float32) for k in waves]
    exports = []
    for mode in (0, 1):
        capi.tune("key_build", mode)
        try:
            t = capi.Table(capi.OPT_FTRL, 1, capacity=cap)
            s = O.Store(O.OPT_FTRL, 1)
            for keys, g in zip(waves, grads):
                t.push(keys, g)
                s.push(keys, g)
                t.defrag()
                for a, e in zip(t.export(), s.export()):
                    same(a, e)
                same(t.pull(keys), s.pull(keys))
            exports.append(t.export())
        finally:
            capi.tune("key_build", 0)
    for a, b in zip(*exports):
        same(a, b)


def _first_minibatches(kind, rng):
    R, nnz = 6000, 30
    rowptr = (np.arange(R + 1) * nnz).astype(np.uint64)
    n = R * nnz
    if kind == "hashed":
        return [synth(rng, R, nnz, 60000) for _ in range(3)]
    if kind == "ragged_zipf":
        return [synth(rng, R, nnz, 40000, 1.2, True) for _ in range(3)]
    out = []
    for step in range(3):
        if kind == "integers":          # every key in the first key range
            keys = rng.randint(0, 50000, size=n).astype(np.uint64)
        elif kind == "reserved":        # hashes, among them the reserved key value
            keys = synth(rng, R, nnz, 60000)[1].copy()
            keys[::211] = np.uint64(2**64 - 1)
        elif kind == "clustered":       # 256 keys next to each of 300 hashes: one home each
            base = synth(rng, R, nnz, 300)[1]
            keys = base + rng.randint(0, 256, size=n).astype(np.uint64)
        elif kind == "one_key":
            keys = np.full(n, capi.hash_decimal_range(7, 1)[0], np.uint64)
        else:
            raise AssertionError(kind)
        out.append((rowptr, keys, rng.randint(0, 2, size=R).astype(np.int32)))
    return out


@pytest.mark.parametrize("kind,settles", [
    ("hashed", True), ("ragged_zipf", True), ("one_key", True),
    ("integers", False), ("reserved", False), ("clustered", False),
])
def test_the_first_minibatch_settles_an_empty_table(kind, settles):
    """xf_keybuild.hip "an empty table": the keys of the first minibatch become the settled tier
    at once (sorted in LDS, range by range) — bit for bit the oracle, and the table that took the
    same keys through the arrival index (key_build = 3) and a defrag.  Keys the LDS sort does not
    take (a run of integers: one key range holds them all, more than its set has room for; the
    reserved key value; hundreds of keys with one home) go the old way, and nothing is settled.
    One key in every nonzero, a power-law head: a range's set holds DISTINCT keys."""
    rng = np.random.RandomState(len(kind))
    raws = _first_minibatches(kind, rng)
    ws = capi.Workspace()
    tabs = []
    for mode in (0, 3):
        capi.tune("key_build", mode)
        try:
            t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 12)     # (grows under the first build)
            s = O.Store(O.OPT_FTRL, 1)
            b0 = capi.LocalBatch(t, *raws[0])
            nu = len(np.unique(raws[0][1]))
            assert len(t) == nu
            assert t.settled == (nu if settles and mode == 0 else 0), (t.settled, nu)
            assert b0.cells_info()["segments"] == 1
            if mode == 3:
                t.defrag()          # (the cells of b0 are built again: its keys are retained)
            ob0 = O.Batch(*raws[0])
            for _ in range(2):
                with O.sum_mode(1):
                    loss_ex, _ = ob0.lr_loss(s.pull(ob0.ukeys))
                    O.lr_update(s, ob0)
                capi.lr_step(t, b0, ws)
                same(ws.fetch_loss(b0.R), loss_ex)
            # the next minibatches: settled keys, holes, an arrival segment
            steps_vs_oracle(t, s, raws[1:], ws, 4, defrag_at=1)
            tabs.append(t.export())
        finally:
            capi.tune("key_build", 0)
    for a, b in zip(tabs[0], tabs[1]):
        same(a, b)


def test_an_empty_sgd_table_with_constant_initial_weights_settles_too():
    """the settled rows' initial weights are the init kind's (k_first_rows), as a first Pull
    would have left them (sgd.h / ftrl.h: the store's default)"""
    rng = np.random.RandomState(77)
    raws = [synth(rng, 5000, 20, 30000) for _ in range(2)]
    outs = []
    for mode in (0, 3):
        capi.tune("key_build", mode)
        try:
            t = capi.Table(capi.OPT_SGD, 1, init=capi.INIT_CONST, init_const=0.25, capacity=1 << 17)
            ws = capi.Workspace()
            b = capi.LocalBatch(t, *raws[0])
            assert t.settled == (len(np.unique(raws[0][1])) if mode == 0 else 0)
            w0 = t.pull(np.unique(raws[0][1]))
            assert np.all(w0 == np.float32(0.25))
            capi.lr_step(t, b, ws)
            b1 = capi.LocalBatch(t, *raws[1])
            capi.lr_step(t, b1, ws)
            t.check()
            outs.append((ws.fetch_loss(b1.R), t.export()))
        finally:
            capi.tune("key_build", 0)
    same(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        same(a, b)


@pytest.mark.parametrize("opt", [capi.OPT_FTRL, capi.OPT_SGD])
def test_keys_the_host_api_pushed_first_do_not_keep_the_first_minibatch_from_settling(opt):
    """lr_worker.cc:180-182 pushes key 0 before the first minibatch: the table is not empty, but
    what it holds is what the host API put there (the table knows: its key count is theirs) — the
    first build takes those keys out of the arrival index, settles the table with the minibatch's
    keys and puts them back with their state: a key of the minibatch into its settled row, the
    others behind the tier.  Bit for bit the oracle, whichever way the keys went in."""
    rng = np.random.RandomState(23)
    raws = [synth(rng, 5000, 20, 40000) for _ in range(3)]
    in_batch = np.unique(raws[0][1])[[5, 700]]
    elsewhere = capi.hash_decimal_range(10**7, 3)
    tabs = []
    for mode in (0, 3):
        capi.tune("key_build", mode)
        try:
            t = capi.Table(opt, 1, capacity=1 << 18)
            s = O.Store(opt, 1)
            ws = capi.Workspace()
            for keys, g in ((np.array([0], np.uint64), np.zeros(1, np.float32)),
                            (np.sort(np.concatenate([in_batch, elsewhere])),
                             np.linspace(-0.5, 0.75, 5).astype(np.float32))):
                t.push(keys, g)
                s.push(keys, g)
            assert len(t) == 6 and t.settled == 0
            b0 = capi.LocalBatch(t, *raws[0])
            nu = len(np.unique(raws[0][1]))
            assert len(t) == nu + 4                      # key 0 and the three from elsewhere
            assert t.settled == (nu if mode == 0 else 0)
            steps_vs_oracle(t, s, raws, ws, 5, defrag_at=2)
            tabs.append(t.export())
        finally:
            capi.tune("key_build", 0)
    for a, b in zip(tabs[0], tabs[1]):
        same(a, b)


def test_a_table_others_hold_rows_of_is_not_renumbered_by_its_first_minibatch():
    """more keys than the table remembers (4096), or keys that did not come through the host
    API: the first minibatch goes through the arrival index"""
    rng = np.random.RandomState(29)
    raw = synth(rng, 3000, 20, 20000)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 17)
    t.pull(capi.hash_decimal_range(10**6, 5000))
    b = capi.LocalBatch(t, *raw)
    assert t.settled == 0 and b.cells_info()["segments"] == 1
    t.check()


@pytest.mark.parametrize("shard,nshards", [(0, 4), (2, 4), (3, 4), (7, 8)])
def test_a_shard_of_the_key_space_settles_on_its_first_minibatch_too(shard, nshards):
    """the uniform key ranges and a range's homes are cut from the SHARD's key span (ps-lite's
    uniform ranges): the first and a middle shard, and the last one, whose span takes
    the division's remainder"""
    rng = np.random.RandomState(40 + shard)
    allk = capi.hash_decimal_range(0, 100000)
    mine = allk[np.array([capi.lib().xf_shard_of(int(k), nshards) for k in allk]) == shard]
    assert len(mine) > 5000
    R, nnz = 4000, 25
    rowptr = (np.arange(R + 1) * nnz).astype(np.uint64)
    raws = [(rowptr, mine[rng.randint(0, len(mine), size=R * nnz)],
             rng.randint(0, 2, size=R).astype(np.int32)) for _ in range(3)]
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 16, shard=shard, nshards=nshards)
    s = O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    b0 = capi.LocalBatch(t, *raws[0])
    assert t.settled == len(np.unique(raws[0][1])) == len(t)
    del b0
    steps_vs_oracle(t, s, raws, ws, 4, defrag_at=2)


def test_a_table_of_rows_of_16_hashed_initial_values_settles_with_them():
    """k_first_rows: the rows of a table settled by its first minibatch carry the init kind's
    values of THEIR keys (FM's v table: hashnorm(seed, key, j), fm_worker.cc's store default) —
    what a Pull through the arrival index puts there; checked against the oracle's store"""
    rng = np.random.RandomState(31)
    raw = synth(rng, 3000, 20, 20000)
    t = capi.Table(capi.OPT_FTRL, 16, capi.INIT_HASHNORM, seed=1234, capacity=1 << 16)
    s = O.Store(O.OPT_FTRL, 16, O.INIT_HASHNORM, 0.0, 1234)
    b = capi.LocalBatch(t, *raw, retain_keys=False)     # (the cells are not stepped: dim 16)
    uk = np.unique(raw[1])
    assert t.settled == len(uk) == len(t)
    assert np.array_equal(t.pull(uk), s.pull(uk))
    fresh = capi.hash_decimal_range(10**6, 100)          # later arrivals: behind the tier
    assert np.array_equal(t.pull(fresh), s.pull(fresh))
    t.check()
    del b


# ------------------------------------------------------------ (key, position) in key order
def _check_sorted(keys, lo=0, span=2**64 - 1, by_hand=None):
    sk, sp, h = capi.sort_key_pos(keys, lo, span)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(sp, order.astype(np.uint32))
    assert np.array_equal(sk, keys[order])
    if by_hand is not None:
        assert h == by_hand
    return h


@pytest.mark.parametrize("n", [1, 5, 3000, 8192, 70001, 2_000_000])
def test_sort_key_pos_hashed_keys(n):
    """xf_sort_key_pos (lr_worker.cc:146-166's std::sort of (fid, sid) as a device routine): hashed
    keys with repeats go through the hand-written sort — uniform key ranges, a range in LDS —
    and come out as numpy's stable argsort orders them."""
    rng = np.random.RandomState(n % 1000)
    pool = rng.randint(0, 2**63, size=max(1, n // 2)).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    keys = pool[rng.randint(0, len(pool), size=n)]
    _check_sorted(keys, by_hand=True)
    # the library's sort (the tests' second implementation) agrees
    capi.tune("key_build", 1)
    try:
        _check_sorted(keys, by_hand=False)
    finally:
        capi.tune("key_build", 0)


def test_sort_key_pos_hot_keys_pieces_of_every_length():
    """keys with 2 ... 6000 occurrences in one minibatch: a range's pieces ranked by comparison,
    sorted by a wavefront, sorted by the workgroup"""
    rng = np.random.RandomState(7)
    n = 1_500_000
    keys = rng.randint(0, 2**63, size=n).astype(np.uint64) * np.uint64(2)
    at = rng.permutation(n)
    o = 0
    for c in [2, 3, 31, 32, 33, 40, 64, 65, 100, 511, 512, 513, 700, 1024, 1560, 2047, 2048, 4000]:
        keys[at[o:o + c]] = np.uint64(rng.randint(0, 2**63)) * np.uint64(2) + np.uint64(1)
        o += c
    # two keys that differ in their lowest bits only (one piece, two keys, hundreds of records)
    k = np.uint64(rng.randint(0, 2**62)) * np.uint64(4)
    keys[at[o:o + 300]] = k
    keys[at[o + 300:o + 700]] = k + np.uint64(1)
    _check_sorted(keys, by_hand=True)


def test_sort_key_pos_key_range_of_a_shard_and_the_extremes():
    """an owner's keys (shard 5 of 8) with the shard's range given, keys outside the given range,
    the reserved key value and key 0"""
    rng = np.random.RandomState(9)
    span = (2**64 - 1) // 8
    lo = 5 * span
    n = 400_000
    keys = (np.uint64(lo) + rng.randint(0, span >> 1, size=n).astype(np.uint64) * np.uint64(2))
    keys[rng.randint(0, n, size=2000)] = keys[0]
    _check_sorted(keys, lo, span, by_hand=True)
    keys[:7] = np.array([0, 1, 2**64 - 1, 2**64 - 2, lo - 1, lo + span + 1, 2**63], dtype=np.uint64)
    keys[100:107] = keys[:7]
    _check_sorted(keys, lo, span, by_hand=True)
    _check_sorted(keys, by_hand=True)   # (no range given: seven eighths of the ranges empty, the
                                        # others beyond a range's LDS: merged)


def test_sort_key_pos_heavy_ranges_are_a_merge_sort():
    """a power-law head (one key with a tenth, a third of the nonzeros; Zipf 1.1), keys that are
    no hashes (every key in one range), all keys equal: the ranges beyond a range's LDS are
    sorted part by part and merged — no library sort"""
    rng = np.random.RandomState(11)
    n = 300_000
    keys = rng.randint(0, 2**63, size=n).astype(np.uint64) * np.uint64(2)
    keys[rng.randint(0, n, size=n // 10)] = np.uint64(12345678901234567)
    _check_sorted(keys, by_hand=True)
    small = rng.randint(0, 5000, size=n).astype(np.uint64)
    _check_sorted(small, by_hand=True)
    _check_sorted(small[:6000], by_hand=True)   # (few enough for one range's LDS)
    _check_sorted(small[:8193], by_hand=True)   # (two parts, the second of one record)
    _check_sorted(np.zeros(20_000, np.uint64) + np.uint64(77), by_hand=True)
    _check_sorted(np.zeros(0, np.uint64))
    n = 3_000_000
    table = rng.randint(0, 2**63, size=1_000_000).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    zipf = table[np.minimum(rng.zipf(1.1, size=n), len(table)) - 1]
    _check_sorted(zipf, by_hand=True)
    zipf[rng.randint(0, n, size=n // 3)] = table[5]
    _check_sorted(zipf, by_hand=True)
    # the shard's range given and not met (all keys in the first of 1000 ranges)
    _check_sorted(zipf >> np.uint64(12), 0, 2**64 - 1, by_hand=True)

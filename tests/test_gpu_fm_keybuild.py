"""The FM key build against the tables' settled tiers (xf_batch_compile_fm_dev, round 4): the
range-partitioned build of xf_keybuild.hip with records that carry a nonzero's row and its
position, the per-super-chunk regroup into the key list / occurrence lists / per-nonzero record
index — no sort of (key, position) pairs.  A minibatch built that way must step exactly like
one built by the sort (xf_batch_compile_dev): loss and both tables bit for bit against the
exact-sum oracle, on uniform and power-law minibatches, ragged and empty rows, several row
windows and super-chunks, heavy keys; and it must fall back to the sort-based build whenever a
key is not settled or the two tables number their rows differently."""
import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

from .test_gpu_parity import same, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu():
    capi.require_gpu()


def _tables(k, opt, nkeys, cap=1 << 18):
    go, oo = (capi.OPT_SGD, O.OPT_SGD) if opt == "sgd" else (capi.OPT_FTRL, O.OPT_FTRL)
    gi, oi = (capi.INIT_CONST, O.INIT_CONST) if opt == "sgd" else (capi.INIT_HASHNORM,
                                                                    O.INIT_HASHNORM)
    tw = capi.Table(go, 1, capacity=cap)
    tv = capi.Table(go, k, gi, 0.001, seed=7, capacity=cap)
    sw, sv = O.Store(oo, 1), O.Store(oo, k, oi, 0.001, 7)
    return tw, tv, sw, sv


def _settle(tw, tv, sw, sv, keys):
    """every key into both tables (a Pull inserts, ftrl.h:56), then the maintenance step"""
    keys = np.unique(keys)
    for t in (tw, tv, sw, sv):
        t.pull(keys)
    tw.defrag()
    tv.defrag()


@pytest.mark.parametrize("k,opt,R,nnz,nkeys", [
    (16, "sgd", 3000, 60, 60000),      # eight super-chunks, one row window
    (64, "ftrl", 1500, 40, 9000),      # k = 64 + FTRL (configs[4]'s model), two super-chunks
    (4, "ftrl", 40000, 12, 150000),    # three row windows
])
def test_keyed_fm_minibatches_step_like_the_oracle(k, opt, R, nnz, nkeys):
    rng = np.random.RandomState(k + R)
    tw, tv, sw, sv = _tables(k, opt, nkeys)
    keytab = capi.hash_decimal_range(0, nkeys)
    _settle(tw, tv, sw, sv, keytab)
    ws = capi.Workspace()
    for step in range(5):
        raw = synth(rng, R, nnz, nkeys, 1.2 if step % 2 else None, True)
        b = capi.FmBatch(tw, tv, *raw)
        assert b.keyed, "every key is settled in both tables: the range-partitioned build"
        ob = O.Batch(*raw)
        assert b.U == ob.U
        if step % 2:
            assert b.H > 0                       # power-law heads: the heavy-key kernels
        with O.sum_mode(1):
            loss_ex, _, _ = ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))
            O.fm_update(sw, sv, ob)
        capi.fm_step(tw, tv, b, ws)
        same(ws.fetch_loss(R), loss_ex)
        tw.check()
        tv.check()
    for t, st in ((tw, sw), (tv, sv)):
        for a, e in zip(t.export(), st.export()):
            same(a, e)


def test_keyed_and_sorted_builds_are_interchangeable_step_by_step():
    """the same minibatches alternately through the two builds on one pair of tables, a third
    pair stepping only sort-built ones: the same tables"""
    k, nkeys, R = 16, 30000, 2500
    rng = np.random.RandomState(3)
    keytab = capi.hash_decimal_range(0, nkeys)
    ta = _tables(k, "ftrl", nkeys)
    tb = _tables(k, "ftrl", nkeys)
    for t in (ta, tb):
        _settle(t[0], t[1], t[2], t[3], keytab)
    ws = capi.Workspace()
    for step in range(6):
        raw = synth(rng, R, 50, nkeys, 1.3 if step == 4 else None, True)
        if step % 2:
            ba = capi.FmBatch(ta[0], ta[1], *raw)
            assert ba.keyed
        else:
            ba = capi.Batch(*raw, on_gpu=True)
        capi.fm_step(ta[0], ta[1], ba, ws)
        capi.fm_step(tb[0], tb[1], capi.Batch(*raw, on_gpu=True), ws)
    for x, y in ((ta[0], tb[0]), (ta[1], tb[1])):
        for a, e in zip(x.export(), y.export()):
            same(a, e)


def test_falls_back_to_the_sort_when_a_key_is_not_settled_or_the_tables_differ():
    k, nkeys, R = 8, 20000, 1200
    rng = np.random.RandomState(5)
    keytab = capi.hash_decimal_range(0, nkeys)
    tw, tv, sw, sv = _tables(k, "sgd", nkeys)
    ws = capi.Workspace()
    raw = synth(rng, R, 30, nkeys, None, True)
    b = capi.FmBatch(tw, tv, *raw)               # empty tables: nothing settled
    assert not b.keyed
    capi.fm_step(tw, tv, b, ws)
    with O.sum_mode(1):
        O.fm_update(sw, sv, O.Batch(*raw))
    tw.defrag()
    tv.defrag()
    raw2 = synth(rng, R, 30, nkeys, None, True)  # mostly keys the tier has not seen
    b2 = capi.FmBatch(tw, tv, *raw2)
    assert not b2.keyed
    capi.fm_step(tw, tv, b2, ws)
    with O.sum_mode(1):
        O.fm_update(sw, sv, O.Batch(*raw2))
    for t, st in ((tw, sw), (tv, sv)):
        for a, e in zip(t.export(), st.export()):
            same(a, e)
    # ... settled now, but the w table holds one key more than the v table: other numbering
    tw.pull(np.array([12345], np.uint64))
    sw.pull(np.array([12345], np.uint64))
    tw.defrag()
    tv.defrag()
    raw3 = (raw[0], raw[1], raw[2])              # keys of the first minibatch: all settled
    b3 = capi.FmBatch(tw, tv, *raw3)
    assert not b3.keyed
    capi.fm_step(tw, tv, b3, ws)
    with O.sum_mode(1):
        O.fm_update(sw, sv, O.Batch(*raw3))
    for t, st in ((tw, sw), (tv, sv)):
        for a, e in zip(t.export(), st.export()):
            same(a, e)


def test_a_keyed_minibatch_scores_and_survives_a_renumbering():
    """xf_fm_predict on a keyed minibatch (the step's forward on the table-resident records) =
    the sort-built one's; after new keys and a defrag the keyed minibatch's rows are looked up
    again and its record index translated: it steps like the oracle; in parity mode it is
    refused (no CSR index of the key list)"""
    k, nkeys, R = 8, 20000, 1000
    rng = np.random.RandomState(6)
    keytab = capi.hash_decimal_range(0, nkeys)
    tw, tv, sw, sv = _tables(k, "ftrl", nkeys)
    _settle(tw, tv, sw, sv, keytab)
    ws = capi.Workspace()
    raw = synth(rng, R, 30, nkeys, 1.2, True)
    b = capi.FmBatch(tw, tv, *raw)
    assert b.keyed
    bs = capi.Batch(*raw, on_gpu=True)
    ob = O.Batch(*raw)
    same(capi.fm_predict(tw, tv, b, ws), capi.fm_predict(tw, tv, bs, ws))
    capi.fm_step(tw, tv, b, ws)
    with O.sum_mode(1):
        O.fm_update(sw, sv, ob)
    same(capi.fm_predict(tw, tv, b, ws), capi.fm_predict(tw, tv, bs, ws))
    new = np.array([777, 778, 1 << 60], np.uint64)   # new keys, then a renumbering
    for t in (tv, tw, sw, sv):
        t.pull(new)
    tw.defrag()
    tv.defrag()
    for step in range(3):                            # the old minibatch, translated
        capi.fm_step(tw, tv, b, ws)
        with O.sum_mode(1):
            loss_ex, _, _ = ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))
            O.fm_update(sw, sv, ob)
        same(ws.fetch_loss(R), loss_ex)
    same(capi.fm_predict(tw, tv, b, ws), capi.fm_predict(tw, tv, bs, ws))
    raw2 = synth(rng, R, 30, nkeys, None, True)      # and a fresh one in the new numbering
    b2 = capi.FmBatch(tw, tv, *raw2)
    assert b2.keyed
    capi.fm_step(tw, tv, b2, ws)
    with O.sum_mode(1):
        O.fm_update(sw, sv, O.Batch(*raw2))
    for t, st in ((tw, sw), (tv, sv)):
        t.check()
        for a, e in zip(t.export(), st.export()):
            same(a, e)
    wp = capi.Workspace()
    wp.parity("reference_order")
    with pytest.raises(capi.XFError, match="table-resident records"):
        capi.fm_step(tw, tv, b2, wp)


def test_factor_widths_without_table_records_take_the_sort_based_build():
    """k = 10 (the reference's default): no table-resident records, so no keyed minibatch"""
    k, nkeys, R = 10, 20000, 1000
    rng = np.random.RandomState(8)
    keytab = capi.hash_decimal_range(0, nkeys)
    tw, tv, sw, sv = _tables(k, "ftrl", nkeys)
    _settle(tw, tv, sw, sv, keytab)
    ws = capi.Workspace()
    for step in range(2):
        raw = synth(rng, R, 30, nkeys, None, True)
        b = capi.FmBatch(tw, tv, *raw)
        assert not b.keyed
        capi.fm_step(tw, tv, b, ws)
        with O.sum_mode(1):
            O.fm_update(sw, sv, O.Batch(*raw))
    for t, st in ((tw, sw), (tv, sv)):
        for a, e in zip(t.export(), st.export()):
            same(a, e)


def test_the_one_shard_trainer_compiles_fm_minibatches_against_its_settled_tables():
    """capi.Sharded (what the worker drives) on one GPU, FM: after the defrag its compile is the
    range-partitioned build (xf_batch_compile_fm) — training steps, scoring and a second defrag
    in between equal the oracle's"""
    k, nkeys, R = 16, 40000, 2000
    rng = np.random.RandomState(11)
    st = capi.Sharded(None, model="fm", optimizer="ftrl", k=k, capacity=1 << 18, seed=7)
    sw, sv = O.Store(O.OPT_FTRL, 1), O.Store(O.OPT_FTRL, k, O.INIT_HASHNORM, 0.001, 7)
    keyed = 0
    for step in range(7):
        raw = synth(rng, R, 40, nkeys, 1.2 if step % 3 == 2 else None, True)
        sb = st.compile(*raw)
        keyed += int(sb.keyed)
        st.step(sb)
        with O.sum_mode(1):
            O.fm_update(sw, sv, O.Batch(*raw))
        if step in (1, 4):                       # every key of the key space, then settle
            allk = capi.hash_decimal_range(0, nkeys)
            for t in (st.w, st.v, sw, sv):
                t.pull(allk)
            st.defrag()
        if step == 5:
            ob = O.Batch(*raw)
            with O.sum_mode(1):
                loss, pctr, _ = ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))
            same(st.predict(st.compile(*raw)), pctr)
    st.check()
    assert keyed >= 4, "after the defrag every minibatch's keys are settled"
    for t, so in ((st.w, sw), (st.v, sv)):
        for a, e in zip(t.export(), so.export()):
            same(a, e)

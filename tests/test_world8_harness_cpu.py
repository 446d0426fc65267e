"""The 8-rank full-size harness (tests/_world8.py) without a GPU, at small sizes: the
generators, the oracle schedules it runs and the shard-by-shard comparison — the part of
tests/test_gpu_world8_fullsize.py that is not the product."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O

from . import _world8 as H


def _fake_rank_files(outdir, world, name, export):
    """what the ranks would write: the oracle's own store cut by the owner rule, every shard
    sorted by key"""
    ks, ws, ns, zs = export
    owner = np.array([O.lib().xo_shard_of(int(k), world) for k in ks])
    for r in range(world):
        m = owner == r
        for f, a in (("k", ks), ("w", ws), ("n", ns), ("z", zs)):
            np.save(os.path.join(outdir, "rank%d_%s_%s.npy" % (r, name, f)), a[m])


def test_generators_hash_decimal_fids_and_are_reproducible():
    pytest.importorskip("xflow_amd.capi").lib()
    a = H.lr_minibatch(3, 1, 50, 20, 10**8)
    b = H.lr_minibatch(3, 1, 50, 20, 10**8)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    rng = np.random.RandomState(H.SEED + 1000 + 3)
    fid = rng.randint(0, 10**8, size=1000)
    assert [int(x) for x in a[1][:5]] == [O.hash_str(str(int(f))) for f in fid[:5]]
    z = H.zipf_minibatch(0, 0, 100, 50, 10**9)
    assert int(z[0][-1]) == 5000 and len(np.unique(z[1])) < 3000     # a power law repeats keys
    assert O.hash_str("0") in set(int(x) for x in z[1])                # its head is fid "0"


def test_rank_ordered_and_concatenated_schedules_and_the_shard_comparison(tmp_path):
    pytest.importorskip("xflow_amd.capi").lib()
    world = 3
    data = [[H.lr_minibatch(r, s, 60 + 7 * r, 12, 400) for r in range(world)] for s in range(2)]
    with O.sum_mode(1):
        w = H.oracle_rank_ordered_lr(O, data, "ftrl", reserve=1000)
        ref = O.Store(O.OPT_FTRL, 1)           # the same schedule, written out
        for mbs in data:
            obs = [O.Batch(*d) for d in mbs]
            pulled = [ref.pull(ob.ukeys) for ob in obs]
            for ob, pw in zip(obs, pulled):
                ref.push(ob.ukeys, ob.lr_grad(ob.lr_loss(pw)[0]))
        for a, b in zip(w.export(), ref.export()):
            H.same(a, b)
        whole = H.oracle_concat_lr(O, data, "ftrl")
        rows = sum(len(d[2]) for d in data[0])
        assert len(H.concat(data[0])[0]) == rows + 1
        assert not np.array_equal(whole.export()[1], w.export()[1])     # different update rules
        out = str(tmp_path)
        _fake_rank_files(out, world, "w", w.export())
        assert H.compare_tables(O, out, world, "w", w.export()) == len(w)
        with pytest.raises(AssertionError):                               # ... and it can fail
            H.compare_tables(O, out, world, "w", whole.export())
        sw, sv = H.oracle_concat_fm(O, data, "ftrl", 4, 7)
        _fake_rank_files(out, world, "v", sv.export())
        assert H.compare_tables(O, out, world, "v", sv.export()) == len(sv)

"""N>1 path on CPU: world_size 2 (and 3) over gloo.  The collectives, the owner bucketing
and the rank-ordered owner updates of xflow_amd.sharded.ShardedTrainer are the code under
test; the arithmetic is the oracle through tests/_cpu_stages.py (a test double — the
product's own stages need a GPU).  Expected result: bit-identical to a single-process
simulation of the same schedule (all workers pull, then push in rank order) on one store."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(rank, step, R=120, nnz=9, nkeys=700):
    from oracle import pyoracle as O
    rng = np.random.RandomState(1000 * step + rank)
    keytab = np.array([O.hash_str(str(i)) for i in range(nkeys)], dtype=np.uint64)
    lens = rng.randint(0, 2 * nnz + 1, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keys = keytab[rng.randint(0, nkeys, size=int(lens.sum()))]
    labels = rng.randint(0, 2, size=R).astype(np.int32)
    return rowptr, keys, labels


def _worker(rank, world, port, model, optimizer, steps, outdir, schedule="sequential",
            merged=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests._cpu_stages import CpuOracleStages, CpuOracleStagesMerged
    from xflow_amd.sharded import ShardedTrainer
    st = (CpuOracleStagesMerged if merged else CpuOracleStages)(model, optimizer, 4, rank, world)
    tr = ShardedTrainer(model=model, optimizer=optimizer, k=4, rank=rank, world=world,
                        stages=st, schedule=schedule)
    for s in range(steps):
        b = tr.compile(*_data(rank, s))
        assert sum(b.send_counts) == b.U
        tr.step(b)
    tr.flush()
    pb = tr.compile(*_data(rank, 99))
    loss = tr.predict(pb).numpy()
    out = {"loss": loss}
    for nm, t in (("w", st.w), ("v", st.v)):
        if t is not None:
            k, w, n, z = t.store.export()
            out.update({nm + "_k": k, nm + "_w": w, nm + "_n": n, nm + "_z": z})
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def _simulate(world, model, optimizer, steps, schedule="sequential"):
    from oracle import pyoracle as O
    opt = O.OPT_FTRL if optimizer == "ftrl" else O.OPT_SGD
    w = O.Store(opt, 1)
    v = None
    if model == "fm":
        v = O.Store(opt, 4, O.INIT_HASHNORM if opt == O.OPT_FTRL else O.INIT_CONST, 0.001, 7)
    outstanding = None      # stale1: the Push of the previous step, applied after this Pull
    for s in range(steps):
        obs = [O.Batch(*_data(r, s)) for r in range(world)]
        pulled = [(w.pull(ob.ukeys), v.pull(ob.ukeys) if v is not None else None) for ob in obs]
        if outstanding is not None:
            for ob, (gw, gv) in outstanding:
                w.push(ob.ukeys, gw)
                if v is not None:
                    v.push(ob.ukeys, gv)
            outstanding = None
        grads = []
        for ob, (pw, pv) in zip(obs, pulled):
            if model == "lr":
                grads.append((ob.lr_grad(ob.lr_loss(pw)[0]), None))
            else:
                loss, _, vsum = ob.fm_loss(4, pw, pv)
                grads.append(ob.fm_grad(4, pv, vsum, loss))
        if schedule == "stale1":
            outstanding = list(zip(obs, grads))
            continue
        for ob, (gw, gv) in zip(obs, grads):       # rank order
            w.push(ob.ukeys, gw)
            if v is not None:
                v.push(ob.ukeys, gv)
    if outstanding is not None:
        for ob, (gw, gv) in outstanding:
            w.push(ob.ukeys, gw)
            if v is not None:
                v.push(ob.ukeys, gv)
    losses = []
    for r in range(world):                          # predict: every rank pulls, no pushes
        ob = O.Batch(*_data(r, 99))
        pw = w.pull(ob.ukeys)
        if model == "lr":
            losses.append(ob.lr_loss(pw)[0])
        else:
            losses.append(ob.fm_loss(4, pw, v.pull(ob.ukeys))[0])
    return w, v, losses


# merged = the owner walks all sources' key lists merged by key (one pass over its shard per
# step) instead of one pass per source; both must give the rank-ordered result
@pytest.mark.parametrize("world,model,optimizer,schedule,merged", [
    (2, "lr", "ftrl", "sequential", False), (2, "fm", "sgd", "sequential", False),
    (3, "fm", "ftrl", "sequential", True), (2, "lr", "ftrl", "stale1", True),
    (3, "fm", "sgd", "stale1", False), (3, "lr", "ftrl", "sequential", True)])
def test_sharded_matches_rank_ordered_schedule(tmp_path, world, model, optimizer, schedule,
                                               merged):
    from oracle import pyoracle as O
    port = 29600 + (os.getpid() % 300) + world + (7 if schedule == "stale1" else 0) + \
        (13 if merged else 0)
    mp.spawn(_worker, args=(world, port, model, optimizer, 4, str(tmp_path), schedule, merged),
             nprocs=world, join=True)
    w, v, losses = _simulate(world, model, optimizer, 4, schedule)
    for nm, store in (("w", w), ("v", v)):
        if store is None:
            continue
        ks, ws, ns, zs = store.export()
        parts = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
        # every key lives on exactly the shard the ps-lite range rule names
        for r, p in enumerate(parts):
            assert all(O.lib().xo_shard_of(int(k), world) == r for k in p[nm + "_k"])
        k = np.concatenate([p[nm + "_k"] for p in parts])
        order = np.argsort(k)
        assert np.array_equal(k[order], ks)
        for f, ref in (("_w", ws), ("_n", ns), ("_z", zs)):
            got = np.concatenate([p[nm + f] for p in parts])[order]
            assert np.array_equal(got, ref), (nm, f)
    for r in range(world):
        got = np.load(str(tmp_path / ("rank%d.npz" % r)))["loss"]
        assert np.array_equal(got, losses[r])


def test_split_counts_are_the_ps_lite_ranges():
    from oracle import pyoracle as O
    from xflow_amd.sharded import split_counts
    keys = np.sort(np.array([O.hash_str(str(i)) for i in range(5000)], dtype=np.uint64))
    for world in (1, 2, 3, 8):
        c = split_counts(keys, world)
        owner = np.array([O.lib().xo_shard_of(int(k), world) for k in keys])
        assert c.tolist() == [int((owner == o).sum()) for o in range(world)]
        assert np.all(np.diff(owner) >= 0)   # contiguous ranges of the sorted list
    edge = np.array([0, (2**64 - 1) // 3 - 1, (2**64 - 1) // 3, 2**64 - 2, 2**64 - 1],
                    dtype=np.uint64)
    assert split_counts(edge, 3).tolist() == [2, 1, 2]


# ---- BASELINE configs[0]: LR + FTRL on data/small_train-0000{0,1}, world_size 2, no GPU -------
def _sample_worker(rank, world, port, train_prefix, epochs, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    from tests._cpu_stages import CpuOracleStages
    from xflow_amd.sharded import ShardedTrainer
    st = CpuOracleStages("lr", "ftrl", 0, rank, world)
    tr = ShardedTrainer(model="lr", optimizer="ftrl", rank=rank, world=world, stages=st)
    for _ in range(epochs):                       # worker r reads <prefix>-%05d (lr_worker.cc:210)
        for rowptr, keys, _, labels in O.read_blocks("%s-%05d" % (train_prefix, rank), 2 << 20):
            tr.step(tr.compile(rowptr, keys, labels))
    k, w, n, z = st.w.store.export()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), k=k, w=w, n=n, z=z)
    dist.barrier()
    dist.destroy_process_group()


def test_config0_sample_data_two_workers(tmp_path, sample_prefixes):
    """Two workers, two key-range shards, the reference's sample shards (which are identical
    files): equals one store receiving both workers' pushes in rank order."""
    from oracle import pyoracle as O
    tr_prefix, _ = sample_prefixes
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_sample_worker, args=(2, port, tr_prefix, 3, str(tmp_path)), nprocs=2, join=True)
    s = O.Store(O.OPT_FTRL, 1)
    for _ in range(3):
        blocks = [list(O.read_blocks("%s-%05d" % (tr_prefix, r), 2 << 20)) for r in range(2)]
        for b0, b1 in zip(*blocks):
            obs = [O.Batch(b[0], b[1], b[3]) for b in (b0, b1)]
            pulled = [s.pull(ob.ukeys) for ob in obs]
            grads = [ob.lr_grad(ob.lr_loss(pw)[0]) for ob, pw in zip(obs, pulled)]
            for ob, g in zip(obs, grads):
                s.push(ob.ukeys, g)
    ks, ws, ns, zs = s.export()
    parts = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    k = np.concatenate([p["k"] for p in parts])
    order = np.argsort(k)
    assert np.array_equal(k[order], ks) and len(ks) == 524        # 524 distinct train fids
    for f, ref in (("w", ws), ("n", ns), ("z", zs)):
        assert np.array_equal(np.concatenate([p[f] for p in parts])[order], ref)

"""Harness of the 8-rank, full-size parity runs (tests/test_gpu_world8_fullsize.py): the synthetic
minibatches of BASELINE configs[2] and configs[4], the rank body (one process per rank, all on
the one GPU of the box, the exchange over the group's host transport), the oracle runs of the
same schedules and the comparison of the ranks' table shards with the oracle's store.

Not a test module (no test_ functions); the comparison and the oracle schedules are exercised
without a GPU by tests/test_world8_harness_cpu.py at small sizes."""
import os
import traceback
from concurrent.futures import ThreadPoolExecutor

import numpy as np

SEED = 20260927


def lr_minibatch(rank, step, rows, nnz, nkeys):
    """configs[2]: `rows` examples of `nnz` fids drawn uniformly from "0" .. str(nkeys-1), keys =
    std::hash<std::string> of the decimal string (io.h:53), labels Bernoulli(1/2)"""
    from xflow_amd import capi
    rng = np.random.RandomState(SEED + 1000 * step + rank)
    fid = rng.randint(0, nkeys, size=rows * nnz).astype(np.uint64)
    rowptr = np.arange(rows + 1, dtype=np.uint64) * np.uint64(nnz)
    return rowptr, capi.hash_decimal_ids(fid), rng.randint(0, 2, size=rows).astype(np.int32)


def zipf_minibatch(rank, step, rows, nnz, nkeys, s=1.1):
    """configs[4]: fids from a power law (Zipf s) over a key space of `nkeys` (the tail beyond it
    lands on the last fid, as in bench.py --zipf)"""
    from xflow_amd import capi
    rng = np.random.RandomState(SEED + 7 + 1000 * step + rank)
    fid = (np.minimum(rng.zipf(s, size=rows * nnz), nkeys) - 1).astype(np.uint64)
    rowptr = np.arange(rows + 1, dtype=np.uint64) * np.uint64(nnz)
    return rowptr, capi.hash_decimal_ids(fid), rng.randint(0, 2, size=rows).astype(np.int32)


def save_minibatch(outdir, tag, rank, step, mb):
    np.save(os.path.join(outdir, "%s_r%d_s%d_rowptr.npy" % (tag, rank, step)), mb[0])
    np.save(os.path.join(outdir, "%s_r%d_s%d_keys.npy" % (tag, rank, step)), mb[1])
    np.save(os.path.join(outdir, "%s_r%d_s%d_labels.npy" % (tag, rank, step)), mb[2])


def load_minibatch(outdir, tag, rank, step):
    return tuple(np.load(os.path.join(outdir, "%s_r%d_s%d_%s.npy" % (tag, rank, step, f)))
                 for f in ("rowptr", "keys", "labels"))


def concat(parts):
    """the ranks' minibatches laid end to end, rank 0 first"""
    rowptr = [np.zeros(1, np.uint64)]
    for rp, _, _ in parts:
        rowptr.append(rp[1:] + rowptr[-1][-1])
    return (np.concatenate(rowptr), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]))


# ------------------------------------------------------------------------------ rank body
def rank_main(rank, world, port, cfg, q):
    """cfg: dict(datadir, tag, outdir, model, optimizer, k, schedule, update, capacity, steps,
    compile_ahead, seed).  Steps the minibatches tag_r<rank>_s<0..steps-1> with one defrag after
    the first step, exports its shard of the tables, then scores minibatch s99."""
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from xflow_amd import capi
        g = capi.Group(rank, world, "127.0.0.1", port, capi.TRANSPORT_HOST, device=0)
        st = capi.Sharded(g, model=cfg["model"], optimizer=cfg["optimizer"], k=cfg["k"],
                          capacity=cfg["capacity"], schedule=cfg["schedule"], seed=cfg["seed"],
                          update=cfg["update"])
        steps = cfg["steps"]
        data = [load_minibatch(cfg["datadir"], cfg["tag"], rank, s) for s in range(steps)]
        alive = []
        if cfg["compile_ahead"]:      # replayed minibatches: compiled before the row renumbering
            alive = [st.compile(*d) for d in data]
        for s in range(steps):
            if not cfg["compile_ahead"]:
                alive.append(st.compile(*data[s]))
            st.step(alive[s])
            if s == 0:
                st.defrag()
        st.check()
        out = {}
        for nm, t in (("w", st.w), ("v", st.v)):
            if t is not None:
                k, w, n, z = t.export()
                o = np.argsort(k)
                out.update({nm + "_k": k[o], nm + "_w": w[o], nm + "_n": n[o], nm + "_z": z[o]})
        rp, ks, lb = load_minibatch(cfg["datadir"], cfg["tag"], rank, 99)
        out["loss"] = st.predict(st.compile(rp, ks, lb)) - lb.astype(np.float32)
        st.check()
        for name, a in out.items():
            np.save(os.path.join(cfg["outdir"], "rank%d_%s.npy" % (rank, name)), a)
        g.barrier()
        del alive
        st.close()
        g.close()
        q.put((rank, None))
    except Exception:
        q.put((rank, traceback.format_exc()))


def start_ranks(world, cfg):
    import multiprocessing as mp
    from .test_group_cpu import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=rank_main, args=(r, world, port, cfg, q)) for r in range(world)]
    for p in ps:
        p.start()
    return ps, q


def join_ranks(ps, q, timeout=900):
    res = [q.get(timeout=timeout) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    errs = [e for _, e in res if e]
    assert not errs, errs[0]


# --------------------------------------------------------------------------- oracle schedules
def _threads():
    try:
        return max(1, min(8, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 4


def oracle_rank_ordered_lr(O, data, optimizer, reserve=0):
    """what ps-lite's server sees when every worker pulls before any pushes and the pushes land
    in rank order: per step all Pulls, every worker's loss + gradient from ITS pull
    (lr_worker.cc:170-173), then the Pushes one after the other (ftrl.h:54-74).  data[s][r]."""
    w = O.Store(O.OPT_FTRL if optimizer == "ftrl" else O.OPT_SGD, 1)
    if reserve:
        w.reserve(reserve)
    for mbs in data:
        with ThreadPoolExecutor(_threads()) as ex:     # (ctypes calls release the GIL)
            obs = list(ex.map(lambda d: O.Batch(*d), mbs))
        pulled = [w.pull(ob.ukeys) for ob in obs]
        with ThreadPoolExecutor(_threads()) as ex:
            grads = list(ex.map(lambda a: a[0].lr_grad(a[0].lr_loss(a[1])[0]), zip(obs, pulled)))
        for ob, g in zip(obs, grads):
            w.push(ob.ukeys, g)
        del obs, pulled, grads
    return w


def oracle_rank_ordered_fm(O, data, optimizer, k, seed):
    """the same rule for FM (fm_worker.cc:226-242): per step every worker pulls w and v, forms
    its loss and both gradients from ITS pull, then the two Pushes of every worker land worker
    after worker (ftrl.h:54-74, :98-149).  data[s][r]."""
    ftrl = optimizer == "ftrl"
    oo = O.OPT_FTRL if ftrl else O.OPT_SGD
    sw = O.Store(oo, 1)
    sv = O.Store(oo, k, O.INIT_HASHNORM if ftrl else O.INIT_CONST, 0.001, seed)
    for mbs in data:
        obs = [O.Batch(*d) for d in mbs]
        pulled = [(sw.pull(ob.ukeys), sv.pull(ob.ukeys)) for ob in obs]

        def grad(a):
            ob, (pw, pv) = a
            loss, _, vsum = ob.fm_loss(k, pw, pv)
            return ob.fm_grad(k, pv, vsum, loss)
        with ThreadPoolExecutor(_threads()) as ex:
            grads = list(ex.map(grad, zip(obs, pulled)))
        for ob, (gw, gv) in zip(obs, grads):
            sw.push(ob.ukeys, gw)
            sv.push(ob.ukeys, gv)
        del obs, pulled, grads
    return sw, sv


def oracle_concat_lr(O, data, optimizer, reserve=0):
    """sum_then_step: one LRWorker::update (lr_worker.cc:145-177) per step on the ranks'
    minibatches laid end to end"""
    w = O.Store(O.OPT_FTRL if optimizer == "ftrl" else O.OPT_SGD, 1)
    if reserve:
        w.reserve(reserve)
    for mbs in data:
        ob = O.Batch(*concat(mbs))
        O.lr_update(w, ob)
        del ob
    return w


def oracle_concat_fm(O, data, optimizer, k, seed, reserve=0):
    """one FMWorker::update (fm_worker.cc:204-245) per step on the concatenation"""
    ftrl = optimizer == "ftrl"
    oo = O.OPT_FTRL if ftrl else O.OPT_SGD
    sw = O.Store(oo, 1)
    sv = O.Store(oo, k, O.INIT_HASHNORM if ftrl else O.INIT_CONST, 0.001, seed)
    if reserve:
        sw.reserve(reserve)
        sv.reserve(reserve)
    for mbs in data:
        ob = O.Batch(*concat(mbs))
        O.fm_update(sw, sv, ob)
        del ob
    return sw, sv


# -------------------------------------------------------------------------------- comparison
def same(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert a.dtype == b.dtype, (what, a.dtype, b.dtype)
    ok = a == b   # (as tests/test_gpu_parity.same: every value identical; NaN never appears)
    if not np.all(ok):
        bad = np.flatnonzero(~ok.ravel())
        raise AssertionError("%s: %d of %d differ, first at %d: %r vs %r" % (
            what, len(bad), ok.size, bad[0], a.ravel()[bad[0]], b.ravel()[bad[0]]))


def shard_slices(O, keys_sorted, world):
    """[lo, hi) of every rank's keys in the oracle's key-sorted export: rank r owns the keys with
    min(key / (UINT64_MAX / world), world - 1) == r (ps-lite's default slicer)"""
    span = np.uint64(0xFFFFFFFFFFFFFFFF // world)
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(keys_sorted, np.uint64(r) * span, side="left")))
    cuts.append(len(keys_sorted))
    for r in range(world):      # the rule itself, on the boundary keys, through the oracle
        for i in {cuts[r], cuts[r + 1] - 1} if cuts[r + 1] > cuts[r] else ():
            assert O.lib().xo_shard_of(int(keys_sorted[i]), world) == r
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def compare_tables(O, outdir, world, name, export):
    """every rank's exported shard of table `name` ("w" / "v") against the oracle's store: the
    same keys (so: the same owner for every key) and the same (w, n, z) bits"""
    ks, ws, ns, zs = export
    total = 0
    for r, (lo, hi) in enumerate(shard_slices(O, ks, world)):
        ld = lambda f: np.load(os.path.join(outdir, "rank%d_%s_%s.npy" % (r, name, f)))
        k = ld("k")
        same(k, ks[lo:hi], "rank %d %s keys" % (r, name))
        for f, ref in (("w", ws), ("n", ns), ("z", zs)):
            same(ld(f).reshape(ref[lo:hi].shape), ref[lo:hi], "rank %d %s.%s" % (r, name, f))
        total += len(k)
    assert total == len(ks)
    return total

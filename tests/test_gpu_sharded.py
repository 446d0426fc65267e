"""The C++ sharded trainer (xf_sharded_*, xf_group_*) on real hardware.

* several ranks of the real kernels as separate processes on ONE GPU, the exchange staged
  through the group's host transport (RCCL refuses two ranks per device): sharded tables,
  owner-side resolve of every source's keys with real duplicates, merged rank-ordered owner
  updates, a defrag between steps, both schedules — bit-exact against the oracle run on the
  same schedule (tests/test_sharded_gloo._simulate);
* the same over RCCL with one rank per GPU, world 2 / 4 / 8, skipped when the box has fewer
  GPUs (the driver's 8-GPU node runs them);
* the sharded checkpoint: saved by 2 ranks, loaded by 3 and by 1."""
import multiprocessing as mp
import ctypes as C
import os
import traceback

import numpy as np
import pytest

from oracle import pyoracle as O
from xflow_amd import capi

from .test_gpu_parity import same
from .test_group_cpu import free_port
from .test_sharded_gloo import _data, _simulate

pytestmark = pytest.mark.gpu


def _big_data(rank, step):
    """minibatches big enough for several row windows per worker (17 408 rows fit one), rows per
    worker that differ, and a key space small enough that chunks of 2048 state rows collect more
    than 8192 nonzeros (they are cut into slices that meet in per-worker accumulators)"""
    rng = np.random.RandomState(77 + 1000 * step + rank)
    R = 20000 + 3000 * rank
    keytab = capi.hash_decimal_range(0, 50000)
    lens = rng.randint(0, 13, size=R)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    keys = keytab[np.minimum(rng.zipf(1.3, size=int(lens.sum())) - 1, 49999)
                  if step % 2 else rng.randint(0, 50000, size=int(lens.sum()))]
    return rowptr, keys, rng.randint(0, 2, size=R).astype(np.int32)


def _rank(rank, world, port, transport, model, optimizer, schedule, steps, outdir, save, q,
          data="small", update="rank_ordered", k=4):
    _data = _big_data if data == "big" else globals()["_data"]
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = C.c_int(0)
        capi.check(capi.lib().xf_device_count(C.byref(ndev)))
        g = capi.Group(rank, world, "127.0.0.1", port, transport,
                       device=0 if transport == capi.TRANSPORT_HOST else rank % ndev.value)
        if transport == capi.TRANSPORT_AUTO:   # RCCL with a GPU per rank, else the fallback
            assert g.transport == (capi.TRANSPORT_RCCL if ndev.value >= world
                                   else capi.TRANSPORT_HOST)
        st = capi.Sharded(g, model=model, optimizer=optimizer, k=k, capacity=64,
                          schedule=schedule, seed=7, update=update)
        alive = []   # freeing a minibatch whose Push is still outstanding would flush it early
        for s in range(steps):
            b = st.compile(*_data(rank, s))
            alive.append(b)
            if not schedule.startswith("owner"):   # (the owner-compute dataflow keeps no key list)
                assert b.U == len(np.unique(_data(rank, s)[1]))
            st.step(b)
            if schedule in ("sequential", "owner") and s == 1:
                st.defrag()          # row renumbering between steps must not change a bit
        st.check()
        rp, ks, lb = _data(rank, 99)
        out = {"loss": st.predict(st.compile(rp, ks, lb)) - lb.astype(np.float32)}
        st.check()
        for nm, t in (("w", st.w), ("v", st.v)):
            if t is not None:
                k, w, n, z = t.export()
                out.update({nm + "_k": k, nm + "_w": w, nm + "_n": n, nm + "_z": z})
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
        if save:
            st.save(os.path.join(outdir, "ckpt"))
        g.barrier()
        st.close()
        g.close()
        q.put((rank, None))
    except Exception:
        q.put((rank, traceback.format_exc()))


def _run(world, transport, model, optimizer, schedule, outdir, save=False, steps=4,
         data="small", update="rank_ordered", k=4):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=_rank, args=(r, world, port, transport, model, optimizer, schedule,
                                          steps, str(outdir), save, q, data, update, k))
          for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    errs = [e for _, e in res if e]
    assert not errs, errs[0]


def _fm_replay_rank(port, outdir, q):
    try:
        os.environ["XF_SHARDED_GENERAL"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        g = capi.Group(0, 1, "127.0.0.1", port, capi.TRANSPORT_HOST, device=0)
        for name, fresh in (("replayed", False), ("fresh", True)):
            st = capi.Sharded(g, model="fm", optimizer="ftrl", k=4, capacity=1 << 12,
                              schedule="owner_stale1", seed=7)
            kept = [st.compile(*_data(0, s)) for s in range(2)]
            alive = []
            for s in range(6):      # 0 0 1 1 0 0: every step but the first meets its own Pushes
                b = st.compile(*_data(0, (s // 2) % 2)) if fresh else kept[(s // 2) % 2]
                alive.append(b)
                st.step(b)
            st.check()
            out = {}
            for nm, t in (("w", st.w), ("v", st.v)):
                k, w, n, z = t.export()
                out.update({nm + "_k": k, nm + "_w": w, nm + "_n": n, nm + "_z": z})
            np.savez(os.path.join(outdir, name + ".npz"), **out)
            del alive, kept
            st.close()
        g.close()
        q.put(None)
    except Exception:
        q.put(traceback.format_exc())


def _check_against_oracle(world, model, optimizer, schedule, outdir, steps=4):
    with O.sum_mode(1):
        w, v, losses = _simulate(world, model, optimizer, steps, schedule)
    parts = [np.load(os.path.join(str(outdir), "rank%d.npz" % r)) for r in range(world)]
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
    return w, v


@pytest.mark.parametrize("world,model,optimizer,schedule", [
    (2, "lr", "ftrl", "sequential"), (2, "lr", "ftrl", "stale1"), (3, "lr", "sgd", "stale1"),
    (3, "fm", "ftrl", "sequential"), (2, "fm", "sgd", "stale1")])
def test_cpp_sharded_ranks_share_one_gpu(tmp_path, world, model, optimizer, schedule):
    _run(world, capi.TRANSPORT_HOST, model, optimizer, schedule, tmp_path)
    _check_against_oracle(world, model, optimizer, schedule, tmp_path)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("model,schedule", [("lr", "sequential"), ("lr", "stale1"),
                                            ("fm", "stale1")])
def test_cpp_sharded_over_rccl(tmp_path, world, model, schedule):
    """one rank per GPU, the exchange over RCCL / xGMI: bit-exact with the same oracle run"""
    import ctypes as C
    n = C.c_int(0)
    capi.check(capi.lib().xf_device_count(C.byref(n)))
    if n.value < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, n.value))
    _run(world, capi.TRANSPORT_RCCL, model, "ftrl", schedule, tmp_path)
    _check_against_oracle(world, model, "ftrl", schedule, tmp_path)


@pytest.mark.parametrize("world,optimizer", [(2, "ftrl"), (3, "sgd"), (3, "ftrl")])
def test_owner_compute_dataflow_ranks_share_one_gpu(tmp_path, world, optimizer):
    """XF_SCHEDULE_OWNER: the nonzeros go to the key owners when the minibatch is compiled, a
    step exchanges fp64 partial row sums and losses instead of weights and gradients, the
    owner applies the workers' gradients in rank order out of one pass over its shard — the
    same numbers as the sequential schedule of the weight/gradient exchange, bit for bit
    (tables after 4 steps with a defrag in between, and the forward of a fifth minibatch)"""
    _run(world, capi.TRANSPORT_HOST, "lr", optimizer, "owner", tmp_path)
    _check_against_oracle(world, "lr", optimizer, "sequential", tmp_path)


@pytest.mark.parametrize("world,optimizer,data", [(2, "ftrl", "small"), (3, "sgd", "small"),
                                                  (2, "ftrl", "big")])
def test_owner_compute_dataflow_overlapped(tmp_path, world, optimizer, data):
    """XF_SCHEDULE_OWNER_STALE1: the gradient + Pushes of step t on a second HIP stream under the
    row-sum / loss exchanges of step t+1; forward(t+1) reads the table before they land, events
    order the reader and the writer — weights exactly one step stale: the numbers of the stale1
    schedule of the weight / gradient exchange and of the oracle run of that rule (every worker
    pulls before the previous step's pushes are applied in rank order), bit for bit."""
    steps = 4
    _run(world, capi.TRANSPORT_HOST, "lr", optimizer, "owner_stale1", tmp_path, steps=steps,
         data=data)
    if data == "small":
        _check_against_oracle(world, "lr", optimizer, "stale1", tmp_path, steps=steps)
        return
    with O.sum_mode(1):
        w = O.Store(O.OPT_FTRL, 1)
        outstanding = []
        for s in range(steps):
            obs = [O.Batch(*_big_data(r, s)) for r in range(world)]
            pulled = [w.pull(ob.ukeys) for ob in obs]
            for ob, g in outstanding:            # step s-1's pushes land after step s's pulls
                w.push(ob.ukeys, g)
            outstanding = [(ob, ob.lr_grad(ob.lr_loss(pw)[0])) for ob, pw in zip(obs, pulled)]
        for ob, g in outstanding:
            w.push(ob.ukeys, g)
        parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
        ks, ws, ns, zs = w.export()
        k = np.concatenate([p["w_k"] for p in parts])
        order = np.argsort(k)
        same(k[order], ks)
        for f, ref in (("w_w", ws), ("w_n", ns), ("w_z", zs)):
            same(np.concatenate([p[f] for p in parts])[order].reshape(ref.shape), ref)


@pytest.mark.parametrize("schedule", ["owner", "sequential"])
def test_several_row_windows_per_worker_and_split_chunks(tmp_path, schedule):
    """two workers with 20 000 and 23 000 rows: more than one row window each (the owner numbers
    them window by window, a window belongs to one worker), chunks with more than 8192 nonzeros
    (slices, per-worker accumulators in HBM, the workers' steps in rank order in the finish
    kernel), uniform and power-law minibatches — against the oracle on the rank-ordered
    schedule, both dataflows"""
    world, steps = 2, 3
    _run(world, capi.TRANSPORT_HOST, "lr", "ftrl", schedule, tmp_path, steps=steps, data="big")
    with O.sum_mode(1):
        w = O.Store(O.OPT_FTRL, 1)
        for s in range(steps):
            obs = [O.Batch(*_big_data(r, s)) for r in range(world)]
            pulled = [w.pull(ob.ukeys) for ob in obs]
            grads = [ob.lr_grad(ob.lr_loss(pw)[0]) for ob, pw in zip(obs, pulled)]
            for ob, g in zip(obs, grads):
                w.push(ob.ukeys, g)
        parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
        ks, ws, ns, zs = w.export()
        k = np.concatenate([p["w_k"] for p in parts])
        order = np.argsort(k)
        same(k[order], ks)
        for f, ref in (("w_w", ws), ("w_n", ns), ("w_z", zs)):
            same(np.concatenate([p[f] for p in parts])[order].reshape(ref.shape), ref)
        for r in range(world):
            ob = O.Batch(*_big_data(r, 99))
            same(parts[r]["loss"], ob.lr_loss(w.pull(ob.ukeys))[0])


@pytest.mark.parametrize("world,optimizer,data", [(2, "ftrl", "small"), (3, "sgd", "small"),
                                                  (2, "ftrl", "big")])
def test_sum_then_step_is_one_update_on_the_concatenated_minibatch(tmp_path, world, optimizer,
                                                                   data):
    """XF_UPDATE_SUM_THEN_STEP on the owner-compute dataflow: the workers' per-key
    sums meet at the owner in fp64 and become ONE optimizer step with 1 / (all rows) — the table
    after every step is what a single LRWorker::update on the ranks' minibatches laid end to end
    gives, bit for bit, whatever the number of GPUs (and what one GPU gives on that
    concatenation)."""
    steps = 3
    gen = _big_data if data == "big" else _data
    _run(world, capi.TRANSPORT_HOST, "lr", optimizer, "owner", tmp_path, steps=steps, data=data,
         update="sum_then_step")

    def concat(step):
        parts = [gen(r, step) for r in range(world)]
        rowptr = [np.zeros(1, np.uint64)]
        for rp, _, _ in parts:
            rowptr.append(rp[1:] + rowptr[-1][-1])
        return (np.concatenate(rowptr), np.concatenate([p[1] for p in parts]),
                np.concatenate([p[2] for p in parts]))
    oo, go = (O.OPT_FTRL, capi.OPT_FTRL) if optimizer == "ftrl" else (O.OPT_SGD, capi.OPT_SGD)
    w = O.Store(oo, 1)
    t = capi.Table(go, 1, capacity=1 << 18)
    ws = capi.Workspace()
    with O.sum_mode(1):
        for s in range(steps):
            raw = concat(s)
            O.lr_update(w, O.Batch(*raw))
            capi.lr_step(t, capi.LocalBatch(t, *raw, retain_keys=False), ws)
            if s == 1:
                t.defrag()
        parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
        ks, wv, ns, zs = w.export()
        k = np.concatenate([p["w_k"] for p in parts])
        order = np.argsort(k)
        same(k[order], ks)
        for f, ref in (("w_w", wv), ("w_n", ns), ("w_z", zs)):
            same(np.concatenate([p[f] for p in parts])[order].reshape(ref.shape), ref)
        for a, e in zip(t.export(), w.export()):       # ... and the single GPU agrees
            same(a, e)
        for r in range(world):
            ob = O.Batch(*gen(r, 99))
            same(parts[r]["loss"], ob.lr_loss(w.pull(ob.ukeys))[0])


@pytest.mark.parametrize("world,optimizer,k,data", [(2, "ftrl", 4, "small"), (3, "sgd", 10, "small"),
                                                    (2, "ftrl", 16, "big")])
def test_fm_on_the_owner_compute_dataflow(tmp_path, world, optimizer, k, data):
    """FM with XF_UPDATE_SUM_THEN_STEP on the owner-compute dataflow: the owners' shares of the
    three row sums (wx, v_sum, v_pow_sum) meet at the rows' worker in fp64, (loss, v_sum) go
    back, every key gets one gradient pass over all workers' rows and one optimizer step on w
    and on v — what one FMWorker::update on the ranks' minibatches laid end to end gives, bit for
    bit (oracle, exact sums), and what one GPU gives on that concatenation."""
    steps = 3
    gen = _big_data if data == "big" else _data
    _run(world, capi.TRANSPORT_HOST, "fm", optimizer, "owner", tmp_path, steps=steps, data=data,
         update="sum_then_step", k=k)

    def concat(step):
        parts = [gen(r, step) for r in range(world)]
        rowptr = [np.zeros(1, np.uint64)]
        for rp, _, _ in parts:
            rowptr.append(rp[1:] + rowptr[-1][-1])
        return (np.concatenate(rowptr), np.concatenate([p[1] for p in parts]),
                np.concatenate([p[2] for p in parts]))
    ftrl = optimizer == "ftrl"
    oo, go = (O.OPT_FTRL, capi.OPT_FTRL) if ftrl else (O.OPT_SGD, capi.OPT_SGD)
    oi, gi = (O.INIT_HASHNORM, capi.INIT_HASHNORM) if ftrl else (O.INIT_CONST, capi.INIT_CONST)
    sw, sv = O.Store(oo, 1), O.Store(oo, k, oi, 0.001, 7)
    tw = capi.Table(go, 1, capacity=1 << 18)
    tv = capi.Table(go, k, gi, 0.001, seed=7, capacity=1 << 18)
    ws = capi.Workspace()
    with O.sum_mode(1):
        for s in range(steps):
            raw = concat(s)
            O.fm_update(sw, sv, O.Batch(*raw))
            capi.fm_step(tw, tv, capi.Batch(*raw), ws)
            if s == 1:
                tw.defrag()
                tv.defrag()
        parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
        for nm, store, t in (("w", sw, tw), ("v", sv, tv)):
            ks, wv, ns, zs = store.export()
            kk = np.concatenate([p[nm + "_k"] for p in parts])
            order = np.argsort(kk)
            same(kk[order], ks)
            for f, ref in (("_w", wv), ("_n", ns), ("_z", zs)):
                same(np.concatenate([p[nm + f] for p in parts])[order].reshape(ref.shape), ref)
            for a, e in zip(t.export(), store.export()):   # ... and the single GPU agrees
                same(a, e)
        for r in range(world):
            ob = O.Batch(*gen(r, 99))
            same(parts[r]["loss"], ob.fm_loss(k, sw.pull(ob.ukeys), sv.pull(ob.ukeys))[0])


@pytest.mark.parametrize("world,optimizer", [(2, "ftrl"), (3, "sgd"), (3, "ftrl")])
def test_fm_on_the_owner_compute_dataflow_rank_ordered(tmp_path, world, optimizer):
    """FM with the reference's update rule (XF_UPDATE_RANK_ORDERED: every worker's Push its own
    optimizer step, fm_worker.cc:226-242 + server.h:24-29) on the owner-compute dataflow: the
    owner keeps one minibatch per worker, every worker's gradient comes from what its Pull
    returned (all Pulls before all Pushes), the Pushes land worker after worker — the tables and
    a held-out forward of the sequential schedule of the weight / gradient exchange and of the
    oracle run of that rule, bit for bit (4 steps, a defrag in between)."""
    _run(world, capi.TRANSPORT_HOST, "fm", optimizer, "owner", tmp_path)
    _check_against_oracle(world, "fm", optimizer, "sequential", tmp_path)


@pytest.mark.parametrize("world,optimizer", [(2, "ftrl"), (3, "sgd"), (3, "ftrl")])
def test_fm_on_the_overlapped_owner_schedule(tmp_path, world, optimizer):
    """XF_SCHEDULE_OWNER_STALE1 for FM (fm_worker.cc:226-242 with every worker's two Pushes one
    step late): the Pushes of step t on the second HIP stream under the row-sum / (loss, v_sum)
    exchanges of step t+1, every worker's gradient from the rows ITS Pull returned — the numbers
    of the stale1 schedule of the weight / gradient exchange and of the oracle run of that rule,
    both tables and a held-out forward, bit for bit."""
    _run(world, capi.TRANSPORT_HOST, "fm", optimizer, "owner_stale1", tmp_path)
    _check_against_oracle(world, "fm", optimizer, "stale1", tmp_path)


def test_fm_overlapped_owner_schedule_replays_a_minibatch(tmp_path):
    """the same compiled minibatch stepped again while its Pushes are outstanding (two workspace
    sets that swap roles), against one rank per step of fresh compiles"""
    import multiprocessing as mp2
    ctx = mp2.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_fm_replay_rank, args=(free_port(), str(tmp_path), q))
    p.start()
    err = q.get(timeout=240)
    p.join(timeout=60)
    assert not err, err
    a, b = np.load(str(tmp_path / "replayed.npz")), np.load(str(tmp_path / "fresh.npz"))
    for f in a.files:
        same(a[f], b[f])


def test_fm_overlapped_owner_schedule_needs_the_rank_ordered_rule():
    g = capi.Group(0, 1, "127.0.0.1", free_port(), capi.TRANSPORT_HOST, device=0)
    with pytest.raises(capi.XFError, match="needs update_rule rank_ordered"):
        capi.Sharded(g, model="fm", optimizer="ftrl", k=4, schedule="owner_stale1",
                     update="sum_then_step")
    g.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_owner_compute_dataflow_over_rccl(tmp_path, world):
    n = C.c_int(0)
    capi.check(capi.lib().xf_device_count(C.byref(n)))
    if n.value < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, n.value))
    _run(world, capi.TRANSPORT_RCCL, "lr", "ftrl", "owner", tmp_path)
    _check_against_oracle(world, "lr", "ftrl", "sequential", tmp_path)


def test_auto_transport_agrees_on_every_rank(tmp_path):
    """XF_TRANSPORT_AUTO at world 2: RCCL when the box has two GPUs; with one, RCCL refuses the
    communicator (two ranks on one device), the ranks agree on that over the bootstrap and the
    run goes through the host transport — bit-exact either way, and nobody hangs"""
    _run(2, capi.TRANSPORT_AUTO, "lr", "ftrl", "stale1", tmp_path)
    _check_against_oracle(2, "lr", "ftrl", "stale1", tmp_path)


def test_rccl_group_of_one_and_the_fused_step():
    """RCCL itself comes up (dlopen, unique id, communicator) with the one GPU there is, the
    exchange of a one-rank group is a plain copy, and the trainer over it IS the fused step."""
    import torch
    g = capi.Group(0, 1, transport=capi.TRANSPORT_RCCL, device=0)
    src = torch.arange(1000, dtype=torch.float32, device="cuda")
    dst = torch.zeros_like(src)
    g.alltoallv_dev(src.data_ptr(), [1000], dst.data_ptr(), [1000], 4,
                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    # the group's two communicators (the side stream of the stale1 schedule has its own), both
    # driven at the same time from two streams: ncclSend / ncclRecv to this very rank
    g.selftest(1 << 20)
    side = torch.cuda.Stream()
    dst2 = torch.zeros_like(src)
    g.alltoallv_dev(src.data_ptr(), [1000], dst2.data_ptr(), [1000], 4, side.cuda_stream,
                    channel=1)
    side.synchronize()
    assert torch.equal(src, dst2)
    st = capi.Sharded(g, model="lr", optimizer="ftrl", capacity=1 << 12)
    t = capi.Table(capi.OPT_FTRL, 1, capacity=1 << 12)
    ws = capi.Workspace()
    for s in range(3):
        raw = _data(0, s)
        st.step(st.compile(*raw))
        capi.lr_step(t, capi.Batch(*raw), ws)
    st.check()
    for a, b in zip(st.w.export(), t.export()):
        same(a, b)
    raw = _data(0, 99)
    same(st.predict(st.compile(*raw)), capi.lr_predict(t, capi.Batch(*raw), ws))


def _load_rank(rank, world, port, outdir, q):
    try:
        g = capi.Group(rank, world, "127.0.0.1", port, capi.TRANSPORT_HOST, device=0)
        st = capi.Sharded(g, model="fm", optimizer="ftrl", k=4, capacity=64, seed=7)
        st.load(os.path.join(outdir, "ckpt"))
        out = {}
        for nm, t in (("w", st.w), ("v", st.v)):
            k, w, n, z = t.export()
            out.update({nm + "_k": k, nm + "_w": w, nm + "_n": n, nm + "_z": z})
        np.savez(os.path.join(outdir, "load%d_of_%d.npz" % (rank, world)), **out)
        g.barrier()
        q.put((rank, None))
    except Exception:
        q.put((rank, traceback.format_exc()))


def test_sharded_checkpoint_reshards_on_load(tmp_path):
    """saved by 2 ranks (one file per shard + manifest), loaded by 3 ranks and by one"""
    _run(2, capi.TRANSPORT_HOST, "fm", "ftrl", "sequential", tmp_path, save=True)
    w, v = _check_against_oracle(2, "fm", "ftrl", "sequential", tmp_path)
    assert sorted(os.listdir(str(tmp_path)))[:3] == ["ckpt.manifest", "ckpt.shard-00000-of-00002",
                                                     "ckpt.shard-00001-of-00002"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=_load_rank, args=(r, 3, port, str(tmp_path), q)) for r in range(3)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert not [e for _, e in res if e], [e for _, e in res if e][0]
    parts = [np.load(str(tmp_path / ("load%d_of_3.npz" % r))) for r in range(3)]
    one = capi.Sharded(None, model="fm", optimizer="ftrl", k=4, capacity=64, seed=7)
    one.load(str(tmp_path / "ckpt"))
    for nm, store, t in (("w", w, one.w), ("v", v, one.v)):
        ks, ws, ns, zs = store.export()
        for r, p in enumerate(parts):
            assert all(O.lib().xo_shard_of(int(k), 3) == r for k in p[nm + "_k"])
        k = np.concatenate([p[nm + "_k"] for p in parts])
        order = np.argsort(k)
        same(k[order], ks)
        for f, ref in (("_w", ws), ("_n", ns), ("_z", zs)):
            same(np.concatenate([p[nm + f] for p in parts])[order].reshape(ref.shape), ref)
        for got, ref in zip(t.export(), (ks, ws, ns, zs)):
            same(got, ref)


def test_training_after_a_model_load_grows_the_table(tmp_path):
    """A loaded model counts towards the table's load: train on unseen keys right after a load
    that left the table about half full (the host-side bound on the key count has to start from
    the loaded size, or the exact check comes too late and the insert fails with XF_EFULL)."""
    rng = np.random.default_rng(5)

    def batch(keys, rows=64, per_row=40):
        ks = rng.choice(keys, size=rows * per_row).astype(np.uint64)
        rp = np.arange(rows + 1, dtype=np.uint64) * per_row
        return rp, ks, rng.integers(0, 2, rows).astype(np.int32)

    pool = rng.choice(1 << 62, size=6000, replace=False).astype(np.uint64)
    first, later = pool[:2000], pool[2000:]
    a = capi.Sharded(None, model="fm", optimizer="ftrl", k=4, capacity=4096, seed=7)
    ba = a.compile(*batch(first, rows=200))
    a.step(ba)
    a.check()
    n_first = len(a.w.export()[0])
    assert 1800 <= n_first <= 2001                       # about half of 4096 slots
    a.save(str(tmp_path / "m"))
    b = capi.Sharded(None, model="fm", optimizer="ftrl", k=4, capacity=4096, seed=7)
    b.load(str(tmp_path / "m"))
    seen = set()
    alive = []
    for i in range(3):                                   # ~1300 unseen keys per minibatch
        rp, ks, lb = batch(later[i * 1300:(i + 1) * 1300])
        seen.update(ks.tolist())
        mb = b.compile(rp, ks, lb)
        alive.append(mb)
        b.step(mb)
        b.check()                                        # raises on XF_EFULL
    keys = set(b.w.export()[0].tolist())
    assert seen <= keys and len(keys) >= n_first + len(seen)


def _general_path_rank(port, schedule, outdir, q):
    try:
        os.environ["XF_SHARDED_GENERAL"] = "1"      # a world-1 trainer runs the exchange path
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        g = capi.Group(0, 1, "127.0.0.1", port, capi.TRANSPORT_RCCL, device=0)
        st = capi.Sharded(g, model="lr", optimizer="ftrl", capacity=1 << 12, schedule=schedule)
        alive = []
        for s in range(3):
            alive.append(st.compile(*_big_data(0, s), keep=(s != 1)))   # one of them one-shot
            st.step(alive[-1])
        st.check()
        k, w, n, z = st.w.export()
        rp, ks, lb = _big_data(0, 99)
        np.savez(os.path.join(outdir, "gp_%s.npz" % schedule), k=k, w=w, n=n, z=z,
                 p=st.predict(st.compile(rp, ks, lb)))
        st.close()
        g.close()
        q.put(None)
    except Exception:
        q.put(traceback.format_exc())


@pytest.mark.parametrize("schedule", ["owner", "sequential"])   # (stale1 is one step stale)
def test_exchange_code_paths_at_world_one_over_rccl_equal_the_fused_step(tmp_path, schedule):
    """XF_SHARDED_GENERAL=1: the N > 1 code paths with the one GPU there is, the all-to-all-v
    through the RCCL transport (a group of one: the self slice) — 20 000-row minibatches (two
    row windows, split chunks), the table afterwards and a forward pass equal to the fused
    single-shard trainer's, bit for bit"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_general_path_rank, args=(free_port(), schedule, str(tmp_path), q))
    p.start()
    err = q.get(timeout=180)
    p.join(timeout=60)
    assert not err, err
    got = np.load(str(tmp_path / ("gp_%s.npz" % schedule)))
    one = capi.Sharded(None, model="lr", optimizer="ftrl", capacity=1 << 12)
    alive = []
    for s in range(3):
        alive.append(one.compile(*_big_data(0, s)))
        one.step(alive[-1])
    one.check()
    for a, e in zip((got["k"], got["w"], got["n"], got["z"]), one.w.export()):
        same(a, e)
    rp, ks, lb = _big_data(0, 99)
    same(got["p"], one.predict(one.compile(rp, ks, lb)))


def _pretend_data(step):
    """rows enough for five row windows, chunks of every kind at once: most hold fewer than 2048
    entries (k_lr_grad_multi), the power-law head chunks more (the general loop) or more than
    8192 (slices + the finish kernel)"""
    rng = np.random.RandomState(500 + step)
    R, K = 72000, 150000
    keytab = capi.hash_decimal_range(0, K)
    lens = rng.randint(0, 4, size=R)
    n = int(lens.sum())
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    fid = rng.randint(0, K, size=n)
    hot = rng.rand(n) < 0.25
    fid[hot] = np.minimum(rng.zipf(1.2, size=int(hot.sum())) - 1, K - 1)
    return rowptr, keytab[fid], rng.randint(0, 2, size=R).astype(np.int32)


def _pretend_rank(port, path, outdir, q):
    try:
        os.environ["XF_SHARDED_GENERAL"] = "1"
        os.environ["XF_OWNER_TIMING_SOURCES"] = "5"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        g = capi.Group(0, 1, "127.0.0.1", port, capi.TRANSPORT_HOST, device=0)
        st = capi.Sharded(g, model="lr", optimizer="ftrl", capacity=1 << 19, schedule="owner")
        capi.tune(*path)
        alive = []
        for s in range(4):
            alive.append(st.compile(*_pretend_data(s % 3)))
            st.step(alive[-1])
            if s == 1:
                st.defrag()      # holes and an arrival segment afterwards
        st.check()
        k, w, n, z = st.w.export()
        np.savez(os.path.join(outdir, "pretend_%s_%d.npz" % path), k=k, w=w, n=n, z=z)
        st.close()
        g.close()
        q.put(None)
    except Exception:
        q.put(traceback.format_exc())


def test_the_several_workers_pass_takes_the_steps_of_the_general_loop(tmp_path):
    """An owner's gradient + Pushes for several workers (XF_UPDATE_RANK_ORDERED): k_lr_grad_multi
    and k_lr_grad_ranked (the chunk's state in LDS; the workers' phases merged — owner_pass = 2,
    and 3 with 32-bit masks — or a phase per worker with the stepping lane picked among the lanes
    that hold the key: 4; 0: the shape picks) against the general loop (owner_pass = 1: state rows
    in registers, a sweep per worker) on one GPU whose rows are dealt out to five pretended
    workers — five optimizer steps per key, chunks below and above one round of registers, split
    chunks, holes after a defrag: the same table, bit for bit.  (That the steps are the
    reference's: the world-2 / 3 / 8 tests against the oracle, which run the same kernel.)
    key_build = 2: the owner's cells through the two-level key build (nonzeros with row
    numbers)."""
    ctx = mp.get_context("spawn")
    paths = [("owner_pass", v) for v in (1, 0, 4, 2, 3)] + [("key_build", 2)]
    for path in paths:
        q = ctx.Queue()
        p = ctx.Process(target=_pretend_rank, args=(free_port(), path, str(tmp_path), q))
        p.start()
        err = q.get(timeout=240)
        p.join(timeout=60)
        assert not err, err
    ref = np.load(str(tmp_path / "pretend_owner_pass_1.npz"))
    assert len(ref["k"]) > 100000 and np.any(ref["w"] != 0)
    for path in paths[1:]:
        got = np.load(str(tmp_path / ("pretend_%s_%d.npz" % path)))
        for f in ("k", "w", "n", "z"):
            same(got[f], ref[f])


def _compile_dev_rank(rank, world, port, dev, outdir, q):
    try:
        import torch
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        g = capi.Group(rank, world, "127.0.0.1", port, capi.TRANSPORT_HOST, device=0)
        st = capi.Sharded(g, model="lr", optimizer="ftrl", capacity=64, schedule="owner")
        alive = []
        for s in range(3):
            rp, ks, lb = _big_data(rank, s)
            if dev:
                d = (torch.from_numpy(ks.view(np.int64).copy()).cuda(),
                     torch.from_numpy(rp.astype(np.uint32).view(np.int32)).cuda(),
                     torch.from_numpy(lb.copy()).cuda())
                torch.cuda.synchronize()
                b = st.compile_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(lb),
                                   len(ks), keep=(s != 1))
                del d                # (the arrays may go as soon as the call has returned)
                torch.cuda.empty_cache()
            else:
                b = st.compile(rp, ks, lb, keep=(s != 1))
            alive.append(b)
            st.step(b)
        st.check()
        k, w, n, z = st.w.export()
        np.savez(os.path.join(outdir, "cd%d_%d.npz" % (dev, rank)), k=k, w=w, n=n, z=z)
        g.barrier()
        st.close()
        g.close()
        q.put((rank, None))
    except Exception:
        q.put((rank, traceback.format_exc()))


def test_compile_from_device_arrays_is_the_compile_from_host_arrays(tmp_path):
    """xf_sharded_compile_dev (raw keys resident in HBM: the stable partition by key owner on the
    device, counts through the group's all-to-all, one host wait) against xf_sharded_compile on
    the same minibatches: world 2, 20 000 / 23 000 rows per worker, uniform and power-law keys —
    the shards of the table bit for bit, and (the host-array run) bit for bit the oracle's."""
    world = 2
    ctx = mp.get_context("spawn")
    for dev in (0, 1):
        q = ctx.Queue()
        port = free_port()
        ps = [ctx.Process(target=_compile_dev_rank, args=(r, world, port, dev, str(tmp_path), q))
              for r in range(world)]
        for p in ps:
            p.start()
        res = [q.get(timeout=240) for _ in ps]
        for p in ps:
            p.join(timeout=60)
        errs = [e for _, e in res if e]
        assert not errs, errs[0]
    for r in range(world):
        a, b = (np.load(str(tmp_path / ("cd%d_%d.npz" % (dev, r)))) for dev in (0, 1))
        for f in ("k", "w", "n", "z"):
            same(a[f], b[f])
    with O.sum_mode(1):
        w = O.Store(O.OPT_FTRL, 1)
        for s in range(3):
            obs = [O.Batch(*_big_data(r, s)) for r in range(world)]
            pulled = [w.pull(ob.ukeys) for ob in obs]
            grads = [ob.lr_grad(ob.lr_loss(pw)[0]) for ob, pw in zip(obs, pulled)]
            for ob, g in zip(obs, grads):
                w.push(ob.ukeys, g)
    parts = [np.load(str(tmp_path / ("cd1_%d.npz" % r))) for r in range(world)]
    ks, ws, ns, zs = w.export()
    k = np.concatenate([p["k"] for p in parts])
    order = np.argsort(k)
    same(k[order], ks)
    for f, ref in (("w", ws), ("n", ns), ("z", zs)):
        same(np.concatenate([p[f] for p in parts])[order].reshape(ref.shape), ref)


def _awkward_minibatch(step):
    """rows of 0 ... 60 nonzeros (a third of them empty), a key twice in a row, a field of 8 hot
    keys, a power-law head that makes its key range a merge sort, a 9000-nonzero row"""
    rng = np.random.RandomState(900 + step)
    R, K = 30000, 400000
    keytab = capi.hash_decimal_range(0, K)
    lens = np.where(rng.rand(R) < 0.33, 0, rng.randint(1, 61, size=R))
    lens[17] = 9000
    n = int(lens.sum())
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    fid = rng.randint(0, K, size=n)
    hot = rng.rand(n) < 0.2
    fid[hot] = np.minimum(rng.zipf(1.15, size=int(hot.sum())) - 1, K - 1)
    sig = rng.rand(n) < 0.03
    fid[sig] = K - 1 - rng.randint(0, 8, size=int(sig.sum()))
    fid[1::97] = fid[0:-1:97][:len(fid[1::97])]      # the neighbour's key again: twice in a row
    return rowptr, keytab[fid], rng.randint(0, 2, size=R).astype(np.int32)


def _worker_side_rank(port, mode, outdir, q):
    try:
        os.environ["XF_SHARDED_GENERAL"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        g = capi.Group(0, 1, "127.0.0.1", port, capi.TRANSPORT_HOST, device=0)
        st = capi.Sharded(g, model="lr", optimizer="ftrl", capacity=1 << 20, schedule="sequential")
        capi.tune("key_build", mode)
        import torch
        alive = []
        for s in range(4):
            rp, ks, lb = _awkward_minibatch(s)
            if s == 2:      # from device arrays; a one-row minibatch behind it
                d = (torch.from_numpy(ks.view(np.int64)).cuda(),
                     torch.from_numpy(rp.astype(np.uint32).view(np.int32)).cuda(),
                     torch.from_numpy(lb).cuda())
                torch.cuda.synchronize()
                alive.append(st.compile_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                            len(lb), len(ks), keep=False))
                st.step(alive[-1])
                alive.append(st.compile(np.array([0, 3], np.uint64), ks[:3].copy(), lb[:1].copy()))
            else:
                alive.append(st.compile(rp, ks, lb, keep=(s != 1)))
            st.step(alive[-1])
        st.step(alive[0])            # a kept minibatch again (its key-sorted copy)
        st.check()
        k, w, n, z = st.w.export()
        rp, ks, lb = _awkward_minibatch(99)
        np.savez(os.path.join(outdir, "ws_%d.npz" % mode), k=k, w=w, n=n, z=z,
                 p=st.predict(st.compile(rp, ks, lb)))
        st.close()
        g.close()
        q.put(None)
    except Exception:
        q.put(traceback.format_exc())


def test_worker_side_lr_build_equals_the_full_minibatch_build(tmp_path):
    """schedule sequential at world one (XF_SHARDED_GENERAL=1): the minibatch of the exchange's
    worker side built by hand — sorted unique keys + cells, xf::batch_compile_lr_dev — trains the
    table the full minibatch build with the library's sorts (xf_tune key_build = 1) trains, bit
    for bit, on minibatches with empty rows, repeated keys, hot keys and a power-law head"""
    ctx = mp.get_context("spawn")
    for mode in (0, 1):
        q = ctx.Queue()
        p = ctx.Process(target=_worker_side_rank, args=(free_port(), mode, str(tmp_path), q))
        p.start()
        err = q.get(timeout=240)
        p.join(timeout=60)
        assert not err, err
    a, b = np.load(str(tmp_path / "ws_0.npz")), np.load(str(tmp_path / "ws_1.npz"))
    assert len(a["k"]) > 100000
    for f in ("k", "w", "n", "z", "p"):
        same(a[f], b[f])

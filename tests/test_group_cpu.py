"""The process group (xf_group_*) without a GPU: TCP bootstrap, rank assignment, the host-side
collectives, and the all-to-all-v over the host transport with host buffers — world 2 and 3,
one OS process per rank.  The RCCL transport itself needs GPUs (tests/test_gpu_group.py)."""
import multiprocessing as mp
import socket
import traceback

import numpy as np
import pytest

from xflow_amd import capi


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, auto_rank, q):
    try:
        g = capi.Group(rank=-1 if auto_rank else rank, world=world, addr="127.0.0.1", port=port,
                       transport=capi.TRANSPORT_HOST)
        r = g.rank
        assert g.world == world and 0 <= r < world
        g.barrier()
        # allgather: every rank contributes a small vector
        got = g.allgather(np.arange(3, dtype=np.int64) + 10 * r)
        want = np.stack([np.arange(3, dtype=np.int64) + 10 * p for p in range(world)])
        assert np.array_equal(got, want)
        # all-to-all-v with ragged, partly empty slices: rank r sends (r + 2p) % 4 items to p
        counts = np.array([(r + 2 * p) % 4 for p in range(world)], np.uint64)
        send = np.concatenate([np.full(int(c), 1000 * r + p, np.uint64)
                               for p, c in enumerate(counts)] + [np.zeros(0, np.uint64)])
        recv, rc = g.alltoallv_host(send, counts)
        exp_counts = np.array([(p + 2 * r) % 4 for p in range(world)], np.uint64)
        assert np.array_equal(rc, exp_counts)
        exp = np.concatenate([np.full(int(c), 1000 * p + r, np.uint64)
                              for p, c in enumerate(exp_counts)] + [np.zeros(0, np.uint64)])
        assert np.array_equal(recv, exp)
        # gatherv to rank 0
        mine = np.full(r + 1, float(r), np.float32)
        allv = g.gatherv(mine)
        if r == 0:
            assert np.array_equal(allv, np.concatenate([np.full(p + 1, float(p), np.float32)
                                                        for p in range(world)]))
        else:
            assert allv is None
        g.barrier()
        g.close()
        q.put((rank, r, None))
    except Exception:
        q.put((rank, -1, traceback.format_exc()))


@pytest.mark.parametrize("world,auto_rank", [(2, False), (3, False), (3, True)])
def test_group_host_transport(world, auto_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=_rank_main, args=(r, world, port, auto_rank, q))
          for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    errs = [e for _, _, e in res if e]
    assert not errs, errs[0]
    assert sorted(r for _, r, _ in res) == list(range(world))   # every rank handed out once


def _bring_up_main(rank, world, port, transport, q):
    try:
        try:
            g = capi.Group(rank=rank, world=world, addr="127.0.0.1", port=port,
                           transport=transport)
        except capi.XFError as e:
            q.put((rank, "error", str(e)))
            return
        recv, _ = g.alltoallv_host(np.full(world, rank, np.int32), [1] * world)
        assert list(recv) == list(range(world))
        q.put((rank, g.transport, None))
        g.close()
    except Exception:
        q.put((rank, "crash", traceback.format_exc()))


@pytest.mark.parametrize("transport", ["auto", "rccl"])
def test_rccl_bring_up_without_gpus_is_collective(transport):
    """No GPU here, so RCCL cannot come up.  The ranks agree on that over the bootstrap: AUTO
    lands every rank on the host transport, the strict RCCL transport fails on every rank with
    the first failing rank's reason — nobody hangs in a half-built communicator."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU behaviour")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    t = capi.TRANSPORT_AUTO if transport == "auto" else capi.TRANSPORT_RCCL
    ps = [ctx.Process(target=_bring_up_main, args=(r, world, port, t, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    assert not [m for _, k, m in res if k == "crash"], res
    if transport == "auto":
        assert [k for _, k, _ in res] == [capi.TRANSPORT_HOST] * world, res
    else:
        assert [k for _, k, _ in res] == ["error"] * world, res
        assert all("rank " in m for _, _, m in res), res


def test_group_world1_and_bad_arguments():
    g = capi.Group(rank=0, world=1, transport=capi.TRANSPORT_HOST)
    assert (g.rank, g.world) == (0, 1)
    g.barrier()
    recv, rc = g.alltoallv_host(np.arange(5, dtype=np.float32), [5])
    assert np.array_equal(recv, np.arange(5, dtype=np.float32)) and list(rc) == [5]
    g.close()
    with pytest.raises(capi.XFError):
        capi.Group(rank=3, world=2, transport=capi.TRANSPORT_HOST)


def _stray(port, q):
    """a connection that is not a rank: keeps trying until rank 0 listens, says the wrong thing"""
    import time
    t0 = time.time()
    while time.time() - t0 < 60:
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=1) as c:
                c.sendall(b"GET / HT")          # 8 bytes, not the group's hello
                time.sleep(0.5)
            q.put("sent")
            return
        except OSError:
            time.sleep(0.02)
    q.put("never connected")


def test_a_stray_connection_does_not_take_a_rank():
    """rank 0 checks a magic word in every hello: something else that connects to the
    rendezvous port is dropped, the group still forms with its real ranks"""
    ctx = mp.get_context("spawn")
    q, sq = ctx.Queue(), ctx.Queue()
    port = free_port()
    stray = ctx.Process(target=_stray, args=(port, sq))
    r0 = ctx.Process(target=_rank_main, args=(0, 2, port, False, q))
    r0.start()
    stray.start()
    assert sq.get(timeout=90) == "sent"
    r1 = ctx.Process(target=_rank_main, args=(1, 2, port, False, q))
    r1.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in (r0, r1, stray):
        p.join(timeout=30)
    errs = [e for _, _, e in res if e]
    assert not errs, errs[0]

"""Multi-GPU driver: one process per GPU, the parameter table sharded by key range, keys /
weights / gradients moved to and from their owning shard with all-to-all collectives
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm).

Replaces ps-lite's worker<->server routing (src/model/lr/lr_worker.cc:170,175;
src/model/fm/fm_worker.cc:228-242): `Pull(keys)` = keys to owners, weights back;
`Push(keys, grads)` = gradients to owners, owner-side FTRL/SGD.  Like ps-lite's default
slicer, a SORTED key list splits into one contiguous range per owner
(owner = min(key / (UINT64_MAX / N), N-1)), so the exchange is a plain all-to-all-v with
no permutation.  xGMI is a full mesh: each peer pair has its own link, an all-to-all is one
message per link.

Update semantics with N workers: every worker's gradient is its own FTRL step
(scaled by its own 1/R, lr_worker.cc:116-118); the owner applies the N pushes of a step in
RANK ORDER after all N pulls — one legal, deterministic serialisation of what ps-lite does
asynchronously.

This module contains no arithmetic: the stages are the HIP kernels behind the C ABI
(`HipStages`).  The stage object is injectable so that the collective plumbing can be
exercised on CPU (`gloo`) in tests with a checker-backed stage object; the default, and the
only one this package ships, is the GPU one, and it raises without a GPU.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi


def owner_boundaries(world):
    """First key of every shard under the ps-lite uniform range rule."""
    span = (2**64 - 1) // world
    return [span * i for i in range(world)]


def split_counts(ukeys_sorted_u64, world):
    """How many of the sorted unique keys each owner gets (contiguous ranges)."""
    bounds = np.array(owner_boundaries(world)[1:], dtype=np.uint64)
    cuts = np.searchsorted(ukeys_sorted_u64, bounds, side="left")
    edges = np.concatenate([[0], cuts, [len(ukeys_sorted_u64)]])
    return np.diff(edges).astype(np.int64)


class HipStages:
    """The product stages: torch CUDA tensors hold the buffers, the HIP kernels do the work
    on torch's current stream (so they order correctly with the RCCL collectives)."""

    def __init__(self, model, optimizer, k, capacity, rank, world, seed=7, **hyper):
        capi.require_gpu()
        if not torch.cuda.is_available():
            raise capi.XFError("HipStages needs a GPU (no CPU fallback)")
        self.dev = torch.device("cuda", torch.cuda.current_device())
        opt = capi.OPT_FTRL if optimizer == "ftrl" else capi.OPT_SGD
        self.model, self.k = model, (k if model == "fm" else 0)
        self.w = capi.Table(opt, 1, capi.INIT_ZERO, capacity=capacity, shard=rank,
                            nshards=world, **hyper)
        self.v = None
        if model == "fm":
            init = capi.INIT_HASHNORM if opt == capi.OPT_FTRL else capi.INIT_CONST
            self.v = capi.Table(opt, k, init, 0.001, seed=seed, capacity=capacity, shard=rank,
                                nshards=world, **hyper)

    # -- buffers -------------------------------------------------------------------------
    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.dev)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    # -- batch ---------------------------------------------------------------------------
    def compile_batch(self, rowptr, keys, labels):
        """Key build on the GPU; the kernels use the batch's own device arrays, only the
        sorted unique key list is also kept as a tensor (it is what the all-to-all sends)."""
        hb = capi.Batch(rowptr, keys, labels, on_gpu=True)
        b = _DeviceBatch()
        b.hb = hb                       # owns the device arrays behind `view`
        b.R, b.NNZ, b.U, b.H = hb.R, hb.NNZ, hb.U, hb.H
        b.view = hb.dev_view()
        b.ukeys_host = hb.host()["ukeys"]
        b.ukeys = self.from_numpy(b.ukeys_host.view(np.int64))
        return b

    # -- table stages (owner side) -------------------------------------------------------
    def resolve(self, table, keys_i64):
        slots = self.empty(keys_i64.numel(), torch.int32)
        table.resolve_dev(keys_i64.data_ptr(), keys_i64.numel(), slots.data_ptr(),
                          self._stream())
        return slots

    def pull(self, table, keys_i64):
        """resolve + weight payload in one pass (dim-1, zero-initialised tables)"""
        n = keys_i64.numel()
        rows = self.empty(n, torch.int32)
        vals = self.empty(n, torch.float32)
        capi.check(capi.lib().xf_table_pull_dev(table.h, keys_i64.data_ptr(), n, rows.data_ptr(),
                                                vals.data_ptr(), self._stream()))
        return rows, vals

    # the same, all source ranks' key lists in one pass over the shard (entries visited in key
    # order, results in the per-source layout); see xf_table_pull_ordered_dev
    def pull_ordered(self, table, keys_sorted, order, want_values):
        n = keys_sorted.numel()
        rows = self.empty(n, torch.int32)
        vals = self.empty(n, torch.float32) if want_values else None
        capi.check(capi.lib().xf_table_pull_ordered_dev(
            table.h, keys_sorted.data_ptr(), order.data_ptr(), n, rows.data_ptr(),
            vals.data_ptr() if want_values else None, self._stream()))
        return rows, vals

    def update_merged(self, table, keys_sorted, order, slots, grads):
        capi.check(capi.lib().xf_table_update_merged_dev(
            table.h, keys_sorted.data_ptr(), order.data_ptr(), keys_sorted.numel(),
            slots.data_ptr(), grads.data_ptr(), self._stream()))

    def gather(self, table, slots):
        vals = self.empty(slots.numel() * table.dim, torch.float32)
        table.gather_dev(slots.data_ptr(), slots.numel(), vals.data_ptr(), self._stream())
        return vals

    def update(self, table, slots, grads):
        table.update_dev(slots.data_ptr(), slots.numel(), grads.data_ptr(), self._stream())

    # -- model stages (worker side) ------------------------------------------------------
    def lr_forward(self, b, wu):
        loss = self.empty(b.R, torch.float32)
        capi.check(capi.lib().xf_lr_forward_dev(capi.C.byref(b.view), wu.data_ptr(),
                                                loss.data_ptr(), None, self._stream()))
        return loss

    def lr_grad(self, b, loss):
        g = self.empty(b.U, torch.float32)
        capi.check(capi.lib().xf_lr_grad_dev(capi.C.byref(b.view), loss.data_ptr(),
                                             g.data_ptr(), self._stream()))
        return g

    def fm_forward(self, b, wu, vu):
        loss = self.empty(b.R, torch.float32)
        vsum = self.empty(b.R, torch.float32)
        capi.check(capi.lib().xf_fm_forward_dev(capi.C.byref(b.view), self.k, wu.data_ptr(),
                                                vu.data_ptr(), loss.data_ptr(), None,
                                                vsum.data_ptr(), self._stream()))
        return loss, vsum

    def fm_grad(self, b, vu, vsum, loss):
        gw = self.empty(b.U, torch.float32)
        gv = self.empty(b.U * self.k, torch.float32)
        capi.check(capi.lib().xf_fm_grad_dev(capi.C.byref(b.view), self.k, vu.data_ptr(),
                                             vsum.data_ptr(), loss.data_ptr(), gw.data_ptr(),
                                             gv.data_ptr(), self._stream()))
        return gw, gv

    def update_multi(self, table, slots, grads, counts):
        """one optimizer pass per source rank, in rank order"""
        off = 0
        for cnt in counts:
            if cnt:
                self.update(table, slots[off:off + cnt],
                            grads[off * table.dim:(off + cnt) * table.dim])
            off += cnt

    # -- streams / events for the stale1 schedule ------------------------------------------
    def _s(self, name):
        if name == "main":
            return torch.cuda.current_stream() if not hasattr(self, "_main") else self._main
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream()
        return self._side

    def on(self, name):
        if not hasattr(self, "_main"):
            self._main = torch.cuda.current_stream()
        return torch.cuda.stream(self._s(name))

    def record(self, stream, event):
        if not hasattr(self, "_main"):
            self._main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(self._s(stream))
        self._events = getattr(self, "_events", {})
        self._events[event] = ev

    def wait(self, stream, event):
        ev = getattr(self, "_events", {}).get(event)
        if ev is not None:
            self._s(stream).wait_event(ev)

    def sync(self):
        torch.cuda.synchronize()

    def check(self):
        torch.cuda.synchronize()
        self.w.check(self._stream())
        if self.v is not None:
            self.v.check(self._stream())

    def tables(self):
        return self.w, self.v


class _DeviceBatch:
    pass


class ShardedTrainer:
    """LRWorker/FMWorker::update across `world` ranks with a key-range-sharded table."""

    def __init__(self, model="lr", optimizer="ftrl", k=10, capacity=1 << 22, rank=None,
                 world=None, stages=None, group=None, schedule="sequential", exchange=None,
                 **hyper):
        """`exchange(out, src, out_splits, in_splits, group)`: the all-to-all-v primitive;
        default torch.distributed.all_to_all_single on the device buffers (RCCL on GPUs).
        Injectable so that tests can run several ranks of the real HIP stages on ONE GPU
        (RCCL refuses two ranks per device) by staging the exchange through gloo."""
        assert schedule in ("sequential", "stale1")
        self.schedule = schedule
        self._xchg = exchange
        self._pending = None
        self._retired = None
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.model = model
        self.stages = stages if stages is not None else HipStages(
            model, optimizer, k, capacity, self.rank, self.world, **hyper)
        self.k = k if model == "fm" else 0
        # with a process group the exchange always goes through the collective, also at
        # world 1 (RCCL handles it; keeps the one-GPU run on the N-GPU code path)
        self._collective = dist.is_available() and dist.is_initialized()
        assert self._collective or self.world == 1, "world > 1 needs a process group"
        self._prof = None
        self._ev = []

    # ---- compile: key build + the (static) exchange plan of this minibatch ---------------
    def compile(self, rowptr, keys, labels):
        b = self.stages.compile_batch(rowptr, keys, labels)
        send = split_counts(b.ukeys_host, self.world)
        recv = torch.empty(self.world, dtype=torch.int64)
        cnt = torch.from_numpy(send.copy())
        if self._collective:
            dev = b.ukeys.device
            cnt_d, recv_d = cnt.to(dev), recv.to(dev)
            self._exchange(recv_d, cnt_d, None, None)
            recv = recv_d.cpu()
        else:
            recv = cnt.clone()
        b.send_counts = send.tolist()
        b.recv_counts = recv.tolist()
        b.n_recv = int(sum(b.recv_counts))
        # The owner-side key lists are part of the compiled minibatch too: like the batch
        # itself they depend only on the input rows, so the key all-to-all runs once here and
        # a step exchanges only what changes — weights one way, gradients the other.
        b.rkeys = self._a2a(b.ukeys, b.send_counts, b.recv_counts)
        # ... and so is the order in which the owner walks them: all sources' lists merged by
        # key (stable, so a key's entries stay in source-rank order).  Walking the lists one
        # source after the other would sweep the shard's index and state once per source.
        b.rkeys_sorted = b.rorder = None
        # (with one source there is nothing to merge: the plain passes are 25 us cheaper)
        if hasattr(self.stages, "pull_ordered") and b.n_recv and self.world > 1:
            flip = torch.iinfo(torch.int64).min           # u64 order through an i64 sort
            srt, order = torch.sort(b.rkeys ^ flip, stable=True)
            b.rkeys_sorted = srt ^ flip
            b.rorder = order.to(torch.int32)
        return b

    def _a2a(self, src, in_counts, out_counts, width=1):
        out = self.stages.empty(sum(out_counts) * width, src.dtype)
        if not self._collective:
            out.copy_(src)
            return out
        self._exchange(out, src, [c * width for c in out_counts], [c * width for c in in_counts])
        return out

    def _exchange(self, out, src, out_splits, in_splits):
        if self._xchg is not None:
            self._xchg(out, src, out_splits, in_splits, self.group)
        else:
            dist.all_to_all_single(out, src, out_splits, in_splits, group=self.group)

    # ---- one minibatch step ----------------------------------------------------------------
    def _resolve(self, table, rkeys, counts):
        """key -> state row on the owner, all source ranks' lists in one call (the same key
        may arrive from several workers; the table's resolve handles that)."""
        if rkeys.numel() == 0:
            return self.stages.empty(0, torch.int32)
        return self.stages.resolve(table, rkeys)

    def _mark(self, name):
        if self._prof is not None and self.schedule == "sequential":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev.append((name, e))

    # ---- one minibatch step -------------------------------------------------------------
    # front half: Pull (owner resolve + gather, weights back) -> forward -> gradient
    # back half : Push (gradients to the owners, owner-side optimizer step per source rank)
    def _front_pull(self, b):
        st = self.stages
        tw, tv = st.tables()
        rkeys = b.rkeys     # keys went to their owners at compile time (ps-lite slicer ranges)
        c = {}
        # owner: key -> state row (insert on first touch, ftrl.h:56) and the weight payload
        merged = getattr(b, "rorder", None) is not None
        if merged:
            c["slots_w"], c["w_recv"] = st.pull_ordered(tw, b.rkeys_sorted, b.rorder, True)
        elif hasattr(st, "pull") and rkeys.numel():
            c["slots_w"], c["w_recv"] = st.pull(tw, rkeys)
        else:
            c["slots_w"] = self._resolve(tw, rkeys, b.recv_counts)
            c["w_recv"] = st.gather(tw, c["slots_w"])
        if self.model == "fm":
            if merged:
                c["slots_v"], _ = st.pull_ordered(tv, b.rkeys_sorted, b.rorder, False)
            else:
                c["slots_v"] = self._resolve(tv, rkeys, b.recv_counts)
        self._mark("resolve")
        if self.model == "fm":
            c["v_recv"] = st.gather(tv, c["slots_v"])
        self._mark("gather")
        return c

    def _front_compute(self, b, c):
        st = self.stages
        # Pull, part 2: weights back, in the order the keys were sent
        wu = self._a2a(c["w_recv"], b.recv_counts, b.send_counts)
        if self.model == "fm":
            vu = self._a2a(c["v_recv"], b.recv_counts, b.send_counts, self.k)
        self._mark("a2a_weights")
        if self.model == "lr":
            loss = st.lr_forward(b, wu)
            self._mark("forward")
            c["g"] = st.lr_grad(b, loss)
        else:
            loss, vsum = st.fm_forward(b, wu, vu)
            self._mark("forward")
            c["g"], c["gv"] = st.fm_grad(b, vu, vsum, loss)
        self._mark("gradient")
        c["wu"], c["loss"] = wu, loss

    def _back(self, b, c):
        """Push: gradients to the owners (keys are already there), owner-side optimizer
        step, one worker after the other in rank order."""
        st = self.stages
        tw, tv = st.tables()
        g_recv = self._a2a(c["g"], b.send_counts, b.recv_counts)
        if self.model == "fm":
            gv_recv = self._a2a(c["gv"], b.send_counts, b.recv_counts, self.k)
        self._mark("a2a_grads")
        c["g_recv"] = g_recv
        return g_recv, (gv_recv if self.model == "fm" else None)

    def _apply(self, b, c, g_recv, gv_recv):
        st = self.stages
        tw, tv = st.tables()
        if getattr(b, "rorder", None) is not None:   # one pass, a key's sources in rank order
            st.update_merged(tw, b.rkeys_sorted, b.rorder, c["slots_w"], g_recv)
            if self.model == "fm":
                st.update_merged(tv, b.rkeys_sorted, b.rorder, c["slots_v"], gv_recv)
        elif hasattr(st, "update_multi"):
            st.update_multi(tw, c["slots_w"], g_recv, b.recv_counts)
            if self.model == "fm":
                st.update_multi(tv, c["slots_v"], gv_recv, b.recv_counts)
        else:
            off = 0
            for cnt in b.recv_counts:
                if cnt:
                    st.update(tw, c["slots_w"][off:off + cnt], g_recv[off:off + cnt])
                    if self.model == "fm":
                        st.update(tv, c["slots_v"][off:off + cnt],
                                  gv_recv[off * self.k:(off + cnt) * self.k])
                off += cnt
        self._mark("update")

    def step(self, b):
        """schedule "sequential" (default): Pull, compute, Push of a step finish before the
        next step's Pull — the order the single-GPU path and the parity tests use.
        schedule "stale1": the Push of step t is applied after the Pull of step t+1 has read
        the table (weights are one step stale — inside ps-lite's asynchronous semantics, and
        still deterministic), on a second stream: its all-to-all overlaps the next Pull and
        its optimizer pass overlaps the next weights exchange, forward and gradient."""
        st = self.stages
        self._mark("begin")
        if self.schedule == "sequential":
            c = self._front_pull(b)
            self._front_compute(b, c)
            g_recv, gv_recv = self._back(b, c)
            self._apply(b, c, g_recv, gv_recv)
            self._last = c
            return
        # ---- stale1 ----
        prev = self._pending                      # (batch, ctx) whose Push is outstanding
        st.wait("main", "applied")                # the Push of step t-2 is in the table
        c = self._front_pull(b)                   # reads the table before Push(t-1) lands
        st.record("main", "pulled")
        if prev is not None:
            pb, pc = prev
            with st.on("side"):
                st.wait("side", "graded")         # gradient of step t-1 is complete
                g_recv, gv_recv = self._back(pb, pc)
                st.wait("side", "pulled")         # do not write while Pull(t) reads
                self._apply(pb, pc, g_recv, gv_recv)
                st.record("side", "applied")
        self._front_compute(b, c)
        st.record("main", "graded")
        self._retired = prev                      # keep step t-1's buffers alive one more step
        self._pending = (b, c)
        self._last = c

    def flush(self):
        """apply the outstanding Push of the stale1 schedule (end of training / before export)"""
        st = self.stages
        if self.schedule != "sequential" and self._pending is not None:
            pb, pc = self._pending
            with st.on("side"):
                st.wait("side", "graded")
                g_recv, gv_recv = self._back(pb, pc)
                st.wait("side", "pulled")
                self._apply(pb, pc, g_recv, gv_recv)
                st.record("side", "applied")
            st.wait("main", "applied")
            self._pending = None
        st.sync()

    def defrag(self):
        """renumber this shard's state rows in key order (table maintenance between steps,
        once the key set has settled; local to the rank, no exchange).  Applies an outstanding
        stale1 Push first, because row numbers held by a step in flight would go stale."""
        self.flush()
        for t in self.stages.tables():
            if t is not None and hasattr(t, "defrag"):
                t.defrag()

    def predict(self, b):
        """forward only (calculate_pctr): pulls insert unseen keys, as in the reference."""
        st = self.stages
        tw, tv = st.tables()
        rkeys = b.rkeys

        def rows_of(table):
            if getattr(b, "rorder", None) is not None:
                return st.pull_ordered(table, b.rkeys_sorted, b.rorder, False)[0]
            return self._resolve(table, rkeys, b.recv_counts)
        wu = self._a2a(st.gather(tw, rows_of(tw)), b.recv_counts, b.send_counts)
        if self.model == "lr":
            return st.lr_forward(b, wu)
        vu = self._a2a(st.gather(tv, rows_of(tv)), b.recv_counts, b.send_counts, self.k)
        return st.fm_forward(b, wu, vu)[0]

    def check(self):
        self.flush()
        self.stages.check()

    # bench hooks: per-stage timing with events recorded on the stream the kernels and the
    # collectives are enqueued on
    def profile(self, enable):
        self._prof = {} if enable else None
        self._ev = []

    def profile_read(self):
        if not self._ev:
            return {}, 0
        torch.cuda.synchronize()
        sums, steps = {}, 0
        for (n0, e0), (n1, e1) in zip(self._ev[:-1], self._ev[1:]):
            if n1 == "begin":
                continue
            sums[n1] = sums.get(n1, 0.0) + e0.elapsed_time(e1)
            steps += n1 == "update"
        self._ev = []
        return sums, steps

"""TEST DOUBLE (lives under tests/, never shipped): the stage interface of
xflow_amd.sharded.HipStages implemented on CPU tensors with the oracle as the arithmetic,
so the collective plumbing of ShardedTrainer (owner bucketing, all-to-all-v of keys /
weights / gradients, rank-ordered owner updates) can run under gloo without a GPU."""
import numpy as np
import torch

from oracle import pyoracle as O


class _Tab:
    """oracle Store + a slot numbering (slot = order of first resolve on this shard)."""

    def __init__(self, opt, dim, init, c, seed, rank, world):
        self.store = O.Store(opt, dim, init, c, seed)
        self.dim = dim
        self.keys = []
        self.index = {}
        self.rank, self.world = rank, world

    def resolve(self, keys):
        out = np.empty(len(keys), dtype=np.int32)
        for i, k in enumerate(keys.tolist()):
            assert O.lib().xo_shard_of(k, self.world) == self.rank, "foreign key"
            if k not in self.index:
                self.index[k] = len(self.keys)
                self.keys.append(k)
            out[i] = self.index[k]
        self.store.pull(keys)  # insert-on-pull
        return out

    def keys_of(self, slots):
        return np.array([self.keys[s] for s in slots.tolist()], dtype=np.uint64)


class CpuOracleStages:
    def __init__(self, model, optimizer, k, rank, world, seed=7):
        opt = O.OPT_FTRL if optimizer == "ftrl" else O.OPT_SGD
        self.model, self.k = model, (k if model == "fm" else 0)
        self.w = _Tab(opt, 1, O.INIT_ZERO, 0.0, 0, rank, world)
        self.v = None
        if model == "fm":
            init = O.INIT_HASHNORM if opt == O.OPT_FTRL else O.INIT_CONST
            self.v = _Tab(opt, k, init, 0.001, seed, rank, world)

    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype)

    def compile_batch(self, rowptr, keys, labels):
        ob = O.Batch(rowptr, keys, labels)

        class B:
            pass
        b = B()
        b.ob = ob
        b.R, b.NNZ, b.U = ob.R, ob.NNZ, ob.U
        b.ukeys_host = ob.ukeys
        b.ukeys = torch.from_numpy(ob.ukeys.view(np.int64).copy())
        return b

    def resolve(self, table, keys_i64):
        return torch.from_numpy(table.resolve(keys_i64.numpy().view(np.uint64)))

    def gather(self, table, slots):
        vals = table.store.pull(table.keys_of(slots.numpy()))
        return torch.from_numpy(np.ascontiguousarray(vals, dtype=np.float32).ravel().copy())

    def update(self, table, slots, grads):
        table.store.push(table.keys_of(slots.numpy()), grads.numpy())

    def lr_forward(self, b, wu):
        return torch.from_numpy(b.ob.lr_loss(wu.numpy())[0])

    def lr_grad(self, b, loss):
        return torch.from_numpy(b.ob.lr_grad(loss.numpy()))

    def fm_forward(self, b, wu, vu):
        loss, _, vsum = b.ob.fm_loss(self.k, wu.numpy(), vu.numpy())
        return torch.from_numpy(loss), torch.from_numpy(vsum)

    def fm_grad(self, b, vu, vsum, loss):
        gw, gv = b.ob.fm_grad(self.k, vu.numpy(), vsum.numpy(), loss.numpy())
        return torch.from_numpy(gw), torch.from_numpy(gv.ravel().copy())

    # the stale1 schedule's stream plumbing: nothing to do on one CPU thread
    class _Null:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def on(self, name):
        return self._Null()

    def record(self, stream, event):
        pass

    def wait(self, stream, event):
        pass

    def sync(self):
        pass

    def check(self):
        pass

    def tables(self):
        return self.w, self.v


class CpuOracleStagesMerged(CpuOracleStages):
    """The same double with the owner-side stages of the merged walk (all sources' key lists
    in key order, xf_table_pull_ordered_dev / xf_table_update_merged_dev): when a stage object
    offers them and world > 1, ShardedTrainer sorts the received keys at compile time and
    calls these instead of one pass per source."""

    def pull_ordered(self, table, keys_sorted, order, want_values):
        ks = keys_sorted.numpy().view(np.uint64)
        od = order.numpy().astype(np.int64)
        assert np.all(ks[:-1] <= ks[1:]), "entries must arrive in (unsigned) key order"
        rows = np.empty(len(ks), dtype=np.int32)
        rows[od] = table.resolve(ks)
        vals = None
        if want_values:
            v = np.empty(len(ks), dtype=np.float32)
            v[od] = np.asarray(table.store.pull(ks), dtype=np.float32).ravel()
            vals = torch.from_numpy(v)
        return torch.from_numpy(rows), vals

    def update_merged(self, table, keys_sorted, order, slots, grads):
        ks = keys_sorted.numpy().view(np.uint64)
        od = order.numpy().astype(np.int64)
        g = grads.numpy().reshape(len(ks), table.dim)
        sl = slots.numpy()
        for i in range(len(ks)):          # key order; a key's sources in rank order (stable)
            assert table.keys[sl[od[i]]] == int(ks[i])
            table.store.push(ks[i:i + 1], g[od[i]])

"""Single-GPU driver of the fused minibatch step (thin wrapper over the C ABI)."""
from . import capi


class SingleGpuTrainer:
    """One shard, one GPU: LRWorker::update / FMWorker::update on device-resident batches."""

    def __init__(self, model="lr", optimizer="ftrl", k=10, capacity=1 << 22, seed=7,
                 rank=0, world=1, **hyper):
        assert world == 1
        opt = capi.OPT_FTRL if optimizer == "ftrl" else capi.OPT_SGD
        self.model = model
        self.w = capi.Table(opt, 1, capi.INIT_ZERO, capacity=capacity, **hyper)
        self.v = None
        if model == "fm":
            init = capi.INIT_HASHNORM if opt == capi.OPT_FTRL else capi.INIT_CONST
            self.v = capi.Table(opt, k, init, 0.001, seed=seed, capacity=capacity, **hyper)
        self.ws = capi.Workspace()

    def compile(self, rowptr, keys, labels):
        if self.model == "lr":   # sort-free key build against the table (cells)
            return capi.LocalBatch(self.w, rowptr, keys, labels)
        return capi.Batch(rowptr, keys, labels).upload()

    def step(self, batch, stream=None):
        if self.model == "lr":
            capi.lr_step(self.w, batch, self.ws, stream)
        else:
            capi.fm_step(self.w, self.v, batch, self.ws, stream)

    def predict(self, batch):
        if self.model == "lr":
            return capi.lr_predict(self.w, batch, self.ws)
        return capi.fm_predict(self.w, self.v, batch, self.ws)

    def defrag(self):
        """renumber state rows in key order (call when the key set has settled)"""
        self.w.defrag()
        if self.v is not None:
            self.v.defrag()

    def check(self):
        self.w.check()
        if self.v is not None:
            self.v.check()

    def profile(self, enable):
        self.ws.profile(enable)

    def profile_read(self):
        return self.ws.profile_read()

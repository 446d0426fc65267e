"""ctypes window onto oracle/liboracle.so (and oracle/_ref/libxflow_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (xflow_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)

OPT_FTRL, OPT_SGD = 0, 1
INIT_ZERO, INIT_CONST, INIT_HASHNORM = 0, 1, 2


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "xflow_oracle.cc")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _ptr(a, t):
    return a.ctypes.data_as(t)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.xo_hash_bytes.restype = C.c_uint64
    L.xo_hash_bytes.argtypes = [C.c_char_p, C.c_size_t]
    L.xo_shard_of.restype = C.c_uint32
    L.xo_shard_of.argtypes = [C.c_uint64, C.c_uint32]
    L.xo_sigmoid.restype = C.c_float
    L.xo_sigmoid.argtypes = [C.c_float]
    L.xo_hashnorm.restype = C.c_float
    L.xo_hashnorm.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.xo_reader_open.restype = C.c_void_p
    L.xo_reader_open.argtypes = [C.c_char_p, C.c_size_t]
    L.xo_reader_close.argtypes = [C.c_void_p]
    L.xo_reader_next.restype = C.c_long
    L.xo_reader_next.argtypes = [C.c_void_p]
    for nm, rt in [("rows", C.c_size_t), ("nnz", C.c_size_t), ("rowptr", _u64p),
                   ("keys", _u64p), ("fgid", _i32p), ("labels", _i32p)]:
        f = getattr(L, "xo_reader_" + nm)
        f.restype = rt
        f.argtypes = [C.c_void_p]
    L.xo_store_create.restype = C.c_void_p
    L.xo_store_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64]
    L.xo_store_destroy.argtypes = [C.c_void_p]
    L.xo_store_set_ftrl.argtypes = [C.c_void_p] + [C.c_float] * 4
    L.xo_store_set_sgd.argtypes = [C.c_void_p, C.c_float]
    L.xo_store_size.restype = C.c_size_t
    L.xo_store_size.argtypes = [C.c_void_p]
    L.xo_store_reserve.restype = None
    L.xo_store_reserve.argtypes = [C.c_void_p, C.c_size_t]
    L.xo_store_pull.argtypes = [C.c_void_p, _u64p, C.c_size_t, _f32p]
    L.xo_store_push.argtypes = [C.c_void_p, _u64p, C.c_size_t, _f32p]
    L.xo_store_export.argtypes = [C.c_void_p, _u64p, _f32p, _f32p, _f32p]
    L.xo_store_import.argtypes = [C.c_void_p, _u64p, C.c_size_t, _f32p, _f32p, _f32p]
    L.xo_ftrl_step.argtypes = [C.c_float] * 5 + [_f32p] * 3
    L.xo_batch_build.restype = C.c_void_p
    L.xo_batch_build.argtypes = [_u64p, _u64p, _i32p, C.c_size_t, C.c_size_t]
    L.xo_batch_free.argtypes = [C.c_void_p]
    for nm, rt in [("rows", C.c_size_t), ("nnz", C.c_size_t), ("nuniq", C.c_size_t),
                   ("ukeys", _u64p), ("rowptr", _u32p), ("uidx", _u32p),
                   ("segptr", _u32p), ("coo_row", _u32p), ("labels", _i32p)]:
        f = getattr(L, "xo_batch_" + nm)
        f.restype = rt
        f.argtypes = [C.c_void_p]
    L.xo_lr_loss.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
    L.xo_lr_grad.argtypes = [C.c_void_p, _f32p, _f32p]
    L.xo_lr_update.argtypes = [C.c_void_p, C.c_void_p]
    L.xo_fm_loss.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p]
    L.xo_fm_grad.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p]
    L.xo_fm_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.xo_auc_logloss.argtypes = [_i32p, _f32p, C.c_size_t, _f32p, _f32p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.xo_set_sum_mode.argtypes = [C.c_int]
    L.xo_lr_update_slices_mt.restype = C.c_long
    L.xo_lr_update_slices_mt.argtypes = [C.c_void_p, _u64p, _u64p, _i32p, C.c_size_t, C.c_int]
    L.xo_train.restype = C.c_long
    L.xo_train.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int,
                           C.c_size_t, C.c_int]
    L.xo_predict.restype = C.c_long
    L.xo_predict.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t,
                             C.c_int, _i32p, _f32p, C.c_size_t]
    _lib = L
    return L


# ----------------------------------------------------------------------------- helpers
class sum_mode:
    """with sum_mode(1): ...  -> the exact-sum variant of the oracle (see xflow_oracle.cc)"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = lib().xo_get_sum_mode()
        lib().xo_set_sum_mode(self.mode)

    def __exit__(self, *a):
        lib().xo_set_sum_mode(self.prev)



def hash_str(s):
    b = s if isinstance(s, bytes) else str(s).encode()
    return int(lib().xo_hash_bytes(b, len(b)))


def sigmoid(x):
    return float(lib().xo_sigmoid(C.c_float(x)))


def hashnorm(seed, key, j):
    return float(lib().xo_hashnorm(seed, key, j))


def read_blocks(path, cap_bytes):
    """Yield (rowptr u64[R+1], keys u64[NNZ], fgid i32[NNZ], labels i32[R]) per block."""
    L = lib()
    h = L.xo_reader_open(path.encode(), cap_bytes)
    if not h:
        raise IOError("cannot open %s" % path)
    try:
        while True:
            rows = L.xo_reader_next(h)
            if rows < 0:
                raise ValueError("malformed input in %s" % path)
            if rows == 0:
                return
            nnz = L.xo_reader_nnz(h)
            yield (np.ctypeslib.as_array(L.xo_reader_rowptr(h), (rows + 1,)).copy(),
                   np.ctypeslib.as_array(L.xo_reader_keys(h), (max(nnz, 1),))[:nnz].copy(),
                   np.ctypeslib.as_array(L.xo_reader_fgid(h), (max(nnz, 1),))[:nnz].copy(),
                   np.ctypeslib.as_array(L.xo_reader_labels(h), (rows,)).copy())
    finally:
        L.xo_reader_close(h)


class Store:
    """The 'server' side: exact key -> {w,n,z} map with FTRL/SGD push and lazy insert."""

    def __init__(self, opt=OPT_FTRL, dim=1, init=INIT_ZERO, init_const=0.0, seed=0):
        self.dim = dim
        self.opt = opt
        self.h = lib().xo_store_create(opt, dim, init, init_const, seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib().xo_store_destroy(self.h)
            self.h = None

    def set_ftrl(self, alpha, beta, l1, l2):
        lib().xo_store_set_ftrl(self.h, alpha, beta, l1, l2)

    def set_sgd(self, lr):
        lib().xo_store_set_sgd(self.h, lr)

    def __len__(self):
        return int(lib().xo_store_size(self.h))

    def reserve(self, nkeys):
        lib().xo_store_reserve(self.h, int(nkeys))

    def pull(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty(len(keys) * self.dim, dtype=np.float32)
        lib().xo_store_pull(self.h, _ptr(keys, _u64p), len(keys), _ptr(out, _f32p))
        return out.reshape(len(keys), self.dim) if self.dim > 1 else out

    def push(self, keys, grads):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        grads = np.ascontiguousarray(grads, dtype=np.float32).ravel()
        assert grads.size == len(keys) * self.dim
        lib().xo_store_push(self.h, _ptr(keys, _u64p), len(keys), _ptr(grads, _f32p))

    def export(self):
        n = len(self)
        keys = np.empty(n, dtype=np.uint64)
        w = np.empty(n * self.dim, dtype=np.float32)
        nn = np.empty(n * self.dim, dtype=np.float32)
        z = np.empty(n * self.dim, dtype=np.float32)
        lib().xo_store_export(self.h, _ptr(keys, _u64p), _ptr(w, _f32p), _ptr(nn, _f32p),
                              _ptr(z, _f32p))
        sh = (n, self.dim) if self.dim > 1 else (n,)
        return keys, w.reshape(sh), nn.reshape(sh), z.reshape(sh)

    def import_(self, keys, w, n=None, z=None):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32).ravel()
                for a in (w, n, z)]
        lib().xo_store_import(self.h, _ptr(keys, _u64p), len(keys),
                              *[None if a is None else _ptr(a, _f32p) for a in arrs])


class Batch:
    """Compiled minibatch (a3): sorted unique keys + CSR uidx + key-grouped COO."""

    def __init__(self, rowptr, keys, labels, row_begin=0, row_end=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if row_end is None:
            row_end = len(rowptr) - 1
        L = lib()
        self.h = L.xo_batch_build(_ptr(rowptr, _u64p), _ptr(keys, _u64p),
                                  _ptr(labels, _i32p), row_begin, row_end)
        self.R = int(L.xo_batch_rows(self.h))
        self.NNZ = int(L.xo_batch_nnz(self.h))
        self.U = int(L.xo_batch_nuniq(self.h))

        def arr(fn, n):
            if n == 0:
                return np.zeros(0, dtype=np.ctypeslib.as_array(fn(self.h), (1,)).dtype)
            return np.ctypeslib.as_array(fn(self.h), (n,)).copy()
        self.ukeys = arr(L.xo_batch_ukeys, self.U)
        self.rowptr = arr(L.xo_batch_rowptr, self.R + 1)
        self.uidx = arr(L.xo_batch_uidx, self.NNZ)
        self.segptr = arr(L.xo_batch_segptr, self.U + 1)
        self.coo_row = arr(L.xo_batch_coo_row, self.NNZ)
        self.labels = arr(L.xo_batch_labels, self.R)

    def __del__(self):
        if getattr(self, "h", None):
            lib().xo_batch_free(self.h)
            self.h = None

    def lr_loss(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32)
        loss = np.empty(self.R, dtype=np.float32)
        pctr = np.empty(self.R, dtype=np.float32)
        lib().xo_lr_loss(self.h, _ptr(w, _f32p), _ptr(loss, _f32p), _ptr(pctr, _f32p))
        return loss, pctr

    def lr_grad(self, loss):
        loss = np.ascontiguousarray(loss, dtype=np.float32)
        g = np.empty(self.U, dtype=np.float32)
        lib().xo_lr_grad(self.h, _ptr(loss, _f32p), _ptr(g, _f32p))
        return g

    def fm_loss(self, k, w, v):
        w = np.ascontiguousarray(w, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32).ravel()
        loss = np.empty(self.R, dtype=np.float32)
        pctr = np.empty(self.R, dtype=np.float32)
        vsum = np.empty(self.R, dtype=np.float32)
        lib().xo_fm_loss(self.h, k, _ptr(w, _f32p), _ptr(v, _f32p), _ptr(loss, _f32p),
                         _ptr(pctr, _f32p), _ptr(vsum, _f32p))
        return loss, pctr, vsum

    def fm_grad(self, k, v, vsum, loss):
        v = np.ascontiguousarray(v, dtype=np.float32).ravel()
        vsum = np.ascontiguousarray(vsum, dtype=np.float32)
        loss = np.ascontiguousarray(loss, dtype=np.float32)
        gw = np.empty(self.U, dtype=np.float32)
        gv = np.empty(self.U * k, dtype=np.float32)
        lib().xo_fm_grad(self.h, k, _ptr(v, _f32p), _ptr(vsum, _f32p), _ptr(loss, _f32p),
                         _ptr(gw, _f32p), _ptr(gv, _f32p))
        return gw, gv.reshape(self.U, k)


def lr_update(store, batch):
    lib().xo_lr_update(store.h, batch.h)


def fm_update(wstore, vstore, batch):
    lib().xo_fm_update(wstore.h, vstore.h, batch.h)


def auc_logloss(labels, pctr, acc=0.0):
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    pctr = np.ascontiguousarray(pctr, dtype=np.float32)
    ll = C.c_float(acc)
    auc = C.c_float(0)
    tp = C.c_int(0)
    fp = C.c_int(0)
    lib().xo_auc_logloss(_ptr(labels, _i32p), _ptr(pctr, _f32p), len(labels), C.byref(ll),
                         C.byref(auc), C.byref(tp), C.byref(fp))
    return ll.value, auc.value, tp.value, fp.value


def lr_update_slices_mt(store, rowptr, keys, labels, core_num):
    """the reference's slice fan-out (lr_worker.cc:186-200) on core_num threads; a timed
    baseline, not a parity path (the slices' pushes race, as in the reference)"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    return lib().xo_lr_update_slices_mt(store.h, _ptr(rowptr, _u64p), _ptr(keys, _u64p),
                                        _ptr(labels, _i32p), len(labels), core_num)


def train(model, wstore, vstore, path, epochs, block_bytes=2 << 20, core_num=1):
    r = lib().xo_train(model, wstore.h, vstore.h if vstore is not None else None, path.encode(),
                       epochs, block_bytes, core_num)
    if r < 0:
        raise RuntimeError("oracle train failed on %s" % path)
    return r


def predict(model, wstore, vstore, path, block_bytes=None, core_num=1, cap=1 << 22):
    if block_bytes is None:  # lr_worker.cc:80 (4 MiB) / fm_worker.cc:106 (2 MiB)
        block_bytes = (4 << 20) if model == 0 else (2 << 20)
    labels = np.empty(cap, dtype=np.int32)
    pctr = np.empty(cap, dtype=np.float32)
    n = lib().xo_predict(model, wstore.h, vstore.h if vstore is not None else None, path.encode(),
                         block_bytes, core_num, _ptr(labels, _i32p), _ptr(pctr, _f32p),
                         cap)
    if n < 0:
        raise RuntimeError("oracle predict failed on %s" % path)
    return labels[:n].copy(), pctr[:n].copy()


def format_auc_line(logloss, auc, tp, fp):
    """The reference's stdout line (base.h:101-108), std::cout default precision 6."""
    def g6(x):
        x = np.float32(x)
        if np.isnan(x):  # iostream prints the sign of a NaN (0 * log2(0) is negative here)
            return "-nan" if np.signbit(x) else "nan"
        return "%g" % float(x)
    if np.isnan(auc):
        return "logloss: %s\ttp_n = %d" % (g6(logloss), tp)
    return "logloss: %s\tauc = %s\ttp = %d fp = %d" % (g6(logloss), g6(auc), tp, fp)


# ------------------------------------------------------------- the real reference subset
_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libxflow_ref.so"))


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(_HERE, "_ref", "libxflow_ref.so"))
        R.ref_hash.restype = C.c_uint64
        R.ref_hash.argtypes = [C.c_char_p, C.c_size_t]
        R.ref_sigmoid.restype = C.c_float
        R.ref_sigmoid.argtypes = [C.c_float]
        R.ref_loader_open.restype = C.c_void_p
        R.ref_loader_open.argtypes = [C.c_char_p, C.c_size_t]
        R.ref_loader_close.argtypes = [C.c_void_p]
        R.ref_loader_next.restype = C.c_long
        R.ref_loader_next.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        R.ref_loader_get.argtypes = [C.c_void_p, _u64p, _u64p, _i32p, _i32p]
        R.ref_auc.argtypes = [_i32p, _f32p, C.c_size_t, _f32p, C.c_char_p, C.c_size_t]
        _ref = R
    return _ref


def ref_read_blocks(path, cap_bytes):
    R = ref()
    h = R.ref_loader_open(path.encode(), cap_bytes)
    try:
        while True:
            nnz = C.c_size_t(0)
            rows = R.ref_loader_next(h, C.byref(nnz))
            if rows <= 0:
                return
            rowptr = np.empty(rows + 1, dtype=np.uint64)
            keys = np.empty(nnz.value, dtype=np.uint64)
            fgid = np.empty(nnz.value, dtype=np.int32)
            labels = np.empty(rows, dtype=np.int32)
            R.ref_loader_get(h, _ptr(rowptr, _u64p), _ptr(keys, _u64p), _ptr(fgid, _i32p),
                             _ptr(labels, _i32p))
            yield rowptr, keys, fgid, labels
    finally:
        R.ref_loader_close(h)


def ref_auc(labels, pctr):
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    pctr = np.ascontiguousarray(pctr, dtype=np.float32)
    ll = C.c_float(0)
    buf = C.create_string_buffer(256)
    ref().ref_auc(_ptr(labels, _i32p), _ptr(pctr, _f32p), len(labels), C.byref(ll), buf, 256)
    return ll.value, buf.value.decode().rstrip("\n")

"""ctypes binding of libxflow_amd.so (include/xflow_amd.h).

The library is the product; this module only marshals pointers.  It never falls back to
a CPU implementation: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XF_LIB") or os.path.join(_HERE, "lib", "libxflow_amd.so")

XF_OK = 0
OPT_FTRL, OPT_SGD = 0, 1
INIT_ZERO, INIT_CONST, INIT_HASHNORM = 0, 1, 2
HEAVY_SEG = 64

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class XFError(RuntimeError):
    pass


class TableConfig(C.Structure):
    _fields_ = [("opt_kind", C.c_int32), ("dim", C.c_int32), ("init_kind", C.c_int32),
                ("init_const", C.c_float), ("seed", C.c_uint64), ("alpha", C.c_float),
                ("beta", C.c_float), ("lambda1", C.c_float), ("lambda2", C.c_float),
                ("lr", C.c_float), ("capacity", C.c_uint64), ("shard", C.c_uint32),
                ("nshards", C.c_uint32)]


class ShardedConfig(C.Structure):
    _fields_ = [("model", C.c_int32), ("optimizer", C.c_int32), ("k", C.c_int32),
                ("schedule", C.c_int32), ("capacity", C.c_uint64), ("seed", C.c_uint64),
                ("alpha", C.c_float), ("beta", C.c_float), ("lambda1", C.c_float),
                ("lambda2", C.c_float), ("lr", C.c_float), ("host_key_build", C.c_int32),
                ("update_rule", C.c_int32)]


class DevBatch(C.Structure):
    _fields_ = [("R", C.c_uint32), ("NNZ", C.c_uint32), ("U", C.c_uint32), ("H", C.c_uint32),
                ("rowptr", vp), ("uidx", vp), ("ukeys", vp), ("segptr", vp), ("coo_row", vp),
                ("labels", vp), ("heavy", vp), ("P", C.c_uint32), ("fwd_ntiles", C.c_uint32),
                ("pptr", vp), ("pidx", vp), ("fwd_scratch", vp), ("fwd_tile_ptr", vp),
                ("fwd_panel_first", vp), ("fwd_grid", C.c_uint32), ("pad3_", C.c_uint32),
                ("ntiles", C.c_uint32),
                ("n_heavy_chunks", C.c_uint32), ("tile_ptr", vp), ("heavy_chunk_ptr", vp),
                ("heavy_scratch", vp)]


# name -> (restype, argtypes); every symbol include/xflow_amd.h declares
SIGNATURES = {
    "xf_last_error": (C.c_char_p, []),
    "xf_version": (C.c_int, []),
    "xf_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "xf_hash_bytes": (C.c_uint64, [C.c_char_p, C.c_size_t]),
    "xf_shard_of": (C.c_uint32, [C.c_uint64, C.c_uint32]),
    "xf_hash_decimal_range": (C.c_int, [C.c_uint64, C.c_size_t, u64p]),
    "xf_hash_decimal_ids": (C.c_int, [u64p, C.c_size_t, u64p]),
    "xf_reader_open": (C.c_int, [C.POINTER(vp), C.c_char_p, C.c_size_t]),
    "xf_reader_open_cached": (C.c_int, [C.POINTER(vp), C.c_char_p, C.c_size_t, C.c_char_p,
                                        C.POINTER(C.c_int)]),
    "xf_reader_close": (C.c_int, [vp]),
    "xf_reader_mapped": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "xf_block_create": (C.c_int, [C.POINTER(vp)]),
    "xf_block_destroy": (C.c_int, [vp]),
    "xf_reader_next_into": (C.c_int, [vp, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                      C.POINTER(u64p), C.POINTER(u64p), C.POINTER(i32p),
                                      C.POINTER(i32p)]),
    "xf_reader_peek_text": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "xf_reader_skip_text": (C.c_int, [vp]),
    "xf_reader_copy_text": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_int]),
    "xf_reader_parse_text": (C.c_int, [vp, vp, C.c_size_t, vp, C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_size_t), C.POINTER(u64p), C.POINTER(u64p),
                                       C.POINTER(i32p), C.POINTER(i32p)]),
    "xf_copy_to_host": (C.c_int, [vp, vp, C.c_size_t]),
    "xf_ingest_create": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "xf_ingest_destroy": (C.c_int, [vp]),
    "xf_ingest_staging": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "xf_ingest_upload": (C.c_int, [vp, C.c_size_t, vp]),
    "xf_ingest_block": (C.c_int, [vp, vp, C.c_size_t, vp, C.POINTER(vp), C.POINTER(vp),
                                  C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_int)]),
    "xf_reader_next": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                 C.POINTER(u64p), C.POINTER(u64p), C.POINTER(i32p),
                                 C.POINTER(i32p)]),
    "xf_scratch_reserve": (C.c_int, [C.c_size_t]),
    "xf_batch_pool_reserve": (C.c_int, [C.c_size_t]),
    "xf_batch_compile": (C.c_int, [C.POINTER(vp), u64p, u64p, i32p, C.c_size_t, C.c_size_t]),
    "xf_batch_free": (C.c_int, [vp]),
    "xf_batch_compile_dev": (C.c_int, [C.POINTER(vp), vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "xf_batch_compile_gpu": (C.c_int, [C.POINTER(vp), u64p, u64p, i32p, C.c_size_t,
                                       C.c_size_t, vp]),
    "xf_batch_compile_fm_dev": (C.c_int, [C.POINTER(vp), vp, vp, vp, vp, vp, C.c_uint32,
                                          C.c_uint32, vp, C.POINTER(C.c_int)]),
    "xf_batch_compile_fm": (C.c_int, [C.POINTER(vp), vp, vp, vp, vp, vp, C.c_size_t,
                                      C.c_size_t, vp, C.POINTER(C.c_int)]),
    "xf_batch_download": (C.c_int, [vp]),
    "xf_batch_compile_local_dev": (C.c_int, [C.POINTER(vp), vp, vp, vp, vp, C.c_uint32,
                                             C.c_uint32, C.c_int, vp]),
    "xf_batch_compile_local": (C.c_int, [C.POINTER(vp), vp, u64p, u64p, i32p, C.c_size_t,
                                         C.c_size_t, C.c_int, vp]),
    "xf_batch_cells_info": (C.c_int, [vp, u32p]),
    "xf_source_hash": (C.c_char_p, []),
    "xf_workspace_capture": (C.c_int, [vp, C.c_int]),
    "xf_workspace_parity": (C.c_int, [vp, C.c_int]),
    "xf_batch_dims": (C.c_int, [vp, u32p, u32p, u32p, u32p]),
    "xf_batch_host": (C.c_int, [vp, C.POINTER(u64p), C.POINTER(u32p), C.POINTER(u32p),
                                C.POINTER(u32p), C.POINTER(u32p), C.POINTER(i32p),
                                C.POINTER(u32p)]),
    "xf_batch_panels": (C.c_int, [vp, u32p, C.POINTER(u32p), C.POINTER(u32p)]),
    "xf_tune": (C.c_int, [C.c_char_p, C.c_double]),
    "xf_batch_tiles": (C.c_int, [vp, u32p, C.POINTER(u32p)]),
    "xf_batch_heavy_chunks": (C.c_int, [vp, u32p, C.POINTER(u32p)]),
    "xf_batch_fwd_tiles": (C.c_int, [vp, u32p, C.POINTER(u32p), C.POINTER(u32p), u32p]),
    "xf_batch_upload": (C.c_int, [vp, vp]),
    "xf_batch_dev_view": (C.c_int, [vp, C.POINTER(DevBatch)]),
    "xf_table_config_default": (None, [C.POINTER(TableConfig)]),
    "xf_table_create": (C.c_int, [C.POINTER(vp), C.POINTER(TableConfig)]),
    "xf_table_destroy": (C.c_int, [vp]),
    "xf_table_size": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "xf_table_settled": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "xf_table_prepare_defrag": (C.c_int, [vp]),
    "xf_table_capacity": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "xf_table_reserve": (C.c_int, [vp, C.c_uint64]),
    "xf_table_defrag": (C.c_int, [vp]),
    "xf_table_set_hyper": (C.c_int, [vp] + [C.c_float] * 5),
    "xf_table_pull": (C.c_int, [vp, u64p, C.c_size_t, f32p]),
    "xf_table_push": (C.c_int, [vp, u64p, C.c_size_t, f32p]),
    "xf_table_resolve_dev": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
    "xf_table_gather_dev": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
    "xf_table_update_dev": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
    "xf_table_pull_dev": (C.c_int, [vp, vp, C.c_size_t, vp, vp, vp]),
    "xf_table_pull_ordered_dev": (C.c_int, [vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "xf_table_update_merged_dev": (C.c_int, [vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "xf_lr_grad_update_dev": (C.c_int, [vp, C.POINTER(DevBatch), vp, vp, vp, vp, vp]),
    "xf_table_check": (C.c_int, [vp, vp]),
    "xf_table_export": (C.c_int, [vp, u64p, f32p, f32p, f32p, C.c_size_t,
                                  C.POINTER(C.c_size_t)]),
    "xf_table_import": (C.c_int, [vp, u64p, C.c_size_t, f32p, f32p, f32p]),
    "xf_table_w_derived": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "xf_lr_forward_dev": (C.c_int, [C.POINTER(DevBatch), vp, vp, vp, vp]),
    "xf_lr_grad_dev": (C.c_int, [C.POINTER(DevBatch), vp, vp, vp]),
    "xf_fm_forward_dev": (C.c_int, [C.POINTER(DevBatch), C.c_int, vp, vp, vp, vp, vp, vp]),
    "xf_fm_grad_dev": (C.c_int, [C.POINTER(DevBatch), C.c_int, vp, vp, vp, vp, vp, vp]),
    "xf_fm_grad_update_dev": (C.c_int, [vp, vp, C.POINTER(DevBatch)] + [vp] * 9),
    "xf_workspace_create": (C.c_int, [C.POINTER(vp)]),
    "xf_workspace_destroy": (C.c_int, [vp]),
    "xf_lr_step": (C.c_int, [vp, vp, vp, vp]),
    "xf_lr_update_dev": (C.c_int, [C.POINTER(vp), vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int,
                                   vp, vp]),
    "xf_fm_step": (C.c_int, [vp, vp, vp, vp, vp]),
    "xf_lr_predict": (C.c_int, [vp, vp, vp, f32p]),
    "xf_fm_predict": (C.c_int, [vp, vp, vp, vp, f32p]),
    "xf_workspace_fetch": (C.c_int, [vp, f32p, f32p, f32p, C.c_size_t, C.c_size_t]),
    "xf_workspace_profile": (C.c_int, [vp, C.c_int]),
    "xf_workspace_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "xf_stream_sync": (C.c_int, [vp]),
    "xf_calib_stream": (C.c_int, [C.c_int, C.c_size_t, C.c_int]),
    "xf_group_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                  C.c_int]),
    "xf_group_destroy": (C.c_int, [vp]),
    "xf_group_abort": (C.c_int, [vp]),
    "xf_group_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "xf_group_barrier": (C.c_int, [vp]),
    "xf_group_allgather_host": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "xf_group_gatherv_host": (C.c_int, [vp, vp, C.c_size_t, vp, u64p]),
    "xf_group_alltoallv": (C.c_int, [vp, vp, u64p, vp, u64p, C.c_size_t, C.c_int, vp]),
    "xf_group_alltoallv_ch": (C.c_int, [vp, C.c_int, vp, u64p, vp, u64p, C.c_size_t, C.c_int,
                                        vp]),
    "xf_group_selftest": (C.c_int, [vp, C.c_size_t]),
    "xf_kb_debug_read": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_size_t, u32p]),
    "xf_sort_key_pos": (C.c_int, [vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp,
                                  C.POINTER(C.c_int)]),
    "xf_sharded_config_default": (None, [C.POINTER(ShardedConfig)]),
    "xf_sharded_create": (C.c_int, [C.POINTER(vp), vp, C.POINTER(ShardedConfig)]),
    "xf_sharded_destroy": (C.c_int, [vp]),
    "xf_sharded_compile": (C.c_int, [vp, C.POINTER(vp), u64p, u64p, i32p, C.c_size_t,
                                     C.c_size_t, C.c_int]),
    "xf_sharded_compile_dev": (C.c_int, [vp, C.POINTER(vp), vp, vp, vp, C.c_uint32, C.c_uint32,
                                         C.c_int]),
    "xf_sbatch_free": (C.c_int, [vp]),
    "xf_sbatch_dims": (C.c_int, [vp, u32p, u32p, u32p, u64p]),
    "xf_sbatch_fm_keyed": (C.c_int, [vp]),
    "xf_sharded_step": (C.c_int, [vp, vp]),
    "xf_sharded_predict": (C.c_int, [vp, vp, f32p]),
    "xf_sharded_flush": (C.c_int, [vp]),
    "xf_sharded_defrag": (C.c_int, [vp]),
    "xf_sharded_check": (C.c_int, [vp]),
    "xf_sharded_set_schedule": (C.c_int, [vp, C.c_int]),
    "xf_sharded_set_parity": (C.c_int, [vp, C.c_int]),
    "xf_sharded_tables": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "xf_sharded_stream": (C.c_int, [vp, C.POINTER(vp)]),
    "xf_sharded_profile": (C.c_int, [vp, C.c_int]),
    "xf_sharded_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "xf_sharded_save": (C.c_int, [vp, C.c_char_p]),
    "xf_sharded_load": (C.c_int, [vp, C.c_char_p]),
    "xf_auc_logloss": (C.c_int, [i32p, f32p, C.c_size_t, f32p, f32p, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "XFCreate": (C.c_int, [C.POINTER(vp), C.c_char_p, C.c_char_p]),
    "XFStartTrain": (C.c_int, [C.POINTER(vp)]),
    "XFDestroy": (C.c_int, [C.POINTER(vp)]),
    "XFSetParam": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
    "XFGetMetric": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double)]),
    "XFGetTables": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "XFSaveModel": (C.c_int, [vp, C.c_char_p]),
    "XFLoadModel": (C.c_int, [vp, C.c_char_p]),
    "XFPredict": (C.c_int, [vp]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The torch wheel bundles its own libamdhip64.so.7; this
    library links the one under /opt/rocm.  Both have the same SONAME, so whichever is mapped
    first serves everybody — and torch finds no GPU when it is handed /opt/rocm's.  When torch
    is installed (it is only plumbing here: device buffers and process groups of the multi-GPU
    driver and of some tests) map ITS runtime first, whether or not torch gets imported later."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        try:
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the shared library; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XFError("libxflow_amd.so is not built: run `python -m xflow_amd.build` "
                          "(there is no CPU fallback)")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        # the library says which sources it was built from: a prebuilt .so that is older than
        # the sources next to it (file times do not survive a snapshot) is an error, not a
        # silent stand-in.  XF_LIB (an experiment's variant build) is taken as it is.
        if "XF_LIB" not in os.environ:
            from . import build as _build
            built, want = L.xf_source_hash().decode(), _build.source_hash()
            if built != want:
                raise XFError("libxflow_amd.so was built from other sources (library %s, "
                              "sources %s): run `python -m xflow_amd.build`" % (built, want))
        _lib = L
    return _lib


def check(rc):
    if rc != XF_OK:
        raise XFError("xflow_amd error %d: %s" % (rc, lib().xf_last_error().decode()))


def require_gpu():
    n = C.c_int(0)
    check(lib().xf_device_count(C.byref(n)))
    if n.value < 1:
        raise XFError("no HIP device visible: the xflow_amd hot path runs only on the GPU")
    return n.value


def _p(a, t):
    return a.ctypes.data_as(t)


def hash_str(s):
    b = s if isinstance(s, bytes) else str(s).encode()
    return int(lib().xf_hash_bytes(b, len(b)))


def tune(name, value):
    check(lib().xf_tune(name.encode(), float(value)))


def hash_decimal_range(start, n):
    out = np.empty(n, dtype=np.uint64)
    check(lib().xf_hash_decimal_range(start, n, _p(out, u64p)))
    return out


def sort_key_pos(keys, lo=0, span=2**64 - 1, repeat=0):
    """xf_sort_key_pos on a host array (through torch: plumbing): (sorted keys, their positions,
    by_hand[, ms per call over `repeat` calls])"""
    import torch
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = len(keys)
    dk = torch.from_numpy(keys.view(np.int64).copy()).cuda() if n else torch.zeros(1, dtype=torch.int64).cuda()
    sk = torch.empty(max(n, 1), dtype=torch.int64, device="cuda")
    sp = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    h = C.c_int(0)
    args = (dk.data_ptr(), n, lo, span, sk.data_ptr(), sp.data_ptr(), None, C.byref(h))
    check(lib().xf_sort_key_pos(*args))
    ms = None
    if repeat:
        import time
        check(lib().xf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(repeat):
            check(lib().xf_sort_key_pos(*args))
        check(lib().xf_stream_sync(None))
        ms = (time.perf_counter() - t0) * 1e3 / repeat
    out = (sk.cpu().numpy().view(np.uint64)[:n], sp.cpu().numpy().view(np.uint32)[:n], bool(h.value))
    return out + (ms,) if repeat else out


def hash_decimal_ids(ids):
    """hash of the decimal string of every id (io.h:53 on str(id))"""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    out = np.empty(len(ids), dtype=np.uint64)
    check(lib().xf_hash_decimal_ids(_p(ids, u64p), len(ids), _p(out, u64p)))
    return out


def read_blocks(path, cap_bytes, cache_path=None, info=None):
    """Yield (rowptr, keys, fgid, labels) numpy copies per text block.  With `cache_path` the
    blocks come from / go to the binarized block cache; info["from_cache"] says which."""
    L = lib()
    h = vp()
    if cache_path is None:
        check(L.xf_reader_open(C.byref(h), path.encode(), cap_bytes))
    else:
        hit = C.c_int(0)
        check(L.xf_reader_open_cached(C.byref(h), path.encode(), cap_bytes, cache_path.encode(),
                                      C.byref(hit)))
        if info is not None:
            info["from_cache"] = bool(hit.value)
    try:
        while True:
            rows, nnz = C.c_size_t(0), C.c_size_t(0)
            rp, ks, fg, lb = u64p(), u64p(), i32p(), i32p()
            check(L.xf_reader_next(h, C.byref(rows), C.byref(nnz), C.byref(rp), C.byref(ks),
                                   C.byref(fg), C.byref(lb)))
            if rows.value == 0:
                return
            r, n = rows.value, nnz.value
            yield (np.ctypeslib.as_array(rp, (r + 1,)).copy(),
                   np.ctypeslib.as_array(ks, (n,)).copy() if n else np.zeros(0, np.uint64),
                   np.ctypeslib.as_array(fg, (n,)).copy() if n else np.zeros(0, np.int32),
                   np.ctypeslib.as_array(lb, (r,)).copy())
    finally:
        L.xf_reader_close(h)


def read_text_blocks(path, cap_bytes):
    """Yield every block of `path` as TEXT (bytes), by the reader's block rule
    (xf_reader_peek_text / xf_reader_skip_text)."""
    L = lib()
    r = vp()
    check(L.xf_reader_open(C.byref(r), path.encode(), cap_bytes))
    try:
        while True:
            t, n = vp(), C.c_size_t()
            check(L.xf_reader_peek_text(r, C.byref(t), C.byref(n)))
            if n.value == 0:
                return
            yield C.string_at(t.value, n.value)
            check(L.xf_reader_skip_text(r))
    finally:
        L.xf_reader_close(r)


def parse_text_block(text, cap_bytes=1 << 26):
    """(rowptr, keys, fgid, labels) of one block's text through the HOST parser
    (xf_reader_parse_text); raises XFError where the host parser rejects the block."""
    import tempfile
    L = lib()
    r, blk = vp(), vp()
    with tempfile.NamedTemporaryFile() as f:     # (a reader to borrow the thread team from)
        f.write(b"0\t0:0:0\n")
        f.flush()
        check(L.xf_reader_open(C.byref(r), f.name.encode(), cap_bytes))
    check(L.xf_block_create(C.byref(blk)))
    try:
        rows, nnz = C.c_size_t(), C.c_size_t()
        rp, ks, fg, lb = u64p(), u64p(), i32p(), i32p()
        buf = C.create_string_buffer(text, len(text) + 16)
        check(L.xf_reader_parse_text(r, buf, len(text), blk, C.byref(rows), C.byref(nnz),
                                     C.byref(rp), C.byref(ks), C.byref(fg), C.byref(lb)))
        R, N = rows.value, nnz.value
        return (np.ctypeslib.as_array(rp, (R + 1,)).copy(),
                np.ctypeslib.as_array(ks, (max(N, 1),))[:N].copy(),
                np.ctypeslib.as_array(fg, (max(N, 1),))[:N].copy(),
                np.ctypeslib.as_array(lb, (max(R, 1),))[:R].copy())
    finally:
        L.xf_block_destroy(blk)
        L.xf_reader_close(r)


class Ingest:
    """The GPU tokeniser (xf_ingest_*): text block -> device arrays (keys, rowptr, labels)."""

    def __init__(self, max_text_bytes=1 << 26):
        self.h = vp()
        check(lib().xf_ingest_create(C.byref(self.h), max_text_bytes))

    def __del__(self):
        if getattr(self, "h", None):
            lib().xf_ingest_destroy(self.h)
            self.h = None

    def block_dev(self, text, stream=None):
        """(ok, rows, nnz, d_keys, d_rowptr, d_labels): device pointers owned by the object"""
        dk, dr, dl = vp(), vp(), vp()
        rows, nnz, ok = C.c_uint32(), C.c_uint32(), C.c_int()
        buf = C.create_string_buffer(text, len(text) + 1)
        check(lib().xf_ingest_block(self.h, buf, len(text), stream, C.byref(dk), C.byref(dr),
                                    C.byref(dl), C.byref(rows), C.byref(nnz), C.byref(ok)))
        return bool(ok.value), rows.value, nnz.value, dk.value, dr.value, dl.value

    def block(self, text):
        """(ok, rowptr, keys, labels) as numpy arrays (copied back from the device); the arrays
        are None when the block is not of the common shape"""
        ok, R, N, dk, dr, dl = self.block_dev(text)
        if not ok:
            return False, None, None, None
        rp, ks, lb = np.zeros(R + 1, np.uint32), np.zeros(N, np.uint64), np.zeros(R, np.int32)
        for a, ptr in ((rp, dr), (ks, dk), (lb, dl)):
            if a.nbytes:
                check(lib().xf_copy_to_host(a.ctypes.data, ptr, a.nbytes))
        return True, rp, ks, lb


class Batch:
    """A compiled minibatch (host arrays; device mirror after upload())."""

    def __init__(self, rowptr, keys, labels, row_begin=0, row_end=None, on_gpu=False):
        """on_gpu=False: host key build (xf_batch_compile); True: the GPU one
        (xf_batch_compile_gpu), the batch is then already device-resident."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if row_end is None:
            row_end = len(rowptr) - 1
        self.h = vp()
        if on_gpu:
            check(lib().xf_batch_compile_gpu(C.byref(self.h), _p(rowptr, u64p), _p(keys, u64p),
                                             _p(labels, i32p), row_begin, row_end, None))
        else:
            check(lib().xf_batch_compile(C.byref(self.h), _p(rowptr, u64p), _p(keys, u64p),
                                         _p(labels, i32p), row_begin, row_end))
        R, N, U, H = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().xf_batch_dims(self.h, C.byref(R), C.byref(N), C.byref(U), C.byref(H)))
        self.R, self.NNZ, self.U, self.H = R.value, N.value, U.value, H.value
        self.on_gpu = on_gpu

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xf_batch_free(self.h)
                self.h = None
        except Exception:      # interpreter shutdown: the module globals may be gone
            pass

    def host(self):
        uk, rp, ui, sp, cr, hv = u64p(), u32p(), u32p(), u32p(), u32p(), u32p()
        lb = i32p()
        check(lib().xf_batch_host(self.h, C.byref(uk), C.byref(rp), C.byref(ui), C.byref(sp),
                                  C.byref(cr), C.byref(lb), C.byref(hv)))

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, (n,)).copy() if n else np.zeros(0, dt)
        return dict(ukeys=arr(uk, self.U, np.uint64), rowptr=arr(rp, self.R + 1, np.uint32),
                    uidx=arr(ui, self.NNZ, np.uint32), segptr=arr(sp, self.U + 1, np.uint32),
                    coo_row=arr(cr, self.NNZ, np.uint32), labels=arr(lb, self.R, np.int32),
                    heavy=arr(hv, self.H, np.uint32))

    def panels(self):
        P, pp, pi = C.c_uint32(0), u32p(), u32p()
        check(lib().xf_batch_panels(self.h, C.byref(P), C.byref(pp), C.byref(pi)))
        if P.value == 0:
            return 0, np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        return (P.value, np.ctypeslib.as_array(pp, (P.value * (self.R + 1),)).copy(),
                np.ctypeslib.as_array(pi, (self.NNZ,)).copy())

    def fwd_tiles(self):
        """(tile_ptr[ntiles+1], panel_first[P+1], grid) of the forward tiles"""
        n, tp, pf, g = C.c_uint32(0), u32p(), u32p(), C.c_uint32(0)
        check(lib().xf_batch_fwd_tiles(self.h, C.byref(n), C.byref(tp), C.byref(pf),
                                       C.byref(g)))
        if n.value == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0
        P = self.panels()[0]
        return (np.ctypeslib.as_array(tp, (n.value + 1,)).copy(),
                np.ctypeslib.as_array(pf, (P + 1,)).copy(), g.value)

    def heavy_chunks(self):
        n, cp = C.c_uint32(0), u32p()
        check(lib().xf_batch_heavy_chunks(self.h, C.byref(n), C.byref(cp)))
        return np.ctypeslib.as_array(cp, (n.value + 1,)).copy()

    def tiles(self):
        n, tp = C.c_uint32(0), u32p()
        check(lib().xf_batch_tiles(self.h, C.byref(n), C.byref(tp)))
        return np.ctypeslib.as_array(tp, (n.value + 1,)).copy()

    def upload(self, stream=None):
        check(lib().xf_batch_upload(self.h, stream))
        return self

    def dev_view(self):
        if not self.on_gpu:
            self.upload()
        v = DevBatch()
        check(lib().xf_batch_dev_view(self.h, C.byref(v)))
        return v


class FmBatch(Batch):
    """xf_batch_compile_fm_dev: the FM key build against the (w, v) tables' settled tiers
    (`keyed` says whether the fast build ran; else it is the sort-based one).  The raw arrays go
    to the device through torch (plumbing)."""

    def __init__(self, wt, vt, rowptr, keys, labels):
        import torch
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        rp = np.ascontiguousarray(rowptr, dtype=np.uint64)
        rp32 = (rp - rp[0]).astype(np.uint32)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        R, NNZ = len(rp32) - 1, int(rp32[-1])
        dk = torch.from_numpy(keys[int(rp[0]):int(rp[0]) + NNZ].view(np.int64).copy()).cuda()
        dr = torch.from_numpy(rp32.view(np.int32)).cuda()
        dl = torch.from_numpy(labels[:max(R, 1)] if R else np.zeros(1, np.int32)).cuda()
        torch.cuda.synchronize()
        self.h = vp()
        keyed = C.c_int(0)
        check(lib().xf_batch_compile_fm_dev(C.byref(self.h), wt.h, vt.h, dk.data_ptr(),
                                            dr.data_ptr(), dl.data_ptr(), R, NNZ, None,
                                            C.byref(keyed)))
        self.keyed = bool(keyed.value)
        Rr, N, U, H = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().xf_batch_dims(self.h, C.byref(Rr), C.byref(N), C.byref(U), C.byref(H)))
        self.R, self.NNZ, self.U, self.H = Rr.value, N.value, U.value, H.value
        self.on_gpu = True


class LocalBatch:
    """A minibatch compiled straight against one table on this GPU (xf_batch_compile_local):
    raw keys -> state rows -> cells; no key list.  Feeds lr_step / lr_predict on that table."""

    def __init__(self, table, rowptr, keys, labels, row_begin=0, row_end=None, retain_keys=True):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if row_end is None:
            row_end = len(rowptr) - 1
        self.h = vp()
        check(lib().xf_batch_compile_local(C.byref(self.h), table.h, _p(rowptr, u64p),
                                           _p(keys, u64p), _p(labels, i32p), row_begin, row_end,
                                           1 if retain_keys else 0, None))
        R, N = C.c_uint32(), C.c_uint32()
        check(lib().xf_batch_dims(self.h, C.byref(R), C.byref(N), None, None))
        self.R, self.NNZ, self.U, self.H = R.value, N.value, 0, 0

    @classmethod
    def update(cls, table, ws, rowptr, keys, labels, retain_keys=True):
        """xf_lr_update_dev: the key build and the step of a fresh minibatch in one call (the
        raw arrays go to the GPU through torch first); returns the compiled minibatch"""
        import torch
        rp = np.ascontiguousarray(rowptr, dtype=np.uint64)
        rp32 = (rp - rp[0]).astype(np.uint32)
        kk = np.ascontiguousarray(keys, dtype=np.uint64)[int(rp[0]):int(rp[-1])]
        lb = np.ascontiguousarray(labels, dtype=np.int32)
        d_rp = torch.from_numpy(rp32.view(np.int32)).cuda()
        d_kk = torch.from_numpy(kk.view(np.int64).copy()).cuda() if len(kk) else None
        d_lb = torch.from_numpy(lb.copy()).cuda() if len(lb) else None
        torch.cuda.synchronize()
        self = cls.__new__(cls)
        self.h = vp()
        check(lib().xf_lr_update_dev(C.byref(self.h), table.h,
                                     d_kk.data_ptr() if d_kk is not None else None,
                                     d_rp.data_ptr(),
                                     d_lb.data_ptr() if d_lb is not None else None,
                                     len(rp32) - 1, len(kk), 1 if retain_keys else 0, ws.h, None))
        check(lib().xf_stream_sync(None))     # (the torch arrays go away with this frame)
        R, N = C.c_uint32(), C.c_uint32()
        check(lib().xf_batch_dims(self.h, C.byref(R), C.byref(N), None, None))
        self.R, self.NNZ, self.U, self.H = R.value, N.value, 0, 0
        return self

    def cells_info(self):
        return cells_info(self)

    def upload(self, stream=None):
        return self

    __del__ = Batch.__del__


def cells_info(batch):
    out = (C.c_uint32 * 8)()
    check(lib().xf_batch_cells_info(batch.h, out))
    names = ["W", "nwin", "nchunk", "G", "nitems", "segments", "nsplit_chunks", "M"]
    return dict(zip(names, list(out)))


class Table:
    """One GPU's shard of the parameter table (ps-lite server replacement)."""

    def __init__(self, opt=OPT_FTRL, dim=1, init=INIT_ZERO, init_const=0.0, seed=0,
                 capacity=1 << 20, shard=0, nshards=1, **hyper):
        require_gpu()
        c = TableConfig()
        lib().xf_table_config_default(C.byref(c))
        c.opt_kind, c.dim, c.init_kind, c.init_const, c.seed = opt, dim, init, init_const, seed
        c.capacity, c.shard, c.nshards = capacity, shard, nshards
        for k, v in hyper.items():
            setattr(c, k, v)
        self.dim, self.opt = dim, opt
        self.h = vp()
        check(lib().xf_table_create(C.byref(self.h), C.byref(c)))

    @classmethod
    def from_handle(cls, h, dim, opt=OPT_FTRL):
        """Non-owning view of a table owned by a worker (XFGetTables)."""
        t = cls.__new__(cls)
        t.h, t.dim, t.opt, t._borrowed = h, dim, opt, True
        return t

    def __del__(self):
        try:
            if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
                lib().xf_table_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        n = C.c_uint64(0)
        check(lib().xf_table_size(self.h, C.byref(n)))
        return n.value

    @property
    def settled(self):
        n = C.c_uint64(0)
        check(lib().xf_table_settled(self.h, C.byref(n)))
        return n.value

    @property
    def capacity(self):
        n = C.c_uint64(0)
        check(lib().xf_table_capacity(self.h, C.byref(n)))
        return n.value

    def reserve(self, capacity):
        check(lib().xf_table_reserve(self.h, capacity))

    def defrag(self):
        check(lib().xf_table_defrag(self.h))

    def pull(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty(len(keys) * self.dim, dtype=np.float32)
        check(lib().xf_table_pull(self.h, _p(keys, u64p), len(keys), _p(out, f32p)))
        return out.reshape(len(keys), self.dim) if self.dim > 1 else out

    def push(self, keys, grads):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        grads = np.ascontiguousarray(grads, dtype=np.float32).ravel()
        assert grads.size == len(keys) * self.dim
        check(lib().xf_table_push(self.h, _p(keys, u64p), len(keys), _p(grads, f32p)))

    def export(self):
        n = C.c_size_t(0)
        check(lib().xf_table_export(self.h, None, None, None, None, 0, C.byref(n)))
        m = n.value
        keys = np.empty(m, dtype=np.uint64)
        w = np.empty(m * self.dim, dtype=np.float32)
        nn = np.empty(m * self.dim, dtype=np.float32)
        z = np.empty(m * self.dim, dtype=np.float32)
        if m:
            check(lib().xf_table_export(self.h, _p(keys, u64p), _p(w, f32p), _p(nn, f32p),
                                        _p(z, f32p), m, C.byref(n)))
        sh = (m, self.dim) if self.dim > 1 else (m,)
        return keys, w.reshape(sh), nn.reshape(sh), z.reshape(sh)

    def import_(self, keys, w, n=None, z=None):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32).ravel()
                for a in (w, n, z)]
        check(lib().xf_table_import(self.h, _p(keys, u64p), len(keys),
                                    *[None if a is None else _p(a, f32p) for a in arrs]))

    def set_hyper(self, alpha, beta, l1, l2, lr=0.001):
        check(lib().xf_table_set_hyper(self.h, alpha, beta, l1, l2, lr))

    def w_derived(self):
        """the gradient + Push kernels derive the old weight from (n, z) (xf_table_w_derived)"""
        yes = C.c_int(0)
        check(lib().xf_table_w_derived(self.h, C.byref(yes)))
        return bool(yes.value)

    # device-pointer API (ints = raw device addresses, e.g. torch.Tensor.data_ptr())
    def resolve_dev(self, d_keys, n, d_slots, stream=None):
        check(lib().xf_table_resolve_dev(self.h, d_keys, n, d_slots, stream))

    def gather_dev(self, d_slots, n, d_vals, stream=None):
        check(lib().xf_table_gather_dev(self.h, d_slots, n, d_vals, stream))

    def update_dev(self, d_slots, n, d_grads, stream=None):
        check(lib().xf_table_update_dev(self.h, d_slots, n, d_grads, stream))

    def check(self, stream=None):
        check(lib().xf_table_check(self.h, stream))


class Workspace:
    def __init__(self, capture=False):
        """capture=True: LR steps also keep the pulled weights and gradients per unique key for
        fetch() (the parity hook; the production step never forms them)."""
        self.h = vp()
        check(lib().xf_workspace_create(C.byref(self.h)))
        if capture:
            check(lib().xf_workspace_capture(self.h, 1))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xf_workspace_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def parity(self, mode):
        """'exact' (default) or 'reference_order' (fp32 running row sums in the reference's
        order: slow, for checking)"""
        check(lib().xf_workspace_parity(self.h, {"exact": 0, "reference_order": 1}[mode]))

    def fetch(self, U, R):
        wu = np.empty(U, np.float32)
        loss = np.empty(R, np.float32)
        g = np.empty(U, np.float32)
        check(lib().xf_workspace_fetch(self.h, _p(wu, f32p), _p(loss, f32p), _p(g, f32p), U, R))
        return wu, loss, g

    def fetch_loss(self, R):
        loss = np.empty(R, np.float32)
        check(lib().xf_workspace_fetch(self.h, None, _p(loss, f32p), None, 0, R))
        return loss

    def profile(self, enable):
        check(lib().xf_workspace_profile(self.h, 1 if enable else 0))

    def profile_read(self):
        ms = (C.c_double * 5)()
        steps = C.c_long(0)
        check(lib().xf_workspace_profile_read(self.h, ms, C.byref(steps)))
        names = ["resolve", "gather", "forward", "gradient", "update"]
        return {k: ms[i] for i, k in enumerate(names)}, steps.value


def lr_step(table, batch, ws, stream=None):
    check(lib().xf_lr_step(table.h, batch.h, ws.h, stream))


def fm_step(wt, vt, batch, ws, stream=None):
    check(lib().xf_fm_step(wt.h, vt.h, batch.h, ws.h, stream))


def lr_predict(table, batch, ws):
    out = np.empty(batch.R, np.float32)
    check(lib().xf_lr_predict(table.h, batch.h, ws.h, _p(out, f32p)))
    return out


def fm_predict(wt, vt, batch, ws):
    out = np.empty(batch.R, np.float32)
    check(lib().xf_fm_predict(wt.h, vt.h, batch.h, ws.h, _p(out, f32p)))
    return out


def stream_sync(stream=None):
    check(lib().xf_stream_sync(stream))


def auc_logloss(labels, pctr, acc=0.0):
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    pctr = np.ascontiguousarray(pctr, dtype=np.float32)
    ll, auc = C.c_float(acc), C.c_float(0)
    tp, fp = C.c_int(0), C.c_int(0)
    nat = C.c_double(0)
    check(lib().xf_auc_logloss(_p(labels, i32p), _p(pctr, f32p), len(labels), C.byref(ll),
                               C.byref(auc), C.byref(tp), C.byref(fp), C.byref(nat)))
    return ll.value, auc.value, tp.value, fp.value, nat.value


TRANSPORT_RCCL, TRANSPORT_HOST, TRANSPORT_AUTO = 0, 1, 2


class Group:
    """The process group of a multi-GPU run (xf_group_*): TCP bootstrap around rank 0, RCCL
    (or, for tests, host-staged) all-to-all-v."""

    def __init__(self, rank=-1, world=0, addr=None, port=0, transport=TRANSPORT_RCCL, device=-1):
        """device: >= 0 a GPU index, -1 the current device, -2 by rank (LOCAL_RANK, else rank
        modulo the visible devices)"""
        self.h = vp()
        check(lib().xf_group_create(C.byref(self.h), rank, world,
                                    addr.encode() if addr else None, port, transport, device))
        r, w, t = C.c_int(), C.c_int(), C.c_int()
        check(lib().xf_group_info(self.h, C.byref(r), C.byref(w), C.byref(t)))
        self.rank, self.world, self.transport = r.value, w.value, t.value

    def close(self):
        if getattr(self, "h", None):
            lib().xf_group_destroy(self.h)
            self.h = None

    __del__ = close

    def barrier(self):
        check(lib().xf_group_barrier(self.h))

    def allgather(self, arr):
        """every rank's (same-shaped) numpy array, stacked along a new first axis"""
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, a.dtype)
        check(lib().xf_group_allgather_host(self.h, a.ctypes.data, a.nbytes, out.ctypes.data))
        return out

    def gatherv(self, arr):
        """rank 0 gets the concatenation of every rank's 1-d array, the others None"""
        a = np.ascontiguousarray(arr)
        sizes = self.allgather(np.array([a.nbytes], np.uint64)).ravel()
        out = np.empty(int(sizes.sum()) // a.itemsize, a.dtype) if self.rank == 0 else None
        check(lib().xf_group_gatherv_host(self.h, a.ctypes.data, a.nbytes,
                                          out.ctypes.data if out is not None else None,
                                          _p(sizes, u64p)))
        return out

    def alltoallv_host(self, send, send_counts):
        """host arrays through the group (any transport that can move host memory)"""
        a = np.ascontiguousarray(send)
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rc = np.ascontiguousarray(self.allgather(sc)[:, self.rank])
        out = np.empty(int(rc.sum()), a.dtype)
        check(lib().xf_group_alltoallv(self.h, a.ctypes.data, _p(sc, u64p), out.ctypes.data,
                                       _p(rc, u64p), a.itemsize, 1, None))
        return out, rc

    def alltoallv_dev(self, d_send, send_counts, d_recv, recv_counts, elem_bytes, stream=None,
                      channel=0):
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        check(lib().xf_group_alltoallv_ch(self.h, channel, d_send, _p(sc, u64p), d_recv,
                                          _p(rc, u64p), elem_bytes, 0, stream))

    def selftest(self, nbytes=1 << 20):
        """COLLECTIVE: both RCCL communicators driven at once from two streams"""
        check(lib().xf_group_selftest(self.h, nbytes))


SCHEDULE_SEQUENTIAL, SCHEDULE_STALE1, SCHEDULE_OWNER, SCHEDULE_OWNER_STALE1 = 0, 1, 2, 3
_SCHEDULES = {"sequential": SCHEDULE_SEQUENTIAL, "stale1": SCHEDULE_STALE1,
              "owner": SCHEDULE_OWNER, "owner_stale1": SCHEDULE_OWNER_STALE1}


class ShardedBatch:
    def __init__(self, h, owner=None):
        self.h = h
        # xf_sbatch_free looks at the trainer that compiled the minibatch (an outstanding
        # stale1 Push): the trainer must outlive its minibatches
        self._owner = owner
        R, N, U, own = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(lib().xf_sbatch_dims(h, C.byref(R), C.byref(N), C.byref(U), C.byref(own)))
        self.R, self.NNZ, self.U, self.n_owned = R.value, N.value, U.value, own.value
        self.keyed = lib().xf_sbatch_fm_keyed(h) == 1

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().xf_sbatch_free(self.h)
                self.h = None
        except Exception:
            pass


class Sharded:
    """xf_sharded_*: LRWorker / FMWorker::update across the ranks of a Group (C++ + RCCL; this
    class only marshals).  group=None: one rank, the fused single-shard step."""

    def __init__(self, group=None, model="lr", optimizer="ftrl", k=10, capacity=1 << 22,
                 schedule="sequential", seed=0, host_key_build=False, update="rank_ordered",
                 **hyper):
        require_gpu()
        c = ShardedConfig()
        lib().xf_sharded_config_default(C.byref(c))
        c.model = 0 if model == "lr" else 1
        c.optimizer = OPT_FTRL if optimizer == "ftrl" else OPT_SGD
        c.k, c.capacity, c.seed = k, capacity, seed
        c.schedule = _SCHEDULES[schedule]
        c.host_key_build = 1 if host_key_build else 0
        c.update_rule = {"rank_ordered": 0, "sum_then_step": 1}[update]
        for name, v in hyper.items():
            setattr(c, name, v)
        self.group = group
        self.model, self.k = model, k
        self.h = vp()
        check(lib().xf_sharded_create(C.byref(self.h), group.h if group is not None else None,
                                      C.byref(c)))
        w, v = vp(), vp()
        check(lib().xf_sharded_tables(self.h, C.byref(w), C.byref(v)))
        self.w = Table.from_handle(w, 1, c.optimizer)
        self.v = Table.from_handle(v, k, c.optimizer) if v else None

    def close(self):
        if getattr(self, "h", None):
            lib().xf_sharded_destroy(self.h)
            self.h = None

    __del__ = close

    def compile(self, rowptr, keys, labels, row_begin=0, row_end=None, keep=True):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if len(labels) == 0:
            labels = np.zeros(1, np.int32)       # a non-null pointer for an empty minibatch
        if row_end is None:
            row_end = len(rowptr) - 1
        h = vp()
        check(lib().xf_sharded_compile(self.h, C.byref(h), _p(rowptr, u64p), _p(keys, u64p),
                                       _p(labels, i32p), row_begin, row_end, 1 if keep else 0))
        return ShardedBatch(h, self)

    def compile_dev(self, d_keys, d_rowptr, d_labels, R, NNZ, keep=True):
        """device pointers (ints): raw keys in CSR order (u64), row offsets (u32, R + 1), labels
        (i32); the arrays may be released when this returns"""
        h = vp()
        check(lib().xf_sharded_compile_dev(self.h, C.byref(h), d_keys, d_rowptr, d_labels, R, NNZ,
                                           1 if keep else 0))
        return ShardedBatch(h, self)

    def step(self, b):
        check(lib().xf_sharded_step(self.h, b.h))

    def predict(self, b):
        out = np.empty(max(b.R, 1), np.float32)
        check(lib().xf_sharded_predict(self.h, b.h, _p(out, f32p)))
        return out[:b.R]

    def flush(self):
        check(lib().xf_sharded_flush(self.h))

    def defrag(self):
        check(lib().xf_sharded_defrag(self.h))

    def check(self):
        check(lib().xf_sharded_check(self.h))

    def set_schedule(self, schedule):
        check(lib().xf_sharded_set_schedule(
            self.h, _SCHEDULES[schedule]))

    def save(self, prefix):
        check(lib().xf_sharded_save(self.h, prefix.encode()))

    def load(self, prefix):
        check(lib().xf_sharded_load(self.h, prefix.encode()))

    def profile(self, enable):
        check(lib().xf_sharded_profile(self.h, 1 if enable else 0))

    def profile_read(self):
        ms = (C.c_double * 6)()
        steps = C.c_long(0)
        check(lib().xf_sharded_profile_read(self.h, ms, C.byref(steps)))
        names = ["owner_pull", "a2a_weights", "forward", "gradient", "a2a_grads", "owner_update"]
        return {k: ms[i] for i, k in enumerate(names)}, steps.value


class XFlow:
    """XFCreate / XFSetParam / XFStartTrain / XFGetMetric / XFDestroy."""

    def __init__(self, train_prefix, test_prefix, **params):
        self.h = vp()
        check(lib().XFCreate(C.byref(self.h), train_prefix.encode(), test_prefix.encode()))
        for k, v in params.items():
            self.set(k, v)

    def set(self, name, value):
        check(lib().XFSetParam(self.h, name.encode(), str(value).encode()))

    def train(self):
        require_gpu()
        check(lib().XFStartTrain(C.byref(self.h)))

    def save(self, path):
        check(lib().XFSaveModel(self.h, path.encode()))

    def load(self, path):
        require_gpu()
        check(lib().XFLoadModel(self.h, path.encode()))

    def predict(self):
        check(lib().XFPredict(self.h))

    def metric(self, name):
        v = C.c_double(0)
        check(lib().XFGetMetric(self.h, name.encode(), C.byref(v)))
        return v.value

    def tables(self):
        w, v = vp(), vp()
        check(lib().XFGetTables(self.h, C.byref(w), C.byref(v)))
        return w, v

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().XFDestroy(C.byref(self.h))
        except Exception:
            pass

"""multicore_hw2_amd — host-side mirror of the reference's operator boundary for ONE hot path:
brute-force nearest-neighbour search behind ``cudaCallback(k, m, n, searchPoints,
referencePoints, results)`` (reference sources/src/core.h:71, sources/src/core.cu:1282-1297).

The product is ``libknn_mi355x.so`` (hand-written HIP for gfx950 + a C-ABI, see
``include/knn_mi355x.h``).  This package only binds it with ctypes so that tests and
``bench.py`` drive exactly the entry points a C/C++ caller would.  There is no CPU fallback:
importing works anywhere, but every compute entry raises if the library is missing or no GPU
is visible.  PyTorch is plumbing only: callers that hold torch tensors pass ``tensor.data_ptr()`` /
``torch.cuda.current_stream().cuda_stream``; ``lib()`` imports torch (when installed) just before the
dlopen so the process ends up with one HIP runtime whatever the import order was.
"""
import ctypes
import os
import sys

import numpy as np

__all__ = ["lib", "lib_path", "cudaCallback", "KnnIndex", "KnnGeom", "KnnError", "KEY_INIT", "set_option",
           "get_option", "trim", "device_count", "EXPORTED_SYMBOLS", "shard_bounds"]

KEY_INIT = 0x7F80000000000000

# every symbol include/knn_mi355x.h declares
EXPORTED_SYMBOLS = [
    "cudaCallback", "knn_device_count", "knn_last_error", "knn_version", "knn_index_create",
    "knn_index_destroy", "knn_keys_init", "knn_index_query_keys", "knn_keys_to_indices",
    "knn_index_query_host", "knn_set_option", "knn_get_option", "knn_index_last_stats",
    "knn_synth_fill_device", "knn_index_timing", "knn_index_timing_read",
    "knn_debug_filter_scores", "knn_index_query_keys_slot", "knn_trim", "knn_keys_allreduce_min",
    "knn_index_query_keys_ex", "knn_index_debug_counters", "knn_debug_scan_plan", "knn_debug_scan_plan_ex", "knn_debug_shard_policy", "knn_debug_plan_shard", "knn_index_query",
    "knn_geom_create", "knn_geom_destroy", "knn_geom_info", "knn_geom_assign", "knn_index_create_sharded",
    "knn_index_seed_export", "knn_index_seed_attach", "knn_geom_first_cell",
]
QUERY_INIT_KEYS = 1   # KNN_QUERY_INIT_KEYS

_HERE = os.path.dirname(os.path.abspath(__file__))
# KNN_MI355X_LIB: A/B hook — load another build of the same C-ABI (e.g. a previous commit's .so)
lib_path = os.environ.get("KNN_MI355X_LIB") or os.path.join(_HERE, "libknn_mi355x.so")
_lib = None


class KnnError(RuntimeError):
    pass


def lib():
    """Load libknn_mi355x.so (built in-tree by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(lib_path):
        raise KnnError(f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    # A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64 (same SONAME as
    # the system one this library is linked against): whichever is mapped first serves both, but if
    # the system one came first torch later finds "No HIP GPUs".  So when torch is installed it is
    # imported before the dlopen — tests, bench.py and smoke() share device memory with it anyway.
    if "torch" not in sys.modules and os.environ.get("KNN_MI355X_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(lib_path)
    c_int, c_ll, c_vp, c_ull = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_ulonglong
    L.cudaCallback.argtypes = [c_int, c_int, c_int, c_vp, c_vp, ctypes.POINTER(ctypes.POINTER(c_int))]
    L.cudaCallback.restype = None
    L.knn_device_count.restype = c_int
    L.knn_last_error.restype = ctypes.c_char_p
    L.knn_version.restype = ctypes.c_char_p
    L.knn_index_create.argtypes = [ctypes.POINTER(c_vp), c_int, c_int, c_ll, c_vp, c_int, c_ll, c_vp]
    L.knn_index_destroy.argtypes = [c_vp]
    L.knn_index_destroy.restype = None
    L.knn_keys_init.argtypes = [c_int, c_vp, c_int, c_vp]
    L.knn_index_query_keys.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp]
    L.knn_index_query_keys_slot.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp]
    L.knn_index_query_keys_ex.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, ctypes.c_uint]
    L.knn_index_query.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_uint]
    L.knn_keys_to_indices.argtypes = [c_int, c_vp, c_int, c_vp, c_vp]
    L.knn_index_query_host.argtypes = [c_vp, c_int, c_vp, c_vp]
    L.knn_set_option.argtypes = [ctypes.c_char_p, c_ll]
    L.knn_get_option.argtypes = [ctypes.c_char_p]
    L.knn_get_option.restype = c_ll
    L.knn_index_last_stats.argtypes = [c_vp, ctypes.POINTER(c_ll)]
    L.knn_synth_fill_device.argtypes = [c_int, c_vp, c_ll, c_ull, c_ll, c_vp]
    L.knn_index_timing.argtypes = [c_vp, c_int]
    L.knn_debug_filter_scores.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_double)]
    L.knn_index_timing_read.argtypes = [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)]
    L.knn_geom_create.argtypes = [ctypes.POINTER(c_vp), c_int, c_ll, c_int, c_vp, c_ll, c_int]
    L.knn_geom_destroy.argtypes = [c_vp]
    L.knn_geom_destroy.restype = None
    L.knn_geom_info.argtypes = [c_vp, ctypes.POINTER(c_ll)]
    L.knn_geom_assign.argtypes = [c_vp, c_int, c_vp, c_ll, c_vp, c_vp]
    L.knn_index_create_sharded.argtypes = [ctypes.POINTER(c_vp), c_int, c_vp, c_int, c_ll, c_vp, c_vp, c_vp]
    L.knn_index_seed_export.argtypes = [c_vp, c_vp, c_vp]
    L.knn_index_seed_attach.argtypes = [c_vp, c_vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise KnnError(f"knn_mi355x error {rc}: {lib().knn_last_error().decode(errors='replace')}")


def device_count():
    return lib().knn_device_count()


def set_option(name, value):
    _check(lib().knn_set_option(name.encode(), int(value)))


def get_option(name):
    return lib().knn_get_option(name.encode())


def trim():
    """Release the pooled device staging buffers of the host-input entry points; bytes released."""
    f = lib().knn_trim
    f.restype = ctypes.c_longlong
    return int(f())


def debug_shard_policy(k, m, n, ndev):
    """knn_debug_shard_policy: GPUs a cudaCallback(k, m, n) uses on a node with ndev devices (host arithmetic)."""
    f = lib().knn_debug_shard_policy
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int]
    rc = f(int(k), int(m), int(n), int(ndev))
    if rc < 0:
        _check(rc)
    return rc


def debug_scan_plan(num_cu, blocks_per_cu, nitems, m, self_lists=False):
    """knn_debug_scan_plan[_ex]: sizes of one scan launch of the cell-pruned path (host arithmetic; works without a GPU)."""
    out = (ctypes.c_longlong * 8)()
    f = lib().knn_debug_scan_plan_ex
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
    _check(f(int(num_cu), int(blocks_per_cu), int(nitems), int(m), 1 if self_lists else 0, out))
    return dict(zip(("blocks", "nlists", "slice", "ovf_base", "ovf_cap", "lds_bytes", "rec_cap", "max_lists"), list(out)))


def debug_plan_shard(k, m, rows):
    """knn_debug_plan_shard: how one shard of a one-shot call would be served (host arithmetic; works without a GPU)."""
    out = (ctypes.c_longlong * 4)()
    f = lib().knn_debug_plan_shard
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    _check(f(int(k), int(m), int(rows), out))
    return dict(zip(("filter", "streamed", "grid", "chunks"), list(out)))


def shard_bounds(n, shards):
    """Contiguous index ranges of the reference set, one per shard: the partition of reference
    core.cu:875-883 (ceil(n/G) per shard, the last takes the rest; possibly empty)."""
    shards = max(1, min(int(shards), int(n)))
    per = -(-n // shards)
    return [(min(g * per, n), min(g * per + per, n)) for g in range(shards)]


def _as_f32(a, count, name):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    if a.size != count:
        raise ValueError(f"{name}: expected {count} floats, got {a.size}")
    return a


def cudaCallback(k, m, n, searchPoints, referencePoints):
    """The drop-in entry with the reference's argument meaning (core.h:71): host fp32 arrays
    ``searchPoints[m*k]``, ``referencePoints[n*k]`` row-major; returns the ``int[m]`` the C
    function hands back through ``int **results`` (the malloc'd buffer is freed here, as the
    reference's caller does at main.cu:98,175).  Runtime errors exit the process like the
    reference's CHECK macro (core.h:77-87)."""
    L = lib()
    q = _as_f32(searchPoints, k * m, "searchPoints")
    r = _as_f32(referencePoints, k * n, "referencePoints")
    res = ctypes.POINTER(ctypes.c_int)()
    L.cudaCallback(k, m, n, q.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p),
                   ctypes.byref(res))
    out = np.ctypeslib.as_array(res, shape=(m,)).astype(np.int32, copy=True)
    ctypes.CDLL(None).free(res)
    return out


class KnnGeom:
    """Global grid of a cell-range sharded set (knn_geom_* in include/knn_mi355x.h): built from a sample of the GLOBAL set,
    identical on every rank that passes the same sample.  Host arithmetic only: works without a GPU (assign needs one)."""

    def __init__(self, k, n_global, nranks, sample, seed_tiles=0):
        self._h = ctypes.c_void_p()
        s = np.ascontiguousarray(sample, dtype=np.float32).reshape(-1)
        self.k, self.nranks = int(k), int(nranks)
        _check(lib().knn_geom_create(ctypes.byref(self._h), self.k, int(n_global), self.nranks,
                                     s.ctypes.data_as(ctypes.c_void_p), s.size // self.k, int(seed_tiles)))
        out = (ctypes.c_longlong * 8)()
        _check(lib().knn_geom_info(self._h, out))
        (self.bits, self.ncells, self.cells_per_rank, self.seed_tiles, self.part_bytes, self.layer_bytes, self.sa, _) = list(out)

    def first_cell(self, rank):
        f = lib().knn_geom_first_cell
        f.argtypes = [ctypes.c_void_p, ctypes.c_int]
        f.restype = ctypes.c_longlong
        return int(f(self._h, int(rank)))

    def assign(self, rows_dev, n, owner_dev, device=0, stream=0):
        """owner_dev[i] (int32, device) = rank whose cell range holds row i of rows_dev."""
        _check(lib().knn_geom_assign(self._h, int(device), ctypes.c_void_p(int(rows_dev)), int(n),
                                     ctypes.c_void_p(int(owner_dev)), ctypes.c_void_p(stream)))

    def close(self):
        if self._h:
            lib().knn_geom_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KnnIndex:
    """Device-resident shard of the reference set (knn_index_* in include/knn_mi355x.h)."""

    @classmethod
    def sharded(cls, geom, rank, refs_dev, gids_dev, n_local, device=0, stream=0, owners=None):
        """A cell-range shard (knn_index_create_sharded): this rank's rows of the global grid `geom` (device pointers;
        gids strictly ascending).  Give it the seed layer (seed_export on every rank, gather, seed_attach) before querying.
        LIFETIME: the index BORROWS refs_dev and gids_dev (and the layer given to seed_attach) until close() — its prep, scan
        and finalise kernels read them on every query.  Pass the objects that own that memory (torch tensors, ...) as
        `owners` and the wrapper keeps them alive; with raw pointers the caller must."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.k, self.device, self.base, self.n, self._keep = geom.k, int(device), 0, int(n_local), owners
        self._layer = None
        _check(lib().knn_index_create_sharded(ctypes.byref(self._h), self.device, geom._h, int(rank), self.n,
                                              ctypes.c_void_p(int(refs_dev)), ctypes.c_void_p(int(gids_dev)),
                                              ctypes.c_void_p(stream)))
        return self

    def seed_export(self, layer_dev, stream=0):
        _check(lib().knn_index_seed_export(self._h, ctypes.c_void_p(int(layer_dev)), ctypes.c_void_p(stream)))

    def seed_attach(self, layer_dev, owner=None):
        """The replicated seed layer (device pointer, borrowed until close(): `owner` keeps its memory alive)."""
        _check(lib().knn_index_seed_attach(self._h, ctypes.c_void_p(int(layer_dev))))
        self._layer = owner

    def __init__(self, k, refs, n_local=None, device=0, base_index=0, refs_on_device=False, stream=0, owners=None):
        """refs_on_device: `refs` is a device pointer the index BORROWS until close() (`owners`: what keeps it alive);
        else host rows, copied."""
        self._h = ctypes.c_void_p()
        self._layer = None
        self.k, self.device, self.base = int(k), int(device), int(base_index)
        if refs_on_device:
            if n_local is None:
                raise ValueError("n_local is required with a device pointer")
            ptr = ctypes.c_void_p(int(refs))
            self._keep = None
        else:
            arr = np.ascontiguousarray(refs, dtype=np.float32).reshape(-1)
            if n_local is None:
                n_local = arr.size // self.k
            if arr.size != n_local * self.k:
                raise ValueError("refs size does not match n_local * k")
            ptr = arr.ctypes.data_as(ctypes.c_void_p)
            self._keep = arr
        self.n = int(n_local)
        _check(lib().knn_index_create(ctypes.byref(self._h), self.device, self.k, self.n, ptr,
                                      1 if refs_on_device else 0, self.base, ctypes.c_void_p(stream)))
        self._keep = owners if refs_on_device else None   # (host rows were copied: nothing to hold)

    def query_keys(self, m, queries_dev, keys_dev, stream=0, slot=0, init_keys=False, indices_dev=None):
        """Async: fold this shard's nearest (distance, global index) keys into keys_dev[m].
        slot 0..7 picks one of the index's eight independent query workspaces.  init_keys: the call starts the
        keys at (+INF, 0) itself (KNN_QUERY_INIT_KEYS) instead of folding into what keys_dev holds.
        indices_dev: also write the int32 indices there (knn_index_query: no separate knn_keys_to_indices launch)."""
        if indices_dev is not None:
            _check(lib().knn_index_query(self._h, int(slot), int(m), ctypes.c_void_p(int(queries_dev)),
                                         ctypes.c_void_p(int(keys_dev)), ctypes.c_void_p(int(indices_dev)),
                                         ctypes.c_void_p(stream), QUERY_INIT_KEYS if init_keys else 0))
        elif init_keys:
            _check(lib().knn_index_query_keys_ex(self._h, int(slot), int(m), ctypes.c_void_p(int(queries_dev)),
                                                 ctypes.c_void_p(int(keys_dev)), ctypes.c_void_p(stream), QUERY_INIT_KEYS))
        else:
            _check(lib().knn_index_query_keys_slot(self._h, int(slot), int(m), ctypes.c_void_p(int(queries_dev)),
                                                   ctypes.c_void_p(int(keys_dev)), ctypes.c_void_p(stream)))

    def query(self, queries):
        """Synchronous host-in/host-out query of this shard alone."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1)
        m = q.size // self.k
        out = np.empty(m, dtype=np.int32)
        _check(lib().knn_index_query_host(self._h, m, q.ctypes.data_as(ctypes.c_void_p),
                                          out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def last_stats(self):
        st = (ctypes.c_longlong * 4)()
        _check(lib().knn_index_last_stats(self._h, st))
        return list(st)

    def debug_counters(self):
        """[seed cells empty, dense cells, cells, rows of the largest cell] of the last cell-pruned batch."""
        out = (ctypes.c_longlong * 4)()
        f = lib().knn_index_debug_counters
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
        _check(f(self._h, out))
        return list(out)

    def debug_filter_scores(self, m, queries_dev, scores_dev, qnorm_dev):
        """Test hook (knn_debug_filter_scores): returns the 8 bound constants."""
        consts = (ctypes.c_double * 8)()
        _check(lib().knn_debug_filter_scores(self._h, int(m), ctypes.c_void_p(int(queries_dev)),
                                             ctypes.c_void_p(int(scores_dev)), ctypes.c_void_p(int(qnorm_dev)),
                                             consts))
        return list(consts)

    def timing(self, enable):
        """True / N > 0: bracket every (N-th) dominant-kernel launch with HIP events; False / 0: stop."""
        _check(lib().knn_index_timing(self._h, int(enable)))

    def timing_read(self):
        """(launches, total_ms) of the dominant kernel since timing(True) / the last read."""
        n, ms = ctypes.c_int(), ctypes.c_double()
        _check(lib().knn_index_timing_read(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def close(self):
        if self._h:
            lib().knn_index_destroy(self._h)
            self._h = ctypes.c_void_p()
        self._keep = self._layer = None   # (the borrowed buffers may go now)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def keys_init(keys_dev, m, device=0, stream=0):
    _check(lib().knn_keys_init(int(device), ctypes.c_void_p(int(keys_dev)), int(m), ctypes.c_void_p(stream)))


def keys_to_indices(keys_dev, m, out_dev, device=0, stream=0):
    _check(lib().knn_keys_to_indices(int(device), ctypes.c_void_p(int(keys_dev)), int(m),
                                     ctypes.c_void_p(int(out_dev)), ctypes.c_void_p(stream)))


def keys_allreduce_min(devices, keys_dev, m, streams=None):
    """RCCL min-reduction of one key array per GPU of this process (knn_keys_allreduce_min)."""
    n = len(devices)
    devs = (ctypes.c_int * n)(*[int(d) for d in devices])
    ptrs = (ctypes.c_void_p * n)(*[int(p) for p in keys_dev])
    strs = (ctypes.c_void_p * n)(*[int(s) for s in streams]) if streams is not None else None
    f = lib().knn_keys_allreduce_min
    f.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                  ctypes.POINTER(ctypes.c_void_p)]
    _check(f(n, devs, ptrs, int(m), strs))


def synth_fill_device(dst_dev, count, seed, first=0, device=0, stream=0):
    _check(lib().knn_synth_fill_device(int(device), ctypes.c_void_p(int(dst_dev)), int(count), int(seed),
                                       int(first), ctypes.c_void_p(stream)))

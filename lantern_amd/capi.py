"""ctypes binding of include/lantern_gpu.h -- what a foreign-language host would write.

Every compute call goes through the C ABI of lantern_amd/lib/liblantern_gpu.so (hand-written
HIP for gfx950).  There is no Python or CPU implementation behind these wrappers: if the
library is missing or no device is present they raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "liblantern_gpu.so")

# usearch_metric_kind_t / usearch_scalar_kind_t (cli.rs:56-69, server.rs:94-101)
METRIC_COS, METRIC_L2SQ, METRIC_HAMMING = 1, 3, 8
SCALAR_F32, SCALAR_F16, SCALAR_I8, SCALAR_B1 = 1, 3, 4, 5
METRICS = {"cos": METRIC_COS, "l2sq": METRIC_L2SQ, "hamming": METRIC_HAMMING}
EMPTY = 0xFFFFFFFF
USEARCH_HEADER_SIZE = 136


class LanternGpuError(RuntimeError):
    pass


class InitOptions(C.Structure):
    _fields_ = [
        ("metric_kind", C.c_int),
        ("metric", C.c_void_p),
        ("quantization", C.c_int),
        ("dimensions", C.c_size_t),
        ("connectivity", C.c_size_t),
        ("expansion_add", C.c_size_t),
        ("expansion_search", C.c_size_t),
        ("num_threads", C.c_size_t),
        ("pq", C.c_bool),
        ("num_centroids", C.c_size_t),
        ("num_subvectors", C.c_size_t),
        ("retriever_ctx", C.c_void_p),
        ("retriever", C.c_void_p),
        ("retriever_mut", C.c_void_p),
    ]


class Metadata(C.Structure):
    _fields_ = [
        ("neighbors_bytes", C.c_size_t),
        ("neighbors_base_bytes", C.c_size_t),
        ("inverse_log_connectivity", C.c_double),
        ("connectivity", C.c_size_t),
        ("dimensions", C.c_size_t),
        ("init_options", InitOptions),
    ]


class GraphInfo(C.Structure):
    _fields_ = [
        ("size", C.c_size_t),
        ("upper_blocks", C.c_size_t),
        ("connectivity", C.c_uint32),
        ("entry_slot", C.c_uint32),
        ("max_level", C.c_int32),
        ("vector_words", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("search_dist_evals", "search_expansions", "search_queries", "add_dist_evals",
                                          "add_expansions", "add_vectors", "add_batches", "add_walk_evals", "add_select_evals",
                                          "add_revlink_evals", "add_reprunes", "search_solo_launches")]


class BuildProfile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("walk_ms", "connect_ms", "group_ms", "revlink_ms", "exchange_ms")] + [("batches", C.c_uint64)]


EXPORTS = [
    "usearch_init", "usearch_free", "usearch_reserve", "usearch_size", "usearch_capacity", "usearch_dimensions",
    "usearch_add", "usearch_add_external", "usearch_search_ef", "lantern_gpu_cursor_open", "lantern_gpu_cursor_search",
    "lantern_gpu_cursor_seen", "lantern_gpu_cursor_close", "usearch_distance", "usearch_index_metadata", "usearch_save",
    "usearch_save_buffer", "usearch_load", "usearch_load_buffer", "usearch_serialized_length",
    "usearch_header_get_entry_slot", "usearch_header_set_entry_slot", "usearch_view_mem_lazy", "usearch_update_header",
    "lantern_gpu_version", "lantern_gpu_device_count",
    "lantern_gpu_set_seed", "lantern_gpu_set_add_batch", "lantern_gpu_add_many", "lantern_gpu_flush",
    "lantern_gpu_add_with_level", "lantern_gpu_search_batch", "lantern_gpu_search_batch_device", "lantern_gpu_search_batch_device_strided",
    "lantern_gpu_set_search_shape", "lantern_gpu_exact_search", "lantern_gpu_dense_profile", "lantern_gpu_distance_gather",
    "lantern_gpu_host_alloc", "lantern_gpu_host_free", "lantern_gpu_save_stream", "lantern_gpu_pq_compact", "lantern_gpu_pq_expand", "lantern_gpu_memory_usage", "lantern_gpu_spec_profile", "lantern_gpu_search_unique_rows", "lantern_gpu_search_row_trace", "lantern_gpu_last_search_grid", "lantern_gpu_last_gather_ms",
    "lantern_gpu_distance_matrix", "lantern_gpu_assign_to_clusters", "lantern_gpu_graph_info_get", "lantern_gpu_export_graph", "lantern_gpu_import_graph",
    "lantern_gpu_export_codes",
    "lantern_gpu_counters_get", "lantern_gpu_set_profiling", "lantern_gpu_build_profile_get", "lantern_gpu_search_phase_profile", "lantern_scan_begin", "lantern_scan_rescan", "lantern_scan_gettuple", "lantern_scan_trace", "lantern_scan_end",
    "lantern_l2sq_dist", "lantern_cos_dist", "lantern_hamming_dist", "lantern_index_server_start", "lantern_index_server_start_tls",
    "lantern_index_server_port", "lantern_index_server_status_port", "lantern_index_server_status",
    "lantern_index_server_served", "lantern_index_server_stop",
    "lantern_gpu_graph_checksum", "lantern_gpu_comm_unique_id", "lantern_gpu_comm_init_rccl", "lantern_gpu_comm_init_host",
    "lantern_gpu_comm_init_local", "lantern_gpu_comm_free", "lantern_gpu_comm_rank", "lantern_gpu_comm_world",
    "lantern_gpu_comm_set_timeout", "lantern_gpu_comm_stats", "lantern_gpu_comm_allgatherv_host",
    "lantern_gpu_comm_allgatherv_device", "lantern_gpu_shard_range", "lantern_gpu_add_sharded", "lantern_gpu_add_row_sharded", "lantern_gpu_search_partitioned", "lantern_gpu_search_batch_lane", "lantern_gpu_search_batch_lane_notify", "lantern_gpu_row_bytes",
    "lantern_gpu_level_for", "lantern_gpu_plan_batch", "lantern_gpu_row_shard_plan",
    "lantern_scan_server_start", "lantern_scan_server_start_fn", "lantern_scan_server_port", "lantern_scan_server_stats",
    "lantern_scan_server_batch_histogram", "lantern_scan_server_timing", "lantern_scan_server_stop", "lantern_scan_client_connect", "lantern_scan_client_search", "lantern_scan_client_search_next",
    "lantern_scan_client_close", "lantern_scan_begin_client",
    "lantern_mirror_acquire", "lantern_mirror_index", "lantern_mirror_version", "lantern_mirror_rebind", "lantern_mirror_advance", "lantern_mirror_release",
    "lantern_mirror_invalidate", "lantern_mirror_set_capacity", "lantern_mirror_stats",
    # Lantern's node-tape helpers (usearch_storage.hpp:9-23), host-only
    "UsearchNodeBytes", "usearch_init_node", "node_tuple_size", "label_from_node", "level_from_node", "reset_node_label", "get_node_neighbors_mut",
    "lantern_quant_bits_scalar_kind",
]

# int fn(void *ctx, const void *queries, size_t nq, size_t vec_bytes, size_t k, size_t ef, u64 *labels, f32 *dists, u32 *counts, const char **err)
BATCH_SEARCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64),
                              C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p))

# int fn(void *ctx, void *host_buf, const size_t *offsets, const size_t *counts, int world, int rank)
QUERIES_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t)  # lantern_gpu_queries_done_fn
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int, C.c_int)
COMM_ID_BYTES = 128

_lib = None


def lib() -> C.CDLL:
    """Load the HIP library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LanternGpuError(f"{LIB_PATH} is missing: run `python -m lantern_amd.build` (needs hipcc); "
                              "there is no CPU fallback")
    # NOTE for hosts that also use PyTorch-ROCm: torch bundles its own libamdhip64.so with the same
    # SONAME, and two HIP runtimes in one process cannot both own the GPU.  `import torch` BEFORE the
    # first call into this module: the loader then resolves this library's libamdhip64.so.7 dependency
    # to the runtime torch already mapped and device pointers / streams are shared.  Without torch the
    # library binds ROCm's own runtime (lantern_amd/hip.py gives buffers, streams and events).
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64, i32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    err = C.POINTER(C.c_char_p)
    sig = {
        "usearch_init": (vp, [C.POINTER(InitOptions), vp, err]),
        "usearch_free": (None, [vp, err]),
        "usearch_reserve": (None, [vp, sz, err]),
        "usearch_size": (sz, [vp, err]),
        "usearch_capacity": (sz, [vp, err]),
        "usearch_dimensions": (sz, [vp, err]),
        "usearch_add": (None, [vp, u64, vp, i32, err]),
        "usearch_add_external": (None, [vp, u64, vp, vp, i32, C.c_int16, u64, err]),
        "usearch_search_ef": (sz, [vp, vp, i32, sz, sz, C.c_bool, vp, vp, err]),
        "lantern_gpu_cursor_open": (vp, [vp, err]),
        "lantern_gpu_cursor_search": (sz, [vp, vp, i32, sz, sz, C.c_bool, vp, vp, err]),
        "lantern_gpu_cursor_seen": (sz, [vp]),
        "lantern_gpu_cursor_close": (None, [vp]),
        "usearch_distance": (f32, [vp, vp, i32, sz, i32, err]),
        "usearch_index_metadata": (Metadata, [vp, err]),
        "usearch_save": (None, [vp, C.c_char_p, err]),
        "usearch_save_buffer": (None, [vp, vp, sz, err]),
        "usearch_load": (None, [vp, C.c_char_p, err]),
        "usearch_load_buffer": (None, [vp, vp, sz, err]),
        "usearch_serialized_length": (sz, [vp, err]),
        "usearch_header_get_entry_slot": (u64, [vp]),
        "usearch_header_set_entry_slot": (None, [vp, u64]),
        "usearch_view_mem_lazy": (None, [vp, vp, err]),
        "usearch_update_header": (None, [vp, vp, err]),
        "lantern_gpu_version": (C.c_char_p, []),
        "lantern_gpu_device_count": (i32, []),
        "lantern_gpu_set_seed": (None, [vp, u64, err]),
        "lantern_gpu_set_add_batch": (None, [vp, sz, sz, err]),
        "lantern_gpu_add_many": (None, [vp, vp, vp, sz, i32, err]),
        "lantern_gpu_flush": (None, [vp, err]),
        "lantern_gpu_add_with_level": (None, [vp, u64, vp, i32, i32, err]),
        "lantern_gpu_search_batch": (None, [vp, vp, sz, i32, sz, sz, vp, vp, vp, err]),
        "lantern_gpu_search_batch_device": (None, [vp, vp, sz, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp, err]),
        "lantern_gpu_search_batch_device_strided": (None, [vp, vp, sz, sz, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp, err]),
        "lantern_gpu_set_search_shape": (None, [vp, i32, i32, err]),
        "lantern_gpu_exact_search": (None, [vp, vp, sz, sz, vp, vp, err]),
        "lantern_gpu_dense_profile": (sz, [i32, vp, vp, vp, vp, sz]),
        "lantern_gpu_distance_gather": (None, [vp, vp, vp, sz, vp, err]),
        "lantern_gpu_spec_profile": (None, [vp, i32, vp, err]),
        "lantern_gpu_search_unique_rows": (None, [vp, i32, C.POINTER(u64), err]),
        "lantern_gpu_search_row_trace": (None, [vp, i32, sz, sz, vp, vp, err]),
        "lantern_gpu_last_search_grid": (i32, [vp, err]),
        "lantern_gpu_last_gather_ms": (f32, [vp, err]),
        "lantern_gpu_save_stream": (None, [vp, vp, vp, err]),
        "lantern_gpu_pq_compact": (None, [vp, err]),
        "lantern_gpu_pq_expand": (None, [vp, err]),
        "lantern_gpu_memory_usage": (None, [vp, C.POINTER(sz), C.POINTER(sz), err]),
        "lantern_gpu_distance_matrix": (None, [vp, sz, vp, sz, i32, sz, i32, i32, vp, err]),
        "lantern_gpu_assign_to_clusters": (None, [vp, sz, sz, sz, sz, vp, sz, i32, vp, vp, err]),
        "lantern_gpu_graph_info_get": (GraphInfo, [vp, err]),
        "lantern_gpu_export_graph": (None, [vp, vp, vp, vp, vp, vp, vp, err]),
        "lantern_gpu_import_graph": (None, [vp, sz, vp, vp, vp, vp, vp, vp, u32, C.c_int32, err]),
        "lantern_gpu_export_codes": (None, [vp, vp, err]),
        "lantern_gpu_counters_get": (Counters, [vp, err]),
        "lantern_gpu_set_profiling": (None, [vp, i32, err]),
        "lantern_gpu_build_profile_get": (BuildProfile, [vp, err]),
        "lantern_gpu_search_phase_profile": (None, [vp, i32, vp, err]),
        "lantern_scan_begin": (vp, [vp, i32, i32, err]),
        "lantern_scan_rescan": (None, [vp, vp, i32, err]),
        "lantern_scan_gettuple": (C.c_bool, [vp, C.POINTER(u64), err]),
        "lantern_scan_trace": (sz, [vp, vp, sz]),
        "lantern_scan_end": (None, [vp]),
        "lantern_l2sq_dist": (f32, [vp, i32, vp, i32, err]),
        "lantern_cos_dist": (f32, [vp, i32, vp, i32, err]),
        "lantern_hamming_dist": (C.c_int32, [vp, i32, vp, i32, err]),
        "lantern_index_server_start": (vp, [C.c_char_p, i32, i32, C.c_char_p, err]),
        "lantern_index_server_start_tls": (vp, [C.c_char_p, i32, i32, C.c_char_p, C.c_char_p, C.c_char_p, err]),
        "lantern_index_server_port": (i32, [vp]),
        "lantern_index_server_status_port": (i32, [vp]),
        "lantern_index_server_status": (i32, [vp]),
        "lantern_index_server_served": (u64, [vp]),
        "lantern_index_server_stop": (None, [vp]),
        "lantern_gpu_graph_checksum": (u64, [vp, err]),
        "lantern_gpu_comm_unique_id": (None, [vp, err]),
        "lantern_gpu_comm_init_rccl": (vp, [i32, i32, vp, err]),
        "lantern_gpu_comm_init_host": (vp, [i32, i32, ALLGATHERV_FN, vp, err]),
        "lantern_gpu_comm_init_local": (None, [i32, C.POINTER(vp), err]),
        "lantern_gpu_comm_free": (None, [vp]),
        "lantern_gpu_comm_rank": (i32, [vp]),
        "lantern_gpu_comm_world": (i32, [vp]),
        "lantern_gpu_comm_set_timeout": (None, [vp, C.c_double]),
        "lantern_gpu_comm_stats": (None, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "lantern_gpu_comm_allgatherv_host": (None, [vp, vp, C.POINTER(sz), C.POINTER(sz), err]),
        "lantern_gpu_comm_allgatherv_device": (None, [vp, vp, C.POINTER(sz), C.POINTER(sz), vp, err]),
        "lantern_gpu_shard_range": (None, [sz, i32, i32, C.POINTER(sz), C.POINTER(sz)]),
        "lantern_gpu_add_sharded": (None, [vp, vp, vp, vp, sz, i32, err]),
        "lantern_gpu_add_row_sharded": (None, [vp, vp, vp, vp, sz, i32, err]),
        "lantern_gpu_search_partitioned": (None, [vp, vp, vp, sz, i32, sz, sz, vp, vp, vp, err]),
        "lantern_gpu_row_bytes": (sz, [vp, err]),
        "lantern_gpu_search_batch_lane": (None, [vp, i32, vp, sz, i32, sz, sz, vp, vp, vp, err]),
        "lantern_gpu_search_batch_lane_notify": (None, [vp, i32, vp, sz, i32, sz, sz, vp, vp, vp, vp, vp, err]),
        "lantern_gpu_level_for": (i32, [u64, u64, u32]),
        "lantern_gpu_plan_batch": (sz, [sz, i32, vp, sz, sz, sz]),
        "lantern_gpu_row_shard_plan": (sz, [vp, i32, u64, u32, sz, sz, vp, vp, vp, sz]),
        "lantern_scan_server_start": (vp, [vp, C.c_char_p, i32, sz, C.c_uint, err]),
        "lantern_scan_server_start_fn": (vp, [BATCH_SEARCH_FN, vp, sz, C.c_char_p, i32, sz, C.c_uint, err]),
        "lantern_scan_server_port": (i32, [vp]),
        "lantern_scan_server_stats": (None, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "lantern_scan_server_batch_histogram": (sz, [vp, C.POINTER(u64), sz]),
        "lantern_scan_server_timing": (None, [vp, C.POINTER(C.c_double)]),
        "lantern_scan_server_stop": (None, [vp]),
        "lantern_scan_client_connect": (vp, [C.c_char_p, i32, err]),
        "lantern_scan_client_search": (sz, [vp, vp, sz, sz, sz, vp, vp, err]),
        "lantern_scan_client_search_next": (sz, [vp, vp, sz, sz, sz, vp, vp, err]),
        "lantern_scan_client_close": (None, [vp]),
        "lantern_scan_begin_client": (vp, [vp, sz, i32, i32, err]),
        "lantern_mirror_acquire": (vp, [u64, u64, C.POINTER(InitOptions), vp, vp, sz, err]),
        "lantern_mirror_index": (vp, [vp]),
        "lantern_mirror_version": (u64, [vp]),
        "lantern_mirror_rebind": (None, [vp, C.POINTER(InitOptions)]),
        "lantern_mirror_advance": (None, [vp, u64]),
        "lantern_mirror_release": (None, [vp]),
        "lantern_mirror_invalidate": (None, [u64]),
        "lantern_mirror_set_capacity": (None, [sz]),
        "lantern_mirror_stats": (None, [C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "UsearchNodeBytes": (u32, [C.POINTER(Metadata), i32, i32]),
        "usearch_init_node": (None, [C.POINTER(Metadata), vp, u64, u32, u64, vp, sz]),
        "node_tuple_size": (u32, [vp, u32, C.POINTER(Metadata)]),
        "label_from_node": (u64, [vp]),
        "level_from_node": (C.c_ulong, [vp]),
        "reset_node_label": (None, [vp]),
        "get_node_neighbors_mut": (vp, [C.POINTER(Metadata), vp, u32, C.POINTER(u32)]),
        "lantern_quant_bits_scalar_kind": (i32, [i32, C.c_bool, err]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError = the library does not export what the header declares
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _check(err: C.c_char_p):
    if err.value is not None:
        raise LanternGpuError(err.value.decode())


def _call(name, *args):
    err = C.c_char_p()
    out = getattr(lib(), name)(*args, C.byref(err))
    _check(err)
    return out


def device_count() -> int:
    return int(lib().lantern_gpu_device_count())


def _rows(x, metric):
    dt = np.uint32 if metric == METRIC_HAMMING else np.float32
    a = np.ascontiguousarray(x, dtype=dt)
    return a.reshape(1, -1) if a.ndim == 1 else a


def _kind(metric):
    return SCALAR_B1 if metric == METRIC_HAMMING else SCALAR_F32


def distance(a, b, metric) -> float:
    """usearch_distance: one pair, on the device."""
    m = METRICS.get(metric, metric)
    A, B = _rows(a, m)[0], _rows(b, m)[0]
    dims = A.size * 32 if m == METRIC_HAMMING else A.size
    return float(_call("usearch_distance", _ptr(A), _ptr(B), _kind(m), dims, m))


def distance_matrix(a, b, metric, exact_order=True):
    m = METRICS.get(metric, metric)
    A, B = _rows(a, m), _rows(b, m)
    dims = A.shape[1] * 32 if m == METRIC_HAMMING else A.shape[1]
    out = np.empty((A.shape[0], B.shape[0]), dtype=np.float32)
    _call("lantern_gpu_distance_matrix", _ptr(A), A.shape[0], _ptr(B), B.shape[0], _kind(m), dims, m,
          1 if exact_order else 0, _ptr(out))
    return out


def assign_to_clusters(dataset, centers, metric, subvector_start=0, subvector_dim=None):
    """product_quantization.c:80-124: nearest centroid of every row's subvector; (cluster ids, distances)."""
    m = METRICS.get(metric, metric)
    X = np.ascontiguousarray(dataset, dtype=np.float32)
    Cn = np.ascontiguousarray(centers, dtype=np.float32)
    sd = Cn.shape[1] if subvector_dim is None else subvector_dim
    idx = np.zeros(X.shape[0], dtype=np.uint32)
    dist = np.zeros(X.shape[0], dtype=np.float32)
    _call("lantern_gpu_assign_to_clusters", _ptr(X), X.shape[0], X.shape[1], subvector_start, sd, _ptr(Cn), Cn.shape[0], m, _ptr(idx),
          _ptr(dist))
    return idx, dist


def version() -> str:
    return lib().lantern_gpu_version().decode()


def experimental_build() -> bool:
    """True when the loaded library contains the walk variants of csrc/experimental/ (LANTERN_BUILD_EXPERIMENTAL=1 at build time)."""
    return "+experimental" in version()


def dense_profile(on: bool):
    """on=True: record HIP events around every fp32-MFMA contraction launch of the exact k-NN.  on=False: stop and return the records
    as a list of dicts {ms, rows, cols, fused} (lantern_gpu_dense_profile)."""
    if on:
        lib().lantern_gpu_dense_profile(1, None, None, None, None, 0)
        return None
    cap = 4096
    ms, rows, cols, fused = np.zeros(cap, np.float32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    n = min(int(lib().lantern_gpu_dense_profile(0, _ptr(ms), _ptr(rows), _ptr(cols), _ptr(fused), cap)), cap)
    return [{"ms": float(ms[i]), "rows": int(rows[i]), "cols": int(cols[i]), "fused": bool(fused[i])} for i in range(n)]


def l2sq_dist(a, b) -> float:
    """SQL l2sq_dist(real[], real[]) (hnsw.c:354-360)."""
    A, B = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return float(_call("lantern_l2sq_dist", _ptr(A), A.size, _ptr(B), B.size))


def cos_dist(a, b) -> float:
    A, B = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return float(_call("lantern_cos_dist", _ptr(A), A.size, _ptr(B), B.size))


def hamming_dist(a, b) -> int:
    A, B = np.ascontiguousarray(a, dtype=np.int32), np.ascontiguousarray(b, dtype=np.int32)
    return int(_call("lantern_hamming_dist", _ptr(A), A.size, _ptr(B), B.size))


class GpuIndex:
    """usearch_index_t over the C ABI.  `dims` = f32 scalars, or u32 WORDS for hamming."""

    def __init__(self, metric, dims, M=16, ef_construction=128, ef=64, seed=42, retriever=None, quantization="f32", retriever_mut=None,
                 pq_codebook=None, num_subvectors=0):
        """retriever: optional Python callable slot(int) -> address(int) of the node tape (the
        ldb_wal_index_node_retriever contract, external_index.c:613-671), used by view_mem_lazy().
        quantization: "f32", "f16" or "i8" storage (reloption quant_bits 32 / 16 / 8, options.c:137-158); vectors
        and queries are still handed over as f32, as Lantern does."""
        self.metric = METRICS.get(metric, metric)
        self.f16 = quantization == "f16"
        self.i8 = quantization == "i8"
        self.b1 = quantization == "b1" and self.metric != METRIC_HAMMING  # quant_bits = 1 on real[]: f32 in, one bit per dimension stored
        self.dims, self.M, self.efc, self.ef = dims, M, ef_construction, ef
        o = InitOptions()
        o.metric_kind = self.metric
        o.metric = None
        o.quantization = SCALAR_F16 if self.f16 else SCALAR_I8 if self.i8 else SCALAR_B1 if self.b1 else _kind(self.metric)
        o.dimensions = dims * 32 if self.metric == METRIC_HAMMING else dims  # scan.c:84-88
        o.connectivity, o.expansion_add, o.expansion_search, o.num_threads = M, ef_construction, ef, 1
        o.pq = False
        self._retriever_cb = None
        if retriever is not None:
            self._retriever_cb = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)(lambda ctx, slot: retriever(int(slot)))
            o.retriever = C.cast(self._retriever_cb, C.c_void_p)
            o.retriever_mut = o.retriever
        self._retriever_mut_cb = None
        if retriever_mut is not None:  # external_index.c:673-697: the same lookup, marking the buffer dirty
            self._retriever_mut_cb = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)(lambda ctx, slot: retriever_mut(int(slot)))
            o.retriever_mut = C.cast(self._retriever_mut_cb, C.c_void_p)
        self.h = None
        cb = None
        if pq_codebook is not None:  # pq = true: [num_centroids][dims] f32, row c = centroid c of every subvector (pqtable.c:194-240)
            cb = np.ascontiguousarray(pq_codebook, dtype=np.float32)
            assert cb.ndim == 2 and cb.shape[1] == dims
            o.pq, o.num_centroids, o.num_subvectors = True, cb.shape[0], num_subvectors
        self.pq_S = num_subvectors if pq_codebook is not None else 0
        self.h = _call("usearch_init", C.byref(o), _ptr(cb))
        _call("lantern_gpu_set_seed", self.h, seed)

    def close(self):
        if getattr(self, "h", None):
            _call("usearch_free", self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(_call("usearch_size", self.h))

    @property
    def capacity(self):
        return int(_call("usearch_capacity", self.h))

    def reserve(self, n):
        _call("usearch_reserve", self.h, n)

    def set_add_batch(self, max_batch, min_ratio):
        _call("lantern_gpu_set_add_batch", self.h, max_batch, min_ratio)

    def set_search_shape(self, waves, max_workgroups=0):
        _call("lantern_gpu_set_search_shape", self.h, waves, max_workgroups)

    def add(self, label, vec, level=None):
        v = _rows(vec, self.metric)[0]
        if v.size != self.dims:
            raise LanternGpuError(f"Wrong number of dimensions: {v.size} instead of {self.dims} expected")  # hnsw_insert.out
        if level is None:
            _call("usearch_add", self.h, int(label), _ptr(v), _kind(self.metric))
        else:
            _call("lantern_gpu_add_with_level", self.h, int(label), _ptr(v), _kind(self.metric), int(level))

    def add_many(self, labels, vecs):
        V = _rows(vecs, self.metric)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        assert V.shape[1] == self.dims and lab.size == V.shape[0]
        _call("lantern_gpu_add_many", self.h, _ptr(lab), _ptr(V), V.shape[0], _kind(self.metric))

    def flush(self):
        _call("lantern_gpu_flush", self.h)

    def add_sharded(self, comm: "Comm", labels, vecs):
        """COLLECTIVE: this rank's shard of the rows (global slot order = rank order); see lantern_gpu_add_sharded."""
        V = _rows(vecs, self.metric) if len(labels) else np.zeros((0, self.dims), dtype=np.uint32 if self.metric == METRIC_HAMMING else np.float32)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        assert V.shape[1] == self.dims and lab.size == V.shape[0]
        _call("lantern_gpu_add_sharded", self.h, comm.h, _ptr(lab), _ptr(V), V.shape[0], _kind(self.metric))

    def add_row_sharded(self, comm: "Comm", labels, vecs):
        """COLLECTIVE on an empty index: the row-sharded build (candidates per shard, merged); see lantern_gpu_add_row_sharded."""
        V = _rows(vecs, self.metric) if len(labels) else np.zeros((0, self.dims), dtype=np.uint32 if self.metric == METRIC_HAMMING else np.float32)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        assert V.shape[1] == self.dims and lab.size == V.shape[0]
        _call("lantern_gpu_add_row_sharded", self.h, comm.h, _ptr(lab), _ptr(V), V.shape[0], _kind(self.metric))

    def spec_profile(self, on, read=False):
        """The instrumented latency-bound walk: {wave role: {section: cycles}} accumulated since the last read (read=True)."""
        out = (C.c_ulonglong * 32)() if read else None
        _call("lantern_gpu_spec_profile", self.h, 1 if on else 0, out)
        if not read:
            return None
        names = ("decision", "neighbour_list", "issue", "role_section", "loads_and_distances", "barrier_wait", "hops", "list_source")
        return {role: {n: int(out[8 * w + i]) for i, n in enumerate(names)} for w, role in enumerate(("visit", "list", "fill", "row"))}

    def unique_rows(self, on, read=False):
        """lantern_gpu_search_unique_rows: switch the row-bitmap instrumentation on / off; with read=True return the number of
        distinct rows evaluated since it was switched on."""
        out = C.c_uint64(0)
        _call("lantern_gpu_search_unique_rows", self.h, 1 if on else 0, C.byref(out) if read else None)
        return int(out.value) if read else None

    def last_gather_ms(self) -> float:
        """kernel time of the last distance_gather launch (HIP events on the index stream)"""
        return float(_call("lantern_gpu_last_gather_ms", self.h))

    def last_search_grid(self) -> int:
        return int(_call("lantern_gpu_last_search_grid", self.h))

    def row_trace_begin(self, nq, per_query_cap):
        """lantern_gpu_search_row_trace(on=1): the following launches of at most nq queries record, per query, the memory objects the
        walk asks for (rows evaluated, adjacency lists read) in order."""
        self._trace_shape = (int(nq), int(per_query_cap))
        _call("lantern_gpu_search_row_trace", self.h, 1, int(nq), int(per_query_cap), None, None)

    def row_trace_end(self):
        """-> (trace [nq][cap] u32, counts [nq] u32) of the last traced launch; switches the tracing off."""
        nq, cap = self._trace_shape
        trace = np.zeros((nq, cap), dtype=np.uint32)
        counts = np.zeros(nq, dtype=np.uint32)
        _call("lantern_gpu_search_row_trace", self.h, 0, nq, cap, _ptr(trace), _ptr(counts))
        return trace, counts

    def pq_compact(self):
        """pq = true: drop the decoded rows from HBM; searches run ADC over the code bytes."""
        _call("lantern_gpu_pq_compact", self.h)

    def pq_expand(self):
        _call("lantern_gpu_pq_expand", self.h)

    def row_bytes(self) -> int:
        """Bytes of one stored row in device memory: the stride of device-resident queries (device_query_rows())."""
        return int(_call("lantern_gpu_row_bytes", self.h))

    def memory_usage(self):
        """(bytes of the vector block or of a compact pq index's code rows, bytes of everything else that grows with the nodes)"""
        a, b = C.c_size_t(), C.c_size_t()
        _call("lantern_gpu_memory_usage", self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def checksum(self) -> int:
        return int(_call("lantern_gpu_graph_checksum", self.h))

    def search(self, query, k, ef=0, streaming=False):
        """usearch_search_ef: (labels, distances) of length <= k."""
        q = _rows(query, self.metric)[0]
        if q.size != self.dims:
            kind = "int" if self.metric == METRIC_HAMMING else "real"
            raise LanternGpuError(f"Expected {kind} array with dimension {self.dims}, got {q.size}")  # hnsw.c:474-476
        labels = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        n = _call("usearch_search_ef", self.h, _ptr(q), _kind(self.metric), k, ef, bool(streaming), _ptr(labels), _ptr(dists))
        return labels[:n], dists[:n]

    def add_external(self, label, vec, node_tape, level, slot):
        """usearch_add_external (insert.c:209): node_tape = address (int) of the new node's tape, slot = its 48-bit page slot."""
        v = _rows(vec, self.metric)[0]
        _call("usearch_add_external", self.h, int(label), _ptr(v), C.c_void_p(node_tape), _kind(self.metric), int(level), int(slot))

    def update_header(self, header: bytes) -> bytes:
        """usearch_update_header (insert.c:214): returns the refreshed 136 bytes."""
        buf = C.create_string_buffer(bytes(header), USEARCH_HEADER_SIZE)
        _call("usearch_update_header", self.h, C.cast(buf, C.c_void_p))
        return buf.raw[:USEARCH_HEADER_SIZE]

    def cursor(self) -> "Cursor":
        return Cursor(self)

    def search_batch(self, queries, k, ef=0):
        Q = _rows(queries, self.metric)
        nq = Q.shape[0]
        labels = np.zeros((nq, k), dtype=np.uint64)
        dists = np.zeros((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        _call("lantern_gpu_search_batch", self.h, _ptr(Q), nq, _kind(self.metric), k, ef, _ptr(labels), _ptr(dists), _ptr(counts))
        return labels, dists, counts

    def search_batch_lane(self, lane, queries, k, ef=0):
        """lantern_gpu_search_batch for a caller that keeps two batches in flight (lane 0 / 1, one thread each)."""
        Q = _rows(queries, self.metric)
        nq = Q.shape[0]
        labels = np.zeros((nq, k), dtype=np.uint64)
        dists = np.zeros((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        _call("lantern_gpu_search_batch_lane", self.h, lane, _ptr(Q), nq, _kind(self.metric), k, ef, _ptr(labels), _ptr(dists), _ptr(counts))
        return labels, dists, counts

    def search_batch_lane_notify(self, lane, queries, k, ef=0):
        """lantern_gpu_search_batch_lane_notify: the answers plus the ORDER in which the queries were handed on (a list of index lists,
        one per callback)."""
        Q = _rows(queries, self.metric)
        nq = Q.shape[0]
        labels = np.zeros((nq, k), dtype=np.uint64)
        dists = np.zeros((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        calls = []
        snapshots = {}

        def on_done(ctx, which, count):
            idx = [int(which[i]) for i in range(count)]
            calls.append(idx)
            for j in idx:  # the rows are filled in when the callback runs
                snapshots[j] = (labels[j].copy(), dists[j].copy(), int(counts[j]))

        cb = QUERIES_DONE_FN(on_done)
        _call("lantern_gpu_search_batch_lane_notify", self.h, lane, _ptr(Q), nq, _kind(self.metric), k, ef, _ptr(labels), _ptr(dists), _ptr(counts),
              C.cast(cb, C.c_void_p), None)
        return labels, dists, counts, calls, snapshots

    def search_partitioned(self, comm: "Comm", queries, k, ef=0):
        """COLLECTIVE: this rank's index holds one share of the rows; every rank passes the same queries and gets the same
        global top-k (lantern_gpu_search_partitioned)."""
        Q = _rows(queries, self.metric)
        nq = Q.shape[0]
        labels = np.zeros((nq, k), dtype=np.uint64)
        dists = np.zeros((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        _call("lantern_gpu_search_partitioned", self.h, comm.h, _ptr(Q), nq, _kind(self.metric), k, ef, _ptr(labels), _ptr(dists), _ptr(counts))
        return labels, dists, counts

    def search_batch_device(self, d_queries, nq, k, ef=0, skip=0, d_labels=None, d_dists=None, d_slots=None, d_counts=None,
                            d_D=None, d_E=None, stream=None, query_stride=None):
        """All pointers are raw device addresses (ints), e.g. torch.Tensor.data_ptr().  `query_stride` = bytes between consecutive
        query rows (device_query_rows(...).strides[0]): checked against the index's stored stride by the library.  Without it the
        call is accepted only for indexes whose stride is unambiguous (no widened rows) -- lantern_gpu.h."""
        if query_stride is None:
            _call("lantern_gpu_search_batch_device", self.h, _ptr(d_queries), nq, k, ef, skip, _ptr(d_labels), _ptr(d_dists),
                  _ptr(d_slots), _ptr(d_counts), _ptr(d_D), _ptr(d_E), _ptr(stream))
        else:
            _call("lantern_gpu_search_batch_device_strided", self.h, _ptr(d_queries), int(query_stride), nq, k, ef, skip, _ptr(d_labels),
                  _ptr(d_dists), _ptr(d_slots), _ptr(d_counts), _ptr(d_D), _ptr(d_E), _ptr(stream))

    def device_query_rows(self, queries) -> np.ndarray:
        """Host rows in the index's STORAGE format at the index's OWN row stride (zero padded): upload them and pass
        `query_stride=rows.strides[0]` to search_batch_device.  Takes the format and the stride from the index itself, so neither
        can be forgotten (bit rows of 65 .. 127 bytes sit at a 128-byte stride)."""
        from . import hip

        return hip.padded_rows(queries, self.metric == METRIC_HAMMING, self.f16, self.i8, self.b1, row_bytes=self.row_bytes())

    def exact_search(self, queries, k):
        Q = _rows(queries, self.metric)
        slots = np.zeros((Q.shape[0], k), dtype=np.uint32)
        dists = np.zeros((Q.shape[0], k), dtype=np.float32)
        _call("lantern_gpu_exact_search", self.h, _ptr(Q), Q.shape[0], k, _ptr(slots), _ptr(dists))
        return slots, dists

    def distance_gather(self, query, slots):
        q = _rows(query, self.metric)[0]
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.zeros(s.size, dtype=np.float32)
        _call("lantern_gpu_distance_gather", self.h, _ptr(q), _ptr(s), s.size, _ptr(out))
        return out

    def graph_info(self):
        return _call("lantern_gpu_graph_info_get", self.h)

    def counters(self):
        c = _call("lantern_gpu_counters_get", self.h)
        return {n: int(getattr(c, n)) for n, _ in Counters._fields_}

    def set_profiling(self, on=True):
        _call("lantern_gpu_set_profiling", self.h, 1 if on else 0)

    def build_profile(self):
        p = _call("lantern_gpu_build_profile_get", self.h)
        return {n: (int(getattr(p, n)) if n == "batches" else float(getattr(p, n))) for n, _ in BuildProfile._fields_}

    def phase_profile(self, on=True, read=False):
        """Diagnostics: instrumented walk kernel; read=True returns and clears {phase: cycles}.  With the two-wave split of
        the register-list walk "merge" is thread 0's pop decision and "pop" the list wave's whole (concurrent) section."""
        out = np.zeros(8, dtype=np.uint64) if read else None
        _call("lantern_gpu_search_phase_profile", self.h, 1 if on else 0, _ptr(out))
        return dict(zip(("visited_compact", "first_barrier", "distances", "merge", "pop", "list_arrival", "descent", "query"), (int(x) for x in out))) if read else None

    def metadata(self):
        return _call("usearch_index_metadata", self.h)

    def export_graph(self, with_vectors=False):
        gi = self.graph_info()
        n, M = gi.size, self.M
        g = {
            "levels": np.zeros(n, dtype=np.uint8),
            "nbr0": np.zeros((n, 2 * M), dtype=np.uint32),
            "upper_off": np.zeros(n, dtype=np.uint32),
            "upper_nbr": np.zeros((max(gi.upper_blocks, 1), M), dtype=np.uint32),
            "labels": np.zeros(n, dtype=np.uint64),
        }
        vecs = None
        if with_vectors:  # storage format: u32 words (hamming), f32, or f16 halves (two per word)
            vecs = np.zeros((n, gi.vector_words), dtype=np.uint32 if (self.metric == METRIC_HAMMING or self.f16 or self.i8 or self.b1) else np.float32)
        _call("lantern_gpu_export_graph", self.h, _ptr(g["levels"]), _ptr(g["nbr0"]), _ptr(g["upper_off"]),
              _ptr(g["upper_nbr"]), _ptr(g["labels"]), _ptr(vecs))
        g["upper_nbr"] = g["upper_nbr"][:gi.upper_blocks]
        g["entry_slot"], g["max_level"] = int(gi.entry_slot), int(gi.max_level)
        if with_vectors:
            g["vectors"] = vecs.view(np.float16)[:, :self.dims] if self.f16 else vecs.view(np.int8)[:, :self.dims] if self.i8 else vecs
        return g

    def export_codes(self):
        """pq indexes: the code bytes [size][num_subvectors]."""
        codes = np.zeros((self.graph_info().size, self.pq_S), dtype=np.uint8)
        _call("lantern_gpu_export_codes", self.h, _ptr(codes))
        return codes

    def import_graph(self, vectors, graph):
        V = _rows(vectors, self.metric)
        if self.f16:  # the importer takes rows in STORAGE format: halves, padded to whole 4-byte words
            h = np.zeros((V.shape[0], (self.dims + 1) // 2 * 2), dtype=np.float16)
            h[:, :self.dims] = V.astype(np.float16)
            V = h
        if self.i8:  # storage format: the quantised bytes, padded to whole 4-byte words; V must already hold integers
            q = np.zeros((V.shape[0], (self.dims + 3) // 4 * 4), dtype=np.int8)
            q[:, :self.dims] = V.astype(np.int8)
            V = q
        levels = np.ascontiguousarray(graph["levels"], dtype=np.uint8)
        nbr0 = np.ascontiguousarray(graph["nbr0"], dtype=np.uint32)
        upper_off = np.ascontiguousarray(graph["upper_off"], dtype=np.uint32)
        upper_nbr = np.ascontiguousarray(graph["upper_nbr"], dtype=np.uint32)
        if upper_nbr.size == 0:
            upper_nbr = np.full((1, self.M), EMPTY, dtype=np.uint32)
        labels = np.ascontiguousarray(graph["labels"], dtype=np.uint64) if graph.get("labels") is not None else None
        _call("lantern_gpu_import_graph", self.h, V.shape[0], _ptr(V), _ptr(labels), _ptr(levels), _ptr(nbr0),
              _ptr(upper_off), _ptr(upper_nbr), int(graph["entry_slot"]), int(graph["max_level"]))

    def view_mem_lazy(self, header: bytes):
        """usearch_view_mem_lazy (scan.c:110): mirror the page-resident graph through the retriever."""
        buf = C.create_string_buffer(bytes(header), USEARCH_HEADER_SIZE)
        _call("usearch_view_mem_lazy", self.h, C.cast(buf, C.c_void_p))

    def save(self, path):
        _call("usearch_save", self.h, path.encode())

    def load(self, path):
        _call("usearch_load", self.h, path.encode())

    def save_buffer(self) -> bytes:
        n = int(_call("usearch_serialized_length", self.h))
        buf = (C.c_char * n)()
        _call("usearch_save_buffer", self.h, C.cast(buf, C.c_void_p), n)
        return bytes(buf)

    def save_stream(self) -> bytes:
        """lantern_gpu_save_stream: the file as the spans the callback is handed, joined (the bytes of save_buffer())."""
        class Span(C.Structure):
            _fields_ = [("data", C.c_void_p), ("size", C.c_size_t)]

        parts = []

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Span), C.c_size_t)
        def write(_ctx, spans, count):
            for i in range(count):
                parts.append(C.string_at(spans[i].data, spans[i].size))
            return 0

        _call("lantern_gpu_save_stream", self.h, C.cast(write, C.c_void_p), None)
        return b"".join(parts)

    def load_buffer(self, data: bytes):
        buf = C.create_string_buffer(data, len(data))
        _call("usearch_load_buffer", self.h, C.cast(buf, C.c_void_p), len(data))


class Mirror:
    """lantern_mirror_*: the cached HBM mirror of a page-resident index.  `.index` is a GpuIndex view of it (not owned)."""

    def __init__(self, relation, version, metric, dims, header, retriever, M=16, ef_construction=128, ef=64, min_vectors=0, retriever_mut=None):
        self.metric = METRICS.get(metric, metric)
        o = InitOptions()
        o.metric_kind, o.metric, o.quantization = self.metric, None, _kind(self.metric)
        o.dimensions = dims * 32 if self.metric == METRIC_HAMMING else dims
        o.connectivity, o.expansion_add, o.expansion_search, o.num_threads = M, ef_construction, ef, 1
        self._cb = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)(lambda ctx, slot: retriever(int(slot)))
        o.retriever = C.cast(self._cb, C.c_void_p)
        self._cb_mut = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)(lambda ctx, slot: (retriever_mut or retriever)(int(slot)))
        o.retriever_mut = C.cast(self._cb_mut, C.c_void_p)
        self._opts = o
        buf = C.create_string_buffer(bytes(header), USEARCH_HEADER_SIZE)
        self.m = _call("lantern_mirror_acquire", relation, version, C.byref(o), None, C.cast(buf, C.c_void_p), min_vectors)
        self.index = None
        if self.m:
            self.index = GpuIndex.__new__(GpuIndex)
            self.index.metric, self.index.dims, self.index.M, self.index.efc, self.index.ef = self.metric, dims, M, ef_construction, ef
            self.index.f16 = self.index.i8 = self.index.b1 = False
            self.index.pq_S = 0
            self.index.h = None  # not owned: GpuIndex.close() must not free it
            self.index.__dict__["h"] = lib().lantern_mirror_index(self.m)
            self.index.close = lambda: None

    @property
    def declined(self):
        return not self.m

    @property
    def version(self):
        return int(lib().lantern_mirror_version(self.m))

    def rebind(self):
        """Bind THIS holder's retriever callbacks to the shared mirror (the latest acquire / rebind wins; any release unbinds)."""
        lib().lantern_mirror_rebind(self.m, C.byref(self._opts))

    def advance(self, version):
        lib().lantern_mirror_advance(self.m, version)

    def release(self):
        if self.m:
            lib().lantern_mirror_release(self.m)
            self.m = None

    @staticmethod
    def invalidate(relation):
        lib().lantern_mirror_invalidate(relation)

    @staticmethod
    def set_capacity(n):
        lib().lantern_mirror_set_capacity(n)

    @staticmethod
    def stats():
        v = [C.c_uint64() for _ in range(4)]
        lib().lantern_mirror_stats(*[C.byref(x) for x in v])
        return dict(zip(("hits", "misses", "rebuilds", "resident"), (int(x.value) for x in v)))


class Cursor:
    """lantern_gpu_cursor_*: one scan's share of usearch_search_ef's streaming contract on a shared index."""

    def __init__(self, index: GpuIndex):
        self.index = index
        self.c = _call("lantern_gpu_cursor_open", index.h)

    def search(self, query, k, ef=0, streaming=False):
        q = _rows(query, self.index.metric)[0]
        labels = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        n = _call("lantern_gpu_cursor_search", self.c, _ptr(q), _kind(self.index.metric), k, ef, bool(streaming), _ptr(labels), _ptr(dists))
        return labels[:n], dists[:n]

    @property
    def seen(self):
        return int(lib().lantern_gpu_cursor_seen(self.c))

    def close(self):
        if self.c:
            lib().lantern_gpu_cursor_close(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def metadata_for(M: int, dims: int, quantization: int = SCALAR_F32, metric: int = METRIC_L2SQ, pq: bool = False, num_subvectors: int = 0,
                 num_centroids: int = 0) -> Metadata:
    """The metadata_t usearch_index_metadata returns for such an index (list sizes in bytes: 4 + 2M*6 at level 0, 4 + M*6 above;
    usearch_storage.cpp:19-32 reads exactly these), built without an index: node-tape arithmetic needs no device."""
    import math

    m = Metadata()
    m.neighbors_bytes, m.neighbors_base_bytes = 4 + M * 6, 4 + 2 * M * 6
    m.inverse_log_connectivity = 1.0 / math.log(M)
    m.connectivity, m.dimensions = M, dims
    o = m.init_options
    o.metric_kind, o.quantization, o.dimensions, o.connectivity = metric, quantization, dims, M
    o.pq, o.num_subvectors, o.num_centroids = pq, num_subvectors, num_centroids
    return m


def header_entry_slot(header: bytes) -> int:
    buf = C.create_string_buffer(bytes(header), USEARCH_HEADER_SIZE)
    return int(lib().usearch_header_get_entry_slot(C.cast(buf, C.c_void_p)))


def level_for(seed: int, slot: int, M: int) -> int:
    return int(lib().lantern_gpu_level_for(seed, slot, M))


def plan_batch(size: int, max_level: int, pending_levels, max_batch: int, min_ratio: int) -> int:
    lv = np.ascontiguousarray(pending_levels, dtype=np.int32)
    return int(lib().lantern_gpu_plan_batch(size, max_level, _ptr(lv), lv.size, max_batch, min_ratio))


def row_shard_plan(shard_sizes, seed: int, M: int, max_batch: int, min_ratio: int):
    """[(first, count, [rows from shard r, ...]), ...]: the batches of lantern_gpu_add_row_sharded (host arithmetic, no device)."""
    sizes = np.ascontiguousarray(shard_sizes, dtype=np.uint64)
    world = int(sizes.size)
    n = int(lib().lantern_gpu_row_shard_plan(_ptr(sizes), world, seed, M, max_batch, min_ratio, None, None, None, 0))
    first, count = np.zeros(max(n, 1), dtype=np.uintp), np.zeros(max(n, 1), dtype=np.uintp)
    share = np.zeros((max(n, 1), world), dtype=np.uintp)
    lib().lantern_gpu_row_shard_plan(_ptr(sizes), world, seed, M, max_batch, min_ratio, _ptr(first), _ptr(count), _ptr(share), n)
    return [(int(first[t]), int(count[t]), [int(x) for x in share[t]]) for t in range(n)]


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    b, e = C.c_size_t(), C.c_size_t()
    lib().lantern_gpu_shard_range(n, world, rank, C.byref(b), C.byref(e))
    return int(b.value), int(e.value)


class Comm:
    """lantern_gpu_comm_t: the exchange transport of the work-sharded build (RCCL, a caller-supplied host
    all-gather, or the in-process hub)."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep  # the ctypes callback must outlive the communicator

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _call("lantern_gpu_comm_unique_id", C.cast(buf, C.c_void_p))
        return buf.raw

    @classmethod
    def rccl(cls, rank: int, world: int, uid: bytes) -> "Comm":
        assert len(uid) == COMM_ID_BYTES
        buf = C.create_string_buffer(uid, COMM_ID_BYTES)
        return cls(_call("lantern_gpu_comm_init_rccl", rank, world, C.cast(buf, C.c_void_p)))

    @classmethod
    def host(cls, rank: int, world: int, allgatherv) -> "Comm":
        """allgatherv(buf: np.ndarray[u8] (whole extent, in place), offsets: list[int], counts: list[int]) -> None"""

        def tramp(ctx, host_buf, offsets, counts, w, r):
            try:
                offs = [int(offsets[i]) for i in range(w)]
                cnts = [int(counts[i]) for i in range(w)]
                extent = max((o + c for o, c in zip(offs, cnts)), default=0)
                view = np.ctypeslib.as_array(C.cast(host_buf, C.POINTER(C.c_uint8)), shape=(max(extent, 1),))
                allgatherv(view, offs, cnts)
                return 0
            except Exception:  # the C side turns a non-zero return into an error string
                import traceback

                traceback.print_exc()
                return 1

        cb = ALLGATHERV_FN(tramp)
        return cls(_call("lantern_gpu_comm_init_host", rank, world, cb, None), keep=cb)

    @classmethod
    def local_world(cls, world: int) -> list["Comm"]:
        arr = (C.c_void_p * world)()
        _call("lantern_gpu_comm_init_local", world, arr)
        return [cls(arr[i]) for i in range(world)]

    @property
    def rank(self):
        return int(lib().lantern_gpu_comm_rank(self.h))

    @property
    def world(self):
        return int(lib().lantern_gpu_comm_world(self.h))

    def set_timeout(self, seconds: float):
        lib().lantern_gpu_comm_set_timeout(self.h, float(seconds))

    def stats(self):
        b, c = C.c_uint64(), C.c_uint64()
        lib().lantern_gpu_comm_stats(self.h, C.byref(b), C.byref(c))
        return {"bytes_received": int(b.value), "collectives": int(c.value)}

    def allgatherv_host(self, buf: np.ndarray, offsets, counts):
        w = len(offsets)
        off = (C.c_size_t * w)(*offsets)
        cnt = (C.c_size_t * w)(*counts)
        _call("lantern_gpu_comm_allgatherv_host", self.h, _ptr(buf), off, cnt)

    def allgatherv_device(self, d_ptr: int, offsets, counts, stream=None):
        w = len(offsets)
        off = (C.c_size_t * w)(*offsets)
        cnt = (C.c_size_t * w)(*counts)
        _call("lantern_gpu_comm_allgatherv_device", self.h, _ptr(d_ptr), off, cnt, _ptr(stream))

    def free(self):
        if self.h:
            lib().lantern_gpu_comm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Scan:
    """lantern_scan_*: the amgettuple paging shim (scan.c:24-338)."""

    def __init__(self, index: GpuIndex = None, init_k=10, ef=0, client: "ScanClient" = None, metric=None, dims=None):
        """A scan on a local index, or -- client=ScanClient, metric=, dims= -- through the scan-side service."""
        self.index, self.client = index, client
        if client is not None:
            self.metric, self.dims = METRICS.get(metric, metric), dims
            qbytes = dims * 4  # f32 scalars, or u32 words for hamming
            self.s = _call("lantern_scan_begin_client", client.c, qbytes, init_k, ef)
        else:
            self.metric, self.dims = index.metric, index.dims
            self.s = _call("lantern_scan_begin", index.h, init_k, ef)

    def rescan(self, query):
        q = _rows(query, self.metric)[0]
        if q.size != self.dims:
            kind = "int" if self.metric == METRIC_HAMMING else "real"
            raise LanternGpuError(f"Expected {kind} array with dimension {self.dims}, got {q.size}")
        _call("lantern_scan_rescan", self.s, _ptr(q), _kind(self.metric))

    def gettuple(self):
        label = C.c_uint64()
        err = C.c_char_p()
        ok = lib().lantern_scan_gettuple(self.s, C.byref(label), C.byref(err))
        _check(err)
        return int(label.value) if ok else None

    def trace(self):
        """The k of every usearch_search_ef issued since the last rescan ("querying index for %d elements", scan.c:219, :272)."""
        n = lib().lantern_scan_trace(self.s, None, 0)
        ks = (C.c_int * max(n, 1))()
        lib().lantern_scan_trace(self.s, ks, n)
        return [int(ks[i]) for i in range(n)]

    def fetch(self, limit):
        out = []
        while len(out) < limit:
            l = self.gettuple()
            if l is None:
                break
            out.append(l)
        return out

    def end(self):
        if self.s:
            lib().lantern_scan_end(self.s)
            self.s = None

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass


class ScanServer:
    """lantern_scan_server_*: one HBM-resident index serving many backends' queries in batched launches."""

    def __init__(self, index=None, host="127.0.0.1", port=0, max_batch=256, max_wait_us=200, batch_fn=None, vec_bytes=0):
        """index: a GpuIndex (the server runs lantern_gpu_search_batch on it), or batch_fn(queries u8[nq, vec_bytes], k, ef)
        -> (labels u64[nq, k], dists f32[nq, k], counts u32[nq]) for tests / custom back ends."""
        self._keep = None
        self.index = index
        if batch_fn is not None:
            def tramp(ctx, queries, nq, vb, k, ef, labels, dists, counts, errp):
                try:
                    q = np.ctypeslib.as_array(C.cast(queries, C.POINTER(C.c_uint8)), shape=(nq, vb))
                    lab, dst, cnt = batch_fn(q, int(k), int(ef))
                    np.ctypeslib.as_array(labels, shape=(nq, k))[:] = lab
                    np.ctypeslib.as_array(dists, shape=(nq, k))[:] = dst
                    np.ctypeslib.as_array(counts, shape=(nq,))[:] = cnt
                    return 0
                except Exception as ex:  # noqa: BLE001 -- becomes the error frame every query of the launch gets
                    self._last_error = C.c_char_p(str(ex).encode())
                    errp[0] = self._last_error
                    return 1

            self._keep = BATCH_SEARCH_FN(tramp)
            self.s = _call("lantern_scan_server_start_fn", self._keep, None, vec_bytes, host.encode(), port, max_batch, max_wait_us)
        else:
            self.s = _call("lantern_scan_server_start", index.h, host.encode(), port, max_batch, max_wait_us)
        self.host = host
        self.port = int(lib().lantern_scan_server_port(self.s))

    def stats(self):
        v = [C.c_uint64() for _ in range(4)]
        lib().lantern_scan_server_stats(self.s, *[C.byref(x) for x in v])
        return dict(zip(("requests", "batches", "launches", "largest_batch"), (int(x.value) for x in v)))

    def timing(self):
        """Mean microseconds of a request on the server by leg (cumulative since start) and the number of requests."""
        v = (C.c_double * 4)()
        lib().lantern_scan_server_timing(self.s, v)
        return {"wait_for_batch_us": v[0], "batch_closed_to_answer_us": v[1], "answer_to_socket_us": v[2], "requests": int(v[3])}

    def stop(self):
        if self.s:
            lib().lantern_scan_server_stop(self.s)
            self.s = None

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass


class ScanClient:
    """lantern_scan_client_*: what a backend's ldb_amgettuple calls where it calls usearch_search_ef today."""

    def __init__(self, host, port):
        self.c = _call("lantern_scan_client_connect", host.encode(), port)

    def search(self, query: np.ndarray, k: int, ef: int = 0):
        q = np.ascontiguousarray(query)
        labels = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        n = _call("lantern_scan_client_search", self.c, _ptr(q), q.nbytes, k, ef, _ptr(labels), _ptr(dists))
        return labels[:n], dists[:n]

    def search_next(self, query: np.ndarray, k: int, ef: int = 0):
        """The next k rows of the scan this connection began with search() (same query)."""
        q = np.ascontiguousarray(query)
        labels = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        n = _call("lantern_scan_client_search_next", self.c, _ptr(q), q.nbytes, k, ef, _ptr(labels), _ptr(dists))
        return labels[:n], dists[:n]

    def close(self):
        if self.c:
            lib().lantern_scan_client_close(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IndexServer:
    """lantern_index_server_*: the external indexing server (B3) on 127.0.0.1 by default."""

    def __init__(self, host="127.0.0.1", port=0, status_port=0, tmp_dir="/tmp", cert=None, key=None):
        """cert / key: PEM files -> the server speaks TLS (lantern_index_server_start_tls)."""
        self.s = _call("lantern_index_server_start_tls", host.encode(), port, status_port, tmp_dir.encode(),
                       cert.encode() if cert else None, key.encode() if key else None)
        self.host = host
        self.port = int(lib().lantern_index_server_port(self.s))
        self.status_port = int(lib().lantern_index_server_status_port(self.s))

    @property
    def status(self):
        return int(lib().lantern_index_server_status(self.s))

    @property
    def served(self):
        return int(lib().lantern_index_server_served(self.s))

    def stop(self):
        if self.s:
            lib().lantern_index_server_stop(self.s)
            self.s = None

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass

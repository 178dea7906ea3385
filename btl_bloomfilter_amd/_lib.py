"""ctypes binding of the C ABI in include/btlbf.h (libbtlbf.so, hand-written HIP for gfx950).

There is no Python or CPU fallback: if the shared library is missing this module raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BTLBF_LIB selects another build of the SAME library (diagnostic / sanitizer builds made by build.py with
# BTLBF_BUILD_TAG, tools/sanitize_host.sh); there is no other implementation to select
LIB_PATH = os.environ.get("BTLBF_LIB") or os.path.join(HERE, "libbtlbf.so")

HOST, DEVICE = 0, 1
BLOOM, COUNTING8 = 0, 1
INCREMENT_MIN, INCREMENT_ALL = 0, 1
ORDER_PARALLEL, ORDER_SERIAL = 0, 1
OK, EINVAL, ENOMEM, EIO, EFORMAT, EHIP = range(6)
INSERT_AUTO, INSERT_DIRECT, INSERT_PARTITIONED = 0, 1, 2
PROF_SLOTS = 10
PROF_NAMES = ["insert_direct", "query_direct", "insert_hash", "insert_split", "insert_apply", "query_hash",
              "query_split", "query_test", "query_resolve", "other"]


class Layout(C.Structure):
    _fields_ = [("starts", C.c_void_p), ("n_seqs", C.c_uint64), ("read_len", C.c_uint32)]


class FastxStats(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_bases", C.c_uint64), ("n_windows", C.c_uint64),
                ("n_hits", C.c_uint64), ("n_batches", C.c_uint64), ("seconds_parse", C.c_double),
                ("seconds_total", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BtlbfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("btlbf error %d: %s" % (code, msg))
        self.code = code


_lib = None

# name -> (restype, argtypes)
_P = C.c_void_p
_PROTOS = {
    "btlbf_last_error": (C.c_char_p, []),
    "btlbf_device_count": (C.c_int, []),
    "btlbf_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.c_int]),
    "btlbf_create_shard": (C.c_int, [C.POINTER(_P), C.c_int, C.c_uint64, C.c_uint, C.c_uint, C.c_uint,
                                     C.c_uint, C.c_uint, C.c_int]),
    "btlbf_destroy": (C.c_int, [_P]),
    "btlbf_load": (C.c_int, [C.POINTER(_P), C.c_int, C.c_char_p, C.c_uint, C.c_int]),
    "btlbf_create_from_header": (C.c_int, [C.POINTER(_P), C.c_int, C.c_char_p, C.c_size_t, C.c_uint, C.c_int]),
    "btlbf_get_dfpr": (C.c_double, [_P]),
    "btlbf_set_dfpr": (None, [_P, C.c_double]),
    "btlbf_store": (C.c_int, [_P, C.c_char_p]),
    "btlbf_header": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "btlbf_store_shard": (C.c_int, [_P, C.c_char_p]),
    "btlbf_kind": (C.c_int, [_P]),
    "btlbf_size": (C.c_uint64, [_P]),
    "btlbf_size_bytes": (C.c_uint64, [_P]),
    "btlbf_local_bytes": (C.c_uint64, [_P]),
    "btlbf_hash_num": (C.c_uint, [_P]),
    "btlbf_kmer_size": (C.c_uint, [_P]),
    "btlbf_threshold": (C.c_uint, [_P]),
    "btlbf_get_n_entry": (C.c_uint64, [_P]),
    "btlbf_get_t_entry": (C.c_uint64, [_P]),
    "btlbf_set_n_entry": (None, [_P, C.c_uint64]),
    "btlbf_set_t_entry": (None, [_P, C.c_uint64]),
    "btlbf_device_ptr": (_P, [_P]),
    "btlbf_device": (C.c_int, [_P]),
    "btlbf_clear": (C.c_int, [_P, _P]),
    "btlbf_upload": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "btlbf_download": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "btlbf_set_insert_mode": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "btlbf_release_scratch": (C.c_int, [_P]),
    "btlbf_set_profiling": (C.c_int, [_P, C.c_int]),
    "btlbf_get_profile": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint), C.c_int]),
    "btlbf_set_query_mode": (C.c_int, [_P, C.c_int]),
    "btlbf_set_spaced_seeds": (C.c_int, [_P, C.POINTER(C.c_char_p), C.c_uint, C.c_uint]),
    "btlbf_digest": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "btlbf_compare": (C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    "btlbf_rank_create": (C.c_int, [C.POINTER(_P), _P]),
    "btlbf_rank_destroy": (None, [_P]),
    "btlbf_rank_ones": (C.c_uint64, [_P]),
    "btlbf_rank_words": (C.c_uint64, [_P]),
    "btlbf_rank_download": (C.c_int, [_P, _P]),
    "btlbf_rank_query": (C.c_int, [_P, _P, C.c_uint64, C.c_int, _P, _P, C.c_int, _P]),
    "btlbf_insert_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), C.c_int, C.c_int, C.c_int, _P]),
    "btlbf_contains_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), _P, _P, _P, C.c_int, _P]),
    "btlbf_insert_and_check_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), _P, _P, _P, C.c_int, _P]),
    "btlbf_min_count_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), _P, _P, C.c_int, _P]),
    "btlbf_insert_hashes": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "btlbf_contains_hashes": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, _P]),
    "btlbf_insert_and_check_hashes": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int, _P]),
    "btlbf_min_count_hashes": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, _P]),
    "btlbf_insert_kmers": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "btlbf_contains_kmers": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, _P]),
    "btlbf_hash_kmers": (C.c_int, [C.c_uint, C.c_uint, _P, C.c_uint64, _P, _P, C.c_int, C.c_int, _P]),
    "btlbf_hash_seqs": (C.c_int, [C.c_uint, C.c_uint, C.POINTER(C.c_char_p), C.c_uint, C.c_uint, _P, C.c_uint64,
                                  C.POINTER(Layout), _P, _P, _P, C.c_int, C.c_int, _P]),
    "btlbf_popcount": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "btlbf_filtered_popcount": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "btlbf_positions_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), C.c_uint, _P, _P, C.c_uint64, _P, _P,
                                       _P]),
    "btlbf_route_plan": (C.c_int, [_P, C.c_uint64, C.POINTER(Layout), C.c_uint, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)]),
    "btlbf_route_windows": (C.c_int, [_P, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "btlbf_plan_read_grid": (C.c_int, [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint32)]),
    "btlbf_route_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), C.c_uint64, C.c_uint, C.c_uint, C.c_int, _P, _P,
                                   _P, _P, _P, _P, C.c_uint64, _P, _P]),
    "btlbf_route_geometry": (C.c_int, [_P, C.c_uint64, C.POINTER(Layout), C.c_uint, C.c_uint, C.POINTER(C.c_uint32)]),
    "btlbf_owner_scratch_bytes": (C.c_int, [_P, C.c_uint64, C.POINTER(Layout), C.c_uint, C.c_uint, C.POINTER(C.c_uint64)]),
    "btlbf_apply_routed_bins": (C.c_int, [_P, _P, _P, C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.POINTER(Layout),
                                          C.c_uint, C.c_int, _P, C.c_uint64, _P, _P]),
    "btlbf_apply_routed": (C.c_int, [_P, _P, _P, C.c_uint, C.c_uint64, C.POINTER(Layout), C.c_uint, C.c_int, _P,
                                     C.c_uint64, _P, _P]),
    "btlbf_apply_spill": (C.c_int, [_P, _P, C.c_uint64, C.c_int, _P, C.c_uint64, _P, _P]),
    "btlbf_resolve_seqs": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), _P, C.c_uint64, _P, _P]),
    "btlbf_popcount_bits": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64), C.c_int, _P]),
    "btlbf_insert_positions": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "btlbf_test_positions": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "btlbf_and_answers": (C.c_int, [_P, _P, C.c_uint64, C.c_uint, _P, C.c_int, _P]),
    "btlbf_synth_reads": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_int, _P]),
    "btlbf_microbench": (C.c_int, [_P, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "btlbf_count_per_seq": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(Layout), C.c_uint, _P, _P, C.c_int, C.c_int, _P]),
    "btlbf_fastx_open": (C.c_int, [C.POINTER(_P), C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "btlbf_fastx_open_range": (C.c_int, [C.POINTER(_P), C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int,
                                         C.c_uint64, C.c_uint64]),
    "btlbf_fastx_next": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "btlbf_fastx_records": (C.c_uint64, [_P]),
    "btlbf_fastx_close": (None, [_P]),
    "btlbf_insert_fastx": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(FastxStats)]),
    "btlbf_contains_fastx": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(FastxStats)]),
}
EXPORTS = sorted(_PROTOS)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's).  A process
    must run ONE HIP runtime: whichever copy is loaded first wins, and torch cannot see the GPU through
    the system copy.  So if torch is installed, load its copy before libbtlbf.so resolves the SONAME
    (without importing torch); a later `import torch` then finds its own runtime already in place."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(path=LIB_PATH):
    """Load libbtlbf.so (no GPU needed to load).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `python -m btl_bloomfilter_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise BtlbfError(rc, load().btlbf_last_error().decode("utf-8", "replace"))

"""ctypes binding of libss_verify.so (include/ss_verify.h).

The library is the product: there is no Python or CPU implementation of the verifier in
this package.  If the shared object is missing the import fails loudly; if no MI355X is
visible, creating a context fails loudly (`SsError`).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libss_verify.so")

def process_defaults() -> int:
    """The rule of ss_process_defaults() (include/ss_verify.h, csrc/ss_env.cpp), stated in Python: ask the HIP runtime for
    24 hardware queues unless the caller's environment already says otherwise (or SS_KEEP_ENV is set) -- streams that share
    one of the runtime's 4 default queues serialise, which is what the pipelines here exist to avoid.  An explicit call
    since ABI 2.4 (the library no longer touches the environment when it is loaded); this module makes it when it is
    IMPORTED, because the runtime reads the variable once, at the process's first HIP call (torch's first CUDA call).
    In Python and not through the library: loading libss_verify.so here, before torch is imported, would bind the
    system's libamdhip64 ahead of the one torch ships, and torch then finds no GPU.  Returns the queue count now in the
    environment (0 = unset).  Verdicts never depend on it."""
    if "SS_KEEP_ENV" not in os.environ:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))
    except ValueError:
        return 0


process_defaults()

SS_OK, SS_ERR_ARG, SS_ERR_HIP, SS_ERR_NO_DEVICE, SS_ERR_WORKSPACE, SS_ERR_NOMEM = 0, -1, -2, -3, -4, -5
MODE_LITERAL, MODE_FIXTURE = 0, 1


class SsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libss_verify error %d: %s" % (code, msg))
        self.code = code


class StwoCfg(C.Structure):
    """ss_stwo_cfg"""
    _fields_ = [("n_cols", C.c_uint32), ("trace_log", C.c_uint32), ("lde_log", C.c_uint32),
                ("n_queries", C.c_uint32), ("n_layers", C.c_uint32), ("mode", C.c_uint32),
                ("pow_target", C.c_uint64), ("hash", C.c_uint32), ("flags", C.c_uint32)]


class S101Shape(C.Structure):
    """ss_s101_shape"""
    _fields_ = [("max_layers", C.c_uint32), ("max_path", C.c_uint32)]


class StwoWsLayout(C.Structure):
    """ss_stwo_ws_layout"""
    _fields_ = [("np", C.c_uint64), ("nip", C.c_uint64), ("ctx", C.c_uint64), ("alpha", C.c_uint64),
                ("leaf", C.c_uint64), ("total_words", C.c_uint64), ("c_queries", C.c_uint32),
                ("c_p", C.c_uint32), ("c_p2", C.c_uint32), ("c_fold", C.c_uint32), ("c_m1", C.c_uint32),
                ("n_pow", C.c_uint32), ("top_levels", C.c_uint32), ("has_plan", C.c_uint32), ("plan", C.c_uint64)]


class InputDesc(C.Structure):
    """ss_input_desc: what ss_verify_inputs takes (family, form of one input, where the n inputs lie)"""
    _fields_ = [("family", C.c_uint32), ("form", C.c_uint32), ("source", C.c_uint32), ("text_fmt", C.c_uint32),
                ("cfg", C.c_void_p), ("shape", C.c_void_p), ("n", C.c_size_t), ("items", C.c_void_p),
                ("lens", C.c_void_p), ("blob", C.c_void_p), ("offs", C.c_void_p)]


FAMILY_STARK101, FAMILY_STWO = 1, 2
FORM_RECORDS, FORM_SHARED_RECORDS, FORM_MINIMAL_RECORDS, FORM_TEXT = 0, 1, 2, 3
SRC_HOST, SRC_PINNED, SRC_FILES = 0, 1, 2


class IngestStats(C.Structure):
    """ss_ingest_stats"""
    _fields_ = [("read_s", C.c_double), ("parse_s", C.c_double), ("total_s", C.c_double),
                ("text_bytes", C.c_uint64), ("record_bytes", C.c_uint64), ("threads", C.c_uint32),
                ("host_parsed", C.c_uint32)]


TEXT_AUTO, TEXT_JSON, TEXT_WIT, TEXT_JSON_SHARED, TEXT_JSON_MINIMAL = 0, 1, 2, 3, 4
STATUS_CONFIG_MISMATCH, STATUS_MALFORMED = 1, 2

EXPORTS = [
    "ss_version", "ss_last_error", "ss_device_count", "ss_abi_sizeof_cfg", "ss_abi_sizeof_shape", "ss_process_defaults",
    "ss_verify_inputs", "ss_stwo_ws_layout_of", "ss_stwo_read_intermediates", "ss_stwo_parse", "ss_s101_parse",
    "ss_stwo_verify_texts", "ss_stwo_verify_files", "ss_s101_verify_texts", "ss_s101_verify_files",
    "ss_s101_record_words", "ss_s101_batch_words", "ss_s101_workspace_bytes", "ss_s101_pack",
    "ss_stwo_record_words", "ss_stwo_batch_words", "ss_stwo_workspace_bytes", "ss_stwo_pack",
    "ss_ctx_create", "ss_ctx_destroy", "ss_s101_verify_batch_dev", "ss_stwo_verify_batch_dev",
    "ss_s101_verify_phase_dev", "ss_stwo_verify_phase_dev", "ss_stwo_pack_dev",
    "ss_s101_verify_records", "ss_stwo_verify_records", "ss_ctx_set_timing", "ss_ctx_collect_timing",
    "ss_selftest", "ss_stwo_write_text", "ss_stwo_text_is_canonical", "ss_stwo_read_texts",
    "ss_s101_write_text", "ss_s101_text_is_canonical", "ss_s101_pack_dev", "ss_s101_read_texts",
    "ss_stwo_shared_fixed_words", "ss_stwo_shared_max_words", "ss_stwo_shared_counts", "ss_stwo_share_record",
    "ss_stwo_unshare_record", "ss_stwo_expand_shared_dev", "ss_stwo_verify_shared_records", "ss_stwo_write_shared_text",
    "ss_stwo_minimal_fixed_words", "ss_stwo_minimal_max_words", "ss_stwo_minimal_counts", "ss_stwo_minimise_record",
    "ss_stwo_minimal_batch_words", "ss_stwo_minimal_workspace_bytes", "ss_stwo_verify_minimal_dev",
    "ss_host_register", "ss_host_unregister", "ss_stwo_verify_texts_pinned", "ss_stwo_verify_records_pinned", "ss_stwo_verify_shared_records_pinned",
    "ss_stwo_verify_minimal_records_pinned", "ss_s101_read_intermediates", "ss_kat", "ss_stwo_verify_minimal_records", "ss_stwo_parse_minimal", "ss_stwo_parse_minimal_route", "ss_stwo_minimal_from_capacity", "ss_stwo_minimal_to_capacity", "ss_stwo_write_minimal_text", "ss_stwo_verify_minimal_texts", "ss_stwo_verify_minimal_texts_pinned", "ss_s101_verify_texts_pinned",
]

_lib = None


def lib() -> C.CDLL:
    """Load the HIP library.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
    pp = C.POINTER(C.c_void_p)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)
    sig("ss_version", C.c_int)
    sig("ss_last_error", C.c_char_p)
    sig("ss_device_count", C.c_int)
    sig("ss_abi_sizeof_cfg", sz)
    sig("ss_abi_sizeof_shape", sz)
    if L.ss_abi_sizeof_cfg() != C.sizeof(StwoCfg) or L.ss_abi_sizeof_shape() != C.sizeof(S101Shape):
        raise ImportError("%s: struct sizes differ from this binding (ss_stwo_cfg %d vs %d)"
                          % (LIB_PATH, L.ss_abi_sizeof_cfg(), C.sizeof(StwoCfg)))
    if (L.ss_version() >> 16) != 2:
        raise ImportError("%s is ABI version 0x%08x; this binding needs 2.x: rebuild" % (LIB_PATH, L.ss_version()))
    sp = C.POINTER(S101Shape)
    sig("ss_s101_record_words", sz, sp)
    sig("ss_s101_batch_words", sz, sp, sz)
    sig("ss_s101_workspace_bytes", sz, sp, sz)
    sig("ss_s101_pack", C.c_int, sp, sz, pp, vp)
    cp = C.POINTER(StwoCfg)
    sig("ss_stwo_record_words", sz, cp)
    sig("ss_stwo_batch_words", sz, cp, sz)
    sig("ss_stwo_workspace_bytes", sz, cp, sz)
    sig("ss_stwo_pack", C.c_int, cp, sz, pp, vp)
    sig("ss_ctx_create", C.c_int, C.c_int, pp)
    sig("ss_ctx_destroy", None, vp)
    sig("ss_s101_verify_batch_dev", C.c_int, vp, sp, sz, vp, vp, sz, vp, vp, vp)
    sig("ss_stwo_verify_batch_dev", C.c_int, vp, cp, sz, vp, vp, sz, vp, vp, vp)
    sig("ss_s101_verify_phase_dev", C.c_int, vp, sp, sz, vp, vp, sz, vp, vp, C.c_int, vp)
    sig("ss_stwo_verify_phase_dev", C.c_int, vp, cp, sz, vp, vp, sz, vp, vp, C.c_int, vp)
    sig("ss_stwo_pack_dev", C.c_int, vp, cp, sz, vp, vp, vp)
    sig("ss_s101_pack_dev", C.c_int, vp, sp, sz, vp, vp, vp)
    sig("ss_process_defaults", C.c_int)
    sig("ss_verify_inputs", C.c_int, vp, C.POINTER(InputDesc), vp, vp)
    sig("ss_s101_verify_records", C.c_int, vp, sp, sz, pp, vp)
    sig("ss_stwo_verify_records", C.c_int, vp, cp, sz, pp, vp)
    sig("ss_ctx_set_timing", C.c_int, vp, C.c_int)
    sig("ss_ctx_collect_timing", C.c_int, vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), u32p)
    cpp, szp, stp = C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(IngestStats)
    sig("ss_stwo_parse", C.c_int, cp, C.c_char_p, sz, C.c_int, vp)
    sig("ss_s101_parse", C.c_int, C.c_char_p, sz, C.c_int, sp, vp)
    sig("ss_stwo_verify_texts", C.c_int, vp, cp, sz, cpp, szp, C.c_int, vp, stp)
    sig("ss_stwo_verify_files", C.c_int, vp, cp, sz, cpp, C.c_int, vp, stp)
    sig("ss_s101_verify_texts", C.c_int, vp, sz, cpp, szp, C.c_int, vp, stp)
    sig("ss_s101_verify_files", C.c_int, vp, sz, cpp, C.c_int, vp, stp)
    sig("ss_stwo_read_texts", C.c_int, vp, cp, sz, cpp, szp, C.c_int, vp, vp)
    sig("ss_s101_read_texts", C.c_int, vp, sz, cpp, szp, C.c_int, vp, vp)
    sig("ss_stwo_ws_layout_of", C.c_int, cp, sz, C.POINTER(StwoWsLayout))
    sig("ss_stwo_read_intermediates", C.c_int, vp, cp, sz, vp, sz, vp, vp, vp, vp, vp, vp)
    sig("ss_s101_read_intermediates", C.c_int, vp, sp, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp)
    sig("ss_selftest", C.c_int, vp, C.c_int, sz, vp, vp)
    sig("ss_kat", C.c_int, vp, C.c_int, sz, vp, sz, vp, sz)
    sig("ss_stwo_verify_texts_pinned", C.c_int, vp, cp, sz, vp, vp, szp, C.c_int, vp, stp)
    sig("ss_s101_verify_texts_pinned", C.c_int, vp, sz, vp, vp, szp, C.c_int, vp, stp)
    sig("ss_host_register", C.c_int, vp, vp, sz)
    sig("ss_host_unregister", C.c_int, vp, vp)
    sig("ss_stwo_verify_records_pinned", C.c_int, vp, cp, sz, vp, vp)
    sig("ss_stwo_verify_shared_records_pinned", C.c_int, vp, cp, sz, vp, vp, vp)
    sig("ss_stwo_verify_minimal_records_pinned", C.c_int, vp, cp, sz, vp, vp, vp)
    sig("ss_stwo_write_text", sz, cp, vp, C.c_int, C.c_int, vp, sz)
    sig("ss_stwo_text_is_canonical", C.c_int, cp, C.c_char_p, sz, C.c_int, vp)
    sig("ss_s101_write_text", sz, vp, C.c_int, C.c_int, vp, sz)
    sig("ss_s101_text_is_canonical", C.c_int, C.c_char_p, sz, C.c_int, vp)
    sig("ss_stwo_shared_fixed_words", sz, cp)
    sig("ss_stwo_shared_max_words", sz, cp)
    sig("ss_stwo_shared_counts", C.c_int, cp, vp, vp)
    sig("ss_stwo_share_record", C.c_int, cp, vp, vp, vp, sz, szp)
    sig("ss_stwo_unshare_record", C.c_int, cp, vp, sz, vp)
    sig("ss_stwo_expand_shared_dev", C.c_int, vp, cp, sz, vp, vp, vp, vp, vp)
    sig("ss_stwo_verify_shared_records", C.c_int, vp, cp, sz, pp, szp, vp)
    sig("ss_stwo_write_shared_text", sz, cp, vp, sz, C.c_int, vp, sz)
    sig("ss_stwo_minimal_fixed_words", sz, cp)
    sig("ss_stwo_minimal_max_words", sz, cp)
    sig("ss_stwo_minimal_counts", C.c_int, cp, vp, vp)
    sig("ss_stwo_minimise_record", C.c_int, cp, vp, vp, vp, sz, szp)
    sig("ss_stwo_minimal_batch_words", sz, cp, sz)
    sig("ss_stwo_minimal_workspace_bytes", sz, cp, sz)
    sig("ss_stwo_verify_minimal_dev", C.c_int, vp, cp, sz, vp, vp, vp, vp, sz, vp, vp, C.c_int, vp)
    sig("ss_stwo_verify_minimal_records", C.c_int, vp, cp, sz, pp, szp, vp)
    sig("ss_stwo_parse_minimal", C.c_int, cp, C.c_char_p, sz, vp, sz, szp)
    sig("ss_stwo_parse_minimal_route", C.c_int, cp, C.c_char_p, sz, C.c_int, vp, sz, szp)
    sig("ss_stwo_minimal_from_capacity", C.c_int, cp, vp, vp, sz, szp)
    sig("ss_stwo_minimal_to_capacity", C.c_int, cp, vp, sz, vp)
    sig("ss_stwo_write_minimal_text", sz, cp, vp, sz, C.c_int, vp, sz)
    sig("ss_stwo_verify_minimal_texts", C.c_int, vp, cp, sz, cpp, szp, vp, stp)
    sig("ss_stwo_verify_minimal_texts_pinned", C.c_int, vp, cp, sz, vp, vp, szp, vp, stp)
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise SsError(rc, lib().ss_last_error().decode("utf-8", "replace"))
    return rc

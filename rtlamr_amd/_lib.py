"""ctypes binding of libamrdemod.so (include/amrdemod.h).  Fails loudly when the library is
missing or no gfx950 device is present -- there is deliberately no fallback path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libamrdemod.so")

AMR_OK, AMR_EINVAL, AMR_ENOMEM, AMR_EHIP, AMR_ENODEV, AMR_EOVERFLOW = 0, -1, -2, -3, -4, -5

# every symbol include/amrdemod.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "amr_create", "amr_plan", "amr_destroy", "amr_device_count", "amr_reset", "amr_get_geometry", "amr_preamble_id", "amr_get_mag_lut", "amr_r900_enable", "amr_set_validation",
    "amr_set_stream", "amr_set_block_base", "amr_decode_batch", "amr_decode_batch_device", "amr_submit_device", "amr_collect", "amr_set_deferral", "amr_flush", "amr_submit_host", "amr_host_alloc", "amr_host_free", "amr_result_device", "amr_prime",
    "amr_comm_test_loopback", "amr_halo_bytes", "amr_prime_blocks", "amr_get_stale_carry", "amr_set_stale_carry", "amr_copy_quantized", "amr_set_timing", "amr_get_timing", "amr_strerror",
    "amr_last_error", "amr_describe", "amr_dev_alloc", "amr_dev_free", "amr_dev_upload", "amr_dev_download",
    "amr_dev_sync", "amr_synth_noise", "amr_synth_uniform", "amr_synth_plant",
    "amr_comm_unique_id", "amr_comm_init", "amr_comm_init_all", "amr_gather_hits_all", "amr_comm_check_all", "amr_comm_destroy", "amr_comm_ranks", "amr_gather_hits", "amr_gather_wait", "amr_gather_fetch",
    "amr_gather_slot_bytes", "amr_gather_wire_bytes", "amr_gather_two_phase", "amr_gather_pack_host", "amr_gather_unpack",
]


class AmrError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


class AmrProtocol(C.Structure):
    _fields_ = [("preamble", C.c_char_p), ("data_rate", C.c_int32), ("chip_length", C.c_int32),
                ("preamble_symbols", C.c_int32), ("packet_symbols", C.c_int32)]


class AmrGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "data_rate", "chip_length", "symbol_length", "sample_rate", "preamble_symbols", "packet_symbols",
        "preamble_length", "packet_length", "block_size", "block_size2", "buffer_length", "n_preambles",
        "pkt_bytes")]


class AmrResult(C.Structure):
    _fields_ = [("n_preambles", C.c_uint32), ("pkt_bytes", C.c_uint32), ("n_hits", C.c_uint64),
                ("preamble_offset", C.POINTER(C.c_uint64)), ("hit_block", C.POINTER(C.c_uint64)),
                ("hit_idx", C.POINTER(C.c_uint32)), ("pkt", C.POINTER(C.c_uint8)),
                ("r900_preamble", C.c_int32), ("r900_digits", C.POINTER(C.c_uint8)),
                ("n_hits_searched", C.c_uint64), ("first_block", C.c_uint64), ("n_blocks", C.c_uint64)]


class AmrCrcCheck(C.Structure):
    _fields_ = [("init", C.c_uint16), ("poly", C.c_uint16), ("residue", C.c_uint16), ("n_spans", C.c_uint16),
                ("span_off", C.c_uint16 * 2), ("span_len", C.c_uint16 * 2)]


class AmrValidator(C.Structure):
    _fields_ = [("n_checks", C.c_int32), ("dedupe_bytes", C.c_int32), ("checks", AmrCrcCheck * 2)]


class AmrGathered(C.Structure):
    _fields_ = [("n_true", C.c_uint64), ("n_hits", C.c_uint64), ("n_preambles", C.c_uint32), ("seq", C.c_uint64),
                ("preamble_offset", C.POINTER(C.c_uint64)), ("hit_block", C.POINTER(C.c_uint64)),
                ("hit_idx", C.POINTER(C.c_uint32))]


class AmrTiming(C.Structure):
    _fields_ = [("demod_ms", C.c_float), ("search_ms", C.c_float), ("total_ms", C.c_float)]


def build(force: bool = False) -> str:
    """Compile libamrdemod.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "amrdemod.h"))
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        jobs = str(min(8, os.cpu_count() or 1))     # one object per kernel family (csrc/launch.h)
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j", jobs, "libamrdemod.so"] + (["-B"] if force else []))
    return SO_PATH


_lib = None


def lib() -> C.CDLL:
    """Load the library (no GPU needed to load; amr_create is what needs the device)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("AMR_LIB_OVERRIDE", SO_PATH)   # developer hook: diagnostic builds of the same ABI
    if not os.path.exists(path):
        raise AmrError(AMR_ENODEV, "rtlamr_amd",
                       f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    L = C.CDLL(path)
    vp, u8p = C.c_void_p, C.POINTER(C.c_uint8)
    L.amr_create.argtypes = [C.POINTER(AmrProtocol), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.amr_plan.argtypes = [C.POINTER(AmrProtocol), C.c_int32, C.POINTER(AmrGeometry), C.POINTER(C.c_int32)]
    L.amr_destroy.argtypes = [vp]
    L.amr_reset.argtypes = [vp]
    L.amr_get_geometry.argtypes = [vp, C.POINTER(AmrGeometry)]
    L.amr_preamble_id.argtypes = [vp, C.c_int32]
    L.amr_preamble_id.restype = C.c_int32
    L.amr_get_mag_lut.argtypes = [vp, C.POINTER(C.c_float)]
    L.amr_set_stream.argtypes = [vp, vp]
    L.amr_set_block_base.argtypes = [vp, C.c_uint64]
    L.amr_decode_batch.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(AmrResult)]
    L.amr_decode_batch_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(AmrResult)]
    L.amr_submit_device.argtypes = [vp, vp, C.c_size_t]
    L.amr_collect.argtypes = [vp, C.POINTER(AmrResult)]
    L.amr_set_deferral.argtypes = [vp, C.c_int32]
    L.amr_flush.argtypes = [vp, C.POINTER(AmrResult)]
    L.amr_prime.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    L.amr_comm_test_loopback.argtypes = [C.c_int32]
    L.amr_get_stale_carry.argtypes = [vp, C.POINTER(C.c_uint8)]
    L.amr_set_stale_carry.argtypes = [vp, C.c_uint8]
    L.amr_halo_bytes.argtypes = [vp]
    L.amr_halo_bytes.restype = C.c_size_t
    L.amr_prime_blocks.argtypes = [vp]
    L.amr_prime_blocks.restype = C.c_size_t
    L.amr_copy_quantized.argtypes = [vp, vp, C.c_size_t]
    L.amr_r900_enable.argtypes = [vp, C.c_int32]
    L.amr_set_validation.argtypes = [vp, C.c_int32, C.POINTER(AmrValidator)]
    L.amr_submit_host.argtypes = [vp, C.c_void_p, C.c_size_t, C.c_size_t]
    L.amr_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.amr_host_free.argtypes = [C.c_void_p]
    L.amr_result_device.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.amr_set_timing.argtypes = [vp, C.c_int32]
    L.amr_get_timing.argtypes = [vp, C.POINTER(AmrTiming)]
    L.amr_strerror.argtypes = [C.c_int]
    L.amr_strerror.restype = C.c_char_p
    L.amr_last_error.restype = C.c_char_p
    L.amr_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.amr_dev_alloc.argtypes = [C.c_int32, C.c_size_t, C.POINTER(vp)]
    L.amr_dev_free.argtypes = [C.c_int32, vp]
    L.amr_dev_upload.argtypes = [C.c_int32, vp, vp, C.c_size_t]
    L.amr_dev_download.argtypes = [C.c_int32, vp, vp, C.c_size_t]
    L.amr_dev_sync.argtypes = [C.c_int32]
    L.amr_synth_noise.argtypes = [C.c_int32, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    L.amr_synth_uniform.argtypes = [C.c_int32, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    L.amr_synth_plant.argtypes = [C.c_int32, vp, C.c_uint64, C.c_uint64, C.c_int32, C.c_uint32, vp, vp,
                                  C.c_uint32, C.c_uint32, vp, vp]
    L.amr_comm_unique_id.argtypes = [vp]
    L.amr_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]
    L.amr_comm_destroy.argtypes = [vp]
    L.amr_comm_init_all.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_uint64]
    L.amr_gather_hits_all.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(C.c_uint64)]
    L.amr_comm_check_all.argtypes = [C.POINTER(vp), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_uint64]
    L.amr_comm_ranks.argtypes = [vp, C.POINTER(C.c_int32)]
    L.amr_gather_hits.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.amr_gather_wait.argtypes = [vp]
    L.amr_gather_fetch.argtypes = [vp, C.c_uint64, C.c_int32, C.POINTER(AmrGathered)]
    L.amr_gather_slot_bytes.argtypes = [C.c_uint64]
    L.amr_gather_slot_bytes.restype = C.c_size_t
    L.amr_gather_wire_bytes.argtypes = [C.c_uint64]
    L.amr_gather_wire_bytes.restype = C.c_size_t
    L.amr_device_count.argtypes = [C.POINTER(C.c_int32)]
    L.amr_gather_two_phase.argtypes = [C.c_uint64]
    L.amr_gather_two_phase.restype = C.c_int32
    L.amr_gather_pack_host.argtypes = [C.POINTER(AmrResult), C.c_uint64, C.c_uint64, vp, C.c_size_t]
    L.amr_gather_unpack.argtypes = [vp, C.c_size_t, C.POINTER(AmrGathered)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("amr_preamble_id", "amr_comm_test_loopback", "amr_halo_bytes", "amr_prime_blocks", "amr_get_stale_carry", "amr_set_stale_carry", "amr_strerror", "amr_last_error",
                        "amr_gather_slot_bytes", "amr_gather_wire_bytes", "amr_gather_two_phase"):
            fn.restype = C.c_int
    _lib = L
    return L


def device_count() -> int:
    """gfx950 devices visible to this process (amr_device_count); 0 without a GPU."""
    n = C.c_int32(0)
    check(lib().amr_device_count(C.byref(n)), "amr_device_count")
    return int(n.value)


def check(status: int, where: str) -> None:
    if status != AMR_OK:
        L = lib()
        raise AmrError(status, where, f"{L.amr_strerror(status).decode()}: {L.amr_last_error().decode()}")

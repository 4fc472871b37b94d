"""ctypes binding of the C-ABI in include/ansx.h (libansx.so, built in-tree by csrc/Makefile).

There is no fallback: if the shared library is missing this module raises, and if no gfx950
device is usable ansx_init reports ANSX_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libansx.so")

FOLD, RFOLD, MSB, INT = 0, 1, 2, 3
FLAG_COMPACT_ALPHABET = 1
OK, ERR_ARG, ERR_CAPACITY, ERR_FORMAT, ERR_HIP, ERR_NO_DEVICE, ERR_DOMAIN, ERR_MODEL = range(8)
SINGLE_STREAM = 0xFFFFFFFF
NO_CHECKPOINTS = 0xFFFFFFFF
DEFAULT_BLOCK_INTS = 16384
DEFAULT_CKPT_INTERVAL = 1024
MAX_FIDELITY = 7
GEN_UNIFORM, GEN_GEOMETRIC, GEN_ZIPF = 0, 1, 2

EXPORTS = [
    "ansx_init", "ansx_destroy", "ansx_strerror", "ansx_last_hip_error", "ansx_codec_name",
    "ansx_bound", "ansx_encode", "ansx_decode", "ansx_encode_dev", "ansx_decode_dev",
    "ansx_container_info", "ansx_profile_enable", "ansx_profile_reset", "ansx_profile_get",
    "ansx_workspace_bytes", "ansx_host_log2", "ansx_selftest_log2", "ansx_selftest_div", "ansx_debug_set",
    "ansx_generate_dev", "ansx_generate_host", "ansx_last_encode_stats", "ansx_merge_containers_dev",
    "ansx_zipf_from_uniform", "ansx_gather_containers", "ansx_last_gather_ranks",
]


class Opts(C.Structure):
    _fields_ = [("block_ints", C.c_uint32), ("ckpt_interval", C.c_uint32), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class ContainerHeader(C.Structure):
    _fields_ = [("magic", C.c_uint8 * 6), ("max_present_m1", C.c_uint16), ("kind", C.c_uint32), ("fidelity", C.c_uint32),
                ("n", C.c_uint64), ("block_ints", C.c_uint32), ("ckpt_interval", C.c_uint32),
                ("nblocks", C.c_uint32), ("max_log2_frame", C.c_uint32), ("max_nsyms", C.c_uint32),
                ("ckpts_per_block", C.c_uint32), ("payload_bytes", C.c_uint64),
                ("payload_offset", C.c_uint64)]


class EncodeStats(C.Structure):
    _fields_ = [("max_nsyms", C.c_uint32), ("max_log2_frame", C.c_uint32), ("near_threshold_decisions", C.c_uint32),
                ("path", C.c_uint32), ("host_redecided", C.c_uint32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("launches", C.c_uint64)]


def build_library(force=False):
    """Compile csrc/ansx.hip for gfx950 into ans_large_alphabet_amd/libansx.so (hipcc)."""
    args = ["make", "-s", "-C", os.path.join(PKG_DIR, "csrc")]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def _preload_shared_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's); if libansx.so pulled in
    /opt/rocm's copy first, a later `import torch` would load a second runtime and find no GPU.
    So when torch is installed, load ITS runtime first (without importing torch); libansx.so's
    NEEDED libamdhip64.so.7 then binds to that already-loaded object."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # a C++-only deployment simply uses /opt/rocm's runtime


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libansx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ans_large_alphabet_amd/csrc` (needs hipcc); there is no CPU fallback")
    _preload_shared_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.ansx_init.restype = C.c_int
    L.ansx_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.ansx_destroy.restype = None
    L.ansx_destroy.argtypes = [vp]
    L.ansx_strerror.restype = C.c_char_p
    L.ansx_strerror.argtypes = [C.c_int]
    L.ansx_last_hip_error.restype = C.c_int
    L.ansx_last_hip_error.argtypes = [vp]
    L.ansx_codec_name.restype = C.c_int
    L.ansx_codec_name.argtypes = [C.c_int, C.c_int, C.c_char_p, sz]
    L.ansx_bound.restype = sz
    L.ansx_bound.argtypes = [C.c_int, C.c_int, sz, C.POINTER(Opts)]
    L.ansx_encode.restype = C.c_int
    L.ansx_encode.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Opts)]
    L.ansx_decode.restype = C.c_int
    L.ansx_decode.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, sz, C.POINTER(Opts)]
    L.ansx_encode_dev.restype = C.c_int
    L.ansx_encode_dev.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Opts), vp]
    L.ansx_decode_dev.restype = C.c_int
    L.ansx_decode_dev.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, sz, C.POINTER(Opts), vp]
    L.ansx_container_info.restype = C.c_int
    L.ansx_container_info.argtypes = [vp, sz, C.POINTER(ContainerHeader)]
    L.ansx_profile_enable.restype = C.c_int
    L.ansx_profile_enable.argtypes = [vp, C.c_int]
    L.ansx_profile_reset.restype = C.c_int
    L.ansx_profile_reset.argtypes = [vp]
    L.ansx_profile_get.restype = C.c_int
    L.ansx_profile_get.argtypes = [vp, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]
    L.ansx_workspace_bytes.restype = sz
    L.ansx_workspace_bytes.argtypes = [vp]
    L.ansx_generate_dev.restype = C.c_int
    L.ansx_generate_dev.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_uint64, vp, sz, vp]
    L.ansx_generate_host.restype = C.c_int
    L.ansx_generate_host.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_uint64, vp, sz]
    L.ansx_merge_containers_dev.restype = C.c_int
    L.ansx_merge_containers_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.c_int, vp, sz, C.POINTER(sz), vp]
    L.ansx_last_encode_stats.restype = C.c_int
    L.ansx_last_encode_stats.argtypes = [vp, C.POINTER(EncodeStats)]
    L.ansx_last_gather_ranks.restype = C.c_int
    L.ansx_last_gather_ranks.argtypes = [vp]
    L.ansx_debug_set.restype = C.c_int
    L.ansx_debug_set.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.ansx_zipf_from_uniform.restype = C.c_int
    L.ansx_zipf_from_uniform.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.ansx_host_log2.restype = C.c_double
    L.ansx_host_log2.argtypes = [C.c_double]
    L.ansx_selftest_log2.restype = C.c_int
    L.ansx_selftest_log2.argtypes = [vp, vp, vp, sz]
    L.ansx_selftest_div.restype = C.c_int
    L.ansx_selftest_div.argtypes = [vp, vp, vp, vp, sz]
    _lib = L
    return L


class AnsxError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().ansx_strerror(status).decode()
        super().__init__("%s: %s (status %d)" % (what, msg, status))

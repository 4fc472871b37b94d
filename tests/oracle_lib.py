"""ctypes bindings for the CHECKERS (test infrastructure only):

* ``oracle/libans_oracle.so``  — clean-room C restatement (oracle/ans_oracle.c)
* ``oracle/_ref/libans_ref.so`` — the unmodified reference headers (oracle/ref_shim.cpp), when built

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FOLD, RFOLD, MSB, INT = 0, 1, 2, 3


class OracleInfo(C.Structure):
    _fields_ = [
        ("max_sym", C.c_uint32),
        ("log2_frame", C.c_uint32),
        ("header_bytes", C.c_uint32),
        ("prelude_bytes", C.c_uint32),
        ("interp_bits", C.c_uint32),
        ("reorder_flag", C.c_uint32),
        ("sigma", C.c_uint64),
        ("final_states", C.c_uint64 * 4),
        ("present_syms", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class PaInfo(C.Structure):
    _fields_ = [("sigma", C.c_uint32), ("universe", C.c_uint32), ("header_bytes", C.c_uint32),
                ("interp_bits", C.c_uint32)]


def build_oracle():
    """Compile the C restatement (and, when /root/reference is present, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def _load_oracle():
    path = os.path.join(ORACLE_DIR, "libans_oracle.so")
    if not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    lib.ans_oracle_fold.restype = C.c_uint32
    lib.ans_oracle_fold.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ans_oracle_unfold.restype = C.c_uint32
    lib.ans_oracle_unfold.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ans_oracle_adjust_freqs.restype = C.c_uint64
    lib.ans_oracle_adjust_freqs.argtypes = [_u64p, C.c_size_t, C.c_uint32, _u32p]
    lib.ans_oracle_write_prelude.restype = C.c_size_t
    lib.ans_oracle_write_prelude.argtypes = [_u32p, C.c_size_t, C.c_uint64, _u8p, C.POINTER(C.c_uint32)]
    lib.ans_oracle_read_prelude.restype = C.c_size_t
    lib.ans_oracle_read_prelude.argtypes = [_u8p, _u32p, C.POINTER(C.c_uint32)]
    lib.ans_oracle_encode.restype = C.c_size_t
    lib.ans_oracle_encode.argtypes = [C.c_int, C.c_uint32, _u32p, C.c_size_t, _u8p, C.c_size_t,
                                      C.POINTER(OracleInfo), C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_size_t)]
    lib.ans_oracle_decode.restype = C.c_int
    lib.ans_oracle_decode.argtypes = [C.c_int, C.c_uint32, _u8p, C.c_size_t, _u32p, C.c_size_t, C.c_int]
    lib.ans_oracle_adjust_freqs_ex.restype = C.c_uint64
    lib.ans_oracle_adjust_freqs_ex.argtypes = [_u64p, C.c_size_t, C.c_uint32, _u32p, C.c_int]
    lib.ans_oracle_pa_encode.restype = C.c_size_t
    lib.ans_oracle_pa_encode.argtypes = [C.c_int, C.c_uint32, _u32p, C.c_size_t, _u8p, C.c_size_t,
                                         C.POINTER(PaInfo), C.POINTER(OracleInfo), C.c_size_t, C.c_void_p,
                                         C.c_void_p, C.POINTER(C.c_size_t)]
    lib.ans_oracle_pa_decode.restype = C.c_int
    lib.ans_oracle_pa_decode.argtypes = [C.c_int, C.c_uint32, _u8p, C.c_size_t, _u32p, C.c_size_t]
    lib.ans_oracle_prelude_hints.restype = None
    lib.ans_oracle_prelude_hints.argtypes = [_u8p, _u32p]
    lib.ans_oracle_bound.restype = C.c_size_t
    lib.ans_oracle_bound.argtypes = [C.c_int, C.c_uint32, C.c_size_t]
    lib.ans_oracle_bound_int.restype = C.c_size_t
    lib.ans_oracle_bound_int.argtypes = [C.c_size_t, C.c_uint32]
    lib.ans_oracle_blocks_digest.restype = C.c_int
    lib.ans_oracle_blocks_digest.argtypes = [C.c_int, C.c_uint32, _u32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _u32p, _u64p, _u64p,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ans_oracle_hash_spans.restype = None
    lib.ans_oracle_hash_spans.argtypes = [_u8p, _u64p, C.c_size_t, C.c_int, _u64p]
    return lib


def _load_ref(name="libans_ref.so"):
    path = os.path.join(ORACLE_DIR, "_ref", name)
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_encode.restype = C.c_size_t
    lib.ref_encode.argtypes = [C.c_int, C.c_int, _u32p, C.c_size_t, _u8p, C.c_size_t]
    lib.ref_decode.restype = None
    lib.ref_decode.argtypes = [C.c_int, C.c_int, _u8p, C.c_size_t, _u32p, C.c_size_t]
    lib.ref_adjust_freqs.restype = C.c_uint64
    lib.ref_adjust_freqs.argtypes = [_u64p, C.c_size_t, C.c_uint32, _u32p]
    lib.ref_serialize_prelude.restype = C.c_size_t
    lib.ref_serialize_prelude.argtypes = [_u32p, C.c_size_t, C.c_uint64, _u8p]
    lib.ref_load_prelude.restype = C.c_size_t
    lib.ref_load_prelude.argtypes = [_u8p, _u32p]
    if hasattr(lib, "ref_blocks_mt"):
        lib.ref_blocks_mt.restype = C.c_int
        lib.ref_blocks_mt.argtypes = [C.c_int, C.c_int, _u32p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_size_t)]
    if hasattr(lib, "ref_zipf_trace"):
        lib.ref_zipf_trace.restype = C.c_size_t
        lib.ref_zipf_trace.argtypes = [C.c_uint32, C.c_double, C.c_uint32, C.c_size_t, _u32p, _u32p,
                                       np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"), C.c_size_t,
                                       C.POINTER(C.c_size_t)]
    if hasattr(lib, "ref_pa_encode"):
        lib.ref_pa_encode.restype = C.c_size_t
        lib.ref_pa_encode.argtypes = [C.c_int, C.c_int, _u32p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    if hasattr(lib, "ref_bwtmtf"):
        lib.ref_bwtmtf.restype = C.c_size_t
        lib.ref_bwtmtf.argtypes = [np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), C.c_size_t, C.c_size_t, _u32p]
    for fn in ("ref_fold_mapping", "ref_fold_undo_mapping", "ref_fold_exception_bytes"):
        getattr(lib, fn).restype = C.c_uint32
        getattr(lib, fn).argtypes = [C.c_int, C.c_uint32]
    return lib


_oracle = None
_ref = {}


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = _load_oracle()
    return _oracle


def ref(name="libans_ref.so"):
    if name not in _ref:
        _ref[name] = _load_ref(name)
    return _ref[name]


def have_ref():
    return ref() is not None


# ---------------------------------------------------------------- convenience wrappers

def host_threads():
    """Threads a whole-list oracle pass may use: the affinity mask, capped at 32."""
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(32, os.cpu_count() or 1))


def oracle_blocks_digest(kind, f, data, block_ints, ckpt_interval, threads=None):
    """Every block of `data` through the oracle on native threads: (sizes u32[nb], stream hashes u64[nb], restart-point
    digests u64[nb], max log2 frame, max alphabet).  See ans_oracle.h."""
    data = np.ascontiguousarray(data, dtype=np.uint32)
    nb = (data.size + block_ints - 1) // block_ints
    sizes = np.zeros(nb, dtype=np.uint32)
    sh = np.zeros(nb, dtype=np.uint64)
    cd = np.zeros(nb, dtype=np.uint64)
    lg, ns = C.c_uint32(0), C.c_uint32(0)
    rc = oracle().ans_oracle_blocks_digest(kind, f, data, data.size, block_ints, ckpt_interval, threads or host_threads(), sizes, sh, cd,
                                           C.byref(lg), C.byref(ns))
    if rc != 0:
        raise RuntimeError("oracle block pass failed")
    return sizes, sh, cd, lg.value, ns.value


def hash_spans(buf, offs, threads=None):
    """ans_oracle_hash of every span [offs[i], offs[i + 1]) of the byte buffer."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    out = np.zeros(offs.size - 1, dtype=np.uint64)
    oracle().ans_oracle_hash_spans(buf, offs, offs.size - 1, threads or host_threads(), out)
    return out


def ckpt_digest(states, offs):
    """The restart-point digest of ans_oracle_blocks_digest from parsed arrays: states [nb, k, 4] u64, offs [nb, k] u32
    (unused slots zero: they add nothing)."""
    nb, k = offs.shape
    with np.errstate(over="ignore"):
        s_idx = np.arange(k, dtype=np.uint64)
        w_state = (np.uint64(2654435761) + np.uint64(2) * (np.uint64(4) * s_idx[:, None] + np.arange(4, dtype=np.uint64)[None, :]))
        w_off = np.uint64(40503) + np.uint64(2) * s_idx
        d = (states.astype(np.uint64) * w_state[None, :, :]).sum(axis=(1, 2), dtype=np.uint64)
        d = d + (offs.astype(np.uint64) * w_off[None, :]).sum(axis=1, dtype=np.uint64)
    return d


def ref_bwtmtf(T, n):
    """src/generate_bwtmtf.cpp:142-173 on the parsed text T (ints, terminated by 0) through oracle/_ref: the reference's
    own suffix sort (include/qsufsort.hpp, unmodified), its BWT and move-to-front statements.  Returns min(len(T) - 1, n)
    ranks."""
    T = np.ascontiguousarray(T, dtype=np.int32)
    out = np.zeros(max(1, min(T.size - 1, n)), dtype=np.uint32)
    m = ref().ref_bwtmtf(T, T.size, n, out)
    return out[:m].copy()


def ansint_large_list(n, vmax, seed, shape):
    """Lists for plain ANSint whose values go far beyond the dense 16384-symbol model (tests/golden/ansint_large.json)."""
    rng = np.random.default_rng(seed)
    if shape == "uniform":
        d = rng.integers(0, vmax, n)
    elif shape == "skew":  # half the list from 50 small values
        d = rng.integers(0, vmax, n)
        d[: n // 2] = rng.integers(0, 50, n // 2)
        rng.shuffle(d)
    else:  # "cluster": eight runs of 300 consecutive values
        base = rng.integers(0, vmax - 300, 8)
        d = base[rng.integers(0, 8, n)] + rng.integers(0, 300, n)
    d[int(rng.integers(0, n))] = vmax - 1
    return d.astype(np.uint32)


def stream_bound(kind, f, data):
    """Worst-case bytes of one stream (ANSint: from the list's largest value, any value below 2^30)."""
    if kind == INT and data.size:
        return oracle().ans_oracle_bound_int(data.size, int(data.max()))
    return oracle().ans_oracle_bound(kind, f, data.size)


def oracle_encode(kind, f, data, ckpt_interval=0):
    """Returns (stream bytes as np.uint8, OracleInfo, ckpt_states[nck,4] u64, ckpt_off[nck] u32)."""
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    cap = stream_bound(kind, f, data)
    out = np.zeros(cap, dtype=np.uint8)
    info = OracleInfo()
    nck_max = (n // ckpt_interval + 1) if ckpt_interval else 1
    st = np.zeros((nck_max, 4), dtype=np.uint64)
    off = np.zeros(nck_max, dtype=np.uint32)
    nck = C.c_size_t(0)
    nb = oracle().ans_oracle_encode(kind, f, data, n, out, cap, C.byref(info), ckpt_interval,
                                    st.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                    C.byref(nck))
    if nb == 0:
        raise RuntimeError("oracle encode failed")
    return out[:nb].copy(), info, st[: nck.value].copy(), off[: nck.value].copy()


def prelude_hints(stream, prelude_offset):
    """Container parse hints (8 x u32) of the codec prelude that starts at stream[prelude_offset]."""
    buf = np.concatenate([np.ascontiguousarray(stream[prelude_offset:], dtype=np.uint8), np.zeros(16, dtype=np.uint8)])
    h = np.zeros(8, dtype=np.uint32)
    oracle().ans_oracle_prelude_hints(buf, h)
    return h


def oracle_decode(kind, f, stream, n, ref_f3_compat=False):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    # the reader may look a few bytes past the prelude; pad defensively
    padded = np.concatenate([stream, np.zeros(16, dtype=np.uint8)])
    out = np.zeros(n, dtype=np.uint32)
    rc = oracle().ans_oracle_decode(kind, f, padded, stream.size, out, n, int(ref_f3_compat))
    if rc != 0:
        raise RuntimeError("oracle decode failed rc=%d" % rc)
    return out


def oracle_pa_encode(kind, f, data, ckpt_interval=0):
    """pseudo_adaptive.cpp:85-130 block: returns (stream, PaInfo, OracleInfo of the codec part, ckpt states, ckpt offsets)."""
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    cap = 8 + 4 * n + 64 + oracle().ans_oracle_bound(kind, f, n)
    out = np.zeros(cap, dtype=np.uint8)
    pinfo, info = PaInfo(), OracleInfo()
    nck_max = (n // ckpt_interval + 1) if ckpt_interval else 1
    st = np.zeros((nck_max, 4), dtype=np.uint64)
    off = np.zeros(nck_max, dtype=np.uint32)
    nck = C.c_size_t(0)
    nb = oracle().ans_oracle_pa_encode(kind, f, data, n, out, cap, C.byref(pinfo), C.byref(info), ckpt_interval,
                                       st.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.byref(nck))
    if nb == 0:
        raise RuntimeError("oracle pa encode failed")
    return out[:nb].copy(), pinfo, info, st[: nck.value].copy(), off[: nck.value].copy()


def oracle_pa_decode(kind, f, stream, n):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    padded = np.concatenate([stream, np.zeros(16, dtype=np.uint8)])
    out = np.zeros(n, dtype=np.uint32)
    rc = oracle().ans_oracle_pa_decode(kind, f, padded, stream.size, out, n)
    if rc != 0:
        raise RuntimeError("oracle pa decode failed rc=%d" % rc)
    return out


def ref_pa_encode(kind, f, data, lib="libans_ref.so"):
    """The bytes src/pseudo_adaptive.cpp writes for one block (oracle/ref_shim.cpp ref_pa_encode)."""
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    cap = 8 + 4 * n + 64 + oracle().ans_oracle_bound(kind, f, n) + 64
    out = np.zeros(cap, dtype=np.uint8)
    hb = C.c_size_t(0)
    nb = ref(lib).ref_pa_encode(kind, f, data, n, out, cap, C.byref(hb))
    return out[:nb].copy(), hb.value


def canonicalize_pa(stream, pinfo, info):
    """Zero the indeterminate padding bits of BOTH interpolative codes of a pseudo_adaptive block: the
    alphabet header's last word and the codec prelude's last word."""
    s = np.array(stream, dtype=np.uint8, copy=True)
    vb = pinfo.interp_bits % 32
    if vb != 0:
        end = pinfo.header_bytes
        w = int.from_bytes(s[end - 4:end].tobytes(), "little") & ((1 << vb) - 1)
        s[end - 4:end] = np.frombuffer(w.to_bytes(4, "little"), dtype=np.uint8)
    if pinfo.sigma != 1:
        hb = pinfo.header_bytes
        s[hb:] = canonicalize(s[hb:], info)
    return s


def ref_encode(kind, f, data, lib="libans_ref.so"):
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    cap = stream_bound(kind, f, data) + 64
    out = np.zeros(cap, dtype=np.uint8)
    nb = ref(lib).ref_encode(kind, f, data, n, out, cap)
    return out[:nb].copy()


def ref_decode(kind, f, stream, n):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    padded = np.concatenate([np.zeros(16, dtype=np.uint8), stream, np.zeros(16, dtype=np.uint8)])
    view = padded[16:16 + stream.size]
    out = np.zeros(n, dtype=np.uint32)
    ref().ref_decode(kind, f, np.ascontiguousarray(view), stream.size, out, n)
    return out


# ---------------------------------------------------------------- seeded input families

def gen_inputs(name, n, seed=0):
    """Deterministic integer-only-friendly input families used across the parity tests."""
    rng = np.random.default_rng(seed)
    if name == "uniform256":
        return rng.integers(1, 257, size=n, dtype=np.uint32)
    if name == "uniform12":
        return rng.integers(0, 1 << 12, size=n, dtype=np.uint32)
    if name == "uniform20":
        return rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
    if name == "uniform24":
        return rng.integers(0, 1 << 24, size=n, dtype=np.uint32)
    if name == "geom0.01":
        return (rng.geometric(0.01, size=n) - 1).astype(np.uint32)
    if name == "geom0.4":
        return (rng.geometric(0.4, size=n) - 1).astype(np.uint32)
    if name.startswith("zipf"):
        # zipf<log2N>[s<exp>], inverse-CDF sampling over {1..N}
        body = name[4:]
        if "s" in body:
            lg, s = body.split("s")
            s = float(s)
        else:
            lg, s = body, 1.0
        N = 1 << int(lg)
        w = 1.0 / np.power(np.arange(1, N + 1, dtype=np.float64), s)
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        u = rng.random(n)
        return (np.searchsorted(cdf, u, side="left") + 1).astype(np.uint32)
    if name == "constant":
        return np.full(n, 7, dtype=np.uint32)
    if name == "distinct":
        return (np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 30)).astype(np.uint32)
    if name == "sparse_large":
        v = rng.integers(0, 1 << 30, size=n, dtype=np.uint32)
        mask = rng.random(n) < 0.9
        v[mask] = rng.integers(0, 50, size=int(mask.sum()), dtype=np.uint32)
        return v
    if name == "boundaries":
        vals = []
        for f in (1, 3, 5):
            T = 1 << (f + 7)
            for k in range(0, 4):
                b = T << (8 * k)
                for d in (-2, -1, 0, 1, 2):
                    x = b + d
                    if 0 <= x < (1 << 30):
                        vals.append(x)
        vals += [0, 1, (1 << 30) - 1, (1 << 30) - 2]
        vals = np.array(vals, dtype=np.uint32)
        return vals[rng.integers(0, vals.size, size=n)]
    raise ValueError(name)


def canonicalize(stream, info):
    """Zero the indeterminate padding bits of the last interpolative word.

    The reference's bit writer leaves bits >= (interp_bits % 32) of the final prelude word
    unspecified (uninitialised stack / stale window contents, include/bits.hpp:146-216), so
    byte parity is defined modulo exactly those bits."""
    s = np.array(stream, dtype=np.uint8, copy=True)
    vb = info.interp_bits % 32
    if vb != 0:
        end = info.header_bytes + info.prelude_bytes
        w = int.from_bytes(s[end - 4:end].tobytes(), "little") & ((1 << vb) - 1)
        s[end - 4:end] = np.frombuffer(w.to_bytes(4, "little"), dtype=np.uint8)
    return s


def ref_zipf_trace(n, q, seed, count):
    """include/zipf_dist.hpp driven by std::mt19937(seed): (values, draws per value, canonical uniforms consumed)."""
    lib = ref()
    vals = np.zeros(count, dtype=np.uint32)
    nd = np.zeros(count, dtype=np.uint32)
    cap = 4 * count + 64
    u = np.zeros(cap, dtype=np.float64)
    done = C.c_size_t(0)
    nu = lib.ref_zipf_trace(n, q, seed, count, vals, nd, u, cap, C.byref(done))
    m = done.value
    return vals[:m], nd[:m], u[:nu]


def ref_blocks_mt(kind, f, data, block_ints, threads):
    """Every block of `data` encoded, then decoded, by `threads` native threads over the compiled reference:
    (ok, encode seconds, decode seconds, total stream bytes)."""
    lib = ref()
    e, d, tot = C.c_double(0), C.c_double(0), C.c_size_t(0)
    bad = lib.ref_blocks_mt(kind, f, np.ascontiguousarray(data, dtype=np.uint32), data.size, block_ints, threads,
                            C.byref(e), C.byref(d), C.byref(tot))
    return bad == 0, e.value, d.value, tot.value


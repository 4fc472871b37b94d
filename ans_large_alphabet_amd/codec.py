"""Host-side mirror of the reference's codec concept for this path.

The reference exposes ``ANSfold<f>`` / ``ANSrfold<f>`` as structs with static ``name()``,
``encode(in, n, out, cap)`` -> bytes written and ``decode(in, bytes, out, n)``
(/root/reference/include/methods.hpp:529-567).  The classes below keep those names and argument
meanings on top of the C-ABI (include/ansx.h); the C++17 mirror with the exact static
signatures is ans_large_alphabet_amd/include/ansx_methods.hpp.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def make_opts(block_ints=0, ckpt_interval=0, flags=0):
    return L.Opts(block_ints, ckpt_interval, flags, 0)


class Context:
    """One per (process, device): owns the HIP stream and the device workspace."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        st = L.lib().ansx_init(device, C.byref(self._h))
        if st != L.OK:
            raise L.AnsxError(st, "ansx_init")

    def close(self):
        if self._h:
            L.lib().ansx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def workspace_bytes(self):
        return L.lib().ansx_workspace_bytes(self._h)

    def merge_containers_dev(self, part_ptrs, part_bytes, out_ptr, out_capacity, stream=None):
        """Native root-side concatenation of rank containers (device pointers, list order) -> bytes written."""
        k = len(part_ptrs)
        ptrs = (C.c_void_p * k)(*part_ptrs)
        sizes = (C.c_size_t * k)(*part_bytes)
        nb = C.c_size_t(0)
        st = L.lib().ansx_merge_containers_dev(self._h, ptrs, sizes, k, out_ptr, out_capacity, C.byref(nb), stream)
        if st != L.OK:
            raise L.AnsxError(st, "ansx_merge_containers_dev")
        return nb.value

    def last_encode_stats(self):
        """dict(max_nsyms, max_log2_frame, near_threshold_decisions, path, host_redecided) of the most recent encode."""
        st = L.EncodeStats()
        L.lib().ansx_last_encode_stats(self._h, C.byref(st))
        return {k: int(getattr(st, k)) for k, _ in L.EncodeStats._fields_}

    def debug_set(self, name, value=None):
        """Select one of the equivalent internal code paths (tests / experiments); value None or "" = default."""
        st = L.lib().ansx_debug_set(self._h, name.encode(), None if value is None else str(value).encode())
        if st != L.OK:
            raise L.AnsxError(st, "ansx_debug_set(%s)" % name)

    # -- per-kernel timing (hipEvents inside the library)
    def profile(self, on=True):
        L.lib().ansx_profile_enable(self._h, int(on))

    def profile_reset(self):
        L.lib().ansx_profile_reset(self._h)

    def profile_get(self):
        arr = (L.KernelTime * 64)()
        cnt = C.c_int(0)
        L.lib().ansx_profile_get(self._h, arr, 64, C.byref(cnt))
        return [(arr[i].name.decode(), arr[i].total_ms, int(arr[i].launches)) for i in range(min(cnt.value, 64))]


class _Codec:
    KIND = L.FOLD
    PREFIX = "ANSfold"

    def __init__(self, fidelity, ctx=None, block_ints=0, ckpt_interval=0, compact=False):
        """compact=True: per-block alphabet compaction (src/pseudo_adaptive.cpp:85-130, ANSX_FLAG_COMPACT_ALPHABET)."""
        self.f = int(fidelity)
        self.ctx = ctx
        self.opts = make_opts(block_ints, ckpt_interval, L.FLAG_COMPACT_ALPHABET if compact else 0)

    def _ctx(self):
        if self.ctx is None:
            self.ctx = Context()
        return self.ctx

    def name(self):  # methods.hpp:530-533 / 550-553
        buf = C.create_string_buffer(32)
        L.lib().ansx_codec_name(self.KIND, self.f, buf, 32)
        return buf.value.decode()

    def bound(self, n):
        return L.lib().ansx_bound(self.KIND, self.f, n, C.byref(self.opts))

    # ---- host buffers (numpy), signature meaning as methods.hpp encode()/decode()
    def encode(self, data, out=None):
        data = np.ascontiguousarray(data, dtype=np.uint32)
        n = data.size
        if out is None:
            out = np.empty(max(self.bound(n), 64), dtype=np.uint8)
        nb = C.c_size_t(0)
        st = L.lib().ansx_encode(self._ctx().handle, self.KIND, self.f, data.ctypes.data, n,
                                 out.ctypes.data, out.size, C.byref(nb), C.byref(self.opts))
        if st != L.OK:
            raise L.AnsxError(st, self.name() + ".encode")
        return out[: nb.value]

    def decode(self, stream, n, out=None):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        if out is None:
            out = np.empty(n, dtype=np.uint32)
        st = L.lib().ansx_decode(self._ctx().handle, self.KIND, self.f, stream.ctypes.data, stream.size,
                                 out.ctypes.data, n, C.byref(self.opts))
        if st != L.OK:
            raise L.AnsxError(st, self.name() + ".decode")
        return out

    # ---- device pointers (HBM resident); ptrs are integers (e.g. torch.Tensor.data_ptr())
    def encode_dev(self, in_ptr, n, out_ptr, out_capacity, stream=None):
        nb = C.c_size_t(0)
        st = L.lib().ansx_encode_dev(self._ctx().handle, self.KIND, self.f, in_ptr, n, out_ptr,
                                     out_capacity, C.byref(nb), C.byref(self.opts), stream)
        if st != L.OK:
            raise L.AnsxError(st, self.name() + ".encode_dev")
        return nb.value

    def decode_dev(self, in_ptr, in_bytes, out_ptr, n, stream=None):
        st = L.lib().ansx_decode_dev(self._ctx().handle, self.KIND, self.f, in_ptr, in_bytes, out_ptr, n,
                                     C.byref(self.opts), stream)
        if st != L.OK:
            raise L.AnsxError(st, self.name() + ".decode_dev")


class ANSfold(_Codec):
    """methods.hpp:529-547"""
    KIND = L.FOLD
    PREFIX = "ANSfold"


class ANSrfold(_Codec):
    """methods.hpp:549-567"""
    KIND = L.RFOLD
    PREFIX = "ANSrfold"


class ANSmsb(_Codec):
    """methods.hpp:499-515 (include/ans_msb.hpp): the fixed-threshold MSB fold; no fidelity."""
    KIND = L.MSB
    PREFIX = "ANSmsb"

    def __init__(self, ctx=None, block_ints=0, ckpt_interval=0, compact=False):
        super().__init__(0, ctx=ctx, block_ints=block_ints, ckpt_interval=ckpt_interval, compact=compact)


class ANSint(_Codec):
    """methods.hpp:484-497 (include/ans_int.hpp), name() == "ANS".  compact=True (default): per-block alphabet
    compaction, any values.  compact=False: the values themselves are the symbols and must be below 16384;
    block_ints=SINGLE_STREAM then gives exactly the bytes of ANSint::encode (ans_int_compress)."""
    KIND = L.INT
    PREFIX = "ANS"

    def __init__(self, ctx=None, block_ints=0, ckpt_interval=0, compact=True):
        super().__init__(0, ctx=ctx, block_ints=block_ints, ckpt_interval=ckpt_interval, compact=compact)


# ---------------------------------------------------------------- container parsing (host)

def unpack_restart_points(raw):
    """29-byte restart points -> (cursors u32[n], states u64[n * 4]): states 0, 1 as one 104-bit little-endian
    integer in bytes 0..12, states 2, 3 in bytes 13..25 (52 bits each), the cursor in bytes 26..28."""
    rec = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, 29).astype(np.uint64)
    n = rec.shape[0]
    st = np.zeros((n, 4), dtype=np.uint64)
    m52 = np.uint64((1 << 52) - 1)
    for pair in range(2):
        b = rec[:, 13 * pair: 13 * pair + 13]
        lo = np.zeros(n, dtype=np.uint64)
        for i in range(8):
            lo |= b[:, i] << np.uint64(8 * i)
        hi = np.zeros(n, dtype=np.uint64)
        for i in range(5):
            hi |= b[:, 8 + i] << np.uint64(8 * i)
        st[:, 2 * pair] = lo & m52
        st[:, 2 * pair + 1] = (lo >> np.uint64(52)) | (hi << np.uint64(12))
    off = (rec[:, 26] | (rec[:, 27] << np.uint64(8)) | (rec[:, 28] << np.uint64(16))).astype(np.uint32)
    return off, st.reshape(-1)


def parse_container(buf):
    """Split a container (np.uint8) into header fields, per-block streams and restart points."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    H = L.ContainerHeader()
    st = L.lib().ansx_container_info(buf.ctypes.data, buf.size, C.byref(H))
    if st != L.OK:
        raise L.AnsxError(st, "ansx_container_info")
    nb = H.nblocks
    idx_off = C.sizeof(L.ContainerHeader)
    boff = np.frombuffer(buf[idx_off: idx_off + 8 * (nb + 1)].tobytes(), dtype=np.uint64)
    ck_off_off = idx_off + 8 * (nb + 1)
    nck = nb * H.ckpts_per_block
    if H.kind & 0x200:  # wide restart points: u32 cursors, then 4 x u64 states (the v2 form)
        ck_off = np.frombuffer(buf[ck_off_off: ck_off_off + 4 * nck].tobytes(), dtype=np.uint32)
        ck_state_off = (ck_off_off + 4 * nck + 7) // 8 * 8
        ck_state = np.frombuffer(buf[ck_state_off: ck_state_off + 32 * nck].tobytes(), dtype=np.uint64)
        hint_off = (ck_state_off + 32 * nck + 15) // 16 * 16
    else:  # packed 29-byte records
        ck_off, ck_state = unpack_restart_points(buf[ck_off_off: ck_off_off + 29 * nck])
        hint_off = (ck_off_off + 29 * nck + 15) // 16 * 16
    hints = np.frombuffer(buf[hint_off: hint_off + 32 * nb].tobytes(), dtype=np.uint32).reshape(nb, 8)
    p0 = int(H.payload_offset)
    streams = [buf[p0 + int(boff[i]): p0 + int(boff[i + 1])] for i in range(nb)]
    return {
        "header": H, "block_off": boff, "streams": streams, "parse_hints": hints,
        "ckpt_off": ck_off.reshape(nb, H.ckpts_per_block) if nck else ck_off.reshape(nb, 0),
        "ckpt_state": ck_state.reshape(nb, H.ckpts_per_block, 4) if nck else ck_state.reshape(nb, 0, 4),
    }


# ---------------------------------------------------------------- synthetic inputs (generate_inputs.cpp)

def parse_dist(spec):
    """'uniform<lo>-<hi>' | 'uniform<bits>' (0 .. 2^bits - 1, generate_inputs.cpp:94-101) | 'geom<p>' |
    'zipf<log2 n>[s<q>]' (values 1 .. 2^log2n, exponent q, default 1.0 as zipf_dist.hpp:39-40)
    -> (dist, a, b)"""
    if spec.startswith("uniform"):
        body = spec[7:]
        if "-" in body:
            lo, hi = body.split("-")
            return L.GEN_UNIFORM, float(int(lo)), float(int(hi))
        return L.GEN_UNIFORM, 0.0, float((1 << int(body)) - 1)
    if spec.startswith("geom"):
        return L.GEN_GEOMETRIC, float(spec[4:]), 0.0
    if spec.startswith("zipf"):
        body = spec[4:]
        lg, q = (body.split("s") + ["1.0"])[:2] if "s" in body else (body, "1.0")
        return L.GEN_ZIPF, float(1 << int(lg)), float(q)
    raise ValueError("unknown distribution %r" % (spec,))


def generate_host(spec, n, seed=0, first_index=0):
    """n values of the named distribution on the CPU (same values as generate_dev)."""
    dist, a, b = parse_dist(spec)
    out = np.empty(n, dtype=np.uint32)
    st = L.lib().ansx_generate_host(dist, a, b, seed, first_index, out.ctypes.data, n)
    if st != L.OK:
        raise L.AnsxError(st, "ansx_generate_host")
    return out


def zipf_from_uniform(n, q, u01):
    """(value, accepted) of one pass of the Zipf generator's rejection loop for the canonical uniform u01."""
    import ctypes as C

    k, acc = C.c_uint32(0), C.c_int(0)
    st = L.lib().ansx_zipf_from_uniform(float(n), float(q), float(u01), C.byref(k), C.byref(acc))
    if st != L.OK:
        raise L.AnsxError(st, "ansx_zipf_from_uniform")
    return int(k.value), bool(acc.value)


def generate_dev(ctx, spec, out_ptr, n, seed=0, first_index=0, stream=None):
    """Fill device memory at out_ptr (n x uint32) with the named distribution; asynchronous on `stream`."""
    dist, a, b = parse_dist(spec)
    st = L.lib().ansx_generate_dev(ctx.handle, dist, a, b, seed, first_index, out_ptr, n, stream)
    if st != L.OK:
        raise L.AnsxError(st, "ansx_generate_dev")

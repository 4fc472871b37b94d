"""Pure-Python builder of the ansx container from ORACLE block streams (test infrastructure):
an independent statement of the container layout documented in DESIGN.md section 3."""
import struct

import numpy as np

import oracle_lib as ol

MAGIC = b"ANSXv3"


def nseg(nb, ckpt):
    nfull = nb - (nb & 3)
    if ckpt == 0 or nfull == 0:
        return 1
    return (nfull + ckpt - 1) // ckpt


def pack_restart_points(states, offs):
    """(n, 4) u64 states + n u32 cursors -> n records of 29 bytes: states 0, 1 as one 104-bit little-endian integer
    (state 0 in the low 52 bits), states 2, 3 likewise, then the cursor in 24 bits."""
    out = bytearray()
    for st, off in zip(states.tolist(), offs.tolist()):
        assert all(x < (1 << 52) for x in st) and off < (1 << 24)
        out += (st[0] | (st[1] << 52)).to_bytes(13, "little")
        out += (st[2] | (st[3] << 52)).to_bytes(13, "little")
        out += off.to_bytes(3, "little")
    return np.frombuffer(bytes(out), dtype=np.uint8)


def build_container(kind, f, data, block, ckpt, wide=None):
    """wide: restart-point format; None = what the library picks (wide for ANSint and once a frame exceeds 2^16)."""
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    if ckpt >= block:
        ckpt = 0
    nblocks = (n + block - 1) // block
    nckf = nseg(block, ckpt) - 1
    streams, cks, cko, hints, maxlg, maxns, maxsig = [], [], [], [], 0, 0, 0
    for b in range(nblocks):
        s, info, st, off = ol.oracle_encode(kind, f, data[b * block:(b + 1) * block], ckpt_interval=ckpt)
        streams.append(s)
        hints.append(ol.prelude_hints(s, info.header_bytes))
        pad_s = np.zeros((nckf, 4), dtype=np.uint64)
        pad_o = np.zeros(nckf, dtype=np.uint32)
        pad_s[:st.shape[0]] = st
        pad_o[:off.shape[0]] = off
        cks.append(pad_s)
        cko.append(pad_o)
        maxlg = max(maxlg, info.log2_frame)
        maxns = max(maxns, info.max_sym + 1)
        maxsig = max(maxsig, int(info.present_syms))  # symbols present in the block's model
    if (kind & 0x1FF) == 3:  # plain ANSint: no parse hints, max_nsyms bounds a block's distinct values (dense and rank-space model alike)
        hints = [np.zeros(8, dtype=np.uint32) for _ in hints]
        maxns = maxsig
    if wide is None:
        wide = (kind & 0xFF) == 3 or maxlg > 16
    index_off = 64
    ckoff_off = index_off + 8 * (nblocks + 1)
    if wide:
        ckstate_off = (ckoff_off + 4 * nblocks * nckf + 7) // 8 * 8
        hint_off = (ckstate_off + 32 * nblocks * nckf + 15) // 16 * 16
    else:
        hint_off = (ckoff_off + 29 * nblocks * nckf + 15) // 16 * 16
    payload_off = hint_off + 32 * nblocks
    sizes = np.array([s.size for s in streams], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    payload = np.concatenate(streams)
    out = np.zeros(payload_off + payload.size, dtype=np.uint8)
    hdr = MAGIC + struct.pack("<HIIQIIIIIIQQ", max(maxsig, 1) - 1, kind | (0x200 if wide else 0), f, n, block, ckpt, nblocks, maxlg, maxns, nckf,
                              int(payload.size), payload_off)
    out[:64] = np.frombuffer(hdr, dtype=np.uint8)
    out[index_off:index_off + 8 * (nblocks + 1)] = boff.view(np.uint8)
    if nckf and wide:
        out[ckoff_off:ckoff_off + 4 * nblocks * nckf] = np.concatenate(cko).view(np.uint8)
        out[ckstate_off:ckstate_off + 32 * nblocks * nckf] = np.concatenate(cks).reshape(-1).view(np.uint8)
    elif nckf:
        out[ckoff_off:ckoff_off + 29 * nblocks * nckf] = pack_restart_points(np.concatenate(cks), np.concatenate(cko))
    out[hint_off:hint_off + 32 * nblocks] = np.concatenate(hints).view(np.uint8)
    out[payload_off:] = payload
    return out

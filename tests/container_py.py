"""Pure-Python builder of the ansx container from ORACLE block streams (test infrastructure):
an independent statement of the container layout documented in DESIGN.md section 3."""
import struct

import numpy as np

import oracle_lib as ol

MAGIC = b"ANSXv2\x00\x00"


def nseg(nb, ckpt):
    nfull = nb - (nb & 3)
    if ckpt == 0 or nfull == 0:
        return 1
    return (nfull + ckpt - 1) // ckpt


def build_container(kind, f, data, block, ckpt):
    data = np.ascontiguousarray(data, dtype=np.uint32)
    n = data.size
    if ckpt >= block:
        ckpt = 0
    nblocks = (n + block - 1) // block
    nckf = nseg(block, ckpt) - 1
    streams, cks, cko, hints, maxlg, maxns = [], [], [], [], 0, 0
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
    index_off = 64
    ckoff_off = index_off + 8 * (nblocks + 1)
    ckstate_off = (ckoff_off + 4 * nblocks * nckf + 7) // 8 * 8
    hint_off = (ckstate_off + 32 * nblocks * nckf + 15) // 16 * 16
    payload_off = hint_off + 32 * nblocks
    sizes = np.array([s.size for s in streams], dtype=np.uint64)
    boff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    payload = np.concatenate(streams)
    out = np.zeros(payload_off + payload.size, dtype=np.uint8)
    hdr = MAGIC + struct.pack("<IIQIIIIIIQQ", kind, f, n, block, ckpt, nblocks, maxlg, maxns, nckf,
                              int(payload.size), payload_off)
    out[:64] = np.frombuffer(hdr, dtype=np.uint8)
    out[index_off:index_off + 8 * (nblocks + 1)] = boff.view(np.uint8)
    if nckf:
        out[ckoff_off:ckoff_off + 4 * nblocks * nckf] = np.concatenate(cko).view(np.uint8)
        out[ckstate_off:ckstate_off + 32 * nblocks * nckf] = np.concatenate(cks).reshape(-1).view(np.uint8)
    out[hint_off:hint_off + 32 * nblocks] = np.concatenate(hints).view(np.uint8)
    out[payload_off:] = payload
    return out

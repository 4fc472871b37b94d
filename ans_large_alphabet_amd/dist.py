"""Multi-GPU plumbing for the ANSfold/ANSrfold path (one process per GPU, torch.distributed;
backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Blocks are independent reference encode() calls, so the input list is partitioned into contiguous
ranges of WHOLE blocks, each rank encodes/decodes its range with no data-path collective, and the
only exchange is the concatenation of the per-rank containers on a root rank:

    all_gather(8-byte sizes)  ->  one direct send per rank into the root's buffer (RCCL has no
    gatherv; every sender uses its own xGMI link to the root, so the step is bound by the largest
    rank segment, not by a ring)  ->  the root rebases the block index (merge_containers).

The reference has no counterpart (single-threaded, no communication).
"""
import struct

import torch

HEADER_BYTES = 64
MAGIC = b"ANSXv3"  # followed by u16 (most symbols present in a block) - 1
KIND_WIDE_RESTART = 0x200  # kind word bit 9: u32 cursors + 4 x u64 states instead of packed 29-byte restart points


def shard_blocks(n, block_ints, rank, world):
    """Contiguous whole-block range of `rank`: returns (first_int, n_ints).  Every rank must own at
    least one block (the codec rejects empty inputs, and an empty rank would leave the others waiting
    in the gather): fewer blocks than ranks is a caller error, raised identically on every rank."""
    nblocks = (n + block_ints - 1) // block_ints
    if nblocks < world:
        raise ValueError("%d blocks cannot be sharded over %d ranks: use fewer ranks or smaller blocks" % (nblocks, world))
    b0 = rank * nblocks // world
    b1 = (rank + 1) * nblocks // world
    lo = min(n, b0 * block_ints)
    hi = min(n, b1 * block_ints)
    return lo, hi - lo


def layout(nblocks, ckpts_per_block, wide=False):
    """Container layout (must match layout_of() in csrc/ansx.hip).  Packed restart points (the default): one array of
    29-byte records where the wide form has its u32 cursors, no separate state array."""
    index_off = HEADER_BYTES
    ckoff_off = index_off + 8 * (nblocks + 1)
    nck = nblocks * ckpts_per_block
    if wide:
        ckstate_off = (ckoff_off + 4 * nck + 7) // 8 * 8
        hint_off = (ckstate_off + 32 * nck + 15) // 16 * 16  # 8 x u32 parse hints per block
    else:
        ckstate_off = ckoff_off
        hint_off = (ckoff_off + 29 * nck + 15) // 16 * 16
    payload_off = hint_off + 32 * nblocks
    return index_off, ckoff_off, ckstate_off, hint_off, payload_off


def parse_header(buf):
    """buf: 1-D uint8 tensor (any device).  Returns a dict of the 64-byte header."""
    raw = bytes(buf[:HEADER_BYTES].cpu().numpy().tobytes())
    if raw[:6] != MAGIC:
        raise ValueError("not an ansx container")
    mp, kind, f, n, block_ints, ckpt, nblocks, maxlg, maxns, nckf, payload_bytes, payload_off = struct.unpack(
        "<HIIQIIIIIIQQ", raw[6:])
    return dict(max_present_m1=mp, kind=kind, f=f, n=n, block_ints=block_ints, ckpt=ckpt, nblocks=nblocks, max_log2_frame=maxlg,
                max_nsyms=maxns, ckpts_per_block=nckf, payload_bytes=payload_bytes, payload_offset=payload_off)


def pack_header(h):
    return MAGIC + struct.pack("<HIIQIIIIIIQQ", h["max_present_m1"], h["kind"], h["f"], h["n"], h["block_ints"], h["ckpt"], h["nblocks"],
                               h["max_log2_frame"], h["max_nsyms"], h["ckpts_per_block"], h["payload_bytes"],
                               h["payload_offset"])


def gather_containers(local, nbytes, dst=0, group=None, async_op=False):
    """Concatenate every rank's first `nbytes` bytes of `local` (1-D uint8 tensor) on `dst`.

    Returns (buffer, sizes, works): on dst `buffer` holds the rank segments back to back in rank
    order (its own segment included) and `sizes` their byte counts; elsewhere (None, sizes, works).
    With async_op the caller must wait on `works` before reading `buffer`."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local.device
    mine = torch.tensor([nbytes], dtype=torch.int64, device=dev)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes_t, mine, group=group)
    sizes = [int(x) for x in sizes_t.tolist()]
    works = []
    buf = None
    if rank == dst:
        buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        buf[offs[rank]:offs[rank + 1]].copy_(local[:nbytes])
        ops = [dist.P2POp(dist.irecv, buf[offs[r]:offs[r + 1]], r, group) for r in range(world) if r != dst]
        if ops:
            works = dist.batch_isend_irecv(ops)
    else:
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, local[:nbytes], dst, group)])
    if not async_op:
        for w in works:
            w.wait()
        works = []
    return buf, sizes, works


def merge_containers(buf, sizes):
    """Root-side index fix-up: turn the back-to-back rank containers in `buf` into ONE container
    whose blocks are the rank block ranges in rank order.  All parts must share kind, fidelity,
    block_ints and ckpt_interval, and every part but the last must hold whole blocks only."""
    dev = buf.device
    parts, off = [], 0
    for s in sizes:
        parts.append(buf[off:off + s])
        off += s
    hs = [parse_header(p) for p in parts]
    h0 = hs[0]
    for i, h in enumerate(hs):
        for key in ("kind", "f", "block_ints", "ckpt", "ckpts_per_block"):
            if h[key] != h0[key]:
                raise ValueError("rank containers disagree on %s" % key)
        if i + 1 < len(hs) and h["n"] % h["block_ints"] != 0:
            raise ValueError("only the last rank may end in a partial block")
    nblocks = sum(h["nblocks"] for h in hs)
    nckf = h0["ckpts_per_block"]
    wide = bool(h0["kind"] & KIND_WIDE_RESTART)  # (the parts agree: their kind words are equal)
    idx_off, ckoff_off, ckstate_off, hint_off, payload_off = layout(nblocks, nckf, wide)
    payload_bytes = sum(h["payload_bytes"] for h in hs)
    out = torch.zeros(payload_off + payload_bytes, dtype=torch.uint8, device=dev)
    merged = dict(h0)
    merged.update(n=sum(h["n"] for h in hs), nblocks=nblocks, payload_bytes=payload_bytes,
                  payload_offset=payload_off, max_log2_frame=max(h["max_log2_frame"] for h in hs),
                  max_nsyms=max(h["max_nsyms"] for h in hs), max_present_m1=max(h["max_present_m1"] for h in hs))
    out[:HEADER_BYTES] = torch.frombuffer(bytearray(pack_header(merged)), dtype=torch.uint8).to(dev)
    blk, pay = 0, 0
    index = []
    for p, h in zip(parts, hs):
        i_off, c_off, s_off, h_off, p_off = layout(h["nblocks"], nckf, wide)
        nb = h["nblocks"]
        boff = p[i_off:i_off + 8 * (nb + 1)].clone().view(torch.int64)
        index.append(boff[:nb] + pay)
        if nckf and wide:
            out[ckoff_off + 4 * blk * nckf: ckoff_off + 4 * (blk + nb) * nckf] = p[c_off:c_off + 4 * nb * nckf]
            out[ckstate_off + 32 * blk * nckf: ckstate_off + 32 * (blk + nb) * nckf] = p[s_off:s_off + 32 * nb * nckf]
        elif nckf:
            out[ckoff_off + 29 * blk * nckf: ckoff_off + 29 * (blk + nb) * nckf] = p[c_off:c_off + 29 * nb * nckf]
        out[hint_off + 32 * blk: hint_off + 32 * (blk + nb)] = p[h_off:h_off + 32 * nb]
        out[payload_off + pay: payload_off + pay + h["payload_bytes"]] = p[p_off:p_off + h["payload_bytes"]]
        blk += nb
        pay += h["payload_bytes"]
    index.append(torch.tensor([pay], dtype=torch.int64, device=dev))
    out[idx_off:idx_off + 8 * (nblocks + 1)] = torch.cat(index).view(torch.uint8)
    return out

"""N > 1 path on CPU (gloo, world_size 2): contiguous block-range sharding, the size
all_gather + send/recv gather to rank 0, and the root-side index rebase.  The rank-local
containers are built from ORACLE block streams (tests/container_py.py), so this also pins the
container layout independently of the HIP code."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_blocks_cover_and_align():
    from ans_large_alphabet_amd import dist as adist

    for n in (1, 5, 4096, 70001, 1 << 20):
        for block in (64, 4096, 16384):
            for world in (1, 2, 3, 8):
                pos = 0
                if (n + block - 1) // block < world:  # an empty rank would hang the gather: refused up front
                    with pytest.raises(ValueError):
                        adist.shard_blocks(n, block, 0, world)
                    continue
                for r in range(world):
                    lo, cnt = adist.shard_blocks(n, block, r, world)
                    assert lo == pos and lo % block == 0 or cnt == 0
                    pos = lo + cnt if cnt else pos
                    if r + 1 < world and cnt:
                        assert cnt % block == 0 or lo + cnt == n
                assert pos == n


def test_layout_matches_python_container(oracle_built):
    import torch

    import container_py as cp
    from ans_large_alphabet_amd import dist as adist

    data = ol.gen_inputs("zipf20s1.2", 20001, seed=2)
    c = cp.build_container(ol.FOLD, 1, data, 4096, 1024)
    h = adist.parse_header(torch.from_numpy(c))
    assert h["n"] == data.size and h["nblocks"] == 5 and h["ckpts_per_block"] == 3
    assert adist.layout(h["nblocks"], h["ckpts_per_block"])[4] == h["payload_offset"]
    assert adist.pack_header(h) == c[:64].tobytes()
    # wide restart points (kind word bit 9): the v2 arrays
    w = cp.build_container(ol.FOLD, 1, data, 4096, 1024, wide=True)
    hw = adist.parse_header(torch.from_numpy(w))
    assert hw["kind"] == (ol.FOLD | adist.KIND_WIDE_RESTART)
    assert adist.layout(hw["nblocks"], hw["ckpts_per_block"], wide=True)[4] == hw["payload_offset"]
    assert w.size - c.size == hw["payload_offset"] - h["payload_offset"] > 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_gather_and_merge(tmp_path, oracle_built, world):
    """world 3: 18 blocks over 3 ranks, the last rank ends in a partial block."""
    port = _free_port()
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_text() == "OK"

"""Worker for tests/test_dist_gloo.py: one rank of the block-range shard + container gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import container_py as cp  # noqa: E402
import oracle_lib as ol  # noqa: E402
from ans_large_alphabet_amd import dist as adist  # noqa: E402


def main():
    out_path = sys.argv[1]
    n, block, ckpt, kind, f = 70001, 4096, 1024, ol.FOLD, 1
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    data = ol.gen_inputs("zipf20s1.2", n, seed=5)
    lo, cnt = adist.shard_blocks(n, block, rank, world)
    local = cp.build_container(kind, f, data[lo:lo + cnt], block, ckpt)
    t = torch.from_numpy(local.copy())
    buf, sizes, _ = adist.gather_containers(t, t.numel(), dst=0)
    ok = True
    if rank == 0:
        merged = adist.merge_containers(buf, sizes).numpy()
        whole = cp.build_container(kind, f, data, block, ckpt)
        ok = merged.size == whole.size and bool(np.array_equal(merged, whole))
        # every block of the merged container still decodes with the oracle
        h = adist.parse_header(torch.from_numpy(merged))
        ok = ok and h["n"] == n and h["nblocks"] == (n + block - 1) // block
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write("OK" if ok else "FAIL")
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

"""Full-load soak of the GPU codec (manual tool, not collected by pytest): 128 Mi-int inputs of six
value distributions x four codecs x four block / restart geometries, encode + decode on the device,
every result compared on the device.  Inputs are produced by torch kernels right before the call and
nothing synchronises explicitly (this is how the default-stream ordering bug of round 1 was found).

    python tests/tools/soak_gpu_big.py        # on an MI355X box, ~1 minute
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import ans_large_alphabet_amd as A
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda", 0)
ctx = A.Context(0)
n = 128 * (1 << 20)
g = torch.Generator(device=dev); g.manual_seed(5)
cap = 8 * n + (64 << 20)
d_out = torch.empty(cap, dtype=torch.uint8, device=dev); d_back = torch.zeros(n, dtype=torch.int32, device=dev)
fails = 0
stream = None  # the context's own stream: must order against torch's default stream by itself
for it in range(24):
    c = it % 6
    if c == 0: d_in = bench.gen_input(torch, A, ctx, "zipf20s1.2", n, 100 + it, dev)
    elif c == 1: d_in = torch.randint(1 << 24, (1 << 30) - 1, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)   # k = 3 everywhere
    elif c == 2: d_in = torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    elif c == 3: d_in = bench.gen_input(torch, A, ctx, "zipf24s1.0", n, 200 + it, dev)
    elif c == 4: d_in = (torch.randint(0, 2, (n,), generator=g, device=dev, dtype=torch.int64) * ((1 << 30) - 1)).to(torch.int32)
    else: d_in = torch.randint(0, 1 << 16, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    kind = [("fold", 1), ("fold", 1), ("fold", 3), ("rfold", 1)][it % 4]
    if kind[0] == "rfold": d_in = torch.clamp(d_in, max=(1 << 30) - 1 - (1 << 8))
    blk, ck = [(16384, 1024), (16384, 512), (32768, 1024), (65536, 1024)][(it // 6) % 4]
    codec = (A.ANSfold if kind[0] == "fold" else A.ANSrfold)(kind[1], ctx=ctx, block_ints=blk, ckpt_interval=ck)
    d_back.zero_()
    t0 = time.time()
    nb = codec.encode_dev(d_in.data_ptr(), n, d_out.data_ptr(), cap, stream=stream)
    codec.decode_dev(d_out.data_ptr(), nb, d_back.data_ptr(), n, stream=stream)
    torch.cuda.synchronize()
    ok = bool(torch.equal(d_back, d_in))
    fails += 0 if ok else 1
    print("it", it, kind, "block", blk, ck, "data", c, "bits/int %.3f" % (8 * nb / n), "ms %.1f" % ((time.time() - t0) * 1e3), "OK" if ok else "FAIL"); sys.stdout.flush()
print("BIG SOAK done, fails", fails)

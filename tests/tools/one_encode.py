"""Kernel-timing probe: config 2 input, a few device encodes, errors ignored (experimental library builds may
produce wrong models); run under `rocprofv3 --kernel-trace --stats`."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import ans_large_alphabet_amd as A

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
spec = sys.argv[2] if len(sys.argv) > 2 else "zipf20s1.2"
f = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = A.Context(0)
d = torch.empty(n, dtype=torch.int32, device="cuda:0")
A.generate_dev(ctx, spec, d.data_ptr(), n, seed=1)
torch.cuda.synchronize()
codec = A.ANSfold(f, ctx=ctx)
out = torch.empty(n * 4 + (1 << 20), dtype=torch.uint8, device="cuda:0")
for i in range(6):
    try:
        codec.encode_dev(d.data_ptr(), n, out.data_ptr(), out.numel())
    except Exception as e:
        print("encode", i, "failed:", str(e)[:80])
torch.cuda.synchronize()
back = torch.empty(n, dtype=torch.int32, device="cuda:0")
try:
    nbytes = codec.encode_dev(d.data_ptr(), n, out.data_ptr(), out.numel())
    for i in range(6):
        try:
            codec.decode_dev(out.data_ptr(), nbytes, back.data_ptr(), n)
        except Exception as e:
            print("decode", i, "failed:", str(e)[:80])
except Exception as e:
    print("encode for the decode leg failed:", str(e)[:80])
torch.cuda.synchronize()
print("done")

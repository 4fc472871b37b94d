import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle_lib as ol
import ans_large_alphabet_amd as A

fam, f, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
data = ol.gen_inputs(fam, n, seed=17 * f)
for th in ("8", "5", "4", None):
    c = A.Context(0)
    c.debug_set("ANSX_NS_HINT", "4096")
    if th: c.debug_set("ANSX_T_HINT", th)
    else: c.debug_set("ANSX_NO_FAST_MODEL", "1")
    codec = A.ANSfold(f, ctx=c, block_ints=16384, ckpt_interval=1024)
    cont = codec.encode(data)
    st = c.last_encode_stats()
    parts = A.parse_container(cont)
    out = []
    for b, s in enumerate(parts["streams"]):
        blk = data[b * 16384:(b + 1) * 16384]
        exp, info, _, _ = ol.oracle_encode(ol.FOLD, f, blk, ckpt_interval=1024)
        # vbyte(max_sym) then log2M
        p = 0
        while s[p] & 128: p += 1
        out.append((int(s[p + 1]), info.log2_frame, info.max_sym + 1, int(info.sigma), bool(np.array_equal(s, exp))))
    print("T_HINT", th, "path", st["path"], out)
    c.close()

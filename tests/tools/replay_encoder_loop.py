"""Manual tool: replay the compiled main loop of k_encode<1, true> from registers (no kernel around it).

    make -C ans_large_alphabet_amd/csrc asm                      # writes ansx_gfx950.s
    python tests/tools/replay_encoder_loop.py /tmp/replay.hip    # generates the benchmark source
    hipcc --offload-arch=gfx950 -O1 /tmp/replay.hip -o replay && ./replay     # on an MI355X

The 32-step loop body is lifted from the ISA listing as it stands (real register allocation, real order),
branches / exec-mask code dropped, and run 64 times by 1024 single-wave workgroups in four forms: VALU + SALU only;
+ its ds_read_u16 pairs (addresses masked into the allocation: out-of-range LDS addresses are ~100 cycles per step
slower and say nothing about the kernel); + its three buffer stores per step; + its input loads.  Operands are
whatever the registers hold, so this measures instruction issue, not the memory system: DESIGN.md section 6 sets
these numbers (193 / 228 / 254 / 276 cycles per step) against the kernel's own ~350."""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LISTING = os.path.join(ROOT, "ans_large_alphabet_amd", "csrc", "ansx_gfx950.s")
KERNEL = "_Z8k_encodeILi1ELb1E"


def loop_body():
    lines = open(LISTING).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().endswith(":") or (l.startswith(KERNEL) and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    fn = lines[start:end]
    ff = [i for i, l in enumerate(fn) if "v_ffbh_u32" in l]
    first, last = ff[3], ff[-1]          # the three before are the prologue's stage A calls
    hdr = max(i for i in range(first) if "Loop Header" in fn[i])
    tail = next(i for i in range(last, len(fn)) if "s_cbranch" in fn[i])
    body = [re.sub(r";.*", "", l).strip() for l in fn[hdr + 1:tail]]
    return [l for l in body if l and not l.startswith(".")]


def build(body, lds, stores, loads):
    keep = []
    for t in body:
        if t.startswith(("s_cbranch", "s_and_saveexec", "s_branch", "global_", "s_barrier", "s_waitcnt")):
            continue
        if "exec" in t and t.startswith("s_"):
            continue
        if t.startswith("ds_read"):
            if lds:
                m = re.match(r"ds_read_u16 (v\d+), (v\d+)(.*)", t)
                keep += ["v_and_b32 v250, 0x7ffc, %s" % m.group(2), "ds_read_u16 %s, v250%s" % (m.group(1), m.group(3))]
            continue
        if t.startswith("buffer_store") and not stores:
            continue
        if t.startswith("buffer_load") and not loads:
            continue
        keep.append(t)
    return keep


def registers(body):
    v, s, rs = {250}, set(), set()
    for t in body:
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
            v.update(range(int(m.group(1)), int(m.group(2)) + 1))
        v.update(int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", t))
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", t):
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        s.update(int(m.group(1)) for m in re.finditer(r"\bs(\d+)\b", t))
        m = re.match(r"buffer_\w+\s+\S+,\s+\S+,\s+s\[(\d+):(\d+)\]", t)
        if m:
            rs.add(int(m.group(1)))
    return sorted(v), sorted(s), sorted(rs)


def main(out):
    body = loop_body()
    v, s, rs = registers(body)
    clob = ", ".join('"v%d"' % i for i in v) + ", " + ", ".join('"s%d"' % i for i in s) + ', "vcc", "memory"'
    src = "#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <stdint.h>\n"
    forms = [("k_valu", (0, 0, 0), "VALU + SALU only"), ("k_lds", (1, 0, 0), "+ ds_read_u16 pairs"),
             ("k_st", (1, 1, 0), "+ 3 buffer stores per step"), ("k_all", (1, 1, 1), "+ input loads")]
    for name, f, _ in forms:
        ks = build(body, *f)
        src += "__global__ __launch_bounds__(64) void %s(uint64_t* out, int iters, unsigned char* scr)\n{\n" % name
        src += "    extern __shared__ unsigned lds[];\n    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i * 2654435761u;\n    __syncthreads();\n"
        for i in v:
            src += '    asm volatile("v_mov_b32 v%d, %d" ::: "v%d");\n' % (i, (i * 37) % 200 + 1, i)
        for i in s:
            src += '    asm volatile("s_mov_b32 s%d, %d" ::: "s%d");\n' % (i, (i * 13) % 60 + 1, i)
        src += "    { const uint64_t ba = (uint64_t)(uintptr_t)(scr + (uint64_t)blockIdx.x * 2097152ull);\n"
        src += "      const unsigned w0 = __builtin_amdgcn_readfirstlane((unsigned)ba), w1 = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xFFFFu);\n"
        for a in rs:
            src += '      asm volatile("s_mov_b32 s%d, %%0\\n\\ts_mov_b32 s%d, %%1\\n\\ts_mov_b32 s%d, 0x200000\\n\\ts_mov_b32 s%d, 0x00020000" :: "s"(w0), "s"(w1) : "s%d", "s%d", "s%d", "s%d");\n' % (
                a, a + 1, a + 2, a + 3, a, a + 1, a + 2, a + 3)
        src += "    }\n    uint64_t t0 = __builtin_amdgcn_s_memtime();\n    for (int it = 0; it < iters; it++) {\n"
        src += '        asm volatile("%s" ::: %s);\n    }\n' % ("\\n\\t".join(ks), clob)
        src += '    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_nop 0" ::: "memory");\n    uint64_t t1 = __builtin_amdgcn_s_memtime();\n'
        src += "    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;\n    if (lds[threadIdx.x] == 0x1234567) out[0] = 0;\n}\n"
    src += "typedef void (*kfn)(uint64_t*, int, unsigned char*);\n"
    src += "static void run(kfn k, const char* what, uint64_t* d, unsigned char* scr)\n{\n    static uint64_t h[1024];\n"
    src += "    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1024), dim3(64), 32768, 0, d, 64, scr);\n"
    src += "    hipDeviceSynchronize();\n    hipMemcpy(h, d, 1024 * 8, hipMemcpyDeviceToHost);\n"
    src += '    printf("%-32s %.1f cycles per step\\n", what, h[5] / (64.0 * 32));\n}\n'
    src += "int main()\n{\n    uint64_t* d; hipMalloc(&d, 1024 * 8);\n    unsigned char* scr; hipMalloc(&scr, (size_t)1024 * 2097152);\n"
    for name, _, desc in forms:
        src += '    run(%s, "%s", d, scr);\n' % (name, desc)
    src += "    return 0;\n}\n"
    open(out, "w").write(src)
    print("%d instructions in the loop body; wrote %s" % (len(body), out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/replay_encoder_loop.hip")

// Micro-benchmark (manual tool, round 4): per-SIMD cost of the operations the decoder step and a producer/consumer
// encoder would be made of, at 1 / 2 / 4 waves per SIMD, and of MIXED pairs (an f64 wave beside an integer wave on
// the same SIMD), plus the LDS hand-over instructions.  Same method as ubench_valu.hip (loops re-executed from the
// instruction cache, 8 independent chains).
//   hipcc --offload-arch=gfx950 -O2 ubench_valu2.hip -o ubench_valu2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 400
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define K64(name, OP)                                                                                        \
__global__ void ind_##name(uint64_t* out, double b0, int role) {                                             \
    uint32_t tmp[8]; double a[8]; for (int j = 0; j < 8; j++) { a[j] = 1.5 + threadIdx.x + j; tmp[j] = j + threadIdx.x; }  \
    double b = b0; uint64_t t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(OP : "+v"(a[j]), "+v"(tmp[j]) : "v"(b)); } } \
    asm volatile("s_nop 0" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();                        \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                        \
    double s = 0; for (int j = 0; j < 8; j++) s += a[j] + tmp[j]; if (s == 12345.678) out[0] = 0; }
K64(fma64, "v_fma_f64 %0, %0, %2, %2")
K64(trunc64, "v_trunc_f64 %0, %0")
K64(rcp64, "v_rcp_f64 %0, %0")
K64(cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
K64(cvt_u32_f64, "v_cvt_u32_f64 %1, %0")
K64(cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
K64(cmp_f64, "v_cmp_ge_f64 s[40:41], %0, %2")
K64(mad_u64_u32, "v_mad_u64_u32 %0, s[40:41], %1, %1, %0")
K64(lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")
K64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %2")
K64(add32, "v_add_u32 %1, %1, %1")
K64(mul_hi_u32, "v_mul_hi_u32 %1, %1, %1")
K64(mul_lo_u32, "v_mul_lo_u32 %1, %1, %1")
K64(mad24, "v_mad_u32_u24 %1, %1, %1, %1")
K64(bcnt, "v_bcnt_u32_b32 %1, %1, %1")
K64(alignbyte, "v_alignbyte_b32 %1, %1, %1, 1")
K64(cndmask, "v_cndmask_b32_e64 %1, %1, %1, s[42:43]")
K64(sad8, "v_sad_u8 %1, %1, %1, %1")
K64(dppadd, "v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
K64(sdwa_sub, "v_sub_u32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
K64(add3, "v_add3_u32 %1, %1, %1, %1")
K64(lshl_or, "v_lshl_or_b32 %1, %1, 3, %1")
K64(perm, "v_perm_b32 %1, %1, %1, %1")
K64(fma32, "v_fma_f32 %1, %1, %1, %1")
K64(pk_fma32, "v_pk_fma_f32 %0, %0, %0, %0")

// a mixed SIMD: waves 0..3 of a 512-thread workgroup run 8 chains of v_fma_f64, waves 4..7 8 chains of v_add_u32
// (role 0: both; role 1: only the f64 waves work; role 2: only the integer waves)
__global__ void mix_f64_int(uint64_t* out, double b0, int role) {
    uint32_t tmp[8]; double a[8]; for (int j = 0; j < 8; j++) { a[j] = 1.5 + threadIdx.x + j; tmp[j] = j + threadIdx.x; }
    double b = b0; const bool second = threadIdx.x >= 256;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (!second) {
        if (role != 2)
        for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile("v_fma_f64 %0, %0, %2, %2" : "+v"(a[j]), "+v"(tmp[j]) : "v"(b)); } }
    } else {
        if (role != 1)
        for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile("v_add_u32 %1, %1, %1" : "+v"(a[j]), "+v"(tmp[j]) : "v"(b)); } }
    }
    asm volatile("s_nop 0" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    double s = 0; for (int j = 0; j < 8; j++) s += a[j] + tmp[j]; if (s == 12345.678) out[0] = 0;
}
// the same with a DEPENDENT f64 chain (what the state wave is) beside independent integer work
__global__ void mix_dep64_int(uint64_t* out, double b0, int role) {
    uint32_t tmp[8]; double a[8]; for (int j = 0; j < 8; j++) { a[j] = 1.5 + threadIdx.x + j; tmp[j] = j + threadIdx.x; }
    double b = b0; const bool second = threadIdx.x >= 256;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (!second) {
        if (role != 2)
        for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile("v_fma_f64 %0, %0, %2, %2" : "+v"(a[0]), "+v"(tmp[0]) : "v"(b)); }
    } else {
        if (role != 1)
        for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile("v_add_u32 %1, %1, %1" : "+v"(a[j]), "+v"(tmp[j]) : "v"(b)); } }
    }
    asm volatile("s_nop 0" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    double s = 0; for (int j = 0; j < 8; j++) s += a[j] + tmp[j]; if (s == 12345.678) out[0] = 0;
}

// LDS hand-over instructions: every wave of the workgroup runs 64 x ITER of them on its own 1 KB / 4 KB slice
#define KLDS(name, OP, BYTES)                                                                                \
__global__ void lds_##name(uint64_t* out, double b0, int role) {                                             \
    extern __shared__ uint32_t sm[];                                                                         \
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;                                           \
    uint32_t addr = wv * 8192 + lane * BYTES; uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;            \
    asm volatile("s_mov_b32 m0, %0" :: "s"(__builtin_amdgcn_readfirstlane(wv * 8192)));                      \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile(OP : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(addr) : "memory"); } \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();           \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                        \
    if (v0 + v1 + v2 + v3 == 123456789u) out[0] = sm[0]; }
KLDS(write_b32, "ds_write_b32 %4, %0", 4)
KLDS(write_addtid, "ds_write_addtid_b32 %0", 4)
KLDS(read_b32, "ds_read_b32 %1, %4", 4)
KLDS(read_addtid, "ds_read_addtid_b32 %1", 4)
KLDS(read_u16, "ds_read_u16 %1, %4", 4)

__global__ void lds_write_b128(uint64_t* out, double b0, int role) {
    extern __shared__ uint32_t sm[];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t addr = wv * 8192 + lane * 16; u32x4 v = { threadIdx.x, 1, 2, 3 };
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile("ds_write_b128 %1, %0" : "+v"(v) : "v"(addr) : "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (v.x == 123456789u) out[0] = sm[0];
}
__global__ void lds_write_b64(uint64_t* out, double b0, int role) {
    extern __shared__ uint32_t sm[];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t addr = wv * 8192 + lane * 8; u32x2 v = { threadIdx.x, 1 };
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile("ds_write_b64 %1, %0" : "+v"(v) : "v"(addr) : "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (v.x == 123456789u) out[0] = sm[0];
}
__global__ void lds_read_b128(uint64_t* out, double b0, int role) {
    extern __shared__ uint32_t sm[];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t addr = wv * 8192 + lane * 16; u32x4 v = { threadIdx.x, 1, 2, 3 };
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile("ds_read_b128 %0, %1" : "+v"(v) : "v"(addr) : "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (v.x == 123456789u) out[0] = sm[0];
}
__global__ void lds_read_b64_rand(uint64_t* out, double b0, int role) {  // a random 8-byte gather, as the decoder's tables
    extern __shared__ uint32_t sm[];
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t addr = wv * 8192 + ((threadIdx.x * 2654435761u) >> 19 & 0x1FF8u); u32x2 v = { threadIdx.x, 1 };
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile("ds_read_b64 %0, %1" : "+v"(v) : "v"(addr) : "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (v.x == 123456789u) out[0] = sm[0];
}

typedef void (*kfn)(uint64_t*, double, int);
struct ent { const char* name; kfn k; };
#define E(n) { #n, ind_##n }
int main()
{
    ent tab[] = { E(fma64), E(trunc64), E(rcp64), E(cvt_f64_u32), E(cvt_u32_f64), E(cvt_f64_f32), E(cmp_f64), E(mad_u64_u32), E(lshrrev_b64), E(lshl_add_u64),
        E(add32), E(mul_hi_u32), E(mul_lo_u32), E(mad24), E(bcnt), E(alignbyte), E(cndmask), E(sad8), E(dppadd), E(sdwa_sub), E(add3), E(lshl_or), E(perm), E(fma32), E(pk_fma32) };
    uint64_t* d; hipMalloc(&d, 1024 * 16 * 8);
    static uint64_t h[1024 * 16];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct cfg { int grid, block; } cfgs[] = { {1024, 64}, {256, 256}, {256, 512}, {256, 1024} };
    auto run = [&](kfn k, int grid, int block, int role, size_t lds, double* ticks_first, double* ticks_last) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d, 1.000001, role);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h, d, (size_t)grid * 16 * 8, hipMemcpyDeviceToHost);
        const double n = (double)ITER * 64;
        *ticks_first = (double)h[0] / n; *ticks_last = (double)h[block / 64 - 1] / n;
        return ms * 1e6 / n;
    };
    for (auto c : cfgs) {
        printf("grid %d x %d threads: per instruction per wave: wall ns | s_memtime ticks (wave 0)\n", c.grid, c.block);
        for (auto& e : tab) {
            double t0, t1; double ns = run(e.k, c.grid, c.block, 0, 0, &t0, &t1);
            printf("  %-14s %7.2f | %6.2f\n", e.name, ns, t0);
        }
    }
    printf("mixed SIMD (512 threads: waves 0-3 f64 fma, waves 4-7 v_add_u32), ticks per instruction: f64 wave | int wave\n");
    for (int role = 0; role < 3; role++) {
        double t0, t1; double ns = run(mix_f64_int, 256, 512, role, 0, &t0, &t1);
        printf("  independent f64 chains, role %d: wall %7.2f ns   f64 %6.2f | int %6.2f\n", role, ns, t0, t1);
        ns = run(mix_dep64_int, 256, 512, role, 0, &t0, &t1);
        printf("  dependent f64 chain,    role %d: wall %7.2f ns   f64 %6.2f | int %6.2f\n", role, ns, t0, t1);
    }
    ent lt[] = { {"write_b32", lds_write_b32}, {"write_b64", lds_write_b64}, {"write_addtid", lds_write_addtid}, {"write_b128", lds_write_b128}, {"read_b32", lds_read_b32},
        {"read_addtid", lds_read_addtid}, {"read_u16", lds_read_u16}, {"read_b128", lds_read_b128}, {"read_b64_rand", lds_read_b64_rand} };
    struct cfg lc[] = { {256, 64}, {256, 256}, {256, 512} };
    for (auto c : lc) {
        printf("LDS, grid %d x %d threads (one workgroup per CU): ticks per wave-instruction (wave 0) | wall ns\n", c.grid, c.block);
        for (auto& e : lt) {
            double t0, t1; double ns = run(e.k, c.grid, c.block, 0, 65536, &t0, &t1);
            printf("  %-14s %6.2f | %7.2f\n", e.name, t0, ns);
        }
    }
    return 0;
}

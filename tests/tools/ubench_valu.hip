// Micro-benchmark (manual tool): per-wave cost of the VALU operations the encoder / decoder steps are made of, on gfx950.
// Every wave runs ITER passes over a 64-deep dependent chain ("dep") or over 8 independent chains of 64 ("ind"); the loop
// bodies are re-executed from the instruction cache (a once-through sequence measures instruction fetch instead: ~8 cycles
// per instruction).  Reported: wall-clock ns and s_memtime ticks per operation per wave, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 400
#define TAIL(T) asm volatile("s_nop 0" ::: "memory"); uint64_t t1 = __builtin_amdgcn_s_memtime();           \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
#define K64(name, OP)                                                                                        \
__global__ void dep_##name(uint64_t* out, double b0) {                                                       \
    uint32_t tmp = threadIdx.x; double a = 1.5 + threadIdx.x, b = b0;                                        \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile(OP : "+v"(a), "+v"(tmp) : "v"(b)); } \
    TAIL() if (a == 12345.678) out[0] = tmp; }                                                               \
__global__ void ind_##name(uint64_t* out, double b0) {                                                       \
    uint32_t tmp[8]; double a[8]; for (int j = 0; j < 8; j++) { a[j] = 1.5 + threadIdx.x + j; tmp[j] = j; }  \
    double b = b0; uint64_t t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(OP : "+v"(a[j]), "+v"(tmp[j]) : "v"(b)); } } \
    TAIL() double s = 0; for (int j = 0; j < 8; j++) s += a[j] + tmp[j]; if (s == 12345.678) out[0] = 0; }
#define K32(name, OP)                                                                                        \
__global__ void dep_##name(uint64_t* out, double b0) {                                                       \
    uint32_t tmp = threadIdx.x; uint32_t a = 5 + threadIdx.x, b = (uint32_t)b0 + 3;                          \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                              \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 64; i++) asm volatile(OP : "+v"(a), "+v"(tmp) : "v"(b)); } \
    TAIL() if (a == 123456789u) out[0] = tmp; }                                                              \
__global__ void ind_##name(uint64_t* out, double b0) {                                                       \
    uint32_t tmp[8]; uint32_t a[8]; for (int j = 0; j < 8; j++) { a[j] = 5 + threadIdx.x + j; tmp[j] = j; }  \
    uint32_t b = (uint32_t)b0 + 3; uint64_t t0 = __builtin_amdgcn_s_memtime();                               \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(OP : "+v"(a[j]), "+v"(tmp[j]) : "v"(b)); } } \
    TAIL() uint32_t s = 0; for (int j = 0; j < 8; j++) s += a[j] + tmp[j]; if (s == 123456789u) out[0] = 0; }
K64(fma64, "v_fma_f64 %0, %0, %2, %2")
K64(add64, "v_add_f64 %0, %0, %2")
K64(mul64, "v_mul_f64 %0, %0, %2")
K64(trunc64, "v_trunc_f64 %0, %0")
K64(ldexp64, "v_ldexp_f64 %0, %0, 1")
K64(rcp64, "v_rcp_f64 %0, %0")
K64(fmaclamp64, "v_fma_f64 %0, %0, %2, %2 clamp")
K64(cvtpair, "v_cvt_u32_f64 %1, %0\n\tv_cvt_f64_u32 %0, %1")
K64(cmp64_ldexp, "v_cmp_ge_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, 1, vcc\n\tv_ldexp_f64 %0, %0, %1")
K64(cmphi_ldexp, "v_cmp_ge_u32 vcc, %1, %1\n\tv_cndmask_b32 %1, 0, 1, vcc\n\tv_ldexp_f64 %0, %0, %1")
K32(add32, "v_add_u32 %0, %0, %2")
K32(mullo32, "v_mul_lo_u32 %0, %0, %2")
K32(mad24, "v_mad_u32_u24 %0, %0, %2, %2")
K32(sad8, "v_sad_u8 %0, %0, %2, %2")
K32(lshladd, "v_lshl_add_u32 %0, %0, 1, %2")
K32(ffbh, "v_ffbh_u32 %0, %0")
K32(dppadd_nop, "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
K32(cmpsel32, "v_cmp_ge_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc")
K32(nop0, "s_nop 0")
K32(nop1, "s_nop 1")
K32(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
K32(rcp32, "v_rcp_f32 %0, %0")
K32(subclamp, "v_sub_u32 %0, %0, %2 clamp")

typedef void (*kfn)(uint64_t*, double);
struct ent { const char* name; kfn dep, ind; int nops; };
#define E(n, c) { #n, dep_##n, ind_##n, c }
int main()
{
    ent tab[] = { E(fma64, 1), E(add64, 1), E(mul64, 1), E(trunc64, 1), E(ldexp64, 1), E(rcp64, 1), E(fmaclamp64, 1), E(cvtpair, 2), E(cmp64_ldexp, 3), E(cmphi_ldexp, 3), E(add32, 1), E(mullo32, 1), E(mad24, 1), E(sad8, 1), E(lshladd, 1), E(ffbh, 1), E(dppadd_nop, 2), E(cmpsel32, 2), E(nop0, 1), E(nop1, 1), E(cvt_f32_u32, 1), E(rcp32, 1), E(subclamp, 1) };
    uint64_t* d; hipMalloc(&d, 1024 * 16 * 8);
    static uint64_t h[1024 * 16];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct cfg { int grid, block; } cfgs[] = { {1024, 64}, {256, 512}, {256, 1024} };
    for (auto c : cfgs) {
        printf("grid %d x %d threads (%d wave(s) per SIMD): per listed sequence per wave: dependent chain [wall ns | s_memtime ticks]   8 independent chains [wall ns | ticks]\n", c.grid, c.block, c.block / 256 ? c.block / 256 : 1);
        for (auto& e : tab) {
            double ns[2], tk[2];
            for (int m = 0; m < 2; m++) {
                kfn k = m ? e.ind : e.dep;
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.block), 0, 0, d, 1.000001);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                hipMemcpy(h, d, (size_t)c.grid * 16 * 8, hipMemcpyDeviceToHost);
                const double n = (double)ITER * 64;
                ns[m] = ms * 1e6 / n; tk[m] = (double)h[0] / n;
            }
            printf("  %-12s (%d instr)  %7.2f | %6.2f      %7.2f | %6.2f\n", e.name, e.nops, ns[0], tk[0], ns[1], tk[1]);
        }
    }
    return 0;
}

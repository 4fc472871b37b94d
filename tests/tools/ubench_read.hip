// Micro-benchmark (manual tool): what a read-only streaming kernel of k_fold_hist's shape reaches on MI355X, with and
// without the LDS histogram work -- the ceiling k_fold_hist is priced against in DESIGN.md section 6.
//   hipcc --offload-arch=gfx950 -O3 ubench_read.hip -o ubench_read && ./ubench_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32;
// KIND 0: sum only.  1: LDS atomics into 4 copies (value & 511).  2: as 1, one copy.
// INFL = independent 16-byte loads per thread in flight; CHUNK_V = uint4 per workgroup.
template <int KIND, int INFL, bool NT> __global__ __launch_bounds__(256) void rd(const uint4* __restrict__ in, u32 nvec_wg, u32* out)
{
    __shared__ u32 h[4 * 520];
    const u32 tid = threadIdx.x;
    if (KIND) { for (u32 s = tid; s < 4 * 520; s += 256) h[s] = 0; __syncthreads(); }
    const uint4* v4 = in + (size_t)blockIdx.x * nvec_wg;
    u32 acc = 0;
    u32* my = h + (KIND == 1 ? (tid & 3) * 520 : 0);
    for (u32 v = tid; v < nvec_wg; v += INFL * 256) {
        uint4 q[INFL];
#pragma unroll
        for (int j = 0; j < INFL; j++) {
            if (NT) {
                const u32* p = (const u32*)&v4[v + j * 256];
                q[j].x = __builtin_nontemporal_load(p); q[j].y = __builtin_nontemporal_load(p + 1);
                q[j].z = __builtin_nontemporal_load(p + 2); q[j].w = __builtin_nontemporal_load(p + 3);
            } else q[j] = v4[v + j * 256];
        }
#pragma unroll
        for (int j = 0; j < INFL; j++) {
            if (KIND == 0) acc += q[j].x + q[j].y + q[j].z + q[j].w;
            else {
                atomicAdd(&my[q[j].x & 511], 1u); atomicAdd(&my[q[j].y & 511], 1u);
                atomicAdd(&my[q[j].z & 511], 1u); atomicAdd(&my[q[j].w & 511], 1u);
            }
        }
    }
    if (KIND) { __syncthreads(); acc = h[tid] + h[tid + 520]; }
    if (acc == 0x12345) out[blockIdx.x] = acc;
}
template <int KIND, int INFL, bool NT> void run(const uint4* d, size_t nvec, u32 nvec_wg, u32* out, const char* what)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, ms;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<KIND, INFL, NT>), dim3(nvec / nvec_wg), dim3(256), 0, 0, d, nvec_wg, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-64s %.3f ms  %.2f TB/s\n", what, best, nvec * 16.0 / best / 1e9);
}
int main()
{
    const size_t bytes = (size_t)1 << 30, nvec = bytes / 16;
    uint4* d; u32* out; hipMalloc(&d, bytes); hipMalloc(&out, 1 << 20);
    u32* hbuf = (u32*)malloc(bytes);
    u32 x = 12345;
    for (size_t i = 0; i < bytes / 4; i++) { x = x * 1664525u + 1013904223u; u32 r = x >> 8; hbuf[i] = (r & 0xff) < 200 ? (r >> 8) & 63 : (r >> 8) & 511; }
    hipMemcpy(d, hbuf, bytes, hipMemcpyHostToDevice);
    run<0, 16, false>(d, nvec, 4096, out, "sum, 64 KiB per workgroup, 16 loads in flight");
    run<0, 8, false>(d, nvec, 4096, out, "sum, 64 KiB per workgroup, 8 loads in flight");
    run<0, 4, false>(d, nvec, 4096, out, "sum, 64 KiB per workgroup, 4 loads in flight");
    run<0, 16, true>(d, nvec, 4096, out, "sum, 64 KiB per workgroup, 16 in flight, nontemporal");
    run<0, 16, false>(d, nvec, 16384, out, "sum, 256 KiB per workgroup, 16 in flight");
    run<0, 8, false>(d, nvec, 65536, out, "sum, 1 MiB per workgroup, 8 in flight");
    run<1, 16, false>(d, nvec, 4096, out, "4-copy LDS histogram, 64 KiB per workgroup, 16 in flight");
    run<1, 8, false>(d, nvec, 4096, out, "4-copy LDS histogram, 64 KiB per workgroup, 8 in flight");
    run<2, 16, false>(d, nvec, 4096, out, "1-copy LDS histogram, 64 KiB per workgroup, 16 in flight");
    run<1, 8, false>(d, nvec, 65536, out, "4-copy LDS histogram, 1 MiB per workgroup, 8 in flight");
    return 0;
}

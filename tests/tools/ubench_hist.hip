// Where does k_fold_hist's time go?  The product kernel on a skewed 256 Mi-int input (4096 workgroups of one
// 16 Ki-int block), built with the HX_* switches of ansx_kernels.h to remove one phase at a time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DHX_NO_ATOM ...] -o ubench_hist ubench_hist.hip
#include "../../ans_large_alphabet_amd/csrc/ansx_kernels.h"
#include <cstdio>
__global__ void k_fill(u32* out, u64 n)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        out[i] = (u32)(((h & 0xFFFFF) >> ((h >> 40) % 21)) + 1);  // about half of the values below 4
    }
}
int main(int argc, char** argv)
{
    const u64 n = 256ull << 20;
    const u32 f = 2, NSP = fold_NSP(f), B = 16384, NB = (u32)(n / B);
    u32 *in, *hist, *gflags;
    ansx_blk* blk;
    hipMalloc(&in, n * 4);
    hipMalloc(&hist, (size_t)NB * NSP * 4);
    hipMalloc(&gflags, 4096);
    hipMalloc(&blk, (size_t)NB * sizeof(ansx_blk));
    hipMemset(blk, 0, (size_t)NB * sizeof(ansx_blk));
    hipMemset(gflags, 0, 4096);
    k_fill<<<4096, 256>>>(in, n);
    ansx_geo g = {};
    g.n = n; g.block_ints = B; g.nblocks = NB; g.f = f; g.kind = 0; g.map = map_fold(f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char* name; u32 mode; size_t lds; } cfg[] = {
        { "4 copies u32 + terms in LDS (exact path)", 1u, (size_t)4 * (NSP + 8) * 4 + (size_t)NSP * 8 + 80 },
        { "4 copies u32, tree sum", 3u, (size_t)4 * (NSP + 8) * 4 },
        { "packed x4, tree sum", 7u | (4u << 4), (size_t)4 * (NSP / 2 + 8) * 4 + 64 },
        { "packed x8, tree sum", 7u | (8u << 4), (size_t)8 * (NSP / 2 + 8) * 4 + 64 },
    };
    for (auto& c : cfg) {
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fold_hist, dim3(NB), dim3(256), c.lds, 0, in, g, B, 1u, NSP, hist, (double*)nullptr, c.mode, blk, gflags, 1u << 30);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-45s %.1f us  (%.2f TB/s of input)\n", c.name, best * 1e3, n * 4 / (best * 1e-3) / 1e12);
    }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}

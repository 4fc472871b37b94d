// K9: ANSrfold remap (ans_reorder_fold.hpp:70-106).  Per block: find the T = 2^(f+7) most
// frequent values, ordered by (count descending, value ascending) — the order std::sort gives
// the reference's (-count, value) pairs (:79-85) — and rewrite every value v as its rank if it is
// one of them, else v + T (:98-106).  If the block has fewer than T distinct values the mapping
// is the identity and the flag is 0 (:94-97).
//
// The reference sorts a (max+1)-entry vector; here the block's values are sorted in LDS (bitonic),
// runs are measured by binary search, the count threshold is bisected, and only the T selected
// runs are sorted by rank.  The remapped block is written to HBM and then flows through the
// same histogram / normalise / encode kernels as ANSfold.
#pragma once

#include "ansx_kernels.h"

__device__ __forceinline__ u32 lds_lower_bound(const u32* a, u32 n, u32 v)
{
    u32 lo = 0, hi = n;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ u32 lds_upper_bound(const u32* a, u32 n, u32 v)
{
    u32 lo = 0, hi = n;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <typename K> __device__ __forceinline__ void lds_bitonic_sort(K* keys, u32 N2, u32 tid)
{
    for (u32 k = 2; k <= N2; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < N2; i += 256) {
                u32 ixj = i ^ j;
                if (ixj > i) {
                    bool asc = (i & k) == 0;
                    K x = keys[i], y = keys[ixj];
                    if ((x > y) == asc) {
                        keys[i] = y;
                        keys[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void k_rfold_remap(const u32* __restrict__ in, ansx_geo g, u32 N2,
    u32* __restrict__ mapped, u32* __restrict__ mostfreq, ansx_blk* __restrict__ blk,
    u32* __restrict__ gflags)
{
    extern __shared__ u8 smem_rf[];
    __shared__ u32 sh_cnt;
    __shared__ u32 sh_max;
    __shared__ u32 sh_part[256];
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const u32 nb = geo_block_n(g, b);
    const u32 T = fold_T(g.f);
    u32* vals = (u32*)smem_rf;
    u16* aux = (u16*)(smem_rf + 4 * (size_t)N2);
    u64* sel = (u64*)(smem_rf + 6 * (size_t)N2);
    const u32* src = in + (u64)b * g.block_ints;
    u32* dst = mapped + (u64)b * g.block_ints;
    if (tid == 0) {
        sh_cnt = 0;
        sh_max = 0;
    }
    __syncthreads();
    u32 lmax = 0;
    for (u32 i = tid; i < N2; i += 256) {
        u32 v = i < nb ? src[i] : 0xFFFFFFFFu;
        vals[i] = v;
        if (i < nb) lmax = v > lmax ? v : lmax;
    }
    atomicMax(&sh_max, lmax);
    __syncthreads();
    lds_bitonic_sort<u32>(vals, N2, tid);
    // run heads and run lengths
    u32 lheads = 0;
    for (u32 i = tid; i < nb; i += 256) {
        u32 v = vals[i];
        bool head = (i == 0) || (vals[i - 1] != v);
        u32 cnt = 0;
        if (head) {
            cnt = lds_upper_bound(vals, nb, v) - i;
            lheads++;
        }
        aux[i] = (u16)cnt;  // cnt <= 16384
    }
    atomicAdd(&sh_cnt, lheads);
    __syncthreads();
    const u32 sigma = sh_cnt;
    const u32 vmax = sh_max;
    if (sigma < T) {  // ans_reorder_fold.hpp:94-97: identity mapping, flag 0
        for (u32 i = tid; i < nb; i += 256) dst[i] = src[i];
        if (tid == 0) {
            blk[b].flag = 0;
            if (vmax >= (1u << 30)) atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
        }
        return;
    }
    if (tid == 0 && (u64)vmax + T >= (1u << 30)) atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
    // largest c with #runs(count >= c) >= T
    auto count_ge = [&](u32 c) -> u32 {
        __syncthreads();
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        u32 l = 0;
        for (u32 i = tid; i < nb; i += 256) l += (aux[i] >= c) ? 1u : 0u;
        atomicAdd(&sh_cnt, l);
        __syncthreads();
        return sh_cnt;
    };
    u32 lo = 1, hi = nb;
    while (lo < hi) {
        u32 mid = (lo + hi + 1) >> 1;
        if (count_ge(mid) >= T) lo = mid;
        else hi = mid - 1;
    }
    const u32 cstar = lo;
    const u32 G = count_ge(cstar + 1);  // runs strictly above the threshold: all selected
    const u32 K = T - G;                // runs at the threshold: the K smallest values
    // prefix count of threshold runs in position (= value) order
    const u32 per = (nb + 255) / 256;
    const u32 plo = tid * per, phi = (plo + per) < nb ? (plo + per) : nb;
    u32 s = 0;
    for (u32 i = plo; i < phi; i++) s += (aux[i] == cstar) ? 1u : 0u;
    __syncthreads();
    sh_part[tid] = s;
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    u32 run = 0;
    for (u32 l = 0; l < tid; l++) run += sh_part[l];
    for (u32 i = plo; i < phi; i++) {
        u32 cnt = aux[i];
        bool take = cnt > cstar;
        if (cnt == cstar) {
            take = run < K;
            run++;
        }
        if (take) {
            u32 slot = atomicAdd(&sh_cnt, 1u);
            sel[slot] = ((u64)(0xFFFFFFFFu - cnt) << 32) | (u64)vals[i];  // (-count, value)
        }
    }
    __syncthreads();
    lds_bitonic_sort<u64>(sel, T, tid);
    for (u32 i = tid; i < nb; i += 256) aux[i] = 0xFFFFu;
    __syncthreads();
    u32* mf = mostfreq + (u64)b * T;
    for (u32 r = tid; r < T; r += 256) {
        u32 v = (u32)sel[r];
        mf[r] = v;  // ans_reorder_fold.hpp:104-105
        aux[lds_lower_bound(vals, nb, v)] = (u16)r;
    }
    __syncthreads();
    for (u32 i = tid; i < nb; i += 256) {
        u32 v = src[i];
        u32 r = aux[lds_lower_bound(vals, nb, v)];
        dst[i] = (r != 0xFFFFu) ? r : v + T;  // :99-103
    }
    if (tid == 0) blk[b].flag = 1;
}

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

template <typename K> __device__ __forceinline__ void lds_bitonic_sort(K* keys, u32 N2, u32 tid, u32 nt = 256)
{
    for (u32 k = 2; k <= N2; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < N2; i += nt) {
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


// ------------------------------------------------------------------------------------------
// K9 (fast form, T <= 4096): the same selection without sorting the block.  Distinct values and
// their counts are collected in an LDS open-addressing hash table; the count threshold c* and,
// among the values with count == c*, the value threshold v* are found by bisection over the
// table; only the T selected (count, value) pairs are sorted (by (-count, value), the order of
// ans_reorder_fold.hpp:79-85); ranks are written back into the table and every input value is
// remapped with one probe.
// ------------------------------------------------------------------------------------------
#define ANSX_RF_SLOTS 20480u  // >= 1.25 x 16384 values per block
#define ANSX_RF_EMPTY 0xFFFFFFFFu

__device__ __forceinline__ u32 rf_slot(u32 v, u32 slots) { return (u32)(((u64)(v * 2654435761u) * slots) >> 32); }

// Launched with 1024 threads: the full-size table (ANSX_RF_SLOTS slots, any block of up to 16 Ki ints) occupies most
// of a CU's LDS, so this one workgroup is all the latency hiding its ~10 passes over the table get (8.9 -> see
// DESIGN.md with 256 threads).  A smaller `slots` is the optimistic form: the context has seen this
// geometry before and sizes the table for 1.5 x the most distinct values a block of it ever had, so that two
// workgroups share a CU (k_rfold_remap_hash2: 64 registers) and every pass is that much shorter; a block that
// does not fit raises the violation flag, writes zeros, and the caller repeats the call with the full table.
__device__ __forceinline__ void rfold_remap_hash_body(const u32* __restrict__ in, const ansx_geo& g, u32 slots,
    u32* __restrict__ mapped, u32* __restrict__ mostfreq, ansx_blk* __restrict__ blk,
    u32* __restrict__ gflags)
{
    extern __shared__ u8 smem_rh[];
    __shared__ u32 sh_cnt;
    __shared__ u32 sh_max;
    __shared__ u32 sh_ovf;
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const u32 b = blockIdx.x;
    const u32 nb = geo_block_n(g, b);
    const u32 T = fold_T(g.f);
    u32* keys = (u32*)smem_rh;                                      // [SLOTS]
    u32* cnt32 = (u32*)(smem_rh + 4 * (size_t)slots);              // [slots/2], two u16 counters each
    u64* sel = (u64*)(smem_rh + 6 * (size_t)slots);                // [T]  (slots is a multiple of 4)
    const u32* src = in + (u64)b * g.block_ints;
    u32* dst = mapped + (u64)b * g.block_ints;
    STAMP_RF(0);
#ifdef ANSX_STAMPS_RF
    const unsigned long long rf_t0 = wall_clock64();
#endif
    for (u32 i = tid; i < slots; i += nt) keys[i] = ANSX_RF_EMPTY;
    for (u32 i = tid; i < slots / 2; i += nt) cnt32[i] = 0;
    if (tid == 0) {
        sh_cnt = 0;
        sh_max = 0;
        sh_ovf = 0;
    }
    __syncthreads();
    auto count_of = [&](u32 slot) -> u32 { return (cnt32[slot >> 1] >> (16 * (slot & 1))) & 0xFFFFu; };
    // ---- insert: value -> count (counts <= 16384 fit 16 bits).  A thread's values (16 at 1024 threads and
    // 16 Ki-int blocks) are requested together: one at a time, each insert waited a global round trip.
    constexpr u32 RF_VPT = 16;
    u32 vals[RF_VPT];
#pragma unroll
    for (u32 q = 0; q < RF_VPT; q++) {
        const u32 i = tid + q * nt;
        vals[q] = i < nb ? src[i] : 0u;
    }
    STAMP_RF(1);
    u32 lmax = 0, ldistinct = 0;
    auto insert_one = [&](u32 v) {
        lmax = v > lmax ? v : lmax;
        u32 slot = rf_slot(v, slots);
        u32 probes = 0;
        for (;;) {
            const u32 old = atomicCAS(&keys[slot], ANSX_RF_EMPTY, v);
            if (old == ANSX_RF_EMPTY) ldistinct++;
            if (old == ANSX_RF_EMPTY || old == v) break;
            slot = slot + 1 == slots ? 0 : slot + 1;
            if (++probes >= slots) {  // table full (only possible with a table below ANSX_RF_SLOTS)
                sh_ovf = 1;
                return;
            }
        }
        atomicAdd(&cnt32[slot >> 1], 1u << (16 * (slot & 1)));
    };
#pragma unroll
    for (u32 q = 0; q < RF_VPT; q++)
        if (tid + q * nt < nb) insert_one(vals[q]);
    for (u32 i = tid + RF_VPT * nt; i < nb; i += nt) insert_one(src[i]);  // (fewer than 1024 threads)
    // (one atomic per wave: 1024 threads on the same two words queue up in the LDS unit)
    lmax = wave_max(lmax);
    ldistinct = wave_sum(ldistinct);
    if ((tid & 63u) == 0) {
        atomicMax(&sh_max, lmax);
        atomicAdd(&sh_cnt, ldistinct);
    }
    __syncthreads();
    STAMP_RF(2);
    if (sh_ovf) {  // optimistic table too small: a valid (all-zero, flag 0) block, and the call is repeated
        for (u32 i = tid; i < nb; i += nt) dst[i] = 0;
        if (tid == 0) {
            blk[b].flag = 0;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
            atomicMax(&gflags[ANSX_G_RFDIST], slots);  // (at least this many distinct values)
        }
        return;
    }
    const u32 sigma = sh_cnt;
    const u32 vmax = sh_max;
    if (tid == 0 && sigma > gflags[ANSX_G_RFDIST]) atomicMax(&gflags[ANSX_G_RFDIST], sigma);
    if (sigma < T) {  // ans_reorder_fold.hpp:94-97: identity mapping, flag 0
        for (u32 i = tid; i < nb; i += nt) dst[i] = src[i];
        if (tid == 0) {
            blk[b].flag = 0;
            if (vmax >= (1u << 30)) atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
        }
        return;
    }
    if (tid == 0 && (u64)vmax + T >= (1u << 30)) atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
    // Thresholds (no sorting of the block, no bisection):
    //   c*  = largest c with #values(count >= c) >= T.  T values of count >= c* exist, so c* <= nb / T (<= 64):
    //         one table pass over a histogram of min(count, CAPC).  Counts 1..4 -- nearly every slot of a skewed
    //         block -- are tallied in registers and reduced through the wave (20 Ki same-address LDS atomics
    //         cost more than everything else in this kernel together); the rest uses LDS atomics.
    //   v*  = K-th smallest value among those with count == c*: up to 3 levels of 10 bits below the largest
    //         value's bit length (levels placed there, not at bit 30, so that the first one already spreads).
    u32* hist = (u32*)sel;  // 1024 bins, reuses the (not yet used) selection buffer (T*8 >= 4096 B
                            // needs T >= 512; for T = 256 the buffer is sized for 512 entries)
    // returns bucket index; *before = total of the buckets passed before it
    auto find_bucket = [&](u32 nbins, u32 target, bool descending, u32* before) -> u32 {
        __syncthreads();
        if (tid < 64) {
            const u32 per = (nbins + 63) / 64;
            u32 loc = 0;
            for (u32 i = 0; i < per; i++) {
                u32 bin = tid * per + i;
                if (descending) bin = nbins - 1 - bin;
                loc += (tid * per + i < nbins) ? hist[bin] : 0u;
            }
            const u32 incl = wave_incl_scan(loc);
            const u32 excl = incl - loc;
            if (excl < target && incl >= target) {  // exactly one lane
                u32 run = excl, found = 0, bef = excl;
                for (u32 i = 0; i < per; i++) {
                    u32 bin = tid * per + i;
                    if (bin >= nbins) break;
                    u32 hb = descending ? nbins - 1 - bin : bin;
                    u32 c = hist[hb];
                    if (run < target && run + c >= target) {
                        found = hb;
                        bef = run;
                    }
                    run += c;
                }
                sh_cnt = found;
                sh_max = bef;
            }
        }
        __syncthreads();
        *before = sh_max;
        return sh_cnt;
    };
    auto clear_hist = [&](u32 nbins) {
        __syncthreads();
        for (u32 i = tid; i < nbins; i += nt) hist[i] = 0;
        __syncthreads();
    };
    u32 before;
    // ---- c*
    const u32 CAPC = nb / T + 1;  // counts >= CAPC exceed every possible c*
    clear_hist(CAPC + 1);
    {
        u32 n1 = 0, n2 = 0, n3 = 0, n4 = 0;
        for (u32 i = tid; i < slots; i += nt) {
            const u32 c = count_of(i);
            n1 += c == 1 ? 1u : 0u;
            n2 += c == 2 ? 1u : 0u;
            n3 += c == 3 ? 1u : 0u;
            n4 += c == 4 ? 1u : 0u;
            if (c > 4) atomicAdd(&hist[c < CAPC ? c : CAPC], 1u);
        }
        n1 = wave_sum(n1);
        n2 = wave_sum(n2);
        n3 = wave_sum(n3);
        n4 = wave_sum(n4);
        if ((tid & 63) == 0) {
            // (bins 1..4 exist: CAPC >= 4 whenever a count of 4 can be below it, else they fold into bin CAPC)
            if (n1) atomicAdd(&hist[1 < CAPC ? 1 : CAPC], n1);
            if (n2) atomicAdd(&hist[2 < CAPC ? 2 : CAPC], n2);
            if (n3) atomicAdd(&hist[3 < CAPC ? 3 : CAPC], n3);
            if (n4) atomicAdd(&hist[4 < CAPC ? 4 : CAPC], n4);
        }
    }
    const u32 cstar = find_bucket(CAPC + 1, T, true, &before);
    STAMP_RF(3);  // bin CAPC itself cannot be the answer: c* < CAPC
    const u32 G = before;            // values with count > c*: all selected
    const u32 K = T - G;             // K smallest values with count == c*
    __syncthreads();
    const u32 ties = hist[cstar];
    u32 vstar = 0xFFFFFFFFu;  // values at the threshold count are selected iff value <= vstar
    if (ties > K) {
        const u32 nbits = 32u - (u32)__builtin_clz(vmax | 1u);
        u32 shift = nbits > 10 ? nbits - 10 : 0;  // level 1: the top 10 bits that occur
        u32 prefix = 0, pshift = 32, need = K;    // candidates so far: (value >> pshift) == prefix
        for (;;) {
            clear_hist(1024);
            for (u32 i = tid; i < slots; i += nt) {
                const u32 k = keys[i];
                if (k != ANSX_RF_EMPTY && count_of(i) == cstar && (pshift >= 32 || (k >> pshift) == prefix))
                    atomicAdd(&hist[(k >> shift) & 1023u], 1u);
            }
            const u32 V = find_bucket(1024, need, false, &before);
            need -= before;
            prefix = pshift >= 32 ? V : ((prefix << (pshift - shift)) | V);
            pshift = shift;
            if (shift == 0) break;
            shift = shift > 10 ? shift - 10 : 0;
        }
        vstar = prefix;
    }
    __syncthreads();
    STAMP_RF(4);
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    // ---- collect the T selected (count, value) pairs: every wave counts its share, reserves it with ONE LDS
    // atomic and fills it in slot order (an atomic per round and wave was a dependent LDS round trip each)
    {
        auto taken = [&](u32 i, u32& k, u32& c) -> bool {
            k = ANSX_RF_EMPTY;
            c = 0;
            if (i < slots) {
                k = keys[i];
                c = count_of(i);
            }
            return k != ANSX_RF_EMPTY && (c > cstar || (c == cstar && k <= vstar));
        };
        u32 mine = 0, k, c;
        for (u32 i0 = 0; i0 < slots; i0 += nt) mine += taken(i0 + tid, k, c) ? 1u : 0u;
        const u32 wtotal = wave_sum(mine);
        u32 run = 0;
        if ((tid & 63) == 0) run = atomicAdd(&sh_cnt, wtotal);
        run = (u32)__shfl((int)run, 0);
        for (u32 i0 = 0; i0 < slots; i0 += nt) {
            const bool take = taken(i0 + tid, k, c);
            const u64 m = __builtin_amdgcn_ballot_w64(take);
            const u32 slot = run + (u32)__builtin_popcountll(m & ((1ull << (tid & 63)) - 1ull));
            if (take && slot < T) sel[slot] = ((u64)(0xFFFFFFFFu - c) << 32) | (u64)k;  // (-count, value)
            run += (u32)__builtin_popcountll(m);
        }
    }
    __syncthreads();
    STAMP_RF(5);
    // ---- sort them by (-count, value).  T <= 1024: one element per thread, bitonic network; a stage whose
    // partner distance is below 64 stays inside the wave (two shuffles, no LDS round trip, no barrier) -- 45 of
    // the 55 stages of a 1024-element sort
    if (T <= nt) {
        u64 key = tid < T ? sel[tid] : ~0ull;
        for (u32 k = 2; k <= T; k <<= 1) {
            for (u32 j = k >> 1; j > 0; j >>= 1) {
                u64 other;
                if (j >= 64) {
                    __syncthreads();
                    if (tid < T) sel[tid] = key;
                    __syncthreads();
                    other = tid < T ? sel[tid ^ j] : ~0ull;
                } else {
                    const u32 lo = (u32)__shfl_xor((int)(u32)key, (int)j);
                    const u32 hi = (u32)__shfl_xor((int)(u32)(key >> 32), (int)j);
                    other = ((u64)hi << 32) | lo;
                }
                const bool asc = (tid & k) == 0, lower = (tid & j) == 0;
                const bool keep_min = asc == lower;
                key = keep_min ? (key < other ? key : other) : (key < other ? other : key);
            }
        }
        __syncthreads();
        if (tid < T) sel[tid] = key;
        __syncthreads();
    } else {
        lds_bitonic_sort<u64>(sel, T, tid, nt);
    }
    STAMP_RF(6);
    for (u32 i = tid; i < slots / 2; i += nt) cnt32[i] = 0xFFFFFFFFu;  // rank 0xFFFF = not selected
    __syncthreads();
    u32* mf = mostfreq + (u64)b * T;
    for (u32 r = tid; r < T; r += nt) {
        const u32 v = (u32)sel[r];
        mf[r] = v;  // ans_reorder_fold.hpp:104-105
        u32 slot = rf_slot(v, slots);
        while (keys[slot] != v) slot = slot + 1 == slots ? 0 : slot + 1;
        // two ranks share a word: clear this half (0xFFFF -> r) with an atomic AND
        atomicAnd(&cnt32[slot >> 1], ~(0xFFFFu << (16 * (slot & 1))) | (r << (16 * (slot & 1))));
    }
    __syncthreads();
    STAMP_RF(7);
    // (the values are still in registers)
    auto remap_one = [&](u32 i, u32 v) {
        u32 slot = rf_slot(v, slots);
        // every value was inserted above, so the probe ends at its slot; the bound only matters if
        // the caller's buffer changes under us (a race on the caller's side must not hang the GPU)
        u32 probes = 0;
        while (keys[slot] != v && probes < slots) {
            slot = slot + 1 == slots ? 0 : slot + 1;
            probes++;
        }
        const u32 r = probes < slots ? count_of(slot) : 0xFFFFu;
        dst[i] = (r != 0xFFFFu) ? r : v + T;  // :99-103
    };
#pragma unroll
    for (u32 q = 0; q < RF_VPT; q++)
        if (tid + q * nt < nb) remap_one(tid + q * nt, vals[q]);
    for (u32 i = tid + RF_VPT * nt; i < nb; i += nt) remap_one(i, src[i]);
#ifdef ANSX_STAMPS_RF
    __syncthreads();
#endif
    STAMP_RF(8);
#ifdef ANSX_STAMPS_RF
    if (threadIdx.x == 0) {
        const unsigned long long d_ = wall_clock64() - rf_t0;
        atomicMax(&g_stamps[4101], d_);
        atomicAdd(&g_stamps[4102], d_);
        if (d_ > 4000) atomicAdd(&g_stamps[4103], 1ull);
    }
    if (threadIdx.x == 0 && blockIdx.x % 61 == 7 && blockIdx.x / 61 < 256) g_stamps[(blockIdx.x / 61) * 16 + 9] = __builtin_amdgcn_s_getreg((23 << 0) | (0 << 6) | (31 << 11));
#endif
    if (tid == 0) blk[b].flag = 1;
}


__global__ __launch_bounds__(1024) void k_rfold_remap_hash(const u32* __restrict__ in, ansx_geo g, u32 slots,
    u32* __restrict__ mapped, u32* __restrict__ mostfreq, ansx_blk* __restrict__ blk, u32* __restrict__ gflags)
{
    rfold_remap_hash_body(in, g, slots, mapped, mostfreq, blk, gflags);
}
// two workgroups per CU: 32 waves, 64 registers each
#ifdef RF_NO_WPE
__global__ __launch_bounds__(1024) void k_rfold_remap_hash2(
#else
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_rfold_remap_hash2(
#endif

    const u32* __restrict__ in, ansx_geo g, u32 slots, u32* __restrict__ mapped, u32* __restrict__ mostfreq,
    ansx_blk* __restrict__ blk, u32* __restrict__ gflags)
{
    rfold_remap_hash_body(in, g, slots, mapped, mostfreq, blk, gflags);
}

// ------------------------------------------------------------------------------------------
// K9 (large blocks, any block_ints < 2^31, e.g. whole-list single-stream mode): the same
// selection with the hash table in HBM.  Table per block: `slots` (power of two >= 2 x block
// size) keys + counts.  k_rfg_insert counts values with global atomics, k_rfg_select (one
// workgroup per block) finds c*, v* by radix selection over the table (8-bit levels, LDS
// histograms), sorts the T selected pairs and writes their ranks over the counts, k_rfg_map
// rewrites every value with one probe.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 rfg_hash(u32 v, u32 mask) { return (v * 2654435761u) & mask; }

__global__ __launch_bounds__(256) void k_rfg_insert(const u32* __restrict__ in, ansx_geo g, u32 slots,
    u32* __restrict__ keys, u32* __restrict__ counts, u32* __restrict__ bstat /* [nb][4]: sigma, vmax */)
{
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    if (gid >= g.n) return;
    const u32 b = (u32)(gid / g.block_ints);
    const u32 v = in[gid];
    u32* K = keys + (u64)b * slots;
    u32* C = counts + (u64)b * slots;
    const u32 mask = slots - 1;
    u32 slot = rfg_hash(v, mask);
    for (;;) {
        const u32 old = atomicCAS(&K[slot], ANSX_RF_EMPTY, v);
        if (old == ANSX_RF_EMPTY) atomicAdd(&bstat[4 * b + 0], 1u);
        if (old == ANSX_RF_EMPTY || old == v) break;
        slot = (slot + 1) & mask;
    }
    atomicAdd(&C[slot], 1u);
    if (__hip_atomic_load(&bstat[4 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v)
        atomicMax(&bstat[4 * b + 1], v);
}

__global__ __launch_bounds__(256) void k_rfg_select(ansx_geo g, u32 slots, const u32* __restrict__ keys,
    u32* __restrict__ counts, const u32* __restrict__ bstat, u32* __restrict__ mostfreq,
    ansx_blk* __restrict__ blk, u32* __restrict__ gflags)
{
    extern __shared__ u64 sel_g[];  // [T]
    __shared__ u32 hist[256];
    __shared__ u32 sh_a, sh_b;
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const u32 T = fold_T(g.f);
    const u32* K = keys + (u64)b * slots;
    u32* C = counts + (u64)b * slots;
    const u32 sigma = bstat[4 * b + 0], vmax = bstat[4 * b + 1];
    if (sigma < T) {  // ans_reorder_fold.hpp:94-97
        if (tid == 0) {
            blk[b].flag = 0;
            if (vmax >= (1u << 30)) atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
        }
        return;
    }
    if (tid == 0 && (u64)vmax + T >= (1u << 30)) atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
    // one radix level: histogram of digit(slot) over the slots passing `pred`, then the bucket
    // where the running total (from the top if descending) reaches `target`
    auto level = [&](auto pred, auto digit, u32 target, bool descending, u32* before) -> u32 {
        __syncthreads();
        hist[tid] = 0;
        __syncthreads();
        for (u32 i = tid; i < slots; i += 256) {
            const u32 k = K[i];
            if (k == ANSX_RF_EMPTY) continue;
            const u32 c = C[i];
            if (pred(k, c)) atomicAdd(&hist[digit(k, c)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            u32 run = 0, found = 0, bef = 0;
            bool done = false;
            for (u32 j = 0; j < 256 && !done; j++) {
                const u32 d = descending ? 255 - j : j;
                const u32 h = hist[d];
                if (run + h >= target) {
                    found = d;
                    bef = run;
                    done = true;
                }
                run += h;
            }
            sh_a = found;
            sh_b = bef;
        }
        __syncthreads();
        *before = sh_b;
        return sh_a;
    };
    // ---- c* = largest c with #(count >= c) >= T: four 8-bit levels from the top
    u32 before, prefix = 0, need = T;
    for (int lv = 3; lv >= 0; lv--) {
        const u32 sh = 8 * lv;
        const u32 pfx = prefix;
        const u32 d = level([&](u32, u32 c) { return lv == 3 || (c >> (sh + 8)) == (pfx >> (sh + 8)); },
            [&](u32, u32 c) { return (c >> sh) & 255u; }, need, true, &before);
        prefix |= d << sh;
        need -= before;
    }
    const u32 cstar = prefix;
    const u32 K_take = need;  // values with count == c* still to take (the smallest ones)
    // number of ties
    __syncthreads();
    if (tid == 0) sh_a = 0;
    __syncthreads();
    {
        u32 l = 0;
        for (u32 i = tid; i < slots; i += 256)
            if (K[i] != ANSX_RF_EMPTY && C[i] == cstar) l++;
        atomicAdd(&sh_a, l);
    }
    __syncthreads();
    const u32 ties = sh_a;
    u32 vstar = 0xFFFFFFFFu;
    if (ties > K_take) {  // K_take-th smallest value among the ties
        u32 vp = 0, vneed = K_take;
        for (int lv = 3; lv >= 0; lv--) {
            const u32 sh = 8 * lv;
            const u32 pfx = vp;
            const u32 d = level([&](u32 k, u32 c) { return c == cstar && (lv == 3 || (k >> (sh + 8)) == (pfx >> (sh + 8))); },
                [&](u32 k, u32) { return (k >> sh) & 255u; }, vneed, false, &before);
            vp |= d << sh;
            vneed -= before;
        }
        vstar = vp;
    }
    __syncthreads();
    if (tid == 0) sh_a = 0;
    __syncthreads();
    for (u32 i = tid; i < slots; i += 256) {
        const u32 k = K[i];
        if (k == ANSX_RF_EMPTY) continue;
        const u32 c = C[i];
        if (c > cstar || (c == cstar && k <= vstar)) {
            const u32 s = atomicAdd(&sh_a, 1u);
            if (s < T) sel_g[s] = ((u64)(0xFFFFFFFFu - c) << 32) | (u64)k;
        }
    }
    __syncthreads();
    lds_bitonic_sort<u64>(sel_g, T, tid);
    for (u32 i = tid; i < slots; i += 256) C[i] = 0xFFFFFFFFu;  // rank, 0xFFFFFFFF = not selected
    __threadfence_block();
    __syncthreads();
    u32* mf = mostfreq + (u64)b * T;
    const u32 mask = slots - 1;
    for (u32 r = tid; r < T; r += 256) {
        const u32 v = (u32)sel_g[r];
        mf[r] = v;
        u32 slot = rfg_hash(v, mask);
        while (K[slot] != v) slot = (slot + 1) & mask;
        C[slot] = r;
    }
    if (tid == 0) blk[b].flag = 1;
}

__global__ __launch_bounds__(256) void k_rfg_map(const u32* __restrict__ in, ansx_geo g, u32 slots,
    const u32* __restrict__ keys, const u32* __restrict__ counts, const ansx_blk* __restrict__ blk,
    u32* __restrict__ mapped)
{
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    if (gid >= g.n) return;
    const u32 b = (u32)(gid / g.block_ints);
    const u32 v = in[gid];
    if (!blk[b].flag) {
        mapped[gid] = v;
        return;
    }
    const u32* K = keys + (u64)b * slots;
    const u32 mask = slots - 1;
    u32 slot = rfg_hash(v, mask);
    u32 probes = 0;  // bounded for the same reason as in k_rfold_remap_hash
    while (K[slot] != v && probes < slots) {
        slot = (slot + 1) & mask;
        probes++;
    }
    const u32 r = probes < slots ? counts[(u64)b * slots + slot] : 0xFFFFFFFFu;
    mapped[gid] = (r != 0xFFFFFFFFu) ? r : v + fold_T(g.f);
}

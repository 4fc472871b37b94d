// Per-block alphabet compaction (src/pseudo_adaptive.cpp:85-130): a block is stored as
//
//   u32 sigma | u32 universe | interpolative code of the running sums of its sigma distinct values |
//   codec stream of the block with every value replaced by its 1-based rank among them
//
// which is byte for byte what the reference's pseudo_adaptive harness writes for the block (it only
// measures sizes and never decodes; the decoder below is this build's own).  A block with a single
// distinct value has no codec stream (:115).  The running sums are kept in 32 bits there, so the sum of a
// block's distinct values must stay below 2^32 - 1 (ANSX_ERR_DOMAIN otherwise).
//
//   encode: k_pa_remap   distinct values (LDS hash set), sorted, ranks by binary search; running sums
//           k_pa_header  interpolative code of the running sums (the prelude writer's machinery)
//           ... then the codec's kernels on the remapped block ...
//   decode: k_pa_parse   alphabet header -> values, one lane per block
//           ... the codec's decoder writes ranks ...
//           k_pa_unmap   rank -> value
#pragma once

#include "ansx_kernels.h"

#include <type_traits>

#define ANSX_PA_SLOTS 20480u  // hash set capacity: >= 1.25 x 16384 values per block
#define ANSX_PA_EMPTY 0xFFFFFFFFu
#define ANSX_PA_MAX_BLOCK 16384u

__device__ __forceinline__ u32 pa_slot(u32 v, u32 slots) { return (u32)(((u64)(v * 2654435761u) * slots) >> 32); }

// One workgroup of 1024 threads per block.  LDS: hash set + distinct values: 80 + 64 KB for any block of 16 Ki ints
// (slots = ANSX_PA_SLOTS, uqcap = ANSX_PA_MAX_BLOCK: the first call of a geometry), or sized from the most distinct
// values a block of the geometry had so far (as k_rfold_remap_hash: two workgroups per CU on the 64-register build
// k_pa_remap2; a block that does not fit raises the violation flag, leaves a valid one-value block, and the call is
// repeated with the full sizes).
__device__ __forceinline__ void pa_remap_body(const u32* __restrict__ in, const ansx_geo& g, u32 slots, u32 uqcap,
    u32* __restrict__ mapped, u32* __restrict__ alpha_sum, ansx_blk* __restrict__ blk, u32* __restrict__ gflags,
    u32 value_limit, u32 mode = 0)
{
    // mode 1 (plain ANSint in rank space, ansx_intsparse.h): alpha_sum receives the distinct VALUES (not their running
    // sums: no 32-bit sum to overflow), ranks are 0-based, the count goes to blk[].sp_sigma -- a one-value block keeps its
    // codec stream -- and the call's largest value to gflags[ANSX_G_VMAX].  Always run with the full table sizes.
    extern __shared__ u32 pa_lds[];
    __shared__ u32 sh_cnt, sh_max, sh_ovf;
    __shared__ u64 sh_part64[20];
    u32* keys = pa_lds;                  // [slots]
    u32* uq = pa_lds + slots;            // [uqcap] distinct values (uqcap: a power of two >= 1024)
    const u32 tid = threadIdx.x, nt = 1024;
    const u32 b = blockIdx.x;
    const u32 nb = geo_block_n(g, b);
    const u32* src = in + (u64)b * g.block_ints;
    u32* dst = mapped + (u64)b * g.block_ints;
    for (u32 i = tid; i < slots; i += nt) keys[i] = ANSX_PA_EMPTY;
    if (tid == 0) {
        sh_cnt = 0;
        sh_max = 0;
        sh_ovf = 0;
    }
    __syncthreads();
    // distinct values: first inserter of a value appends it to uq.  A thread's 16 values (blocks hold at most
    // 16 Ki ints) are requested together and stay in registers for the ranking pass below: one at a time, each
    // waited a global round trip that this one workgroup per CU has nothing to hide behind.
    constexpr u32 PA_VPT = ANSX_PA_MAX_BLOCK / 1024;
    u32 vals[PA_VPT];
    u32 lmax = 0;
    // (mode 1 takes blocks of any length -- a whole list in single-stream mode -- 16 Ki ints at a time: what bounds it is the
    // number of DISTINCT values, uqcap)
    const u32 nchunks = mode == 1 ? (nb + ANSX_PA_MAX_BLOCK - 1) / ANSX_PA_MAX_BLOCK : 1u;
    for (u32 ch = 0; ch < nchunks; ch++) {
    const u32 cbase = ch * ANSX_PA_MAX_BLOCK;
#pragma unroll
    for (u32 q = 0; q < PA_VPT; q++) vals[q] = cbase + tid + q * nt < nb ? src[cbase + tid + q * nt] : 0u;
#pragma unroll
    for (u32 q = 0; q < PA_VPT; q++) {
        if (cbase + tid + q * nt >= nb) break;
        const u32 v = vals[q];
        lmax = v > lmax ? v : lmax;
        u32 slot = pa_slot(v, slots);
        u32 probes = 0;
        for (; probes < slots; probes++) {
            const u32 old = atomicCAS(&keys[slot], ANSX_PA_EMPTY, v);
            if (old == ANSX_PA_EMPTY) {
                const u32 at = atomicAdd(&sh_cnt, 1u);
                if (at < uqcap) uq[at] = v;
            }
            if (old == ANSX_PA_EMPTY || old == v) break;
            slot = slot + 1 == slots ? 0 : slot + 1;
        }
        if (probes == slots) sh_ovf = 1;  // table full (only possible below the full size)
    }
    if (nchunks > 1) {  // (a list with too many distinct values: give up after the chunk that showed it, not after the list)
        __syncthreads();
        if (sh_ovf || sh_cnt > uqcap) break;
    }
    }
    atomicMax(&sh_max, lmax);
    __syncthreads();
    if (mode == 1 && (sh_ovf || sh_cnt > uqcap)) {  // more distinct values than the rank-space model has symbols: ANSX_ERR_DOMAIN
        for (u32 i = tid; i < nb; i += nt) dst[i] = 0;
        if (tid == 0) {
            alpha_sum[(u64)b * g.block_ints] = 0;
            blk[b].sp_sigma = 1;
            atomicOr(&gflags[ANSX_G_ERR], 1u << 6);
        }
        return;
    }
    if (sh_ovf || sh_cnt > uqcap) {  // optimistic sizes too small: a valid one-value block, and the call is repeated
        for (u32 i = tid; i < nb; i += nt) dst[i] = 1;
        if (tid == 0) {
            alpha_sum[(u64)b * g.block_ints] = 0;
            blk[b].pa_sigma = 1;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
            atomicMax(&gflags[ANSX_G_RFDIST], sh_cnt > slots ? sh_cnt : slots);
        }
        return;
    }
    const u32 sigma = sh_cnt;
    if (tid == 0 && sh_max >= value_limit) atomicOr(&gflags[ANSX_G_ERR], 1u << 6 /* ANSX_ERR_DOMAIN */);
    if (tid == 0 && sigma > gflags[ANSX_G_RFDIST]) atomicMax(&gflags[ANSX_G_RFDIST], sigma);
    if (mode == 1 && tid == 0 && sh_max > gflags[ANSX_G_VMAX]) atomicMax(&gflags[ANSX_G_VMAX], sh_max);
    // sort the distinct values (bitonic, padded to a power of two).  Every thread keeps E = N2 / 1024 consecutive
    // elements in registers: a stage whose partner distance is below E is a compare-exchange inside the thread, below
    // 64 E a shuffle inside the wave, and only the others (10 of the 78 stages at N2 = 4096) go through LDS and a
    // barrier -- with every stage through LDS the sort was half of this kernel's time.
    u32 N2 = 1024;
    while (N2 < sigma) N2 <<= 1;
    for (u32 i = sigma + tid; i < N2; i += nt) uq[i] = 0xFFFFFFFFu;
    __syncthreads();
    auto sort_regs = [&](auto tag) {
        constexpr u32 E = decltype(tag)::value;
        u32 r[E];
#pragma unroll
        for (u32 e = 0; e < E; e++) r[e] = uq[tid * E + e];
        for (u32 k = 2; k <= N2; k <<= 1) {
            for (u32 j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64 * E) {  // partner in another wave
                    __syncthreads();
#pragma unroll
                    for (u32 e = 0; e < E; e++) uq[tid * E + e] = r[e];
                    __syncthreads();
#pragma unroll
                    for (u32 e = 0; e < E; e++) {
                        const u32 i = tid * E + e;
                        const u32 o = uq[i ^ j];
                        const bool keep_min = ((i & k) == 0) == ((i & j) == 0);
                        r[e] = keep_min ? (r[e] < o ? r[e] : o) : (r[e] < o ? o : r[e]);
                    }
                } else if (j >= E) {  // partner in another lane of this wave, same register
#pragma unroll
                    for (u32 e = 0; e < E; e++) {
                        const u32 i = tid * E + e;
                        const u32 o = (u32)__shfl_xor((int)r[e], (int)(j / E));
                        const bool keep_min = ((i & k) == 0) == ((i & j) == 0);
                        r[e] = keep_min ? (r[e] < o ? r[e] : o) : (r[e] < o ? o : r[e]);
                    }
                } else {  // partner in this thread
#pragma unroll
                    for (u32 e = 0; e < E; e++) {
                        if ((e & j) == 0) {
                            const u32 i = tid * E + e;
                            const u32 x = r[e], y = r[e | j];
                            const bool asc = (i & k) == 0;
                            const bool swap = (x > y) == asc;
                            r[e] = swap ? y : x;
                            r[e | j] = swap ? x : y;
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 e = 0; e < E; e++) uq[tid * E + e] = r[e];
        __syncthreads();
    };
    switch (N2 / 1024) {
    case 1: sort_regs(std::integral_constant<u32, 1>{}); break;
    case 2: sort_regs(std::integral_constant<u32, 2>{}); break;
    case 4: sort_regs(std::integral_constant<u32, 4>{}); break;
    case 8: sort_regs(std::integral_constant<u32, 8>{}); break;
    default: sort_regs(std::integral_constant<u32, 16>{}); break;
    }
    if (mode == 1) {
        u32* as = alpha_sum + (u64)b * g.block_ints;
        for (u32 j = tid; j < sigma; j += nt) as[j] = uq[j];
        if (tid == 0) blk[b].sp_sigma = sigma;
    } else
    // running sums of the alphabet (pseudo_adaptive.cpp:103-105, u32 there: the exact sum must fit)
    {
        const u32 per = (sigma + nt - 1) / nt;
        const u32 lo = tid * per, hi = (lo + per) < sigma ? (lo + per) : sigma;
        u64 sum = 0;
        for (u32 j = lo; j < hi; j++) sum += uq[j];
        u64 total;
        u64 run = block_excl_scan<u64>(sum, sh_part64, tid, nt, &total);
        u32* as = alpha_sum + (u64)b * g.block_ints;
        for (u32 j = lo; j < hi; j++) {
            run += uq[j];
            as[j] = (u32)run;
        }
        if (tid == 0) {
            if (total >= 0xFFFFFFFFull) atomicOr(&gflags[ANSX_G_ERR], 1u << 6 /* ANSX_ERR_DOMAIN */);
            blk[b].pa_sigma = sigma;
        }
    }
    // 1-based rank of every value (:91-103): branch-free lower bound over the padded, sorted array (the padding
    // compares above every value), the thread's 16 searches advancing together -- 16 independent LDS reads per step
    __syncthreads();
    for (u32 ch = 0; ch < nchunks; ch++) {
        const u32 cbase = ch * ANSX_PA_MAX_BLOCK;
        if (nchunks > 1) {  // (a single chunk's values are still in registers)
#pragma unroll
            for (u32 q = 0; q < PA_VPT; q++) vals[q] = cbase + tid + q * nt < nb ? src[cbase + tid + q * nt] : 0u;
        }
        u32 lo[PA_VPT];
#pragma unroll
        for (u32 q = 0; q < PA_VPT; q++) lo[q] = 0;
        for (u32 st = N2 >> 1; st > 0; st >>= 1) {
            u32 probe[PA_VPT];
#pragma unroll
            for (u32 q = 0; q < PA_VPT; q++) probe[q] = uq[lo[q] + st - 1];
#pragma unroll
            for (u32 q = 0; q < PA_VPT; q++) lo[q] += probe[q] < vals[q] ? st : 0u;
        }
#pragma unroll
        for (u32 q = 0; q < PA_VPT; q++)
            if (cbase + tid + q * nt < nb) dst[cbase + tid + q * nt] = lo[q] + (mode == 1 ? 0u : 1u);
    }
}

__global__ __launch_bounds__(1024) void k_pa_remap(const u32* __restrict__ in, ansx_geo g, u32 slots, u32 uqcap,
    u32* __restrict__ mapped, u32* __restrict__ alpha_sum, ansx_blk* __restrict__ blk, u32* __restrict__ gflags, u32 value_limit,
    u32 mode)
{
    pa_remap_body(in, g, slots, uqcap, mapped, alpha_sum, blk, gflags, value_limit, mode);
}
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_pa_remap2(const u32* __restrict__ in,
    ansx_geo g, u32 slots, u32 uqcap, u32* __restrict__ mapped, u32* __restrict__ alpha_sum, ansx_blk* __restrict__ blk,
    u32* __restrict__ gflags, u32 value_limit)
{
    pa_remap_body(in, g, slots, uqcap, mapped, alpha_sum, blk, gflags, value_limit);
}

// Alphabet header: one workgroup of 256 threads per block (the prelude writer's generic path).  `cap` = words per
// LDS array: ANSX_PA_MAX_BLOCK (offsets + bit buffer = 128 KB: one workgroup per CU, the running sums read from
// HBM), or the value-list capacity k_pa_remap ran with on an optimistic call (a few thousand: three arrays --
// the running sums are staged too -- and three workgroups per CU; a block k_pa_remap gave up on has sigma = 1).
__global__ __launch_bounds__(256) void k_pa_header(ansx_geo g, const u32* __restrict__ alpha_sum,
    ansx_blk* __restrict__ blk, u8* __restrict__ scratch, u64 scr_stride, u32 cap)
{
    extern __shared__ u32 lds32[];
    __shared__ u32 sh_part[8];
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    ansx_blk* B = &blk[b];
    const u32 sigma = B->pa_sigma;
    const u32* as = alpha_sum + (u64)b * g.block_ints;
    u32* off = lds32;           // [sigma]
    u32* bits = lds32 + cap;    // bit buffer (the code of sigma ascending values below 2^32 takes at most sigma words)
    const u32* inc = as;
    if (cap < ANSX_PA_MAX_BLOCK) {
        u32* inc_lds = lds32 + 2 * cap + 16;
        for (u32 i = tid; i < sigma; i += 256) inc_lds[i] = as[i];
        __syncthreads();
        inc = inc_lds;
    }
    const u64 uni = (u64)inc[sigma - 1] + 1;    // block_alphabet.back() + 1 (:108)
    if (cap <= 4096)  // at most 16 items per thread: the writer's register-resident form
        prelude_emit<16, true>(g, B, sigma, 0, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, nullptr, b, tid, uni);
    else
        prelude_emit<0, true>(g, B, sigma, 0, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, nullptr, b, tid, uni);
}

// Decoder side: alphabet header -> distinct values, one lane per block (serial bit parsing; values and the
// universe need up to 33 bits, so the interval arithmetic is 64-bit).  Writes the values (not the running
// sums) to alpha[b][0..sigma), and pa_info[b] = {sigma, header bytes, error}.
__global__ __launch_bounds__(64) void k_pa_parse(const u8* __restrict__ cont, ansx_geo g,
    const u64* __restrict__ block_off, u64 payload_off, u32* __restrict__ alpha, uint4* __restrict__ pa_info,
    u32* __restrict__ gflags)
{
    __shared__ uint4 stkA[24][64];  // pending right subtrees: (a, n) and the bounds as two 64-bit halves
    __shared__ uint2 stkB[24][64];
    constexpr u32 PA_WIN = 64;      // window words per lane
    __shared__ u32 win[PA_WIN][64];
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x * 64 + lane;
    if (b >= g.nblocks) return;
    const u64 boff = block_off[b];
    const u8* stream = cont + payload_off + boff;
    const u32 sbytes = (u32)(block_off[b + 1] - boff);
    const u32 nb = geo_block_n(g, b);
    u32 err = 0, sigma = 0, hb = 0;
    u32* out = alpha + (u64)b * g.block_ints;
    if (sbytes < 8) err = 1;
    if (!err) {
        sigma = ld_u32_unaligned(stream);
        const u64 uni = ld_u32_unaligned(stream + 4);
        if (sigma == 0 || sigma > nb || sigma > ANSX_PA_MAX_BLOCK) err = 1;
        const u8* bp = stream + 8;
        const u64 maxbits = (u64)(sbytes - 8) * 8;
        u64 bitpos = 0;
        // The code is read strictly front to back: every lane keeps the next 256 bytes of its header in its own
        // column of an LDS window; when any lane of the wave is about to run out, ALL lanes refill theirs from
        // where they stand, with 64 independent loads each (one global round trip per ~100 items for the whole
        // wave; an 8-byte global load per item, each waiting on the position the previous one produced, made this
        // kernel 3.3 ms, and lanes refilling one by one 6.3).  Bytes past the block's own are never read and
        // count as zero.
        const u32 nbytes = sbytes - 8;
        u64 wbit0 = 0;
        auto refill = [&]() {
            wbit0 = bitpos & ~31ull;
            const u32 byte0 = (u32)(wbit0 >> 3);
            if (byte0 + 4 * PA_WIN <= nbytes) {
#pragma unroll
                for (u32 k = 0; k < PA_WIN; k++) win[k][lane] = ld_u32_unaligned(bp + byte0 + 4 * k);
            } else {
                for (u32 k = 0; k < PA_WIN; k++) {
                    const u32 off = byte0 + 4 * k;
                    u32 v = 0;
                    for (u32 i = 0; i < 4 && off + i < nbytes; i++) v |= (u32)bp[off + i] << (8 * i);
                    win[k][lane] = v;
                }
            }
            wave_lds_sync();
        };
        refill();
        auto getbits = [&](u32 nbits) -> u32 {  // nbits in [0, 32]; the caller keeps 96 bits of window ahead
            if (nbits == 0) return 0u;
            const u32 rel = (u32)(bitpos - wbit0);
            const u32 word = rel >> 5;
            const u64 w = (u64)win[word][lane] | ((u64)win[word + 1][lane] << 32);
            const u32 v = (u32)((w >> (rel & 31u)) & (nbits >= 32 ? 0xFFFFFFFFull : ((1ull << nbits) - 1ull)));
            bitpos += nbits;
            return v;
        };
        u32 sp = 0, a = 0, n = sigma;
        u64 low = 1, high = uni + 1;
        u64 prev_sum = 0;  // values come out in index order only within a subtree: differences are taken afterwards
        (void)prev_sum;
        for (u32 it = 0; it < sigma && !err; it++) {
            // (an item reads at most 33 bits; 96 keeps word + 1 inside the window)
            if (__any((u32)(bitpos - wbit0) + 96 > PA_WIN * 32)) refill();
            if (n == 0) {
                if (sp == 0) {
                    err = 1;
                    break;
                }
                sp--;
                const uint4 e = stkA[sp][lane];
                const uint2 e2 = stkB[sp][lane];
                a = e.x;
                n = e.y;
                low = (u64)e.z | ((u64)e.w << 32);
                high = (u64)e2.x | ((u64)e2.y << 32);
            }
            const u32 h = (n + 1) >> 1;
            const u32 n1 = h - 1, n2 = n - h;
            const u64 U = high - n2 - low - n1 + 1;
            if (U == 0 || U > uni + 1 || bitpos > maxbits) {
                err = 1;
                break;
            }
            u64 val = 1;  // read_center_mid (interp.hpp:47-63)
            if (U != 1) {
                const u32 bb = 64 - __clzll((unsigned long long)(U - 1));  // hi(U-1)+1, <= 33
                const u64 m = (1ull << bb) - U;
                const u64 dh = U - (1ull << (bb - 1));
                val = (u64)getbits(bb - 1) + 1;
                if (val > m) val = (2 * val + getbits(1)) - m - 1;
                val += dh;
                if (val > U) val -= U;
            }
            const u64 v = low + n1 - 1 + val;
            out[a + h - 1] = (u32)(v - 1);  // running sum of the alphabet up to this value
            if (n2) {
                stkA[sp][lane] = make_uint4(a + h, n2, (u32)(v + 1), (u32)((v + 1) >> 32));
                stkB[sp][lane] = make_uint2((u32)high, (u32)(high >> 32));
                sp++;
            }
            n = n1;
            high = v - 1;
        }
        if (bitpos > maxbits) err = 1;
        hb = 8 + 4 * (u32)((bitpos + 31) >> 5);
        if (hb > sbytes) err = 1;
        if (!err) {  // running sums -> values (this lane wrote every entry itself)
            u32 prev = 0;
            for (u32 j0 = 0; j0 < sigma; j0 += 16) {  // 16 independent loads per round trip
                u32 cur[16];
#pragma unroll
                for (u32 u = 0; u < 16; u++) cur[u] = j0 + u < sigma ? out[j0 + u] : 0u;
#pragma unroll
                for (u32 u = 0; u < 16; u++) {
                    if (j0 + u < sigma) {
                        if ((j0 + u) && cur[u] <= prev) err = 1;  // distinct ascending values: sums strictly increase (value 0 only first)
                        out[j0 + u] = cur[u] - prev;
                        prev = cur[u];
                    }
                }
            }
        }
    }
    pa_info[b] = make_uint4(sigma, hb, err, 0);
    if (err) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
}

// rank -> value on the decoded block; a one-value block is filled here (it has no codec stream)
__global__ __launch_bounds__(256) void k_pa_unmap(ansx_geo g, const u32* __restrict__ alpha,
    const uint4* __restrict__ pa_info, u32* __restrict__ out, u32* __restrict__ gflags)
{
    const u32 b = blockIdx.x;
    const uint4 pi = pa_info[b];
    if (pi.z) return;
    const u32 nb = geo_block_n(g, b);
    const u32* al = alpha + (u64)b * g.block_ints;
    u32* o = out + (u64)b * g.block_ints;
    const u32 sigma = pi.x;
    u32 bad = 0;
    if (sigma == 1) {
        const u32 v = al[0];
        for (u32 i = threadIdx.x; i < nb; i += 256) o[i] = v;
        return;
    }
    // 16 ranks per thread and round are requested together, then their 16 values: two round trips per round
    // instead of two per int (the kernel was a chain of 128 dependent round trips per block)
    u32 done = 0;
    if ((((uintptr_t)o) & 15u) == 0) {
        uint4* o4 = (uint4*)o;
        const u32 nvec = nb >> 2;
        for (u32 v0 = 0; v0 < nvec; v0 += 4 * 256) {
            uint4 r4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 v = v0 + j * 256 + threadIdx.x;
                r4[j] = v < nvec ? o4[v] : make_uint4(1u, 1u, 1u, 1u);
            }
            u32 r[16], x[16];
#pragma unroll
            for (int j = 0; j < 4; j++) r[4 * j] = r4[j].x, r[4 * j + 1] = r4[j].y, r[4 * j + 2] = r4[j].z, r[4 * j + 3] = r4[j].w;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const bool ok = r[j] >= 1 && r[j] <= sigma;
                bad |= ok ? 0u : 1u;
                x[j] = al[ok ? r[j] - 1 : 0u];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 v = v0 + j * 256 + threadIdx.x;
                // (a rank out of range: the int is left as decoded, as before, and the call fails with ERR_FORMAT)
                const bool a0 = r[4 * j] >= 1 && r[4 * j] <= sigma, a1 = r[4 * j + 1] >= 1 && r[4 * j + 1] <= sigma;
                const bool a2 = r[4 * j + 2] >= 1 && r[4 * j + 2] <= sigma, a3 = r[4 * j + 3] >= 1 && r[4 * j + 3] <= sigma;
                if (v < nvec)
                    o4[v] = make_uint4(a0 ? x[4 * j] : r[4 * j], a1 ? x[4 * j + 1] : r[4 * j + 1], a2 ? x[4 * j + 2] : r[4 * j + 2],
                        a3 ? x[4 * j + 3] : r[4 * j + 3]);
            }
        }
        done = nvec << 2;
    }
    for (u32 i = done + threadIdx.x; i < nb; i += 256) {
        const u32 r = o[i];
        if (r < 1 || r > sigma) bad = 1;
        else o[i] = al[r - 1];
    }
    if (bad) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
}

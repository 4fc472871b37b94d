// ansx device kernels (gfx950).  One block = one independent reference encode() call
// (SURVEY F1); kernels are organised so that lanes = ANS states:
//
//   encode:  K1 k_fold_hist        per-block folded-symbol histogram (4 LDS copies), entropy terms + H   ans_fold.hpp:70-78, util.hpp:271-282
//            K2 k_sort_entropy     stable counting sort of (freq, sym)                                    ans_util.hpp:114-124
//               k_scale_attempts   one lane per (block, frame size): scale_freqs + cross entropy          ans_util.hpp:77-95, util.hpp:284-298
//               k_select_model     stop rule + compact encoder table                                      ans_util.hpp:127-153, ans_fold.hpp:82-91
//            K3 k_write_prelude    vbyte + log2 M + parallel interpolative coder                          ans_util.hpp:46-63, interp.hpp:28-79
//            K5 k_encode<MODE, POW2>  quad of lanes per block, 4 interleaved states, branch-free step;     ans_fold.hpp:100-120,249-278
//                                  MODE 1: main loop written out operation by operation (4 strands)
//            K6 k_scan_sizes / k_compact / k_write_header   container assembly
//   decode:  K7 k_parse_prelude_par (8 lanes per block on the container's parse hints) / _win / _fast / generic   ans_util.hpp:25-42, interp.hpp:47-63,81-118
//            K8 k_decode_rank<RFOLD, RING> (k_decode: slot->symbol fallback)   quad per restart segment   ans_fold.hpp:179-228,283-311
//   rfold:   ansx_rfold.h (value remap in front of K1, ans_reorder_fold.hpp:70-106)
// DESIGN.md section 5 describes each kernel and the hardware facts they are built around.
#pragma once

#include <type_traits>

#include "ansx_dev.h"

#ifdef ANSX_STAMPS  // development: per-phase wall-clock stamps of a few workgroups (printed by ansx_last_encode_stats)
__device__ unsigned long long g_stamps[16 * 256 + 16];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x % 61 == 7 && blockIdx.x / 61 < 256) g_stamps[(blockIdx.x / 61) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifdef ANSX_STAMPS_RF
#define STAMP_RF(i) STAMP(i)
#else
#define STAMP_RF(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// per-block metadata (device)
// ------------------------------------------------------------------------------------------
struct ansx_blk {
    u32 n;              // ints in this block (sum of histogram)
    u32 max_sym;        // largest folded symbol
    u32 sigma;          // number of distinct folded symbols
    u32 m0_log2;        // log2 of the first frame size tried
    double H;           // entropy (bits/symbol)
    double thr;         // H * 1.001
    u32 resolved;       // model chosen
    int prev;           // last successful-but-rejected attempt (global attempt index), -1 = none
    u32 logM;           // log2 of the chosen frame size
    u32 status;         // 0 ok, else ansx_status
    u32 prelude_bytes;  // header + prelude = offset of the first payload byte
    u32 stream_bytes;   // total bytes of this block's reference stream
    u32 hdr_bytes;      // rfold header bytes (4 or 4+4T), 0 for fold
    u32 flag;           // rfold reorder flag
    // per-block alphabet compaction (ansx_pa.h, src/pseudo_adaptive.cpp:85-130)
    u32 pre_bytes;      // bytes of the alphabet header in front of the codec stream (0 without compaction)
    u32 pa_sigma;       // distinct values of the block (0 without compaction; 1: the block has no codec stream)
    u32 sp_sigma;       // plain ANSint modelled in rank space (ansx_intsparse.h): distinct values of the block, else 0
    u32 pad_[1];
};

// encoder table entry (ans_fold.hpp:30-34 enc_entry_fold, plus the reciprocal used for the
// exact state/freq division)
struct __attribute__((aligned(16))) ansx_enc_entry {
    u32 base;
    u32 freq;
    double rcp;
};

typedef u32 ansx_u32x4 __attribute__((ext_vector_type(4)));

#define ANSX_G_VIOL_BIT 8u  // gflags[ANSX_G_ERR]: the optimistic (hint-sized) path does not apply to this input
enum { ANSX_G_MAXLOGM = 0, ANSX_G_MAXNSYMS = 1, ANSX_G_ERR = 2, ANSX_G_PAD = 3,
    // (words 4, 5 hold the 64-bit payload size)
    ANSX_G_NEAR = 6,    // stop-rule comparisons XH < 1.001 H closer than 1e-12 relative (see ansx_near_threshold)
    ANSX_G_RFDIST = 7,  // rfold: the most distinct values any block of the call had (sizes the next call's hash tables)
    ANSX_G_MAXT = 8,    // largest chosen candidate index t (frame = M0 * 2^t) of the call: lanes per block of k_candidates
    ANSX_G_MAXSIGMA = 9,
    ANSX_G_VMAX = 10 };  // plain ANSint in rank space: the call's largest value (decides whether the dense model applies)  // most symbols PRESENT in any block (<= its alphabet size): goes into the container header and
                            // sizes the decoder's per-present-symbol table

// The one step of the path whose parity with the reference is empirical rather than by construction:
// log2 is libm's there and ansx_log2_portable here (<= 1 ulp apart), so the decision XH < H * 1.001
// (ans_util.hpp:149) could differ when the two sides agree to ~15 digits.  Every such comparison is
// counted (expected: none, ever -- the sums differ by ~1e-3 relative or more in practice) and reported
// through ansx_last_encode_stats; the blocks concerned are listed for the host, which decides them again with
// libm's log2 (resolve_near in ansx.hip) and, should it ever disagree, repeats the call with its decisions forced.
#define ANSX_NEAR_BAND 1e-12
#define ANSX_NEAR_CAP 1024u  // blocks with a close call the device lists for the host (more: the host looks at every block)
__device__ __forceinline__ bool ansx_near_threshold(double XH, double thr, double band = ANSX_NEAR_BAND)
{
    // (a single-symbol block has H = XH = 0 exactly on both sides -- log2(1) -- and is not a close call)
    const double d = XH - thr;
    return thr > 0.0 && (d < 0 ? -d : d) <= band * thr;
}
enum { ANSX_ATTEMPTS = 8 };  // frame sizes tried per batch

// K0: everything an encode call needs zeroed, in one launch instead of four memsets (each a few microseconds of an
// otherwise idle stream): the call's flag words, the per-64-block size sums, the per-block metadata and the
// container's header / index / restart-point area (unused slots and padding are defined to be zero).
// Regions are 16-byte aligned; lengths in 16-byte units.
struct ansx_zero4 {
    uint4* p[4];
    u64 n16[4];
};
__global__ __launch_bounds__(256) void k_begin_encode(ansx_zero4 Z)
{
    const u64 stride = (u64)gridDim.x * 256;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int r = 0; r < 4; r++)
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < Z.n16[r]; i += stride) Z.p[r][i] = z;
}

// ------------------------------------------------------------------------------------------
// K1: folded-symbol histogram.  One workgroup per chunk of a block; LDS bins; coalesced 16 B
// loads.  Replaces the first pass of ans_fold_encode<f>::create (ans_fold.hpp:74-78).
// ------------------------------------------------------------------------------------------
#define ANSX_HCOPY_PAD 8u
// 16-byte load of data that is read once per pass (global_load_dwordx4 ... nt): measured on MI355X with
// tests/tools/ubench_read.hip, a 1 GiB read-only stream reaches 6.9 TB/s this way against 6.0-6.5 TB/s plain.
typedef u32 ansx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_stream(const uint4* p)
{
    const ansx_u32x4 v = __builtin_nontemporal_load((const ansx_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// PACKED (alphabets above 16 Ki slots, f = 6, 7): one copy of 16-bit counters, two symbols per LDS word -- a chunk
// holds at most 16384 values, so a count cannot carry into its neighbour -- NSP * 2 bytes of LDS instead of NSP * 4.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_fold_hist(const u32* __restrict__ in, ansx_geo g,
    u32 chunk, u32 cpb, u32 NSP, u32* __restrict__ hist, double* __restrict__ hterm, u32 sum_mode,
    ansx_blk* __restrict__ blk, u32* __restrict__ gflags, u32 value_limit)
{
    // sum_mode bit 0: four histogram copies, entropy terms kept in LDS (alphabets <= 2048 slots);
    //          bit 1: the fast model path -- H is the workgroup's tree sum of the terms instead of the reference's
    //                 left-to-right sum (within ~1e-13 of it; the stop rule of that path keeps a 1e-9 guard band
    //                 around its threshold and sends anything closer to the exact path, see k_model_finish)
    const u32 sum_here = sum_mode & 1u;
    const bool tree_sum = (sum_mode & 2u) != 0;
    extern __shared__ u32 lds_hist[];
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x / cpb, c = blockIdx.x % cpb;
    const u32 nb = geo_block_n(g, b);
    const u64 start = (u64)c * chunk;
    if (start >= nb) return;
    const u32 len = (u32)((nb - start) < chunk ? (nb - start) : chunk);
    const u32* src = in + (u64)b * g.block_ints + start;
    // sum_here (alphabets <= 2048 slots): 4 histogram copies, lane l counts into copy l & 3.  With
    // skewed data a dozen lanes of every ds_add hit the hottest bin and the LDS unit serialises
    // them; copies are ANSX_HCOPY_PAD words apart modulo the 32 banks so the same symbol's four
    // counters sit in different banks.  Layout: [4][NSP + pad] u32 | [NSP + 8] f64 terms | max word.
    const u32 cstride = sum_here ? NSP + ANSX_HCOPY_PAD : 0;
    const u32 hwords = sum_here ? 4 * cstride : (PACKED ? NSP / 2 : NSP);
    auto hcount = [&](u32 sy) -> u32 {  // symbol sy's count in (copy 0 of) the LDS histogram
        if constexpr (PACKED) return (lds_hist[sy >> 1] >> (16u * (sy & 1u))) & 0xFFFFu;
        else return lds_hist[sy];
    };
    for (u32 s = tid; s < hwords; s += 256) lds_hist[s] = 0;
    u32* const aux = lds_hist + hwords;  // terms (f64, 8-byte aligned: hwords is even), then the max word
    if (sum_here && !tree_sum && tid < 17) aux[2 * NSP + tid] = 0;  // 8 pad terms + the max word (see below)
    u32* const my_hist = lds_hist + (tid & 3) * cstride;
    __syncthreads();
    const ansx_map mp = g.map;
    const bool ident = g.kind == 3;  // ANSint (wave-uniform)
    u32 lmax = 0, bad = 0;
    auto take = [&](u32 x) {
        bad |= (x >= value_limit) ? 1u : 0u;
        u32 k = map_nbytes(mp, x);
        u32 s = map_sym(mp, x, k);
        // (fold / msb maps keep every 32-bit value inside the symbol array; ANSint's identity map does not: a value
        // at or above value_limit = NSP is a domain error, counted as symbol 0 so that nothing is written out of range)
        if (ident) s = s < NSP ? s : 0u;
        if constexpr (PACKED) atomicAdd(&my_hist[s >> 1], 1u << (16u * (s & 1u)));
        else atomicAdd(&my_hist[s], 1u);
        lmax = s > lmax ? s : lmax;
    };
    u32 done = 0;
    if ((((uintptr_t)src) & 15u) == 0) {
        const uint4* v4 = (const uint4*)src;
        const u32 nvec = len >> 2;
        // 16 (a whole 16 Ki-int chunk: one round trip) independent 16-byte loads in flight per thread before any of them is consumed: the
        // kernel is a latency chain otherwise (measured: 4 % VALU activity, 64 % of the wave's
        // cycles in s_waitcnt with one load per iteration)
        u32 v = tid;
        for (; v + 15 * 256 < nvec; v += 16 * 256) {
            uint4 q[16];
#pragma unroll
            for (int j = 0; j < 16; j++) q[j] = ld16_stream(v4 + v + j * 256);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                take(q[j].x);
                take(q[j].y);
                take(q[j].z);
                take(q[j].w);
            }
        }
        for (; v + 7 * 256 < nvec; v += 8 * 256) {  // (smaller chunks)
            uint4 q[8];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = ld16_stream(v4 + v + j * 256);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                take(q[j].x);
                take(q[j].y);
                take(q[j].z);
                take(q[j].w);
            }
        }
        for (; v < nvec; v += 256) {
            uint4 q = ld16_stream(v4 + v);
            take(q.x);
            take(q.y);
            take(q.z);
            take(q.w);
        }
        done = nvec << 2;
    }
    for (u32 i = done + tid; i < len; i += 256) take(src[i]);
    __syncthreads();
    u32* h = hist + (u64)b * NSP;
    if (cpb == 1) {
        // the block's histogram is complete here: also evaluate the entropy terms p*log2(p)
        // (util.hpp:276-279), 256 at a time; an absent symbol contributes +0.0, which leaves the
        // in-order sum (never -0.0) unchanged.  Small alphabets (sum_here): the terms go to LDS
        // and thread 0 adds them left to right, exactly like the scalar loop of util.hpp:271-282
        // (~5 cycles per dependent add; the other workgroups of the CU keep streaming).  Large
        // ones: the terms go to HBM and k_scale_attempts sums them.
        const double nd = (double)nb;
        double* lds_term = (double*)aux;
        double* ht = (sum_here || tree_sum) ? nullptr : hterm + (u64)b * NSP;
        double part = 0.0;
        auto term = [&](u32 fr) -> double {
            if (!fr) return 0.0;
            const double p = ansx_div_int31((double)fr, nd);
            return p * ansx_log2_portable(p);
        };
        for (u32 s = tid; s < NSP; s += 256) {
            u32 fr = hcount(s);
            if (sum_here) fr += lds_hist[cstride + s] + lds_hist[2 * cstride + s] + lds_hist[3 * cstride + s];
            h[s] = fr;
            const double t = term(fr);
            if (tree_sum) part = part + t;
            else if (sum_here) lds_term[s] = t;
            else if (ht != nullptr) ht[s] = t;  // (no array: a launch-site error must not become a write to address 0)
        }
        if (tree_sum) {
            part = wave_sum(part);
            __syncthreads();  // every histogram word has been read: the first 4 doubles of the LDS are free
            double* wpart = (double*)lds_hist;
            if ((tid & 63) == 0) wpart[tid >> 6] = part;
            __syncthreads();
            if (tid == 0) {
                const double H = -((wpart[0] + wpart[1]) + (wpart[2] + wpart[3]));
                blk[b].H = H;
                blk[b].thr = H * (1.0 + (double)1 / (double)1000);  // ans_util.hpp:124
            }
        } else if (sum_here) {
            // symbols above the block's largest one are absent (+0.0 terms): stop there
            const u32 wmax = wave_max(lmax);
            if ((tid & 63) == 0) atomicMax(&aux[2 * NSP + 16], wmax);  // word after the padded terms
            __syncthreads();
            if (tid == 0) {
                const u32 ns8 = (aux[2 * NSP + 16] + 8u) & ~7u;  // <= NSP (a multiple of 8)
                double acc = 0.0;
                double t8[8], n8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = lds_term[u];
                for (u32 i = 0; i < ns8; i += 8) {
                    // next 8 terms in flight while these 8 are added (rows are NSP + 8 terms long)
#pragma unroll
                    for (int u = 0; u < 8; u++) n8[u] = lds_term[i + 8 + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) acc = acc + t8[u];
#pragma unroll
                    for (int u = 0; u < 8; u++) t8[u] = n8[u];
                }
                const double H = -acc;
                blk[b].H = H;
                blk[b].thr = H * (1.0 + (double)1 / (double)1000);  // ans_util.hpp:124
            }
        }
    } else {
        for (u32 s = tid; s < NSP; s += 256) {
            u32 v = hcount(s);
            if (v) atomicAdd(&h[s], v);
        }
    }
    lmax = wave_max(lmax);
    bad = wave_or(bad);
    if ((tid & 63) == 0) {
        atomicMax(&blk[b].max_sym, lmax);
        if (bad) atomicOr(&gflags[ANSX_G_ERR], 1u << 6 /* ANSX_ERR_DOMAIN */);
    }
}

// ------------------------------------------------------------------------------------------
// K2a: order the non-zero (freq, sym) pairs ascending (ans_util.hpp:114-122) and evaluate the
// entropy H (util.hpp:271-282).  One wave per block.
//
// The order is a STABLE sort by frequency of symbols that are already in index order, and the
// frequencies of a block are mostly tiny, so it is done as a counting sort: frequencies below
// ANSX_VMAX are binned in LDS; within a 64-symbol pass the lanes that share a frequency are found
// with ballots (rank = popcount of the matching lanes below), and the running per-value cursor
// carries the order across passes.  The few symbols with freq >= ANSX_VMAX ("big", at most
// n/ANSX_VMAX of them) are ranked among themselves and appended.
// Entropy: p*log2(p) terms lane-parallel, summed serially in index order so that the rounding
// matches a scalar left-to-right accumulation.
// ------------------------------------------------------------------------------------------
#define ANSX_VMAX 256u
#define ANSX_MASKV 256u

// Single-wave workgroups: LDS operations of one wave execute in order, so a "barrier" only has
// to stop the compiler from reordering and wait for the LDS queue; __syncthreads() would also
// drain every outstanding global store (vmcnt(0)) at each call.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// cap = entries of the staged histogram row: NSP, or the alphabet hint on an optimistic call (this one-wave-per-block
// kernel is bound by how many blocks a CU's LDS holds: 2300-symbol alphabets, 28 -> 21 KB per block).  A block
// above the hint is left as the call's memset made it -- no model, no stream -- and the call is repeated.
// pairs != nullptr (the fast model path, blocks of at most 65535 ints): instead of sortF / sortSym the kernel writes
// one uint2 per rank: { freq | sym << 16, fs_rem } with fs_rem = n - (sum of the frequencies ranked before it),
// the divisor of that symbol's scale_freqs step (ans_util.hpp:83), so that k_candidates can prepare its reciprocal.
// HT: type of the staged row -- u16 for blocks of at most 65535 values (a count cannot exceed the block length): half
// the dynamic LDS, i.e. more blocks per CU for this occupancy-bound kernel.
// STAGED = false (alphabets whose row does not fit the LDS, f = 6, 7): the three passes read the row from HBM instead.
template <typename HT, bool STAGED = true>
__global__ __launch_bounds__(64) void k_sort_entropy(ansx_geo g, u32 NSP, u32 nbig_cap, u32 h_deferred,
    const u32* __restrict__ hist, u32* __restrict__ sortF, u16* __restrict__ sortSym,
    ansx_blk* __restrict__ blk, u32 cap, uint2* __restrict__ pairs, u32* __restrict__ gflags)
{
    extern __shared__ u64 lds_k2a[];  // [nbig_cap] big keys (freq << 16 | sym), then the staged row
    __shared__ u32 cnt[ANSX_VMAX];
    __shared__ u16 cnt0[ANSX_VMAX];  // number of symbols per frequency value (before the scan)
    __shared__ unsigned long long vmask[ANSX_MASKV];  // lanes of the current pass per frequency value
    __shared__ u32 wadj[ANSX_VMAX];  // packed output: (frequency mass ranked before the value's bin) - (bin start) * value
    __shared__ u32 sh_nbig;
    u64* big_keys = lds_k2a;
    HT* hrow = (HT*)(lds_k2a + nbig_cap);  // [cap] this block's histogram row
    double* terms = (double*)(hrow + (STAGED ? cap : 0u));   // [512], only allocated when the entropy is summed here
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    const u32* h = hist + (u64)b * NSP;
    // stage the histogram row once: its first 640 entries are requested together with the alphabet size they are
    // then cut to (a row has NSP entries whatever the block's alphabet) -- one round trip instead of two or more
    constexpr u32 SORT_PRE = 10;
    u32 hp[SORT_PRE];
#pragma unroll
    for (u32 r = 0; r < SORT_PRE; r++) hp[r] = r * 64 + lane < NSP ? h[r * 64 + lane] : 0u;
    const u32 ns = blk[b].max_sym + 1;
    auto hget = [&](u32 sy) -> u32 {
        if constexpr (STAGED) return hrow[sy];
        else return h[sy];
    };
    if (STAGED && ns > cap) {
        // the block outgrew the alphabet hint this (optimistic) call is sized for: no model, no stream -- said
        // explicitly, so that the repeat does not hang on what the later kernels make of an untouched block
        if (lane == 0) {
            blk[b].status = 9;
            blk[b].resolved = 1;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
        }
        return;
    }
    u32* oF = sortF + (u64)b * NSP;
    u16* oS = sortSym + (u64)b * NSP;
    uint2* oP = pairs ? pairs + (u64)b * NSP : nullptr;
    if constexpr (STAGED) {
#pragma unroll
        for (u32 r = 0; r < SORT_PRE; r++)
            if (r * 64 + lane < ns) hrow[r * 64 + lane] = (HT)hp[r];
        for (u32 s = SORT_PRE * 64 + lane; s < ns; s += 64) hrow[s] = (HT)h[s];
    }
    for (u32 v = lane; v < ANSX_VMAX; v += 64) cnt[v] = 0;
    for (u32 v = lane; v < ANSX_MASKV; v += 64) vmask[v] = 0;
    if (lane == 0) sh_nbig = 0;
    wave_lds_sync();
    // pass 1: bin the frequencies
    u32 sigma = 0;
    u64 total = 0;
    for (u32 s0 = 0; s0 < ns; s0 += 64) {
        const u32 s = s0 + lane;
        const u32 fr = s < ns ? hget(s) : 0u;
        if (fr) {
            sigma++;
            total += fr;
            if (fr < ANSX_VMAX) atomicAdd(&cnt[fr], 1u);
            else {
                const u32 slot = atomicAdd(&sh_nbig, 1u);
                if (slot < nbig_cap) big_keys[slot] = ((u64)fr << 16) | s;
            }
        }
    }
    sigma = wave_sum(sigma);
    total = wave_sum(total);
    wave_lds_sync();
    // pass 2: exclusive scan of the bins -> first output position of every frequency value
    u32 nsmall, small_mass = 0;
    {
        const u32 per = ANSX_VMAX / 64;
        u32 loc = 0, locw = 0;
        for (u32 i = 0; i < per; i++) {
            const u32 t = cnt[lane * per + i];
            loc += t;
            locw += t * (lane * per + i);
        }
        const u32 incl = wave_incl_scan(loc), inclw = wave_incl_scan(locw);
        nsmall = wave_last(incl);
        small_mass = wave_last(inclw);  // sum of all frequencies below ANSX_VMAX
        u32 run = incl - loc, runw = inclw - locw;
        for (u32 i = 0; i < per; i++) {
            const u32 v = lane * per + i;
            u32 t = cnt[v];
            cnt0[v] = (u16)(t > 0xFFFFu ? 0xFFFFu : t);
            cnt[v] = run;
            wadj[v] = runw - run * v;  // (wrapping) mass before rank pos of value v = wadj[v] + pos * v
            run += t;
            runw += t * v;
        }
    }
    wave_lds_sync();
    const u32 ntot = (u32)total;
    // one store per placed symbol: (frequency, symbol) as before, or the packed pair with the remaining mass
    auto place = [&](u32 pos, u32 fr, u32 s) {
        if (oP) oP[pos] = make_uint2((fr & 0xFFFFu) | (s << 16), ntot - (wadj[fr] + pos * fr));
        else {
            oF[pos] = fr;
            oS[pos] = (u16)s;
        }
    };
    // pass 3: stable placement, 64 symbols at a time in index order
    for (u32 s0 = 0; s0 < ns; s0 += 64) {
        const u32 s = s0 + lane;
        const u32 fr = s < ns ? hget(s) : 0u;
        const bool small = fr != 0 && fr < ANSX_VMAX;
        // a frequency value that occurs once in the whole block needs no ranking: its symbol goes
        // to the start of its bin (this covers nearly all "hot" symbols, whose values are all
        // distinct and would otherwise cost one loop iteration each)
        const bool uniq = small && cnt0[fr] == 1;
        if (uniq) place(cnt[fr], fr, s);
        // Rank among the lanes of this pass that share a frequency value.  Values below
        // ANSX_MASKV (practically all that occur more than once): every lane ORs its lane bit
        // into the value's 64-bit LDS mask, reads the mask back, and its rank is the number of
        // set bits below it -- independent of the order in which the LDS unit applies the
        // atomics; the lowest lane then advances the value's cursor and clears the mask.  (The
        // LDS operations of one wave execute in program order.)
        const bool masked = small && !uniq && fr < ANSX_MASKV;
        if (masked) atomicOr(&vmask[fr], 1ull << lane);
        wave_lds_sync();
        if (masked) {
            const unsigned long long m = vmask[fr];
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            place(cnt[fr] + (u32)__popcll(below), fr, s);
            if (below == 0) {
                cnt[fr] += (u32)__popcll(m);
                vmask[fr] = 0;
            }
        }
        // larger repeated values (rare): one ballot round per distinct value
        unsigned long long todo = __ballot(small && !uniq && !masked);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const u32 v0 = (u32)__builtin_amdgcn_readlane((int)fr, leader);  // leader is wave-uniform
            const unsigned long long m = __ballot(small && !uniq && !masked && fr == v0);
            if (small && !uniq && !masked && fr == v0) {
                place(cnt[v0] + (u32)__popcll(m & ((1ull << lane) - 1ull)), fr, s);
            }
            // one wave: LDS operations execute in program order, so the cursor update below
            // follows the reads above without a barrier (a barrier here would also drain the
            // outstanding global stores on every iteration)
            if ((int)lane == leader) cnt[v0] += (u32)__popcll(m);
            todo &= ~m;
        }
    }
    wave_lds_sync();
    // big symbols: rank among themselves by (freq, sym)
    const u32 nbig = sh_nbig < nbig_cap ? sh_nbig : nbig_cap;
    for (u32 i = lane; i < nbig; i += 64) {
        const u64 key = big_keys[i];
        u32 rank = 0, before = small_mass;
        for (u32 j = 0; j < nbig; j++) {
            const bool lt = big_keys[j] < key;
            rank += lt ? 1u : 0u;
            before += lt ? (u32)(big_keys[j] >> 16) : 0u;
        }
        if (oP) oP[nsmall + rank] = make_uint2(((u32)(key >> 16) & 0xFFFFu) | ((u32)(key & 0xFFFFu) << 16), ntot - before);
        else {
            oF[nsmall + rank] = (u32)(key >> 16);
            oS[nsmall + rank] = (u16)(key & 0xFFFFu);
        }
    }
    // entropy, util.hpp:271-282
    const double nd = (double)total;
    double acc = 0.0;
    // h_deferred: the terms were produced by k_fold_hist and are summed by k_scale_attempts
    for (u32 base = 0; base < (h_deferred ? 0u : ns); base += 512) {
        wave_lds_sync();
        for (u32 u = lane; u < 512; u += 64) {
            u32 i = base + u;
            u32 fr = i < ns ? hget(i) : 0u;
            // an absent symbol adds p*log2(1) = +0.0: the sum (never -0.0) is unchanged
            double p = fr ? (double)fr / nd : 0.0;
            double q = fr ? p : 1.0;
            terms[u] = p * ansx_log2_portable(q);
        }
        wave_lds_sync();
        if (lane == 0) {
            u32 lim = ns - base < 512 ? ns - base : 512;
            u32 u = 0;
            for (; u + 4 <= lim; u += 4) {
                acc = acc + terms[u];
                acc = acc + terms[u + 1];
                acc = acc + terms[u + 2];
                acc = acc + terms[u + 3];
            }
            for (; u < lim; u++) acc = acc + terms[u];
        }
    }
    if (lane == 0) {
        double H = -acc;
        double approx = 1.0 + (double)1 / (double)1000;  // ans_util.hpp:124
        u32 m0 = 0;  // ans_util.hpp:109-112
        if (sigma != 0 && (sigma & (sigma - 1)) == 0) m0 = 31 - __clz(sigma);
        else m0 = sigma == 0 ? 0 : (32 - __clz(sigma));
        ansx_blk* B = &blk[b];
        B->n = (u32)total;
        B->sigma = sigma;
        B->m0_log2 = m0;
        if (!h_deferred) {  // otherwise k_fold_hist / k_scale_attempts own the entropy
            B->H = H;
            B->thr = H * approx;
        }
        B->resolved = B->pa_sigma == 1 ? 1u : 0u;  // (compaction: a one-value block has no model)
        B->prev = -1;
        B->status = 0;
    }
}

// Stage-1 results of ansx_log2_portable for x = 1 .. 65535 (entry 0 unused): one table per context.
// 16 bytes per entry = one load per lookup: the exponent part of stage 1 is cheap to derive from the operand
// (ansx_log2_e_of_int, checked against ansx_log2_stage1 for every x), and the candidate kernel is bound by the
// address unit's rate on exactly these 64-line gathers.
struct __attribute__((aligned(16))) ansx_log2_ent {
    double y, ylo;
};
// e of ansx_log2_stage1 for an integer 1 <= x <= 65535: its exponent, + 1 when the mantissa exceeds sqrt(2)
__device__ __forceinline__ int ansx_log2_e_of_int(u32 x)
{
    // x > 1.4142135623730951 * 2^msb  <=>  x << (31 - msb) > 1.4142135623730951 * 2^31 = 3037000499.97...; the left
    // side is an integer (< 2^32): integer operations only, this sits in the candidate kernel's inner loop
    const u32 lz = (u32)__builtin_clz(x | 1u);
    return (int)(31u - lz) + ((x << lz) >= 3037000500u ? 1 : 0);
}
__global__ void k_build_log2_lut(ansx_log2_ent* __restrict__ lut)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 65536u) return;
    ansx_log2_ent r;
    int e;
    ansx_log2_stage1((double)(i ? i : 1u), &e, &r.y, &r.ylo);
    lut[i] = r;
}

// ------------------------------------------------------------------------------------------
// K2b: scale_freqs (ans_util.hpp:77-95) + cross_entropy (util.hpp:284-298) for one candidate
// frame size per LANE.  The recurrence over symbols is serial (each S depends on the remaining
// frame), and the entropy sums must be taken left to right in index order to round like the
// scalar reference, but different frame sizes M0*2^t and different blocks are independent: one
// lane per (block, t), both loops in registers, inputs prefetched 8 symbols ahead.  Doubles
// are evaluated exactly as written (the TU is built with -ffp-contract=off).
// attMeta[(b,t)] = {ok, maxS, XH bits lo, XH bits hi}.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale_attempts(ansx_geo g, u32 NSP, u32 batch,
    const u32* __restrict__ hist, const u32* __restrict__ sortF, const u16* __restrict__ sortSym,
    ansx_blk* __restrict__ blk, u16* __restrict__ attS, u32* __restrict__ attMeta,
    const double* __restrict__ hterm, const ansx_log2_ent* __restrict__ l2lut, u32 stage_ok)
{
    __shared__ u32 scale_stage[256 / ANSX_ATTEMPTS][256];  // 32 KB: 256 (freq | sym << 16) words per block
    const u32 gid = blockIdx.x * 256 + threadIdx.x;
    const u32 b = gid / ANSX_ATTEMPTS, t = gid % ANSX_ATTEMPTS;
    if (b >= g.nblocks) return;
    const ansx_blk B = blk[b];
    if (B.resolved) return;  // (all ANSX_ATTEMPTS lanes of a block leave together)
    if (hterm != nullptr && batch == 0 && t == 0) {
        // entropy H = -sum p*log2(p), left to right in index order (util.hpp:271-282); the terms
        // come from k_fold_hist.  One lane per block, 8 blocks per wave in parallel.
        const double* ht = hterm + (u64)b * NSP;
        const u32 ns0 = B.max_sym + 1;
        double acc = 0.0;
        u32 i = 0;
        for (; i + 4 <= ns0; i += 4) {
            const double t0 = ht[i], t1 = ht[i + 1], t2 = ht[i + 2], t3 = ht[i + 3];
            acc = acc + t0;
            acc = acc + t1;
            acc = acc + t2;
            acc = acc + t3;
        }
        for (; i < ns0; i++) acc = acc + ht[i];
        const double H = -acc;
        blk[b].H = H;
        blk[b].thr = H * (1.0 + (double)1 / (double)1000);  // ans_util.hpp:124
    }
    const u32 T = batch * ANSX_ATTEMPTS + t;
    const u32 sh = B.m0_log2 + T;
    u32* meta = attMeta + ((u64)b * ANSX_ATTEMPTS + t) * 4;
    const bool dead = sh > 31;  // frame sizes beyond 2^31 are unreachable for valid inputs
    if (dead) {
        meta[0] = 0;
        meta[1] = 0;
        if (!stage_ok) return;  // (with staging the lane still helps to load its block's symbols)
    }
    // M and freq_sum are integers < 2^53: carried as doubles (exact), which keeps the
    // int64/uint64 -> double conversions of ans_util.hpp:83 out of the dependency chain
    double Md = (double)((i64)1 << (dead ? 0u : sh));
    double fsd = (double)B.n;
    const u32* F = sortF + (u64)b * NSP;
    const u16* Sy = sortSym + (u64)b * NSP;
    // candidate frequencies, layout [block][symbol][attempt]: the ANSX_ATTEMPTS lanes of a block
    // write / read 16 contiguous bytes per symbol (scattered 2-byte stores into per-attempt rows
    // cost 12x write amplification, measured with WRITE_SIZE)
    u16* S = attS + (u64)b * ANSX_ATTEMPTS * NSP + t;
    const bool wide = g.kind == 3;  // candidate frequencies as u32 (the buffer is sized for it)
    u32 maxS = 0;
    const u32 sigma = B.sigma;
    bool stop = dead;
    auto scale_one = [&](u32 fr, u32 sy) {  // one step of scale_freqs (ans_util.hpp:80-92)
        const double frd = (double)fr;
        double aratio = ansx_div_int31(Md, fsd);
        double v = aratio * frd;
        v = 0.5 + v;
        u32 sc = (u32)v;
        if (sc == 0) sc = 1;
        if (wide) ((u32*)attS)[((u64)b * NSP + sy) * ANSX_ATTEMPTS + t] = sc;  // ANSint: 32-bit frequencies (ans_int.hpp:30-34)
        else S[sy * ANSX_ATTEMPTS] = (u16)(sc > 65535u ? 65535u : sc);
        maxS = sc > maxS ? sc : maxS;
        Md = Md - (double)sc;
        fsd = fsd - frd;
        if (Md < 0.0) stop = true;  // ans_util.hpp:90-91
    };
    if (stage_ok) {
        // Blocks of at most 65535 ints (frequencies fit 16 bits): the (frequency, symbol) pairs are
        // staged 256 at a time in LDS by the block's 8 lanes, packed into one word.  The recurrence
        // stores one S per symbol, and a global load consumed behind those stores waits for them
        // (in-order vmcnt): per 8-symbol register chunk that was half of the wave's cycles; per
        // 256-symbol stage it is 2-3 drains per block.
        u32* st = &scale_stage[threadIdx.x / ANSX_ATTEMPTS][0];
        for (u32 c0 = 0; c0 < sigma; c0 += 256) {
            wave_lds_sync();  // previous stage fully consumed by the block's lanes (same wave)
#pragma unroll
            for (u32 k = 0; k < 256 / (4 * ANSX_ATTEMPTS); k++) {
                u32 j = c0 + 4 * t + 4 * ANSX_ATTEMPTS * k;
                j = j < NSP - 4 ? j : NSP - 4;  // rows are NSP entries long
                const uint4 f4 = *(const uint4*)(F + j);
                const uint2 s2 = *(const uint2*)(Sy + j);
                const uint4 pk = make_uint4((f4.x & 0xFFFFu) | (s2.x << 16), (f4.y & 0xFFFFu) | (s2.x & 0xFFFF0000u),
                    (f4.z & 0xFFFFu) | (s2.y << 16), (f4.w & 0xFFFFu) | (s2.y & 0xFFFF0000u));
                *(uint4*)(st + 4 * t + 4 * ANSX_ATTEMPTS * k) = pk;
            }
            wave_lds_sync();
            const u32 lim = sigma - c0 < 256 ? sigma - c0 : 256;
            for (u32 j0 = 0; j0 < lim && !stop; j0 += 8) {
                u32 e8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) e8[u] = st[j0 + u];  // 256-entry rows: no overrun
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (!stop && j0 + u < lim) scale_one(e8[u] & 0xFFFFu, e8[u] >> 16);
            }
            // (a lane that has stopped keeps staging: the other lanes of its block need the data)
        }
        if (dead) return;
    } else {
        // rows are 16-byte aligned and NSP (>= sigma rounded up to 8) entries long: register chunks
        // of 8 symbols, loaded two chunks ahead of the recurrence
        struct chunk {
            uint4 fa, fb, sy;
        };
        const u32 lastc = NSP - 8;
        auto load_chunk = [&](u32 j0) -> chunk {
            j0 = j0 < lastc ? j0 : lastc;  // prefetches past the row's end are clamped into it
            chunk c;
            c.fa = *(const uint4*)(F + j0);
            c.fb = *(const uint4*)(F + j0 + 4);
            c.sy = *(const uint4*)(Sy + j0);
            return c;
        };
        auto run_chunk = [&](const chunk& c, u32 j0) {
            const u32 fr8[8] = { c.fa.x, c.fa.y, c.fa.z, c.fa.w, c.fb.x, c.fb.y, c.fb.z, c.fb.w };
            const u32 sy8[8] = { c.sy.x & 0xFFFFu, c.sy.x >> 16, c.sy.y & 0xFFFFu, c.sy.y >> 16,
                c.sy.z & 0xFFFFu, c.sy.z >> 16, c.sy.w & 0xFFFFu, c.sy.w >> 16 };
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (!stop && j0 + u < sigma) scale_one(fr8[u], sy8[u]);
        };
        chunk c0 = load_chunk(0), c1 = load_chunk(8), c2;
        for (u32 j0 = 0; j0 < sigma && !stop; j0 += 24) {
            c2 = load_chunk(j0 + 16);
            run_chunk(c0, j0);
            c0 = load_chunk(j0 + 24);
            run_chunk(c1, j0 + 8);
            c1 = load_chunk(j0 + 32);
            run_chunk(c2, j0 + 16);
        }
    }
    const u32 ok = (Md == 0.0) ? 1u : 0u;
    meta[0] = ok;
    meta[1] = maxS;
    if (!ok || (maxS >= ANSX_U16_LIMIT && g.kind != 3)) return;  // failed, or the u16 exit: XH is not consulted
    // cross entropy in index order (util.hpp:284-298); note the int accumulators there.
    // Branch-free: an absent symbol contributes p = 0, q = 1 -> +0.0, which leaves the running
    // sum unchanged (the sum is never -0.0), so the 8 log2 evaluations of a chunk overlap and
    // only the final additions form the serial chain.
    const u32 ns = B.max_sym + 1;
    const u32* h = hist + (u64)b * NSP;
    const double nd = (double)(int)B.n;
    // q = S / M with M = 2^sh: the division by a power of two is exact, so it is a multiply
    // (sh == 31 would make (int)M negative in util.hpp:289; such frames never reach this point)
    // p = h / n: a block of 2^k ints (every full block of the default geometry) makes that h * 2^-k, exactly
    const bool n_pow2 = __all((B.n & (B.n - 1u)) == 0);
    const double inv_nd = 1.0 / nd;  // (exact when used)
    double acc = 0.0;
    uint4 ha = *(const uint4*)(h), hb = *(const uint4*)(h + 4);
    for (u32 i0 = 0; i0 < ns; i0 += 8) {
        const u32 h8[8] = { ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w };
        u32 s8[8];
#pragma unroll
        for (int u = 0; u < 8; u++)  // rows are NSP >= ns+8 long
            s8[u] = wide ? ((const u32*)attS)[((u64)b * NSP + i0 + u) * ANSX_ATTEMPTS + t] : (u32)S[(u64)(i0 + u) * ANSX_ATTEMPTS];
        if (i0 + 8 < ns) {
            ha = *(const uint4*)(h + i0 + 8);
            hb = *(const uint4*)(h + i0 + 12);
        }
        double tm[8];
        // log2(q), q = S * 2^-sh with integer S <= 65535: the mantissa-dependent part of
        // ansx_log2_portable comes from the per-context table (built by the same code, so the
        // value is bit-identical to evaluating the function on q), the exponent part is e(S) - sh
        ansx_log2_ent le[8];
#pragma unroll
        for (int u = 0; u < 8; u++) le[u] = l2lut[s8[u] < 65536u ? s8[u] : 0u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool valid = (i0 + u < ns) && (h8[u] != 0);
            const double p = !valid ? 0.0 : n_pow2 ? (double)h8[u] * inv_nd : ansx_div_int31((double)h8[u], nd);
            double lg = ansx_log2_stage2(ansx_log2_e_of_int(s8[u]) - (int)sh, le[u].y, le[u].ylo);
            // (ANSint frequencies above the table: the function itself; q = S * 2^-sh is exact)
            if (s8[u] >= 65536u) lg = ansx_log2_portable((double)s8[u] * ansx_bits_to_f64((u64)(1023 - (int)sh) << 52));
            tm[u] = p * (valid ? lg : 0.0);  // absent: p * log2(1) = +0.0
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = acc + tm[u];
    }
    const u64 xb = ansx_f64_to_bits(-acc);
    meta[2] = (u32)xb;
    meta[3] = (u32)(xb >> 32);
}

// ------------------------------------------------------------------------------------------
// K2c: stop rule of adjust_freqs (ans_util.hpp:127-153) over the batch's candidates, then the
// encoder table (ans_fold.hpp:82-91).  One wave per block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_select_model(ansx_geo g, u32 NSP, u32 batch,
    const u32* __restrict__ hist, const u16* __restrict__ attS, const u32* __restrict__ attMeta,
    u16* __restrict__ prevS, ansx_blk* __restrict__ blk, ansx_enc_entry* __restrict__ table,
    u32* __restrict__ tab32, u32* __restrict__ gflags, u32 last_batch, u32 always16,
    u32* __restrict__ nearlist, const u32* __restrict__ force, double near_band, u32 test_flip)
{
    // nearlist: blocks with a comparison inside near_band of its threshold (at most ANSX_NEAR_CAP listed; gflags
    // counts them all).  force: per-block log2 frame decided by the host for such blocks (0 = none): taken as is.
    // test_flip (tests): a close call is decided the WRONG way, so that the host's re-decision has something to fix.
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    ansx_blk* B = &blk[b];
    if (B->resolved) return;
    const bool wide = g.kind == 3;  // ANSint: candidate frequencies are u32 (k_scale_attempts)
    const u32 ns = B->max_sym + 1;
    const u32* h = hist + (u64)b * NSP;
    const double thr = B->thr;
    int prev = B->prev;
    int chosen = -2;
    u32 near = 0;
    // all candidates' results in one round trip (lane t holds attempt t), then the sequential rule
    uint4 mt = make_uint4(0u, 0u, 0u, 0u);
    if (lane < ANSX_ATTEMPTS) mt = *(const uint4*)(attMeta + ((u64)b * ANSX_ATTEMPTS + lane) * 4);
    const u32 forced = force != nullptr ? force[b] : 0u;
    for (u32 t = 0; t < ANSX_ATTEMPTS && !forced; t++) {
        const u32 m0 = __shfl(mt.x, (int)t), m1 = __shfl(mt.y, (int)t);
        const u32 m2 = __shfl(mt.z, (int)t), m3 = __shfl(mt.w, (int)t);
        if (!m0) continue;  // scale_freqs failed: M *= 2 (ans_util.hpp:131-135)
        const u32 T = batch * ANSX_ATTEMPTS + t;
        if (m1 >= ANSX_U16_LIMIT && g.kind != 3) {  // ans_util.hpp:141-145 (ANSint: require_u16 = false, ans_int.hpp:50)
            chosen = prev;
            break;
        }
        const double XH = ansx_bits_to_f64((u64)m2 | ((u64)m3 << 32));
        const bool close = ansx_near_threshold(XH, thr, near_band);
        near += close ? 1u : 0u;
        if ((XH < thr) != (close && test_flip != 0)) {  // ans_util.hpp:149
            chosen = (int)T;
            break;
        }
        prev = (int)T;
    }
    if (forced) {  // the host's decision for this block: the candidate with that frame, once its batch is here
        const int Tf = (int)forced - (int)B->m0_log2;
        if (forced == 0xFFFFFFFFu) chosen = -1;  // (the reference's degenerate exit)
        else if (Tf >= (int)(batch * ANSX_ATTEMPTS) && Tf < (int)((batch + 1) * ANSX_ATTEMPTS)) chosen = Tf;
    }
    if (lane == 0 && near) {
        const u32 i = atomicAdd(&gflags[ANSX_G_NEAR], 1u);
        if (i < ANSX_NEAR_CAP) nearlist[i] = b;
    }
    if (chosen == -2) {
        // still undecided after this batch: remember the last rejected success
        if (prev >= (int)(batch * ANSX_ATTEMPTS)) {
            const u32 pt = (u32)prev - batch * ANSX_ATTEMPTS;
            if (wide) {
                const u32* S = (const u32*)attS + (u64)b * ANSX_ATTEMPTS * NSP + pt;
                u32* P = (u32*)prevS + (u64)b * NSP;
                for (u32 s = lane; s < ns; s += 64) P[s] = S[(u64)s * ANSX_ATTEMPTS];
            } else {
                const u16* S = attS + (u64)b * ANSX_ATTEMPTS * NSP + pt;
                u16* P = prevS + (u64)b * NSP;
                for (u32 s = lane; s < ns; s += 64) P[s] = S[(u64)s * ANSX_ATTEMPTS];
            }
        }
        if (lane == 0) {
            B->prev = prev;
            atomicAdd(&gflags[ANSX_G_PAD], 1u);  // encode: blocks still undecided after this batch
            if (last_batch) {
                B->resolved = 1;
                B->status = 7;  // ANSX_ERR_MODEL
                atomicOr(&gflags[ANSX_G_ERR], 1u << 7);
            }
        }
        return;
    }
    // (ANSint frames may exceed 2^16 -- 32-bit frequencies, ans_int.hpp:30-34,50: the 16-byte table entries, the
    // integer-state encoder and the slot -> symbol decoder take any frame the reference's own arithmetic survives;
    // beyond 2^27 its 64-bit renormalisation bound K * RADIX * freq overflows, so that is where this build stops.
    // The 16-bit codecs stop there too: their state update (quotient below 2^36) << log2 M leaves 64 bits at 2^28 --
    // in the reference as well -- and the prelude writer's 32-bit code arithmetic relies on frames below 2^30.)
    if (chosen >= 0 && B->m0_log2 + (u32)chosen > 27) chosen = -1;
    if (chosen < 0) {  // "prev" is the all-zero vector: reference's degenerate exit
        if (lane == 0) {
            B->resolved = 1;
            B->status = 7;
            B->logM = 0;
            atomicOr(&gflags[ANSX_G_ERR], 1u << 7);
        }
        return;
    }
    const bool from_batch = chosen >= (int)(batch * ANSX_ATTEMPTS);
    const u16* S = from_batch ? attS + (u64)b * ANSX_ATTEMPTS * NSP + (chosen - batch * ANSX_ATTEMPTS)
                              : prevS + (u64)b * NSP;
    const u32* S32 = from_batch ? (const u32*)attS + (u64)b * ANSX_ATTEMPTS * NSP + (chosen - batch * ANSX_ATTEMPTS)
                                : (const u32*)prevS + (u64)b * NSP;
    const u32 sstride = from_batch ? ANSX_ATTEMPTS : 1u;
    auto chosen_freq = [&](u32 s) -> u32 { return wide ? S32[(u64)s * sstride] : (u32)S[(u64)s * sstride]; };
    // exclusive scan of the chosen frequencies -> encoder table (ans_fold.hpp:82-91):
    // lane = symbol, 64 symbols per pass (coalesced), wave prefix sum, carry between passes
    ansx_enc_entry* tab = table + (u64)b * NSP;
    u32* t32 = tab32 + (u64)b * NSP;
    // the 16-byte entries (32-bit base/freq + reciprocal) are only read by the integer-state encoder
    // and the generic prelude writer: frames above 2^16 always get them, other blocks only when the
    // host already knows such a consumer will run (always16) -- or later through k_table16_from32
    const bool write16 = always16 || (B->m0_log2 + (u32)chosen) > 16;
    u32 carry = 0;
    // (alphabets of up to 640 symbols: the histogram and the chosen frequencies of all rounds are requested before
    // the first scan -- a round trip per 64 symbols made this 1-wave-per-block kernel 47 us per block)
    constexpr u32 SEL_PRE = 10;
    u32 hpre[SEL_PRE], spre[SEL_PRE];
    const bool pre = ns <= 64 * SEL_PRE;
    if (pre) {
#pragma unroll
        for (u32 r = 0; r < SEL_PRE; r++) {
            const u32 s = r * 64 + lane;
            hpre[r] = s < ns ? h[s] : 0u;
            spre[r] = s < ns ? chosen_freq(s) : 0u;
        }
    }
    for (u32 s0 = 0, r = 0; s0 < ns; s0 += 64, r++) {
        const u32 s = s0 + lane;
        u32 hv, sv;
        if (pre) {
            hv = 0;
            sv = 0;
#pragma unroll
            for (u32 q = 0; q < SEL_PRE; q++)
                if (q == r) {
                    hv = hpre[q];
                    sv = spre[q];
                }
        } else {
            hv = s < ns ? h[s] : 0u;
            sv = s < ns ? chosen_freq(s) : 0u;
        }
        const u32 fr = hv ? sv : 0u;
        const u32 incl = wave_incl_scan(fr);
        const u32 base = carry + incl - fr;
        if (s < ns) {
            if (write16) {
                ansx_enc_entry e;
                e.base = base;
                e.freq = fr;
                e.rcp = fr ? 1.0 / (double)fr : 0.0;
                tab[s] = e;
            }
            t32[s] = (base << 16) | fr;  // valid while M <= 65536 (base < 2^16, freq < 65535)
        }
        carry += wave_last(incl);
    }
    if (lane == 0) {
        u32 logM = B->m0_log2 + (u32)chosen;
        B->logM = logM;
        B->resolved = 1;
        // same-address atomics serialise in L2 (16 K blocks): only the few blocks that raise the
        // running maximum issue one
        if (__hip_atomic_load(&gflags[ANSX_G_MAXLOGM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < logM)
            atomicMax(&gflags[ANSX_G_MAXLOGM], logM);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXNSYMS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ns)
            atomicMax(&gflags[ANSX_G_MAXNSYMS], ns);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXSIGMA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B->sigma)
            atomicMax(&gflags[ANSX_G_MAXSIGMA], B->sigma);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)chosen)
            atomicMax(&gflags[ANSX_G_MAXT], (u32)chosen);
    }
}

// 16-byte table entries for the blocks whose frame is at most 2^16 (k_select_model skipped them),
// rebuilt from the compact entries.  Only launched in the rare mixed case: some block of the call
// needs a frame above 2^16, so the integer-state encoder runs for all of them.
__global__ __launch_bounds__(256) void k_table16_from32(ansx_geo g, u32 NSP, const ansx_blk* __restrict__ blk,
    const u32* __restrict__ tab32, ansx_enc_entry* __restrict__ table)
{
    const u32 b = blockIdx.x;
    if (blk[b].status || blk[b].logM > 16) return;
    const u32 ns = blk[b].max_sym + 1;
    for (u32 s = threadIdx.x; s < ns; s += 256) {
        const u32 w = tab32[(u64)b * NSP + s];
        ansx_enc_entry e;
        e.base = w >> 16;
        e.freq = w & 0xFFFFu;
        e.rcp = e.freq ? 1.0 / (double)e.freq : 0.0;
        table[(u64)b * NSP + s] = e;
    }
}

// ------------------------------------------------------------------------------------------
// K3: prelude writer (ans_util.hpp:46-63).  The binary interpolative code (interp.hpp:65-79)
// is a pre-order walk of a balanced tree whose node contexts depend only on input values, so
// every item's codeword is computed independently: lane i finds its node by descending from
// the root (rank = pre-order position), codeword lengths are prefix-summed in rank order, and
// codewords are OR-ed into an LDS bit buffer (LSB-first u32 words, bits.hpp:84-105).  Padding
// bits of the last word are written as zero (canonical form).
// ------------------------------------------------------------------------------------------
struct ansx_code {
    u32 code, len, rank;
};

// Exclusive prefix sum of one value per thread over a workgroup of NT threads (NT a multiple of
// 64, <= 1024): wave scan over shuffles, wave totals through LDS.  *total = sum over the group.
// wsum: NT/64 + 1 words of LDS scratch.  Contains two barriers (none for a single-wave workgroup).
template <typename T> __device__ __forceinline__ T block_excl_scan(T v, T* wsum, u32 tid, u32 nt, T* total)
{
    const T incl = wave_incl_scan(v);
    const u32 nw = nt >> 6;
    if (nw == 1) {  // a single wave: no LDS, no barrier
        *total = wave_last(incl);
        return incl - v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    if (tid < 64) {
        const T w = tid < nw ? wsum[tid] : (T)0;
        const T wi = wave_incl_scan(w);  // nw <= 16: the first row would do
        if (tid < nw) wsum[tid] = wi - w;
        if (tid == nw - 1) wsum[nw] = wi;
    }
    __syncthreads();
    *total = wsum[nw];
    return wsum[tid >> 6] + incl - v;
}

// The tree node of item i in the interpolative code of ns items: the run [a, a + n) whose middle it is, and its
// pre-order rank.  A function of (ns, i) only -- k_build_interp_geo tabulates it per context for alphabets up to 4096
// slots (geo: triangular, row ns at ns (ns - 1) / 2, entry = {a | n << 16, rank}); the descent is the fallback.
struct ansx_node {
    u32 a, n, rank;
};
__device__ __forceinline__ ansx_node interp_node(u32 ns, u32 i)
{
    u32 a = 0, n = ns, rank = 0;
    for (;;) {
        const u32 h = (n + 1) >> 1;
        const u32 mid = a + h - 1;
        if (i == mid) break;
        if (i < mid) {
            n = h - 1;
            rank += 1;
        } else {
            rank += h;
            a += h;
            n -= h;
        }
    }
    return ansx_node{ a, n, rank };
}
ANSX_HD u64 interp_geo_row(u32 ns) { return (u64)ns * (ns - 1) / 2; }
__global__ __launch_bounds__(256) void k_build_interp_geo(u32 max_ns, uint2* __restrict__ geo)
{
    const u32 ns = blockIdx.x + 1;  // 1 .. max_ns
    if (ns > max_ns) return;
    uint2* row = geo + interp_geo_row(ns);
    for (u32 i = threadIdx.x; i < ns; i += 256) {
        const ansx_node nd = interp_node(ns, i);
        row[i] = make_uint2(nd.a | (nd.n << 16), nd.rank);
    }
}

// W: the arithmetic type -- u64 in general; u32 where the universe is below 2^31 (frames up to 2^30: every quantity
// below is at most 2 u), half the instructions on a machine without 64-bit integer VALU operations.
// the arithmetic of one item once its node (a, n, rank) and the three inc[] values around it are known:
// inc_lo = inc[a - 1] (unused when a == 0), inc_hi = inc[a + n] (unused when a + n == ns), inc_mid = inc[i]
template <typename W = u64>
__device__ __forceinline__ ansx_code interp_code(u32 ns, W u, u32 a, u32 n, u32 rank, u32 inc_lo, u32 inc_hi, u32 inc_mid)
{
    const u32 h = (n + 1) >> 1;
    const W one = 1;
    const W n1 = h - 1, n2 = n - h;
    const W low = (a == 0) ? one : (W)inc_lo + 2;            // parent v + 1
    const W high = (a + n == ns) ? (u + one) : (W)inc_hi;    // parent v - 1
    const W v = (W)inc_mid + one;                            // interp.hpp:73
    W val = v - low - n1 + one;
    const W U = high - n2 - low - n1 + one;
    ansx_code c;
    c.rank = rank;
    if (U == 1) {  // interp.hpp:31-32
        c.code = 0;
        c.len = 0;
        return c;
    }
    u32 bb;  // hi(U-1)+1
    if constexpr (sizeof(W) == 8) bb = 64 - __clzll((unsigned long long)(U - 1));
    else bb = 32 - __clz((u32)(U - 1));
    const W d = 2 * U - (one << bb);
    val = val + (U - (d >> 1));
    if (val > U) val -= U;
    const W m = (one << bb) - U;
    if (val <= m) {
        c.code = (u32)(val - one);
        c.len = bb - 1;
    } else {
        val += m;
        c.code = (u32)(((val - one) >> 1) | (((val - one) & one) << (bb - 1)));
        c.len = bb;
    }
    return c;
}
__device__ __forceinline__ ansx_node interp_node_of(u32 ns, u32 i, const uint2* __restrict__ geo_row)
{
    if (geo_row != nullptr) {
        const uint2 e = geo_row[i];
        return ansx_node{ e.x & 0xFFFFu, e.x >> 16, e.y };
    }
    return interp_node(ns, i);
}
// W: the arithmetic type -- u64 in general; u32 where the universe is below 2^31 (frames up to 2^30: every quantity
// below is at most 2 u), half the instructions on a machine without 64-bit integer VALU operations.
template <typename W = u64>
__device__ __forceinline__ ansx_code interp_item(const u32* __restrict__ inc, u32 ns, W u, u32 i,
    const uint2* __restrict__ geo_row = nullptr)
{
    const ansx_node nd = interp_node_of(ns, i, geo_row);
    const u32 inc_lo = nd.a == 0 ? 0u : inc[nd.a - 1];
    const u32 inc_hi = nd.a + nd.n == ns ? 0u : inc[nd.a + nd.n];
    return interp_code<W>(ns, u, nd.a, nd.n, nd.rank, inc_lo, inc_hi, inc[i]);
}
// Eight items i0, i0 + 256, ... at once (items at or beyond ns: length 0): all nodes first, then all 24 inc[] reads in
// flight together, then the arithmetic -- for callers whose inc[] is in HBM, where an item's reads are a round trip
template <typename W = u64>
__device__ __forceinline__ void interp_items8(const u32* __restrict__ inc, u32 ns, W u, u32 i0, const uint2* __restrict__ geo_row,
    ansx_code (&c8)[8])
{
    ansx_node nd[8];
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 i = i0 + 256 * j;
        nd[j] = interp_node_of(ns, i < ns ? i : i0, geo_row);
    }
    u32 lo[8], hi[8], mid[8];
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 i = i0 + 256 * j;
        lo[j] = inc[nd[j].a == 0 ? 0u : nd[j].a - 1];                       // (always a valid address; unused where a == 0)
        hi[j] = inc[nd[j].a + nd[j].n >= ns ? ns - 1 : nd[j].a + nd[j].n];  // (likewise at the right edge)
        mid[j] = inc[i < ns ? i : i0];
    }
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        c8[j] = interp_code<W>(ns, u, nd[j].a, nd[j].n, nd[j].rank, lo[j], hi[j], mid[j]);
        if (i0 + 256 * j >= ns) c8[j].len = 0;
    }
}

// Interpolative prelude of one block from inc[] in LDS (workgroup of 256 threads, every thread calls
// this): codes -> lengths -> exclusive scan -> bit buffer -> bytes at `out` (rfold header first).
// IPT > 0: at most 256 * IPT symbols, every thread keeps the codes of its <= IPT items from the length
// pass for the packing pass (one tree descent per item); IPT == 0: any alphabet, two descents.
// off: [ns] u32, bits: [ns + 1] u32 (LDS; 31 bits per item is the format's ceiling), sh_part: 8 u32.
// PA == false: the codec prelude (rfold header, vbyte(max_sym), log2 M, code over universe 2^logM + ns + 1),
//               written behind the block's B->pre_bytes.
// PA == true:  the alphabet header of per-block compaction (pseudo_adaptive.cpp:106-113): u32 ns, u32 `uni`,
//               code of the ns running sums over universe `uni`; sets B->pre_bytes.
// W32: the caller guarantees a universe below 2^31 (32-bit code arithmetic, see interp_item).
// KEEP8 (IPT == 0 only): the caller guarantees ns <= 2048 * KEEP8; the codes of a thread's <= 8 KEEP8 items stay in
// registers between the two passes (two words each) instead of being derived again -- on alphabets beyond the tabulated
// tree geometry (> 4096 slots) a derivation is a 14-level descent per item, half of the writer's time there.
template <int IPT, bool PA = false, bool W32 = false, int KEEP8 = 0, int NTH = 256>
__device__ __forceinline__ void prelude_emit(const ansx_geo& g, ansx_blk* B, u32 ns, u32 logM, const u32* inc,
    u32* off, u32* bits, u32* sh_part, u8* __restrict__ out, const u32* __restrict__ mostfreq, u32 b, u32 tid,
    u64 uni = 0, u32* __restrict__ hints = nullptr, const uint2* __restrict__ geo = nullptr)
{
    const uint2* const geo_row = (geo != nullptr && ns >= 1) ? geo + interp_geo_row(ns) : nullptr;  // (tabulated tree nodes)
    constexpr bool SMALL = IPT > 0;
    static_assert(NTH == 256 || IPT > 0, "the eight-items-per-round form strides by 256 threads");
    typedef typename std::conditional<W32, u32, u64>::type W;
    const W u = (W)(PA ? uni : ((u64)1 << logM) + ns + 1);  // ans_util.hpp:60
    ansx_code mine[SMALL ? IPT : 1];
    u32 kept_code[KEEP8 > 0 ? 8 * KEEP8 : 1], kept_lr[KEEP8 > 0 ? 8 * KEEP8 : 1];
    (void)kept_code, (void)kept_lr;
    if (SMALL) {
#pragma unroll
        for (int j = 0; j < (SMALL ? IPT : 1); j++) {
            const u32 i = tid + NTH * j;
            mine[j].len = 0;
            if (i < ns) {
                mine[j] = interp_item<W>(inc, ns, u, i, geo_row);
                off[mine[j].rank] = mine[j].len;
            }
        }
    } else {
        // (eight items per round: inc[] may live in HBM on this path -- k_model_finish<0> --, where an item's three loads
        // are a dependent round trip; one item at a time that was 116 of the kernel's 136 us per block on 8000-symbol alphabets)
        if constexpr (KEEP8 > 0) {
#pragma unroll
            for (int r = 0; r < KEEP8; r++) {
                const u32 i0 = tid + 8u * NTH * (u32)r;
#pragma unroll
                for (u32 j = 0; j < 8; j++) kept_code[8 * r + j] = 0, kept_lr[8 * r + j] = 0;
                if (i0 < ns) {
                    ansx_code c8[8];
                    interp_items8<W>(inc, ns, u, i0, geo_row, c8);
#pragma unroll
                    for (u32 j = 0; j < 8; j++) {
                        if (i0 + NTH * j < ns) off[c8[j].rank] = c8[j].len;
                        kept_code[8 * r + j] = c8[j].code;
                        kept_lr[8 * r + j] = c8[j].len | (c8[j].rank << 8);  // (len 0 beyond ns)
                    }
                }
            }
        } else
        for (u32 i0 = tid; i0 < ns; i0 += NTH * 8) {
            ansx_code c8[8];
            interp_items8<W>(inc, ns, u, i0, geo_row, c8);
#pragma unroll
            for (u32 j = 0; j < 8; j++)
                if (i0 + NTH * j < ns) off[c8[j].rank] = c8[j].len;
        }
        __threadfence_block();  // (off[] and the bit buffer may live in HBM on this path)
    }
    __syncthreads();
    // exclusive scan of off[0..ns)
    const u32 per = (ns + NTH - 1) / NTH;
    const u32 lo = tid * per, hi = (lo + per) < ns ? (lo + per) : ns;
    u32 sum = 0;
    for (u32 s = lo; s < hi; s++) sum += off[s];
    u32 total_bits;
    u32 run = block_excl_scan<u32>(sum, sh_part, tid, NTH, &total_bits);
    for (u32 s = lo; s < hi; s++) {
        u32 t = off[s];
        off[s] = run;
        run += t;
    }
    const u32 nwords = (total_bits + 31) >> 5;
    for (u32 w = tid; w <= nwords; w += NTH) bits[w] = 0;
    if (!SMALL) __threadfence_block();
    __syncthreads();
    if (!PA && hints != nullptr && tid < 8) {
        // Parse hints for the container index: where the decoder may enter this code in parallel.  The code
        // is a pre-order walk, so the right subtree of a node (pre-order rank r, n items) begins at the
        // item of rank r + h, h = (n + 1) / 2; off[] holds every item's bit offset by rank.  hints[0] = valid
        // bits, hints[1 + i] = offset of the right subtree of top node i (first three levels, breadth
        // first: children of i are 2i + 1, 2i + 2), 0 where it is empty.
        u32 hv = total_bits;
        if (tid >= 1) {
            const u32 node = tid - 1;  // 0 .. 6
            u32 r = 0, n = ns;
            // path from the root: bits of (node + 1) below its leading one, most significant first
            const u32 depth = 31 - __clz(node + 1);
            for (u32 k = depth; k-- > 0;) {
                const u32 h = (n + 1) >> 1;
                if ((((node + 1) >> k) & 1u) == 0) {  // left child
                    r = r + 1;
                    n = n ? h - 1 : 0;
                } else {  // right child
                    r = r + h;
                    n = n ? n - h : 0;
                }
            }
            const u32 h = (n + 1) >> 1;
            hv = (n != 0 && n - h != 0) ? off[r + h] : 0u;
        }
        hints[(u64)b * 8 + tid] = hv;
    }
    auto place = [&](const ansx_code& c) {
        if (c.len) {
            u32 o = off[c.rank];
            u32 w = o >> 5, sh = o & 31;
            atomicOr(&bits[w], c.code << sh);
            if (sh + c.len > 32) atomicOr(&bits[w + 1], c.code >> (32 - sh));
        }
    };
    if (SMALL) {
#pragma unroll
        for (int j = 0; j < (SMALL ? IPT : 1); j++) place(mine[j]);
    } else {
        if constexpr (KEEP8 > 0) {
#pragma unroll
            for (int q = 0; q < 8 * KEEP8; q++) {
                ansx_code c;
                c.code = kept_code[q], c.len = kept_lr[q] & 0xFFu, c.rank = kept_lr[q] >> 8;
                place(c);
            }
        } else
        for (u32 i0 = tid; i0 < ns; i0 += NTH * 8) {
            ansx_code c8[8];
            interp_items8<W>(inc, ns, u, i0, geo_row, c8);
#pragma unroll
            for (u32 j = 0; j < 8; j++) place(c8[j]);
        }
        __threadfence_block();
    }
    __syncthreads();
    const u32 nbytes = nwords * 4;
    if (PA) {
        if (tid == 0) {
            st_u32_unaligned(out, ns);
            st_u32_unaligned(out + 4, (u32)uni);
            B->pre_bytes = 8 + nbytes;
        }
        for (u32 j = tid; j < nbytes; j += NTH) out[8 + j] = (u8)(bits[j >> 2] >> (8 * (j & 3)));
        return;
    }
    const u32 p0 = B->pre_bytes;
    u32 p = p0;
    if (g.kind == 1) {  // ans_reorder_fold.hpp:132-154
        const u32 T = fold_T(g.f);
        const u32 flag = B->flag;
        if (tid == 0) st_u32_unaligned(out + p, flag);
        p += 4;
        if (flag) {
            const u32* mf = mostfreq + (u64)b * T;
            for (u32 i = tid; i < T; i += NTH) st_u32_unaligned(out + p + 4 * (u64)i, mf[i]);
            p += 4 * T;
        }
    }
    const u32 hdr = p - p0;
    // vbyte(max_sym) (vbyte.hpp:57-80) + log2(M) byte (ans_util.hpp:51)
    u32 ms = ns - 1;
    u32 vb = 1;
    for (u32 t = ms; t >= 128; t >>= 7) vb++;
    if (tid == 0) {
        u32 t = ms;
        u32 q = p;
        while (t >= 128) {
            out[q++] = (u8)((t & 127) | 128);
            t >>= 7;
        }
        out[q++] = (u8)(t & 127);
        out[q] = (u8)logM;
    }
    p += vb + 1;
    for (u32 w = tid; w < nwords; w += NTH) st_u32_unaligned(out + p + 4 * w, bits[w]);  // (any byte alignment)
    if (tid == 0) {
        B->hdr_bytes = hdr;
        B->prelude_bytes = p + nbytes;
    }
}

// IPT > 0: alphabets of at most 256 * IPT slots in frames up to 2^16 -- inc[] lives in LDS and is
// built from the compact 4-byte table entries (the workgroup is a latency chain otherwise: 16-byte
// table entries, inc[] through global memory behind a fence, two tree descents per item).
template <int IPT>
__global__ __launch_bounds__(256) void k_write_prelude(ansx_geo g, u32 NSP,
    const ansx_enc_entry* __restrict__ table, const u32* __restrict__ tab32, u32* __restrict__ incbuf,
    ansx_blk* __restrict__ blk, u8* __restrict__ scratch, u64 scr_stride, const u32* __restrict__ mostfreq,
    u32* __restrict__ hints, u32 cap, const uint2* __restrict__ geo, u32* __restrict__ g_work = nullptr)
{
    // g_work (IPT == 0 only; alphabets above 16 Ki slots, whose two arrays do not fit the LDS): 2 cap + 16 words of
    // HBM per block for off[] and the bit buffer instead.
    // cap = words per LDS array: NSP, or on an optimistic call the alphabet hint the whole call is sized for (a
    // 2300-symbol alphabet then takes 28 KB instead of the 48 KB of its 4096 slots: 5 instead of 3 workgroups per
    // CU); a block above it writes nothing -- the call is repeated anyway (ANSX_G_MAXNSYMS > hint).
    constexpr bool SMALL = IPT > 0;
    extern __shared__ u32 lds32[];
    __shared__ u32 sh_part[8];
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    ansx_blk* B = &blk[b];
    // (small alphabets: the table row is requested together with the block's fields, not after them)
    const u32* t32 = tab32 + (u64)b * NSP;
    u32 epre[4] = {};
    if (SMALL) {
#pragma unroll
        for (u32 r = 0; r < 4; r++) epre[r] = tid + 256 * r < NSP ? t32[tid + 256 * r] : 0u;
    }
    if (B->status || !B->resolved || B->pa_sigma == 1) {  // (unresolved: optimistic single-batch call, the host repeats it)
        if (tid == 0) {
            B->prelude_bytes = B->pa_sigma == 1 ? B->pre_bytes : 0;  // a one-value block is its alphabet header
        }
        return;
    }
    const u32 ns = B->max_sym + 1;
    const u32 logM = B->logM;
    if (ns > cap) {
        if (tid == 0) B->prelude_bytes = 0;
        return;
    }
    u32* off = (!SMALL && g_work != nullptr) ? g_work + (u64)b * (2 * (u64)cap + 16) : lds32;  // [ns]
    u32* bits = off + cap;    // bit buffer
    u32* inc = SMALL ? lds32 + 2 * cap : incbuf + (u64)b * NSP;
    if (SMALL) {
#pragma unroll
        for (u32 r = 0; r < 4; r++) {
            const u32 s = tid + 256 * r;
            if (s < ns) inc[s] = (epre[r] >> 16) + (epre[r] & 0xFFFFu) + s;  // ans_util.hpp:54-58: inc[s] = inc[s-1] + nfreq[s] + 1
        }
        for (u32 s = tid + 1024; s < ns; s += 256) {
            const u32 e = t32[s];
            inc[s] = (e >> 16) + (e & 0xFFFFu) + s;
        }
    } else {
        const ansx_enc_entry* tab = table + (u64)b * NSP;
        for (u32 s = tid; s < ns; s += 256) {
            ansx_enc_entry e = tab[s];
            inc[s] = e.base + e.freq + s;
        }
        __threadfence_block();
    }
    __syncthreads();
    prelude_emit<IPT, false, true>(g, B, ns, logM, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, mostfreq, b, tid, 0, hints, geo);  // (universe 2^logM + ns + 1 < 2^31: frames end at 2^27)
}

// ------------------------------------------------------------------------------------------
// K5: the 4-state rANS encoder (ans_fold.hpp:100-120, 249-278).  A quad of lanes owns one
// block; lane q carries state q and encodes in[4g+3-q] of every group g, walking the block
// backwards.  Bytes emitted by the four lanes of a step (k exception bytes, then the 32-bit
// renormalisation word) are laid out by a quad prefix sum over DPP.  state/freq is evaluated
// exactly through the reciprocal stored in the table plus a +-1 correction.
// ------------------------------------------------------------------------------------------
struct enc_lane {
    u64 st;     // integer form (global-table variant, restart points, flush)
    double sd;  // f64 form (LDS-table variant): exact, state < 2^36 * M <= 2^52
    u32 p;      // byte cursor relative to the block's stream start (uniform within the quad)
};

// one symbol of one state.  x: value (its exception-byte count k in e.k); e: its table entry,
// fetched ahead of the dependency chain.
struct enc_ent {
    u32 freq, base, k;
    double rcp;
};

__device__ __forceinline__ void enc_update(enc_lane& L, u32 x, const enc_ent e, bool active, u32 ql,
    u32 logM, u8* __restrict__ out)
{
    const u32 k = e.k, freq = e.freq;
    const u32 eb = x & ((1u << (8 * k)) - 1u);
    u64 st = L.st;
    // renormalise: state >= K*RADIX*freq  <=>  (state >> 36) >= freq   (ans_fold.hpp:105-110)
    const bool rn = active && ((st >> 36) >= (u64)freq);
    const u32 w = (u32)st;
    if (rn) st >>= 32;
    // exact q = st / freq, r = st % freq; st < 2^36 * freq <= 2^52 (ans_fold.hpp:111).
    // rcp is 1/freq to < 2^-40 relative, so trunc(st * rcp) is within +-1 of the quotient;
    // the remainder of the estimate lies in (-2^16, 2^17), so 32-bit arithmetic decides it.
    if (logM > 16) {
        // frames above 2^16 (ANSint, whole-list modes): the state no longer fits a double -- plain 64-bit division
        const u64 q = st / freq, r = st % freq;
        st = (q << logM) + r + e.base;
    } else {
        const u32 st_hi = (u32)(st >> 32), st_lo = (u32)st;
        const double std_ = __builtin_fma((double)st_hi, 4294967296.0, (double)st_lo);
        const double qd = std_ * e.rcp;                             // < 2^36 + 1
        const u32 q_hi = (u32)(qd * (1.0 / 4294967296.0));          // trunc
        const u32 q_lo = (u32)__builtin_fma(-(double)q_hi, 4294967296.0, qd);
        int r = (int)(st_lo - q_lo * freq);
        const int adj = (r < 0) ? -1 : ((r >= (int)freq) ? 1 : 0);
        r -= adj * (int)freq;
        const u64 q = (((u64)q_hi << 32) | q_lo) + (u64)(i64)adj;
        st = (q << logM) + (u64)((u32)r + e.base);
    }
    if (active) L.st = st;
    const u32 c = active ? (k + (rn ? 4u : 0u)) : 0u;
    u32 total;
    const u32 incl = quad_incl_scan(c, ql, &total);
    u8* a = out + (L.p + (incl - c));
    if (active) {
        if (k == 1) a[0] = (u8)eb;
        if (k >= 2) st_u16_unaligned(a, (u16)eb);
        if (k == 3) a[2] = (u8)(eb >> 16);
        if (rn) st_u32_unaligned(a + k, w);
    }
    L.p += total;
}

// The same step with the state carried as a double (LDS-table variant, M <= 2^16).  Every
// quantity is an integer below 2^53, so all operations are exact:
//   renorm   state >= 2^36*freq  <=>  sd >= thr;   hi = trunc(sd * 2^-32), w = sd - hi*2^32
//   divide   q ~ trunc(s * rcp) (within +-1), r = fma(-q, F, s) exact, +-1 correction
//   update   sd' = fma(q, M, r + base)  < 2^52 + 2^17
// No int<->f64 conversion, 64-bit shift or 32-bit integer multiply sits on the dependency chain
// (those slow-rate operations cost ~0.37 ms of the 1.40 ms integer form, measured by ablation).
struct enc_ent_d {
    double Fd, based, rcp, thr;
    u32 k;
};

__device__ __forceinline__ void enc_update_d(enc_lane& L, u32 x, const enc_ent_d e, bool active, u32 ql,
    double Md, u8* __restrict__ out)
{
    const u32 k = e.k;
    const u32 eb = x & ((1u << (8 * k)) - 1u);
    const double sd = L.sd;
    const bool rn = active && (sd >= e.thr);
    const double hi = __builtin_trunc(sd * (1.0 / 4294967296.0));
    const double wd = __builtin_fma(-hi, 4294967296.0, sd);
    const u32 w = (u32)wd;
    const double s0 = rn ? hi : sd;
    double qd = __builtin_trunc(s0 * e.rcp);
    double rd = __builtin_fma(-qd, e.Fd, s0);
    const double adj = (rd < 0.0) ? -1.0 : ((rd >= e.Fd) ? 1.0 : 0.0);
    qd = qd + adj;
    rd = __builtin_fma(-adj, e.Fd, rd);
    const double nsd = __builtin_fma(qd, Md, rd + e.based);
    if (active) L.sd = nsd;
    const u32 c = active ? (k + (rn ? 4u : 0u)) : 0u;
    u32 total;
    const u32 incl = quad_incl_scan(c, ql, &total);
    u8* a = out + (L.p + (incl - c));
    if (active) {
        if (k == 1) a[0] = (u8)eb;
        if (k >= 2) st_u16_unaligned(a, (u16)eb);
        if (k == 3) a[2] = (u8)(eb >> 16);
        if (rn) st_u32_unaligned(a + k, w);
    }
    L.p += total;
}

// Branch-free form of the same step for the pipelined main loop (LDS-table variant).
//
// Stores.  hipcc keeps an s_and_saveexec / s_cbranch_execz / s_or_b64 region around every
// predicated global store (VMEM is never if-converted): 4 branches and 8 tiny basic blocks per
// step, each an issue bubble for a lone wave and a barrier for the scheduler.  Raw buffer stores
// make the predicate part of the ADDRESS instead: a lane with nothing to write gets offset 2^31,
// beyond num_records, and the hardware drops the access.  Three stores per step, none of which
// needs its value masked: byte 0 of x at a (k odd), 16 bits of x >> 8(k&1) at a + (k&1)
// (k >= 2), the renormalisation word at a + k.
//
// Divide.  rcp is a deliberate UNDER-estimate of 1/F (relative deficit in (2^-39.01, 2^-38), see
// enc_tab<true>::getp), so q' = trunc(s0 * rcp) is q or q-1, never above: one comparison
// r' = s0 - q'F >= F decides the correction.  The update needs no corrected remainder:
// q*M + (s0 - q*F) + base = q*(M - F) + (s0 + base), all integers below 2^53, so one fma is exact.
//
// Byte cursor.  Each lane puts its byte count c <= 7 into byte `ql` of a word; two DPP adds give
// every lane the quad's packed counts S, and v_sad_u8 (sum of the four bytes of a word, plus an
// accumulator) turns S & lomask into the lane's address and S into the next cursor.
#define ANSX_BUF_OOB 0x80000000u
struct enc_ent_n {
    double Fd, rcp, thr, MF, based;
    u32 k, kpos;
};
struct enc_quad_const {
    u32 four_pos;  // 4 << (8 * ql)
    u32 lomask;    // (1 << (8 * ql)) - 1
};

template <int P0, int P1, int P2, int P3> __device__ __forceinline__ u32 quad_add_perm(u32 v)
{
    // r = v[quad_perm] + v in one VOP2-DPP; s_nop 1 = the 2 wait states a DPP read needs after a
    // VALU write of the same VGPR
    u32 r;
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf"
                 : "=v"(r)
                 : "v"(v), "n"(P0), "n"(P1), "n"(P2), "n"(P3));
    return r;
}

__device__ __forceinline__ double f64_from_hi(u32 hi) { return __builtin_bit_cast(double, (u64)hi << 32); }

__device__ __forceinline__ void enc_update_n(enc_lane& L, u32 x, const enc_ent_n e, const enc_quad_const qc,
    __amdgpu_buffer_rsrc_t rsrc)
{
    const u32 k = e.k;
    const double sd = L.sd;
    const bool rn = sd >= e.thr;
    // renormalise: s0 = rn ? state >> 32 : state;  w = low 32 bits (only stored when rn)
    const double s0 = __builtin_trunc(sd * f64_from_hi(rn ? 0x3DF00000u : 0x3FF00000u));
    const u32 w = (u32)__builtin_fma(-s0, 4294967296.0, sd);
    double qd = __builtin_trunc(s0 * e.rcp);
    const double rd = __builtin_fma(-qd, e.Fd, s0);
    qd = qd + f64_from_hi((rd >= e.Fd) ? 0x3FF00000u : 0u);
    L.sd = __builtin_fma(qd, e.MF, s0 + e.based);
    // byte cursor (L.p carries the lane's buffer offset bias inside the pipelined loop)
    const u32 v = e.kpos + (rn ? qc.four_pos : 0u);
    const u32 s1 = quad_add_perm<1, 0, 3, 2>(v);
    const u32 S = quad_add_perm<2, 3, 0, 1>(s1);
    const u32 a = __builtin_amdgcn_sad_u8(S & qc.lomask, 0u, L.p);
    L.p = __builtin_amdgcn_sad_u8(S, 0u, L.p);
    const u32 t = k & 1u;
    __builtin_amdgcn_raw_buffer_store_b8((u8)x, rsrc, t ? a : ANSX_BUF_OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b16((u16)(x >> (t << 3)), rsrc, (k >= 2) ? a + t : ANSX_BUF_OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(w, rsrc, rn ? a + k : ANSX_BUF_OOB, 0, 0);
}

__device__ __forceinline__ u64 f64_to_u64_exact(double d)  // d is an integer in [0, 2^53)
{
    const double hi = __builtin_trunc(d * (1.0 / 4294967296.0));
    const double lo = __builtin_fma(-hi, 4294967296.0, d);
    return ((u64)(u32)hi << 32) | (u64)(u32)lo;
}

// table access: LDS-resident compact entries (base << 16 | freq) or the 16-byte global entries
template <bool LDS_TABLE> struct enc_tab;
template <> struct enc_tab<false> {
    const ansx_enc_entry* t;
    typedef enc_ent ent;
    __device__ __forceinline__ enc_ent get(const ansx_map& mp, u32 x) const
    {
        enc_ent r;
        r.k = map_nbytes(mp, x);
        ansx_enc_entry e = t[map_sym(mp, x, r.k)];
        r.freq = e.freq;
        r.base = e.base;
        r.rcp = e.rcp;
        return r;
    }
    __device__ __forceinline__ void step(enc_lane& L, u32 x, const enc_ent& e, bool active, u32 ql, u32 logM,
        double, u8* __restrict__ out) const
    {
        enc_update(L, x, e, active, ql, logM, out);
    }
    typedef enc_ent entp;
    __device__ __forceinline__ enc_ent getp(const ansx_map& mp, u32 x, double, u32) const { return get(mp, x); }
    __device__ __forceinline__ u32 fetchp(const ansx_map&, u32, u32& l) const
    {
        l = 0;
        return 0;
    }
    __device__ __forceinline__ enc_ent finishp(const ansx_map& mp, u32 x, u32, double, u32) const { return get(mp, x); }
    __device__ __forceinline__ void step_nb(enc_lane& L, u32 x, const enc_ent& e, const enc_quad_const&, u32 ql,
        u32 logM, __amdgpu_buffer_rsrc_t, u8* __restrict__ out) const
    {
        enc_update(L, x, e, true, ql, logM, out);
    }
    __device__ __forceinline__ void init(enc_lane& L, u64 Lb) const { L.st = Lb; }
    __device__ __forceinline__ u64 state(const enc_lane& L) const { return L.st; }
};
template <> struct enc_tab<true> {
    const u32* t;  // LDS (MODE 1) or the block's row of compact entries in HBM (MODE 2)
    // MODE 2: the first `nhot` entries of the row are also held in LDS (hot[nhot] = 0 is a sentinel); the rest
    // is read through a buffer descriptor over the wave's 16 rows, rowoff = this lane's row offset in bytes
    const u32* hot;
    u32 nhot, sent, rowoff;  // sent: index of the zero sentinel
    ansx_u32x4 trs;
    typedef enc_ent_d ent;
    __device__ __forceinline__ enc_ent_d get(const ansx_map& mp, u32 x) const
    {
        enc_ent_d r;
        r.k = map_nbytes(mp, x);
        const u32 e = t[map_sym(mp, x, r.k)];
        r.Fd = (double)(e & 0xFFFFu);
        r.based = (double)(e >> 16);
        // 1/freq: hardware seed + one Newton step (relative error ~2^-50, far below the 2^-37
        // the +-1 correction needs)
        const double r0 = __builtin_amdgcn_rcp(r.Fd);
        r.rcp = __builtin_fma(__builtin_fma(-r.Fd, r0, 1.0), r0, r0);
        r.thr = r.Fd * 68719476736.0;  // 2^36 * freq = K * RADIX * freq (ans_fold.hpp:89)
        return r;
    }
    __device__ __forceinline__ void step(enc_lane& L, u32 x, const enc_ent_d& e, bool active, u32 ql, u32,
        double Md, u8* __restrict__ out) const
    {
        enc_update_d(L, x, e, active, ql, Md, out);
    }
    typedef enc_ent_n entp;
    __device__ __forceinline__ enc_ent_n getp(const ansx_map& mp, u32 x, double Md, u32 ql8) const
    {
        enc_ent_n r;
        r.k = map_nbytes(mp, x);
        const u32 e = t[map_sym(mp, x, r.k)];
        r.Fd = (double)(e & 0xFFFFu);
        r.based = (double)(e >> 16);
        // 1/F, under-estimated on purpose: hardware seed r0 (relative error <= 2^-24.5, measured)
        // and one Newton step against 1 - 2^-39 instead of 1:
        //   r1 = r0 + r0 * (1 - 2^-39 - F*r0) = (1/F) * (1 - d0^2 - 2^-39 (1 + d0)),  d0^2 <= 2^-49
        // so the relative deficit lies in (2^-39.01, 2^-38): s0 * r1 < s0 / F always, and by less
        // than 2^36 * 2^-38 = 1/4 (+ 2^-17 of rounding), i.e. trunc() is q or q - 1.
        const double r0 = __builtin_amdgcn_rcp(r.Fd);
        r.rcp = __builtin_fma(__builtin_fma(-r.Fd, r0, 1.0 - 1.8189894035458565e-12), r0, r0);
        r.thr = r.Fd * 68719476736.0;  // 2^36 * freq = K * RADIX * freq (ans_fold.hpp:89)
        r.MF = Md - r.Fd;
        r.kpos = r.k << ql8;
        return r;
    }
    // MODE 2 of k_encode: alphabets too large for 16 LDS tables per wave.  A skewed list spends most of its
    // steps on the first few hundred symbols: those entries sit in LDS, the others come from HBM / L2
    // through an inline-asm buffer load whose offset is out of range for the hot lanes (no request leaves
    // the CU for them, the load returns 0; with every lane going to L2 the 1024 waves issued ~150 requests per
    // clock, more than its channels take).  The word is hot | cold (one of them is 0); the caller owns the
    // vmcnt wait of the cold part, conversion happens one sub-batch later.
    __device__ __forceinline__ u32 fetchp(const ansx_map& mp, u32 x, u32& l) const
    {
        const u32 k = map_nbytes(mp, x);
        const u32 sym = map_sym(mp, x, k);
        const bool is_hot = sym < nhot;
        // hot rows are u16 RUNNING SUMS (round 4: twice the symbols in the same LDS -- 1151 instead of 576 per block, which
        // halves the lanes that go to L2 on 2300-symbol alphabets): freq = next - current, base = current; the sentinel
        // pair {0, 0} gives the word 0
        typedef __attribute__((address_space(3))) const u16 lds_cu16;
        lds_cu16* cp = (lds_cu16*)hot + (is_hot ? sym : sent);
        const u32 c0 = cp[0], c1 = cp[1];
        l = ((c1 - c0) & 0xFFFFu) | (c0 << 16);
        const u32 voff = is_hot ? ANSX_BUF_OOB : rowoff + 4 * sym;
        u32 e;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(e) : "v"(voff), "s"(trs) : "memory");
        return e;
    }
    __device__ __forceinline__ enc_ent_n finishp(const ansx_map& mp, u32 x, u32 e, double Md, u32 ql8) const
    {
        enc_ent_n r;
        r.k = map_nbytes(mp, x);
        r.Fd = (double)(e & 0xFFFFu);
        r.based = (double)(e >> 16);
        const double r0 = __builtin_amdgcn_rcp(r.Fd);
        r.rcp = __builtin_fma(__builtin_fma(-r.Fd, r0, 1.0 - 1.8189894035458565e-12), r0, r0);  // see getp
        r.thr = r.Fd * 68719476736.0;
        r.MF = Md - r.Fd;
        r.kpos = r.k << ql8;
        return r;
    }
    __device__ __forceinline__ void step_nb(enc_lane& L, u32 x, const enc_ent_n& e, const enc_quad_const& qc,
        u32, u32, __amdgpu_buffer_rsrc_t rsrc, u8* __restrict__) const
    {
        enc_update_n(L, x, e, qc, rsrc);
    }
    __device__ __forceinline__ void init(enc_lane& L, u64 Lb) const { L.sd = (double)Lb; }
    __device__ __forceinline__ u64 state(const enc_lane& L) const { return f64_to_u64_exact(L.sd); }
};


// ---- explicitly scheduled main loop of the LDS-table encoder (MODE 1) ---------------------------------
// One wave per SIMD runs alone (64 Ki chains = 1024 waves), so its own instruction stream is all that fills the
// SIMD.  The step is written out operation by operation -- state chain (8 dependent f64 operations: compare,
// select, ldexp, trunc, mul, trunc, fma with clamp, fma), byte emission, the table-entry arithmetic of the
// NEXT symbol (B) and the fold map + LDS reads of the symbol three steps ahead (A) -- with a scheduling barrier
// after each line, 44 VALU instructions per symbol (58.8 before).  What was measured on the way
// (tests/tools/ubench_valu.hip, ubench_distance.hip, replay_encoder_loop.py, in-kernel s_memtime traces):
//  * a lone wave issues one VALU instruction per ~4.6 cycles; 8.4 / 6.1 / 5.5 if a source was written 1 / 2 / 3
//    instructions earlier;
//  * replayed from registers, this step costs 210 cycles (254 with its LDS reads and stores); inside the kernel
//    it averages ~350: a third of the steps stall 400-1200 cycles on the vector-memory path (4 scattered
//    operations per step and wave, 16 cache lines each, one address unit per CU) -- the order of the VALU work
//    changes nothing measurable, the placement of the loads does (bursts of 8 neighbouring loads: -5 %).
struct encp_a {   // after stage A: exception bytes k, 8k, and the two halves of the table word (LDS reads in flight)
    u32 k, sh, F16, B16;
};
struct encp_e {   // after stage B
    double Fd, rcp, MF, based, omF;  // freq, 1/freq (under-estimate), M - freq, base, 1 - freq
    u32 thr_hi, k, sh;               // high word of 2^36 * freq; exception bytes k; 8k
    u32 off1, off2;                  // buffer-offset deltas of the byte / short store: 0 / k & 1, or out of range
};
#define ANSX_SLOT() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ u32 f64_hi(double d) { return (u32)(__builtin_bit_cast(u64, d) >> 32); }
__device__ __forceinline__ u32 f64_lo(double d) { return (u32)__builtin_bit_cast(u64, d); }
__device__ __forceinline__ u32 ffbh_u32(u32 x)  // leading zeros; 0xFFFFFFFF for 0 (__builtin_clz(0) is undefined)
{
    u32 r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
template <int CTRL> __device__ __forceinline__ u32 quad_add_dpp(u32 v)
{
    // v + v[quad_perm]: the DPP-combine pass folds the move into the add; hazards are the compiler's
    return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

#define ANSX_ENC_HOT 577  // MODE 2: words per LDS row = 1152 u16 running sums (1151 hot symbols) + a zero sentinel pair (odd stride; 16 rows = 36.9 KB per wave)
#define ANSX_ENC_U 8    // table entries per lane fetched ahead
#define ANSX_ENC_XB 32  // inputs per lane fetched ahead (one super-batch)

// MODE 0: 16-byte table entries in HBM, integer state (any frame size)
// MODE 1: compact tables of the wave's 16 blocks in LDS, f64 state, branch-free step (M <= 2^16)
// MODE 2: compact tables in HBM (alphabets too large for LDS), otherwise as MODE 1: the entries of
//         the next sub-batch are fetched with inline-asm loads and consumed behind exact vmcnt
//         waits (24 = the stores of one sub-batch, 32 = the input prefetch of a super-batch)
template <int MODE, bool POW2 = false>
__global__ __launch_bounds__(256) void k_encode(const u32* __restrict__ in, ansx_geo g, u32 NSP,
    const ansx_enc_entry* __restrict__ table, const u32* __restrict__ tab32, u32 lds_stride,
    ansx_blk* __restrict__ blk, u8* __restrict__ scratch, u64 scr_stride,
    u64* __restrict__ ckpt_state, u32* __restrict__ ckpt_off, u32* __restrict__ sizes,
    unsigned long long* __restrict__ gsums, u32 first_block = 0)
{
    extern __shared__ u32 lds_tab[];
    // sizes[b] = bytes of block b's stream, gsums[b / 64] += them: what k_assemble needs to place a block without
    // a scan kernel in between
    auto publish_size = [&](ansx_blk* B_, u32 b_, u32 sz) {
        B_->stream_bytes = sz;
        sizes[b_] = sz;
        if (sz) atomicAdd(&gsums[b_ >> 6], (unsigned long long)sz);
    };
    // A workgroup is 1 to 4 waves, each with its own 16 blocks and tables.  With 1024 single-wave workgroups the
    // dispatcher does not put a CU's four on four different SIMDs, and waves that run the same long loop out of
    // phase compete for the CU's instruction fetch: four waves per workgroup land one per SIMD (0.94 -> 0.79 ms on
    // the headline workload) and a barrier per super-batch keeps them in step (-> 0.70 ms).
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    // first block of this wave (uniform); first_block: the blocks before it belong to another launch (k_encode_pc)
    const u32 wb0 = (u32)__builtin_amdgcn_readfirstlane((int)(first_block + (blockIdx.x * (blockDim.x >> 6) + wv) * 16));
    u32* const wtab = lds_tab + wv * 16 * lds_stride;
    const u32 gt = wb0 * 4 + lane;
    const u32 b = gt >> 2, ql = gt & 3;
    constexpr bool LDS_TABLE = MODE == 1;
    constexpr bool F64 = MODE != 0;
    if (LDS_TABLE || MODE == 2) {
        // the wave stages the compact tables of its 16 blocks (coalesced 4-byte entries): a whole row of
        // lds_stride entries (MODE 1; what lies beyond a block's own alphabet is never looked up) or the first
        // lds_stride - 1 entries plus a zero sentinel (MODE 2).  Two rows per round, every load of a round in
        // flight before the first LDS write (row by row behind a dependent load of the alphabet size this took
        // 90 k cycles, 6 % of the kernel).
        const u32 b0 = wb0;
        if constexpr (MODE == 2) {
            // rows of 2 * lds_stride u16: running sums cum[0 .. H] of the first H = 2 * lds_stride - 3 symbols (cum[0] = 0,
            // cum[s + 1] = base(s) + freq(s) mod 2^16: the compact entries hold a valid base for absent symbols too), then the
            // sentinel pair cum[H + 1] = cum[H + 2] = 0.  One row per round, all its loads in flight together.
            typedef __attribute__((address_space(3))) u16 lds_u16;
            const u32 H = 2 * lds_stride - 3;
            const u32 limh = H < NSP ? H : NSP;
            for (u32 j = 0; j < 16; j++) {
                if (b0 + j >= g.nblocks) break;
                const u32* r0 = tab32 + (u64)(b0 + j) * NSP;
                lds_u16* cum = (lds_u16*)(wtab + j * lds_stride);
                for (u32 e0 = 0; e0 < limh; e0 += 64 * 10) {
                    u32 v0[10];
#pragma unroll
                    for (int i = 0; i < 10; i++) {
                        const u32 e = e0 + lane + 64 * i;
                        v0[i] = e < limh ? r0[e] : 0u;
                    }
#pragma unroll
                    for (int i = 0; i < 10; i++) {
                        const u32 e = e0 + lane + 64 * i;
                        if (e < H) cum[e + 1] = (u16)(e < limh ? (v0[i] >> 16) + (v0[i] & 0xFFFFu) : 0u);
                    }
                }
                if (lane == 0) cum[0] = 0;
                if (lane < 2) cum[H + 1 + lane] = 0;
            }
        } else {
        const u32 take = lds_stride;
        const u32 lim = take < NSP ? take : NSP;
        for (u32 j = 0; j < 16; j += 2) {
            if (b0 + j >= g.nblocks) break;
            const bool two = b0 + j + 1 < g.nblocks;
            const u32* r0 = tab32 + (u64)(b0 + j) * NSP;
            const u32* r1 = r0 + (two ? NSP : 0);
            u32 v0[10], v1[10];
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const u32 e = lane + 64 * i;
                v0[i] = e < lim ? r0[e] : 0u;
                v1[i] = e < lim ? r1[e] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const u32 e = lane + 64 * i;
                if (e < take) {
                    wtab[j * lds_stride + e] = v0[i];
                    if (two) wtab[(j + 1) * lds_stride + e] = v1[i];
                }
            }
        }
        }
        __syncthreads();
    }
    if (b >= g.nblocks) return;
    ansx_blk* B = &blk[b];
    if (B->status || !B->resolved || B->pa_sigma == 1) {
        if (ql == 0) publish_size(B, b, B->pa_sigma == 1 && !B->status ? B->pre_bytes : 0u);
        return;
    }
    const u32 nb = geo_block_n(g, b);
    const u32* src = in + (u64)b * g.block_ints;
    enc_tab<F64> tab;
    if constexpr (MODE == 1) tab.t = wtab + (lane >> 2) * lds_stride;
    else if constexpr (MODE == 2) {
        tab.t = tab32 + (u64)b * NSP;
        tab.hot = wtab + (lane >> 2) * lds_stride;
        const u32 own = B->max_sym + 1;
        const u32 H = 2 * lds_stride - 3;
        tab.nhot = own < H ? own : H;  // symbols of THIS block whose running sums were staged
        tab.sent = H + 1;              // u16 index of the zero sentinel pair
        tab.rowoff = (lane >> 2) * NSP * 4;
        const u64 ba = (u64)(uintptr_t)(tab32 + (u64)wb0 * NSP);
        tab.trs = ansx_u32x4{ (u32)ba, (u32)(ba >> 32) & 0xFFFFu, 16u * NSP * 4u, 0x00020000u };
    } else tab.t = table + (u64)b * NSP;
    u8* out = scratch + (u64)b * scr_stride;
    // buffer view of the wave's 16 scratch slots (the host guarantees 16 * scr_stride < 2^31 for
    // the LDS variant): wave-uniform descriptor, per-lane 32-bit offsets
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        scratch + (u64)wb0 * scr_stride, 0, (int)(16 * scr_stride), 0x00020000);
    const u32 obase = (u32)((lane >> 2) * scr_stride);
    const ansx_map f = g.map;  // value -> symbol map
    const u32 logM = B->logM;
    const u64 Lb = (u64)16 << logM;
    enc_lane L;
    tab.init(L, Lb);
    L.p = B->prelude_bytes;
    const double Md = (double)(1u << logM);
    const u32 r = nb & 3;
    // tail symbols all go to state 0 (ans_fold.hpp:257-261)
    for (u32 t = 0; t < r; t++) {
        u32 x = src[nb - 1 - t];
        tab.step(L, x, tab.get(f, x), ql == 0, ql, logM, Md, out);
    }
    // groups of four, backwards (ans_fold.hpp:262-272): in[4g+3-q] -> state q
    const u32 G = nb >> 2;
    const u32 cg = g.ckpt >> 2;  // groups per restart interval (0 = none)
    u32 ck_seg = (cg && G) ? (G - 1) / cg : 0;  // next restart point to record: segment index
    u32 ck_g = ck_seg * cg;                     // ... and its group index
    u32 pbias = 0;  // inside the pipelined loop of the LDS variant L.p is a buffer offset (+ obase)
    auto record = [&](u32 gidx) {
        // every symbol with index >= 4*gidx is now encoded: decoder restart point of segment
        // ck_seg (the decoder of that segment starts with exactly these states and cursor)
        if (ck_seg && gidx == ck_g) {
            const u64 idx = (u64)b * g.nckf + (ck_seg - 1);
            const u64 stv = tab.state(L);
            if (g.ckw) {
                ckpt_state[idx * 4 + ql] = stv;
                if (ql == 0) ckpt_off[idx] = L.p - pbias;
            } else {
                // packed record (ansx_dev.h): the even lane of a pair stores both states as one 104-bit integer.  (A frame
                // above 2^16 would not fit: the host repeats such a call with wide restart points.)
                u8* rec = (u8*)ckpt_state + idx * ANSX_CK_RECORD;
                const u64 pst = ((u64)quad_perm<ANSX_QP(1, 0, 3, 2)>((u32)(stv >> 32)) << 32) | quad_perm<ANSX_QP(1, 0, 3, 2)>((u32)stv);
                if ((ql & 1u) == 0) {
                    u8* pp = rec + 13u * (ql >> 1);
                    st_u64_unaligned(pp, (stv & ((1ull << ANSX_CK_STATE_BITS) - 1ull)) | (pst << ANSX_CK_STATE_BITS));
                    st_u32_unaligned(pp + 8, (u32)(pst >> 12));
                    pp[12] = (u8)(pst >> 44);
                }
                if (ql == 0) {
                    const u32 cur = L.p - pbias;
                    st_u16_unaligned(rec + 26, (u16)cur);
                    rec[28] = (u8)(cur >> 16);
                }
            }
            ck_seg--;
            ck_g -= cg;
        }
    };
    u32 gi = G;  // groups [0, gi) remain
    // leading remainder so that the pipelined part covers a multiple of XB groups
    for (u32 t = G % ANSX_ENC_XB; t > 0; t--) {
        const u32 gidx = --gi;
        u32 x = src[4 * gidx + 3 - ql];
        tab.step(L, x, tab.get(f, x), true, ql, logM, Md, out);
        record(gidx);
    }
    // Software pipeline.  Inputs are fetched XB groups (one "super-batch") ahead into registers;
    // table entries (LDS) U groups ahead.  On CDNA4 stores share the in-order vmcnt with loads
    // and the emitted-byte stores are predicated, so consuming a prefetched input costs a full
    // drain of the store queue: doing that once per XB = 32 steps instead of once per U = 8 is
    // what the deep input prefetch buys.  The per-symbol dependency chain
    // (renorm test -> divide -> state) never waits on memory.
    if (gi) {
        // restart points fall on sub-batch boundaries when the interval is a multiple of U groups
        const bool ck_per_batch = (cg % ANSX_ENC_U) == 0;
        const u32* base = src + 3 - ql;
        const enc_quad_const qc = { 4u << (8 * ql), (1u << (8 * ql)) - 1u };
        if constexpr (F64) {
            pbias = obase;
            L.p += obase;
        }
        bool piped = false;
        if constexpr (MODE == 1) {
            if (ck_per_batch) {
                piped = true;
                // Inputs come through a buffer view of the wave's 16 blocks, one load per step straight into the
                // registers a sub-batch of 8 steps just released: they are first read 21 steps (>= 84 vector-memory
                // operations) later.  vmcnt counts at most
                // 63 operations in flight, so anything with more than 63 younger operations has completed: no
                // wait is needed (and none could name it).  A lane whose block has no further group computes a
                // negative offset, i.e. one beyond num_records: the load returns 0, whose table lookups are harmless.
                const u64 iba = (u64)(uintptr_t)(in + (u64)wb0 * g.block_ints);
                const u64 irem = g.n - (u64)wb0 * g.block_ints;
                const u32 inrec = (u32)((irem < 16ull * g.block_ints ? irem : 16ull * g.block_ints) * 4);
                const ansx_u32x4 irs = ansx_u32x4{ (u32)__builtin_amdgcn_readfirstlane((u32)iba),
                    (u32)__builtin_amdgcn_readfirstlane((u32)(iba >> 32) & 0xFFFFu), (u32)__builtin_amdgcn_readfirstlane(inrec), 0x00020000u };
                // byte offset of group (gi - 32), this lane's state, in that view
                u32 vcur = (lane >> 2) * g.block_ints * 4 + 4 * (3 - ql) + 16 * (gi - ANSX_ENC_XB);
                u32 xa[ANSX_ENC_XB];
#define ANSX_XLOAD(dst, voff, j) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(irs), "s"(16 * (ANSX_ENC_XB - 1 - (j))) : "memory")
#pragma unroll
                for (int j = 0; j < ANSX_ENC_XB; j++) ANSX_XLOAD(xa[j], vcur, j);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < ANSX_ENC_XB; i += 8)
                    asm volatile("" : "+v"(xa[i]), "+v"(xa[i + 1]), "+v"(xa[i + 2]), "+v"(xa[i + 3]), "+v"(xa[i + 4]),
                                 "+v"(xa[i + 5]), "+v"(xa[i + 6]), "+v"(xa[i + 7]));
                // the quad's table as 16-bit halves (freq, base), read with two ds_read_u16: no unpacking on the
                // VALU.  Issued as inline asm: the halves are consumed two steps later behind ONE hand-counted wait.
                const u32 tbase = (u32)(uintptr_t)(__attribute__((address_space(3))) const u32*)wtab + 4 * (lane >> 2) * lds_stride;
                const u32 c32f = 8u + (u32)__builtin_clz(f.t1);  // POW2 maps: t1 = 2^(f+7); 32 - f
                // stage A: value -> (k, symbol), the two halves of its table word requested from LDS
                auto stage_a = [&](u32 x) {
                    encp_a a;
                    if constexpr (POW2) {
                        const u32 d = __builtin_elementwise_sub_sat(c32f, ffbh_u32(x));
                        a.k = d >> 3;
                        a.sh = d & 0x18u;
                    } else {
                        a.k = map_nbytes(f, x);
                        a.sh = a.k << 3;
                    }
                    const u32 la = tbase + 4 * (__umul24(a.k, f.D) + (x >> a.sh));
                    asm volatile("ds_read_u16 %0, %1" : "=v"(a.F16) : "v"(la) : "memory");
                    asm volatile("ds_read_u16 %0, %1 offset:2" : "=v"(a.B16) : "v"(la) : "memory");
                    return a;
                };
                // stage B: the arithmetic form of a table entry (see enc_tab<true>::getp for the reciprocal)
                auto stage_b = [&](const encp_a& a) {
                    encp_e e;
                    e.Fd = (double)a.F16;
                    e.based = (double)a.B16;
                    const double r0 = __builtin_amdgcn_rcp(e.Fd);
                    e.rcp = __builtin_fma(__builtin_fma(-e.Fd, r0, 1.0 - 1.8189894035458565e-12), r0, r0);
                    e.thr_hi = f64_hi(e.Fd) + (36u << 20);  // 2^36 * freq: same mantissa, low word 0
                    e.MF = Md - e.Fd;
                    e.omF = 1.0 - e.Fd;
                    e.k = a.k;
                    e.sh = a.sh;
                    e.off1 = (a.k << 31) + ANSX_BUF_OOB;
                    e.off2 = (a.k - 2u) & 0x80000001u;
                    return e;
                };
                encp_a a0 = stage_a(xa[0]);
                encp_a an = stage_a(xa[1]);   // consumed by stage B one step ahead of its symbol ...
                encp_a an2 = stage_a(xa[2]);  // ... two steps after its LDS reads were issued
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0.F16), "+v"(a0.B16) : : "memory");
                encp_e ec = stage_b(a0);
                double sd = L.sd;
                u32 pcur = L.p;
                // every block of the workgroup is a full one: the same number of super-batches for all its waves
                const bool wg_sync = blockDim.x > 64 && ((u64)first_block + (u64)(blockIdx.x + 1) * (blockDim.x >> 2)) * g.block_ints <= g.n;
                while (gi) {
                    if (wg_sync) __syncthreads();
                    const u32 top = gi;
                    const u32 vnext = vcur - 16 * ANSX_ENC_XB;
#pragma unroll
                    for (int j = 0; j < ANSX_ENC_XB; j++) {
                        if (j % ANSX_ENC_U == 0) {
                            // the eight registers the previous sub-batch released take the inputs of their steps in
                            // the next super-batch (the last eight of THIS one when requested at step 0): eight
                            // neighbouring loads share their cache lines
#pragma unroll
                            for (int i = 0; i < ANSX_ENC_U; i++) {
                                const int e = (j + ANSX_ENC_XB - ANSX_ENC_U + i) % ANSX_ENC_XB;
                                if (j == 0) ANSX_XLOAD(xa[e], vcur, e);
                                else ANSX_XLOAD(xa[e], vnext, e);
                            }
                        }
                        const u32 x = xa[j];
                        const u32 x2 = xa[(j + 3) % ANSX_ENC_XB];
                        // One operation per line, a scheduling barrier after each: the order written is the order
                        // issued.  A lone wave issues an instruction every ~4.6 cycles if none of its sources was
                        // written by one of the THREE instructions before it; otherwise it loses a slot
                        // (tests/tools/ubench_distance.hip: 8.4 / 6.1 / 5.5 / 4.6 cycles per operation at dependency
                        // distance 1 / 2 / 3 / 4).  Four strands are therefore rotated, one operation each per
                        // row, so that consecutive operations of a strand are four slots apart:
                        //   [C] the state chain of this symbol      [E] its byte emission
                        //   [B] the table entry of the next symbol  [A] map + LDS reads of the symbol 3 steps ahead
                        encp_a a2;
                        encp_e en;
                        const u32 k = ec.k;
#define Q ANSX_SLOT();
                        // row 0.  The halves of `an` were requested two steps ago; the only younger LDS
                        // operations are the two of `an2`.
                        Q asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(an.F16), "+v"(an.B16) : : "memory");
                        Q const bool rn = f64_hi(sd) >= ec.thr_hi;                                        // C1 renormalise?
                        Q const u32 w = f64_lo(sd + 4503599627370496.0);                                   // E  low word of the state (< 2^52)
                        Q en.Fd = (double)an.F16;                                                          // B
                        Q u32 lz2 = 0, d2 = 0;
                        if constexpr (POW2) lz2 = ffbh_u32(x2);                                            // A  (-1 for 0)
                        else a2.k = map_nbytes(f, x2);
                        // row 1
                        Q int ex = rn ? -32 : 0;                                                           // C2
                        asm("" : "+v"(ex));  // (keeps ldexp(sd, select) from becoming select(ldexp, sd): 3 operations)
                        Q const u32 fp = rn ? qc.four_pos : 0u;                                            // E
                        Q const double r0 = __builtin_amdgcn_rcp(en.Fd);                                   // B
                        Q if constexpr (POW2) d2 = __builtin_elementwise_sub_sat(c32f, lz2);               // A  8k + (0..7)
                        // row 2
                        Q const double t = __builtin_ldexp(sd, ex);                                        // C3
                        Q const u32 v = (k << (8 * ql)) + fp;                                              // E  packed byte counts
                        Q en.based = (double)an.B16;                                                       // B
                        Q if constexpr (POW2) a2.k = d2 >> 3;                                              // A
                        // row 3
                        Q const double s0 = __builtin_trunc(t);                                            // C4 state after renormalisation
                        Q const u32 s1 = quad_add_dpp<0xB1>(v);                                            // E
                        Q const double nt = __builtin_fma(-en.Fd, r0, 1.0 - 1.8189894035458565e-12);      // B
                        Q if constexpr (POW2) a2.sh = d2 & 0x18u;                                          // A
                        else a2.sh = a2.k << 3;
                        // row 4
                        Q double q = s0 * ec.rcp;                                                          // C5
                        Q const u32 S = quad_add_dpp<0x4E>(s1);                                            // E  the quad's counts, a byte per lane
                        Q const double sb = s0 + ec.based;                                                 // C
                        Q const u32 xs2 = x2 >> a2.sh;                                                     // A
                        // row 5
                        Q q = __builtin_trunc(q);                                                          // C6 quotient or quotient - 1
                        Q const double u = s0 + ec.omF;                                                    // C
                        Q en.rcp = __builtin_fma(nt, r0, r0);                                              // B
                        Q const u32 sym2 = __umul24(a2.k, f.D) + xs2;                                      // A
                        // row 6
                        Q const double base2 = __builtin_fma(q, ec.MF, sb);                                // C
                        Q const u32 m = S & qc.lomask;                                                     // E
                        Q const u32 pnext = __builtin_amdgcn_sad_u8(S, 0u, pcur);                          // E  cursor after this step
                        Q const u32 la2 = tbase + 4 * sym2;                                                // A
                        // row 7
                        Q const double one_short = __builtin_fmin(__builtin_fmax(__builtin_fma(-q, ec.Fd, u), 0.0), 1.0);  // C7
                        Q const u32 a = __builtin_amdgcn_sad_u8(m, 0u, pcur);                              // E  this lane's bytes start here
                        Q en.thr_hi = f64_hi(en.Fd) + (36u << 20);                                         // B
                        Q const u32 t8 = ec.sh & 8u;                                                       // E  8 (k & 1)
                        // row 8
                        Q sd = __builtin_fma(one_short, ec.MF, base2);                                     // C8 new state
                        Q en.MF = Md - en.Fd;                                                              // B
                        Q en.k = an.k;
                        en.sh = an.sh;
                        const u32 km2 = an.k - 2u;                                                         // B
                        Q asm volatile("ds_read_u16 %0, %1" : "=v"(a2.F16) : "v"(la2) : "memory");        // A
                        // row 9
                        Q const u32 addr1 = a + ec.off1;                                                   // E
                        Q const u32 addr2 = a + ec.off2;                                                   // E
                        Q en.omF = 1.0 - en.Fd;                                                            // B
                        Q asm volatile("ds_read_u16 %0, %1 offset:2" : "=v"(a2.B16) : "v"(la2) : "memory");  // A
                        // row 10
                        Q const u32 ak = a + k;                                                            // E
                        Q const u32 xv = x >> t8;                                                          // E
                        Q en.off1 = (an.k << 31) + ANSX_BUF_OOB;                                           // B  0 when k is odd
                        Q en.off2 = km2 & 0x80000001u;                                                     // B  k & 1 when k >= 2
                        // row 11
                        Q __builtin_amdgcn_raw_buffer_store_b8((u8)x, rsrc, addr1, 0, 0);                  // E  k odd: byte 0
                        Q const u32 addr3 = rn ? ak : ANSX_BUF_OOB;                                        // E
                        Q pcur = pnext;
                        Q __builtin_amdgcn_raw_buffer_store_b16((u16)xv, rsrc, addr2, 0, 0);               // E  k >= 2: the two high bytes
                        // row 12
                        Q __builtin_amdgcn_raw_buffer_store_b32(w, rsrc, addr3, 0, 0);                     // E  renormalisation word
                        Q;
#undef Q
                        ec = en;
                        an = an2;
                        an2 = a2;
                        if (j % ANSX_ENC_U == ANSX_ENC_U - 1) {
                            L.sd = sd;
                            L.p = pcur;
                            record(top - (u32)(j + 1));
                        }
                    }
                    gi = top - ANSX_ENC_XB;
                    vcur = vnext;
                }
                // nothing may still be on its way into a register when the loop is left
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < ANSX_ENC_XB; i += 8)
                    asm volatile("" : "+v"(xa[i]), "+v"(xa[i + 1]), "+v"(xa[i + 2]), "+v"(xa[i + 3]), "+v"(xa[i + 4]),
                                 "+v"(xa[i + 5]), "+v"(xa[i + 6]), "+v"(xa[i + 7]));
                asm volatile("" : "+v"(an.F16), "+v"(an.B16), "+v"(an2.F16), "+v"(an2.B16));
#undef ANSX_XLOAD
                L.sd = sd;
                L.p = pcur;
            }
        }
        if (!piped) {
        u32 xa[ANSX_ENC_XB], xb[ANSX_ENC_XB];
#pragma unroll
        for (int j = 0; j < ANSX_ENC_XB; j++) xa[j] = base[4 * (gi - 1 - j)];
        u32 raw1[ANSX_ENC_U] = {}, raw1l[ANSX_ENC_U] = {};
        if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < ANSX_ENC_U; j++) raw1[j] = tab.fetchp(f, xa[j], raw1l[j]);
        }
        // (waves of one workgroup stay in step, see the scheduled loop above)
        const bool wg_sync2 = blockDim.x > 64 && ((u64)first_block + (u64)(blockIdx.x + 1) * (blockDim.x >> 2)) * g.block_ints <= g.n;
        while (gi) {
            if (wg_sync2) __syncthreads();
            const u32 top = gi;  // this super-batch encodes groups top-1 ... top-XB
            const bool any_next = __builtin_amdgcn_ballot_w64(top >= 2 * ANSX_ENC_XB) != 0;  // wave uniform
            if (top >= 2 * ANSX_ENC_XB) {
                // Issued as inline asm so that hipcc's waitcnt pass does not see these loads: it
                // would put s_waitcnt vmcnt(31..0) in front of their first use, and because vmcnt
                // is in-order and shared with stores that is a full drain of the 96 stores of this
                // super-batch (measured: 30 % of the wave's cycles in s_waitcnt).  The loads are
                // consumed a whole super-batch later, behind those 96 stores; vmcnt is a 6-bit
                // counter, so anything older than 63 operations has completed — enc_xb_ready()
                // below states that explicitly with a wait that never touches recent stores.
#pragma unroll
                for (int j = 0; j < ANSX_ENC_XB; j++)
                    asm volatile("global_load_dword %0, %1, off"
                                 : "=v"(xb[j])
                                 : "v"(base + 4 * (top - ANSX_ENC_XB - 1 - j))
                                 : "memory");
            }
            typename enc_tab<F64>::entp e1[ANSX_ENC_U];
            if constexpr (MODE != 2) {
#pragma unroll
                for (int j = 0; j < ANSX_ENC_U; j++) e1[j] = tab.getp(f, xa[j], Md, 8 * ql);
            }
#pragma unroll
            for (int sb = 0; sb < ANSX_ENC_XB / ANSX_ENC_U; sb++) {
                typename enc_tab<F64>::entp e0[ANSX_ENC_U];
                if constexpr (MODE == 2) {
                    // raw1 = table words of this sub-batch, requested one sub-batch ago (sb >= 1: the
                    // 24 stores of that sub-batch are the only younger operations) or at the end of
                    // the previous super-batch (sb == 0: only the input prefetch above is younger)
                    if (sb == 0) {
                        if (any_next) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    }
                    asm volatile("" : "+v"(raw1[0]), "+v"(raw1[1]), "+v"(raw1[2]), "+v"(raw1[3]), "+v"(raw1[4]),
                                 "+v"(raw1[5]), "+v"(raw1[6]), "+v"(raw1[7]));
#pragma unroll
                    for (int j = 0; j < ANSX_ENC_U; j++)
                        e0[j] = tab.finishp(f, xa[sb * ANSX_ENC_U + j], raw1[j] | raw1l[j], Md, 8 * ql);
                    if (sb + 1 < ANSX_ENC_XB / ANSX_ENC_U) {
#pragma unroll
                        for (int j = 0; j < ANSX_ENC_U; j++) raw1[j] = tab.fetchp(f, xa[(sb + 1) * ANSX_ENC_U + j], raw1l[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < ANSX_ENC_U; j++) e0[j] = e1[j];
                    if (sb + 1 < ANSX_ENC_XB / ANSX_ENC_U) {
#pragma unroll
                        for (int j = 0; j < ANSX_ENC_U; j++)
                            e1[j] = tab.getp(f, xa[(sb + 1) * ANSX_ENC_U + j], Md, 8 * ql);
                    }
                }
                const u32 sbtop = top - sb * ANSX_ENC_U;  // groups sbtop-1 ... sbtop-U
                if (ck_per_batch) {
#pragma unroll
                    for (int j = 0; j < ANSX_ENC_U; j++)
                        tab.step_nb(L, xa[sb * ANSX_ENC_U + j], e0[j], qc, ql, logM, rsrc, out);
                    record(sbtop - ANSX_ENC_U);
                } else {
#pragma unroll
                    for (int j = 0; j < ANSX_ENC_U; j++) {
                        tab.step_nb(L, xa[sb * ANSX_ENC_U + j], e0[j], qc, ql, logM, rsrc, out);
                        record(sbtop - 1 - j);
                    }
                }
            }
            gi = top - ANSX_ENC_XB;
            // >= 96 buffer stores were issued after the xb loads: waiting until at most 40 VMEM
            // operations are outstanding proves the loads done (in-order counter) and only
            // involves stores that are >= 13 steps old.  The "+v" operands keep every use of xb
            // behind the wait.
            asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
#pragma unroll
            for (int j = 0; j < ANSX_ENC_XB; j += 8)
                asm volatile(""
                             : "+v"(xb[j]), "+v"(xb[j + 1]), "+v"(xb[j + 2]), "+v"(xb[j + 3]), "+v"(xb[j + 4]),
                             "+v"(xb[j + 5]), "+v"(xb[j + 6]), "+v"(xb[j + 7]));
#pragma unroll
            for (int j = 0; j < ANSX_ENC_XB; j++) xa[j] = xb[j];
            if constexpr (MODE == 2) {
                if (gi) {
#pragma unroll
                    for (int j = 0; j < ANSX_ENC_U; j++) raw1[j] = tab.fetchp(f, xa[j], raw1l[j]);
                }
            }
        }
        }  // !piped
        L.p -= pbias;
        pbias = 0;
    }
    // flush state - L, order 0,1,2,3 (ans_fold.hpp:275-278,115-120)
    st_u64_unaligned(out + L.p + 8 * ql, tab.state(L) - Lb);
    if (ql == 0) publish_size(B, b, L.p + 32);
}

// ---- k_encode_pc: the LDS-table encoder as a producer / consumer pair of waves per 16 blocks --------------------
// k_encode<1> runs ONE wave per SIMD (4 chains per block is all the parallelism the format gives) and that wave issues
// an instruction every ~7 cycles: it is a lone in-order stream with its own LDS / VMEM waits, and a SIMD gives a lone wave
// one vector instruction per ~5.1 cycles at best (tests/tools/ubench_valu2.hip: two waves share a SIMD at 4.3 cycles per
// f64 / VOP3 instruction and 2.4 per 4-byte-encoded 32-bit one).  Of its 44 vector instructions per symbol only the state
// chain and the byte emission depend on the state; the fold map, the table look-up and the reciprocal depend on the input
// alone.  Here a second wave on the same SIMD (the PRODUCER, waves 4..7 of a 512-thread workgroup) does that part a batch
// of S steps ahead and hands every (lane, step) 16 bytes through LDS:
//     word 0, 1  1/F as an f64, under-estimated as in enc_tab<true>::getp, its two lowest mantissa bits replaced by k
//                (the number of exception bytes; 3 * 2^-52 relative, far inside the (2^-39.01, 2^-38) deficit window)
//     word 2     F | base << 16
//     word 3     the exception bytes laid out for the stores: byte 0 = x & 0xFF, bits 16..31 = (x >> 8 (k & 1)) & 0xFFFF
// and the CONSUMER (waves 0..3, same lanes = same chains) runs the state chain and the emission: 35 vector instructions
// per symbol instead of 44, no LDS gathers, no input loads.  The tables are u16 RUNNING SUMS, one per symbol plus the
// frame size (freq = next - current; two adjacent ds_read_u16 as before): half the LDS of the 4-byte entries, which is
// what makes room for the hand-over buffers (2 x S x 1 KB per pair).  One workgroup barrier per batch; double-buffered.
// Host-checked: block_ints % 128 == 0 and <= 2^22, restart interval a multiple of 4 S.  The call's last workgroup may hold
// blocks that do not exist and the partial last block: see "irregular" in the kernel.
#define ANSX_PC_S 8  // (steps per batch of the default form; the kernel is a template over it)
__device__ __forceinline__ void quad_transpose4(u32 (&v)[4], u32 ql);  // (defined with the decoder)
__device__ __forceinline__ void pc_store_short_hi(u32 v, ansx_u32x4 rs, u32 voff)
{
    asm volatile("buffer_store_short_d16_hi %0, %1, %2, 0 offen" : : "v"(v), "v"(voff), "s"(rs) : "memory");
}
// Launch shapes (blockDim.x = 128 * pairs; wave w < pairs is the consumer of pair w, wave pairs + w its producer):
//   4 pairs, S = 8   one workgroup per CU, a producer and a consumer on every SIMD: as fast as k_encode<1>, no faster
//                    (the pair's 53 vector instructions per symbol fill the SIMD as the lone wave's 44 do) -- opt-in;
//   2 pairs, S = 4   alphabets of up to ~2400 symbols (BASELINE config 3: 2296) with EVERY table entry in LDS: 32 blocks per
//                    CU instead of 64, so the call takes two rounds, but every wave has a SIMD to itself and the consumer's
//                    35 instructions per symbol run at the lone-wave rate -- faster than the compact-table mode (MODE 2 of
//                    k_encode), whose cold look-ups were 4-byte gathers into a 150 MB working set;
//   1 pair,  S = 8   short lists (at most two such workgroups per CU): the state chain of a lone wave is what such a call
//                    waits for, and the consumer's is 20 % shorter.
template <bool POW2, int S>
__global__ __launch_bounds__(512) void k_encode_pc(const u32* __restrict__ in, ansx_geo g, u32 NSP,
    const u32* __restrict__ tab32, u32 ns_cap, u32 rowwords, ansx_blk* __restrict__ blk, u8* __restrict__ scratch,
    u64 scr_stride, u64* __restrict__ ckpt_state, u32* __restrict__ ckpt_off, u32* __restrict__ sizes,
    unsigned long long* __restrict__ gsums)
{
    static_assert(S == 4 || S == 8, "the input ring is 32 steps deep: 4 or 8 batches per super-batch");
    extern __shared__ u32 lds_pc[];
    typedef __attribute__((address_space(3))) u16 lds_u16;
    typedef __attribute__((address_space(3))) ansx_u32x4 lds_x4;
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 pairs = blockDim.x >> 7;
    const bool producer = wv >= pairs;
    const u32 pair = producer ? wv - pairs : wv;
    const u32 wb0 = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * pairs + pair) * 16));  // first block of this pair (uniform)
    u32* const ptab = lds_pc + pair * 16 * rowwords;  // the pair's 16 rows of running sums (u16), rowwords words each
    constexpr u32 HANDW = 2 * S * 64 * 4 + (S == 8 ? 16 * 36 : 0);  // words per pair: two hand-over buffers (+ the producer's input strips)
    u32* const hand = lds_pc + pairs * 16 * rowwords + pair * HANDW;
    {
        // Both waves of a pair stage its tables, eight rows each, a row at a time with all loads of a 640-entry piece in
        // flight: cum[0] = 0, cum[s + 1] = base(s) + freq(s) mod 2^16 (the compact entries hold a valid base for absent symbols
        // too; what lies beyond a block's own alphabet is never looked up)
        const u32 r0 = producer ? 8u : 0u;
        const u32 lim = ns_cap < NSP ? ns_cap : NSP;
        for (u32 j = 0; j < 8; j++) {
            if (wb0 + r0 + j >= g.nblocks) break;  // (the list ends inside this workgroup: no such block, its lanes only idle along)
            const u32* row0 = tab32 + (u64)(wb0 + r0 + j) * NSP;
            lds_u16* c0 = (lds_u16*)(ptab + (r0 + j) * rowwords);
            for (u32 e0 = 0; e0 < lim; e0 += 640) {
                u32 v0[10];
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    const u32 e = e0 + lane + 64 * i;
                    v0[i] = e < lim ? row0[e] : 0u;
                }
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    const u32 e = e0 + lane + 64 * i;
                    if (e < lim) c0[e + 1] = (u16)((v0[i] >> 16) + (v0[i] & 0xFFFFu));
                }
            }
            if (lane == 0) c0[0] = 0;
        }
    }
    __syncthreads();
    const u32 b = wb0 + (lane >> 2), ql = lane & 3u;
    const u32 G = g.block_ints >> 2;   // groups of four per block
    // A workgroup whose 16 x pairs blocks are not all full blocks of the list (the list's last workgroup: blocks that do not exist, the
    // partial last block) is IRREGULAR: it runs 32 extra steps in front, and its producers hand over NEUTRAL entries -- frequency M,
    // base 0, no exception bytes: state and cursor stay as they are (x / M * M + x % M = x; 16 M < 2^36 M: no renormalisation) --
    // wherever a lane has no symbol at a step.  A block of nbk ints has its nbk / 4 groups in the LAST steps, the nbk % 4 tail
    // symbols (all on state 0, ans_fold.hpp:257-261) in the steps before them on lane 0, neutral entries before that.
    const bool irr = (u64)(blockIdx.x + 1) * 16u * pairs * g.block_ints > g.n;  // (workgroup-uniform)
    const u32 Gtot = G + (irr ? (u32)ANSX_ENC_XB : 0u);  // (a whole super-batch of the producer's input ring: its loops keep their shape)
    const u32 NBATCH = Gtot / (u32)S;
    const bool exists = b < g.nblocks;
    const u32 nbk = exists ? geo_block_n(g, b) : 0u;
    const u32 Gp = nbk >> 2, rt = nbk & 3u;  // this block's groups and tail symbols
    if (producer) {
        // ---------------------------------------------------------------- producer
        const ansx_map f = g.map;
        const u64 iba = (u64)(uintptr_t)(in + (u64)wb0 * g.block_ints);
        const u64 ifirst = (u64)wb0 * g.block_ints;
        const u64 irem = ifirst < g.n ? g.n - ifirst : 0;  // ints of the list from this pair's first block on
        const u32 inrec = (u32)((irem < 16ull * g.block_ints ? irem : 16ull * g.block_ints) * 4);
        const ansx_u32x4 irs = ansx_u32x4{ (u32)__builtin_amdgcn_readfirstlane((u32)iba),
            (u32)__builtin_amdgcn_readfirstlane((u32)(iba >> 32) & 0xFFFFu), (u32)__builtin_amdgcn_readfirstlane(inrec), 0x00020000u };
        const u32 tbase = (u32)(uintptr_t)(__attribute__((address_space(3))) const u32*)ptab + 4 * (lane >> 2) * rowwords;
        const u32 hbase = (u32)(uintptr_t)(__attribute__((address_space(3))) const u32*)hand + 16 * lane;
        // irregular workgroups: the neutral entry of this lane's block.  A frame of 2^16 does not fit the 16-bit frequency field:
        // its stand-in has frequency 65535 -- no bytes, no renormalisation either, but the state creeps up by a factor
        // 1 + 2^-16 per step, so the consumer sets the state back to its initial value at the lane's first live step.
        const u32 lgm = (exists && irr) ? blk[b].logM : 8u;
        ansx_u32x4 neutral = ansx_u32x4{ 0u, 0u, 0u, 0u };
        if (irr) {
            const u32 Mn = lgm <= 15 ? (1u << lgm) : 65535u;
            const double Md_ = (double)Mn;
            const double r0 = __builtin_amdgcn_rcp(Md_);
            const double rc = __builtin_fma(__builtin_fma(-Md_, r0, 1.0 - 1.8189894035458565e-12), r0, r0);
            neutral = ansx_u32x4{ f64_lo(rc) & ~3u, f64_hi(rc), Mn, 0u };
        }
        const u32* const tailp = in + (u64)(exists ? b : 0) * g.block_ints + 3u * Gp;  // tail symbol at step gi: tailp[gi]
        // x and the entry of (this lane, the step of group index gi) in an irregular workgroup
        auto fix_x = [&](u32 x, u32 gi) -> u32 {
            const bool real = gi < Gp, tail = !real && ql == 0 && gi < Gp + rt;
            u32 tv = 0;
            if (__builtin_amdgcn_ballot_w64(tail) != 0) tv = tail ? tailp[gi] : 0u;
            return real ? x : tv;
        };
        auto fix_h = [&](const ansx_u32x4& h, u32 gi) -> ansx_u32x4 {
            const bool live = gi < Gp || (ql == 0 && gi < Gp + rt);
            return live ? h : neutral;
        };
        const u32 c32f = 8u + (u32)__builtin_clz(f.t1);  // POW2 maps: t1 = 2^(f+7); 32 - f
        // one symbol's hand-over entry from its value and the two running sums around its symbol
        auto entry = [&](u32 x, u32 k, u32 sh, u32 cur, u32 nxt) {
            const u32 F = (nxt - cur) & 0xFFFFu;
            const double Fd = (double)F;
            // 1/F, under-estimated on purpose (enc_tab<true>::getp)
            const double r0 = __builtin_amdgcn_rcp(Fd);
            const double rc = __builtin_fma(__builtin_fma(-Fd, r0, 1.0 - 1.8189894035458565e-12), r0, r0);
            ansx_u32x4 h;
            h.x = (f64_lo(rc) & ~3u) | k;
            h.y = f64_hi(rc);
            h.z = F | (cur << 16);
            const u32 xs = x >> (sh & 8u);
            h.w = (x & 0xFFu) | (xs << 16);
            return h;
        };
        auto fold_of = [&](u32 x, u32& k, u32& sh) {
            if constexpr (POW2) {
                const u32 d = __builtin_elementwise_sub_sat(c32f, ffbh_u32(x));
                k = d >> 3;
                sh = d & 0x18u;
            } else {
                k = map_nbytes(f, x);
                sh = k << 3;
            }
            return tbase + 2 * (__umul24(k, f.D) + (x >> sh));
        };
        u32 pb = 0;  // batch being produced
        const u32 rtl = ql == 0 ? rt : 0u;  // tail steps of this lane (state 0 takes them all)
        // the S hand-over entries of batch pb (groups gtop - 1 .. gtop - S) from this lane's inputs.  Irregular workgroups: a lane's
        // batches are all neutral, then one or two boundary batches, then all live.  Neutral lanes store their entries in the first
        // batch of either buffer (every lane is neutral in the 32 extra steps) and idle afterwards -- masked off, they cost no issue
        // slot --, boundary batches (at most two per lane, and only the partial block has any that are not at a batch boundary)
        // take the per-step form for the whole wave.
        auto emit = [&](u32 (&x)[S], u32 gtop, u32 hb) {
            bool act = true;
            if (irr) {
                const bool real = gtop <= Gp, neut = gtop - (u32)S >= Gp + rtl;
                if (__builtin_amdgcn_ballot_w64(!real && !neut) != 0) {
                    u32 kk[S], shh[S], cur[S], nxt[S];
#pragma unroll
                    for (int i = 0; i < S; i++) x[i] = fix_x(x[i], gtop - 1u - (u32)i);
#pragma unroll
                    for (int i = 0; i < S; i++) {
                        const u32 la = fold_of(x[i], kk[i], shh[i]);
                        cur[i] = *(lds_u16*)(size_t)la;
                        nxt[i] = *(lds_u16*)(size_t)(la + 2);
                    }
#pragma unroll
                    for (int i = 0; i < S; i++) *(lds_x4*)(size_t)(hb + i * 1024) = fix_h(entry(x[i], kk[i], shh[i], cur[i], nxt[i]), gtop - 1u - (u32)i);
                    return;
                }
                if (pb < 2 && neut) {
#pragma unroll
                    for (int i = 0; i < S; i++) *(lds_x4*)(size_t)(hb + i * 1024) = neutral;
                }
                act = real;
            }
            if (act) {
                u32 kk[S], shh[S], cur[S], nxt[S];
#pragma unroll
                for (int i = 0; i < S; i++) {
                    const u32 la = fold_of(x[i], kk[i], shh[i]);
                    cur[i] = *(lds_u16*)(size_t)la;
                    nxt[i] = *(lds_u16*)(size_t)(la + 2);
                }
#pragma unroll
                for (int i = 0; i < S; i++) *(lds_x4*)(size_t)(hb + i * 1024) = entry(x[i], kk[i], shh[i], cur[i], nxt[i]);
            }
        };
        if constexpr (S == 8) {
            // Inputs as 16-byte loads: the 8 groups of a batch are 128 contiguous bytes of the quad's block, lane l of the quad
            // loads bytes [32 l, 32 l + 32) (a quad reads one whole line per batch instead of eight 16-byte pieces of it through
            // eight instructions), parks them in a 128-byte LDS strip of the quad and every lane picks its dword of each step.
            // (The decoder's lesson, DESIGN.md section 6 round 4: it is the number of scattered requests that costs, not the bytes.)
            // Three batches in flight: registers w[(t + 3) % 4] are requested while batch t is produced.
            u32* const strip = hand + 2 * S * 64 * 4 + 0;  // (the producer's strips follow the pair's two hand-over buffers)
            const u32 sbase = (u32)(uintptr_t)(__attribute__((address_space(3))) const u32*)strip + (lane >> 2) * 144;
            const u32 vq = (lane >> 2) * g.block_ints * 4 + 32 * ql;  // this lane's 32 bytes of a batch, relative to the batch's lowest group
            ansx_u32x4 w[4][2];
            auto request = [&](u32 batch, ansx_u32x4 (&d)[2]) {
                // groups G - 8 batch - 8 .. G - 8 batch - 1; a batch before the block's first group: an offset in the previous
                // block or beyond num_records -- never consumed
                const u32 off = vq + 16u * (Gtot - 8u * batch - 8u);
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d[0]) : "v"(off), "s"(irs) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(d[1]) : "v"(off), "s"(irs) : "memory");
            };
            request(0, w[0]);
            request(1, w[1]);
            request(2, w[2]);
            for (u32 sb = 0; sb < Gtot / ANSX_ENC_XB; sb++) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    request(pb + 3, w[(t + 3) % 4]);
                    asm volatile("s_waitcnt vmcnt(6)" : "+v"(w[t][0]), "+v"(w[t][1]) : : "memory");  // three younger requests of two loads
                    *(lds_x4*)(size_t)(sbase + 32 * ql) = w[t][0];
                    *(lds_x4*)(size_t)(sbase + 32 * ql + 16) = w[t][1];
                    u32 xx[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) xx[i] = *(__attribute__((address_space(3))) const u32*)(size_t)(sbase + 16 * (7 - i) + 4 * (3 - ql));
                    emit(xx, Gtot - 8u * pb, hbase + (pb & 1u) * (S * 1024));
                    pb++;
                    __syncthreads();
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]), "+v"(w[3][1]) : : "memory");
            return;
        }
        if constexpr (S == 4) {
        // S = 4 (two pairs per workgroup, each wave alone on its SIMD: LDS has no room for input strips, the producer has
        // instruction slots to spare): the 4 groups of a batch are 64 contiguous bytes of the quad's block; lane l loads group
        // (base + 3 - l) as ONE 16-byte load, reverses its elements (a renaming) and the quad transposes 4 x 4 (quad_transpose4,
        // 8 DPP moves + 8 selects per batch): v[i] = element (3 - l) of group base + 3 - i = this lane's input of step i.  A quarter
        // of the load requests of the dword-per-step form, each a whole 16-byte piece.  Eight batches in flight.
        const u32 vq4 = (lane >> 2) * g.block_ints * 4 + 16u * (3u - ql);
        ansx_u32x4 wq[8];
        auto request4 = [&](u32 batch, ansx_u32x4& d) {
            // groups Gtot - 4 batch - 4 .. Gtot - 4 batch - 1; a batch behind the last one: an offset beyond num_records -- never consumed
            const u32 off = vq4 + 16u * (Gtot - 4u * batch - 4u);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(off), "s"(irs) : "memory");
        };
#pragma unroll
        for (int j = 0; j < 7; j++) request4((u32)j, wq[j]);
        for (u32 sb = 0; sb < Gtot / ANSX_ENC_XB; sb++) {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                request4(pb + 7, wq[(t + 7) % 8]);
                asm volatile("s_waitcnt vmcnt(7)" : "+v"(wq[t]) : : "memory");  // seven younger requests
                u32 xs_[4] = { wq[t].w, wq[t].z, wq[t].y, wq[t].x };
                quad_transpose4(xs_, ql);
                emit(xs_, Gtot - 4u * pb, hbase + (pb & 1u) * (S * 1024));
                pb++;
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(wq[4]), "+v"(wq[5]), "+v"(wq[6]), "+v"(wq[7]) : : "memory");
        }
        return;
    }
    // -------------------------------------------------------------------- consumer
    ansx_blk* B = &blk[exists ? b : 0];
    // a block without a model (an error, an unresolved model, a one-value block of the compaction layer) goes through the
    // motions with every store out of range and publishes its special size at the end; a block that does not exist (the list
    // ends inside this workgroup) likewise, and publishes nothing
    const bool skip = !exists || B->status || !B->resolved || B->pa_sigma == 1;
    const u32 logM = exists ? B->logM : 8u;
    const u64 Lb = (u64)16 << logM;
    const double Md = (double)(1u << logM);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        scratch + (u64)wb0 * scr_stride, 0, (int)(16 * scr_stride), 0x00020000);
    const u64 sba = (u64)(uintptr_t)(scratch + (u64)wb0 * scr_stride);
    const ansx_u32x4 srs = ansx_u32x4{ (u32)__builtin_amdgcn_readfirstlane((u32)sba), (u32)__builtin_amdgcn_readfirstlane((u32)(sba >> 32) & 0xFFFFu),
        (u32)__builtin_amdgcn_readfirstlane((u32)(16 * scr_stride)), 0x00020000u };
    const u32 obase = (u32)((lane >> 2) * scr_stride);
    u8* out = scratch + (u64)b * scr_stride;
    const u32 cg = g.ckpt >> 2;  // groups per restart interval (0 = none)
    u32 ck_seg = (cg && Gp && !skip) ? (Gp - 1) / cg : 0;  // next restart point to record: segment index (of THIS block's groups)
    u32 ck_g = ck_seg * cg;                               // ... and its group index
    const u32 pbias = obase;
    double sd = (double)Lb;
    u32 pcur = skip ? 0x40000000u : B->prelude_bytes + obase;
    auto record = [&](u32 gidx) {
        // every symbol with index >= 4*gidx is now encoded: decoder restart point of segment ck_seg (as in k_encode)
        if (ck_seg && gidx == ck_g) {
            const u64 idx = (u64)b * g.nckf + (ck_seg - 1);
            const u64 stv = f64_to_u64_exact(sd);
            if (g.ckw) {
                ckpt_state[idx * 4 + ql] = stv;
                if (ql == 0) ckpt_off[idx] = pcur - pbias;
            } else {
                u8* rec = (u8*)ckpt_state + idx * ANSX_CK_RECORD;
                const u64 pst = ((u64)quad_perm<ANSX_QP(1, 0, 3, 2)>((u32)(stv >> 32)) << 32) | quad_perm<ANSX_QP(1, 0, 3, 2)>((u32)stv);
                if ((ql & 1u) == 0) {
                    u8* pp = rec + 13u * (ql >> 1);
                    st_u64_unaligned(pp, (stv & ((1ull << ANSX_CK_STATE_BITS) - 1ull)) | (pst << ANSX_CK_STATE_BITS));
                    st_u32_unaligned(pp + 8, (u32)(pst >> 12));
                    pp[12] = (u8)(pst >> 44);
                }
                if (ql == 0) {
                    const u32 cur = pcur - pbias;
                    st_u16_unaligned(rec + 26, (u16)cur);
                    rec[28] = (u8)(cur >> 16);
                }
            }
            ck_seg--;
            ck_g -= cg;
        }
    };
    const u32 hbase = (u32)(uintptr_t)(__attribute__((address_space(3))) const u32*)hand + 16 * lane;
    const u32 four_pos = 4u << (8 * ql), lomask = (1u << (8 * ql)) - 1u, ql8 = 8 * ql;
    u32 gi = Gtot;  // groups the steps still to come could hold (an irregular workgroup's first 32 are neutral for every full block)
    const u32 gstart = Gp + (ql == 0 ? rt : 0u);  // irregular: this lane's first live step is the one of group index gstart - 1
    const double sd_first = sd;
    const bool creeps = irr && logM == 16;  // (this lane's neutral steps are the stand-in: see the producer)
    auto steps = [&](auto RESET, u32 hb) {
        ansx_u32x4 h[S];
#pragma unroll
        for (int i = 0; i < S; i++) h[i] = *(lds_x4*)(size_t)(hb + i * 1024);
#pragma unroll
        for (int i = 0; i < S; i++) {
            // the arithmetic form of the table entry (stage B of k_encode, minus the reciprocal)
            const u32 k = h[i].x & 3u;
            const double rcp = __builtin_bit_cast(double, ((u64)h[i].y << 32) | h[i].x);
            const double Fd = (double)(h[i].z & 0xFFFFu);
            const double based = (double)(h[i].z >> 16);
            const u32 thr_hi = f64_hi(Fd) + (36u << 20);  // 2^36 * freq: same mantissa, low word 0
            const double MF = Md - Fd, omF = 1.0 - Fd;
            const u32 off1 = (h[i].x << 31) + ANSX_BUF_OOB;  // 0 when k is odd
            const u32 off2 = (k - 2u) & 0x80000001u;          // k & 1 when k >= 2
            // state chain (enc_update_n / the scheduled loop of k_encode)
            if constexpr (decltype(RESET)::value) sd = gi - (u32)i == gstart ? sd_first : sd;  // (the neutral steps before it have moved the state)
            const bool rn = f64_hi(sd) >= thr_hi;
            const u32 w = f64_lo(sd + 4503599627370496.0);
            int ex = rn ? -32 : 0;
            asm("" : "+v"(ex));
            const double s0 = __builtin_trunc(__builtin_ldexp(sd, ex));
            const double q = __builtin_trunc(s0 * rcp);
            const double base2 = __builtin_fma(q, MF, s0 + based);
            const double one_short = __builtin_fmin(__builtin_fmax(__builtin_fma(-q, Fd, s0 + omF), 0.0), 1.0);
            sd = __builtin_fma(one_short, MF, base2);
            // byte emission
            const u32 v = (k << ql8) + (rn ? four_pos : 0u);
            const u32 s1 = quad_add_dpp<0xB1>(v);
            const u32 Sq = quad_add_dpp<0x4E>(s1);
            const u32 a = __builtin_amdgcn_sad_u8(Sq & lomask, 0u, pcur);
            pcur = __builtin_amdgcn_sad_u8(Sq, 0u, pcur);
#ifndef ANSX_PC_ABL_NOSTORE  // (timing-only ablations, never built into the product: -DANSX_PC_ABL_*)
            __builtin_amdgcn_raw_buffer_store_b8((u8)h[i].w, rsrc, a + off1, 0, 0);
            pc_store_short_hi(h[i].w, srs, a + off2);
            __builtin_amdgcn_raw_buffer_store_b32(w, rsrc, rn ? a + k : ANSX_BUF_OOB, 0, 0);
#else
            asm volatile("" :: "v"(a + off1), "v"(a + off2), "v"(rn ? a + k : ANSX_BUF_OOB), "v"(w));
#endif
        }
    };
    for (u32 cb = 0; cb < NBATCH; cb++) {
        __syncthreads();
        const u32 hb = hbase + (cb & 1u) * (S * 1024);
        // (the batch with a creeping lane's first live step -- one or two per irregular workgroup -- runs the form that sets the state back)
        if (irr && __builtin_amdgcn_ballot_w64(creeps && gstart <= gi && gstart + (u32)S > gi) != 0) steps(std::true_type{}, hb);
        else steps(std::false_type{}, hb);
        gi -= (u32)S;
        record(gi);
    }
    if (!exists) return;
    if (skip) {
        if (ql == 0) {
            const u32 sz = (B->pa_sigma == 1 && !B->status) ? B->pre_bytes : 0u;
            B->stream_bytes = sz;
            sizes[b] = sz;
            if (sz) atomicAdd(&gsums[b >> 6], (unsigned long long)sz);
        }
        return;
    }
    // flush state - L, order 0,1,2,3 (ans_fold.hpp:275-278,115-120)
    const u32 pfin = pcur - pbias;
    st_u64_unaligned(out + pfin + 8 * ql, f64_to_u64_exact(sd) - Lb);
    if (ql == 0) {
        const u32 sz = pfin + 32;
        B->stream_bytes = sz;
        sizes[b] = sz;
        atomicAdd(&gsums[b >> 6], (unsigned long long)sz);
    }
}

// ------------------------------------------------------------------------------------------
// K6: container assembly.  Exclusive scan of the block stream sizes, then a gather copy of
// every block's bytes from its worst-case-strided scratch slot to its final offset.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan_sizes(ansx_geo g, const ansx_blk* __restrict__ blk,
    u64* __restrict__ block_off, u64* __restrict__ result, u64 payload_off, u64 capacity,
    u32* __restrict__ gflags)
{
    // One workgroup; every wave owns a contiguous range of blocks and walks it 64 blocks at a time with its lanes
    // on consecutive blocks, so the sizes of a round are one independent load per lane (a thread reading its 16
    // consecutive blocks one after the other spent 47 us in dependent round trips).
    __shared__ u64 part[20];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 NB = g.nblocks;
    const u32 per = ((NB + 15) / 16 + 63) & ~63u;  // blocks per wave, a multiple of 64
    const u32 lo = wave * per < NB ? wave * per : NB, hi = (lo + per) < NB ? (lo + per) : NB;
    u64 sum = 0;
    for (u32 i = lo + lane; i < hi; i += 64) sum += blk[i].stream_bytes;
    sum = wave_sum(sum);  // the wave's total in every lane
    u64 total;
    u64 run = block_excl_scan<u64>(lane == 63 ? sum : 0ull, part, tid, 1024, &total);  // bytes before this wave's range
    run = wave_last(run);
    if (tid == 0) {
        block_off[NB] = total;
        result[0] = total;  // payload bytes
        if (payload_off + total > capacity) atomicOr(&gflags[ANSX_G_ERR], 1u << 2 /* CAPACITY */);
    }
    for (u32 i0 = lo; i0 < hi; i0 += 64) {
        const u32 i = i0 + lane;
        const u64 v = i < hi ? (u64)blk[i].stream_bytes : 0ull;
        const u64 incl = wave_incl_scan(v);
        if (i < hi) block_off[i] = run + incl - v;
        run += wave_last(incl);
    }
}

__global__ __launch_bounds__(256) void k_compact(ansx_geo g, const ansx_blk* __restrict__ blk,
    const u64* __restrict__ block_off, const u8* __restrict__ scratch, u64 scr_stride,
    u8* __restrict__ payload, const u32* __restrict__ gflags)
{
    // A block's stream is ~18 KB: with 4 bytes per thread and round the copy was 18 dependent round trips
    // (0.10 ms, 91 % of the wave's cycles waiting).  16-byte pieces, four of them requested before the first is
    // stored, and the three scalars below requested together.
    const u32 b = blockIdx.x;
    const u32 tid = threadIdx.x;
    const u32 err = gflags[ANSX_G_ERR];
    const u32 size = blk[b].stream_bytes;
    const u64 off = block_off[b];
    if (err) return;  // capacity / domain / model error: nothing is copied
    const u8* src = scratch + (u64)b * scr_stride;
    u8* dst = payload + off;
    // head: bytes until dst is 16-byte aligned
    u32 head = (u32)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > size) head = size;
    if (tid < head) dst[tid] = src[tid];
    const u32 nq = (size - head) >> 4;
    uint4* d16 = (uint4*)(dst + head);
    const u8* s1 = src + head;
    auto ld16 = [&](u32 j) {
        const u8* p = s1 + 16 * (u64)j;
        return make_uint4(ld_u32_unaligned(p), ld_u32_unaligned(p + 4), ld_u32_unaligned(p + 8), ld_u32_unaligned(p + 12));
    };
    u32 j = tid;
    for (; j + 3 * 256 < nq; j += 4 * 256) {
        const uint4 v0 = ld16(j), v1 = ld16(j + 256), v2 = ld16(j + 512), v3 = ld16(j + 768);
        d16[j] = v0;
        d16[j + 256] = v1;
        d16[j + 512] = v2;
        d16[j + 768] = v3;
    }
    for (; j < nq; j += 256) d16[j] = ld16(j);
    const u32 done = head + 16 * nq;
    if (done + tid < size) dst[done + tid] = src[done + tid];
}

__global__ void k_write_header(ansx_geo g, u8* __restrict__ out, const u32* __restrict__ gflags,
    const u64* __restrict__ result, u64 payload_off)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // ansx_container_header, little endian (include/ansx.h)
    const char magic[8] = { 'A', 'N', 'S', 'X', 'v', '3', 0, 0 };
    for (int i = 0; i < 6; i++) out[i] = (u8)magic[i];
    {
        const u32 ms = gflags[ANSX_G_MAXSIGMA];  // bytes 6, 7: (most symbols present in a block) - 1
        *(u16*)(out + 6) = (u16)(ms ? ms - 1u : 0u);
    }
    u32* w = (u32*)(out + 8);
    w[0] = g.kind | (g.pa ? 0x100u : 0u) | (g.ckw ? ANSX_KIND_WIDE_RESTART : 0u);  // bit 8: per-block alphabet compaction, bit 9: wide restart points
    w[1] = g.f;
    *(u64*)(out + 16) = g.n;
    w = (u32*)(out + 24);
    w[0] = g.block_ints;
    w[1] = g.ckpt;
    w[2] = g.nblocks;
    w[3] = gflags[ANSX_G_MAXLOGM];
    // (plain ANSint: the bound on a block's DISTINCT values -- the same whether the call ran the dense model or the rank-space one)
    w[4] = (g.kind == 3 && !g.pa) ? gflags[ANSX_G_MAXSIGMA] : gflags[ANSX_G_MAXNSYMS];
    w[5] = g.nckf;
    *(u64*)(out + 48) = result[0];
    *(u64*)(out + 56) = payload_off;
}

// K6, fused form (up to 65536 blocks): one workgroup per block places and copies its stream without a scan kernel
// in front -- the bytes before block b are the sums of the 64-block groups before it (gsums, accumulated by the
// encoder) plus the sizes of the blocks of its own group below it: at most 1024 + 63 words, one or two loads per
// thread.  The workgroup of the last block also writes the index's final entry, the payload size and the header.
__global__ __launch_bounds__(256) void k_assemble(ansx_geo g, const u32* __restrict__ sizes,
    const unsigned long long* __restrict__ gsums, u64* __restrict__ block_off, u64* __restrict__ result,
    const u8* __restrict__ scratch, u64 scr_stride, u8* __restrict__ out, u64 payload_off, u64 capacity,
    u32* __restrict__ gflags, u32 with_header)
{
    __shared__ u64 part[8];
    const u32 b = blockIdx.x;
    const u32 tid = threadIdx.x;
    const u32 size = sizes[b];
    u64 acc = 0;
    for (u32 i = tid; i < (b >> 6); i += 256) acc += gsums[i];
    {
        const u32 i = (b & ~63u) + tid;
        if (tid < 64 && i < b) acc += sizes[i];
    }
    u64 tot;
    (void)block_excl_scan<u64>(acc, part, tid, 256, &tot);
    const u64 off = tot;
    const bool fits = payload_off + off + size <= capacity;
    if (tid == 0) {
        block_off[b] = off;
        if (!fits) atomicOr(&gflags[ANSX_G_ERR], 1u << 2 /* CAPACITY */);
        if (b == g.nblocks - 1) {
            const u64 total = off + size;
            block_off[g.nblocks] = total;
            result[0] = total;  // payload bytes
            if (with_header) {
                // ansx_container_header, little endian (include/ansx.h)
                const char magic[8] = { 'A', 'N', 'S', 'X', 'v', '3', 0, 0 };
                for (int i = 0; i < 6; i++) out[i] = (u8)magic[i];
                {
                    const u32 ms = __hip_atomic_load(&gflags[ANSX_G_MAXSIGMA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(u16*)(out + 6) = (u16)(ms ? ms - 1u : 0u);  // bytes 6, 7: (most symbols present in a block) - 1
                }
                u32* w = (u32*)(out + 8);
                w[0] = g.kind | (g.pa ? 0x100u : 0u) | (g.ckw ? ANSX_KIND_WIDE_RESTART : 0u);  // bit 8: per-block alphabet compaction, bit 9: wide restart points
                w[1] = g.f;
                *(u64*)(out + 16) = g.n;
                w = (u32*)(out + 24);
                w[0] = g.block_ints;
                w[1] = g.ckpt;
                w[2] = g.nblocks;
                w[3] = __hip_atomic_load(&gflags[ANSX_G_MAXLOGM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w[4] = __hip_atomic_load(&gflags[(g.kind == 3 && !g.pa) ? ANSX_G_MAXSIGMA : ANSX_G_MAXNSYMS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (see k_write_header)
                w[5] = g.nckf;
                *(u64*)(out + 48) = total;
                *(u64*)(out + 56) = payload_off;
            }
        }
    }
    // (a capacity / domain / model error makes the output invalid anyway: blocks that still fit are copied, the
    // call reports the error)
    if (!fits || size == 0) return;
    const u8* src = scratch + (u64)b * scr_stride;
    u8* dst = out + payload_off + off;
    // head: bytes until dst is 16-byte aligned; then 16-byte pieces, four of them requested before the first is stored
    u32 head = (u32)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > size) head = size;
    if (tid < head) dst[tid] = src[tid];
    const u32 nq = (size - head) >> 4;
    uint4* d16 = (uint4*)(dst + head);
    const u8* s1 = src + head;
    auto ld16 = [&](u32 j) {
        const u8* p = s1 + 16 * (u64)j;
        return make_uint4(ld_u32_unaligned(p), ld_u32_unaligned(p + 4), ld_u32_unaligned(p + 8), ld_u32_unaligned(p + 12));
    };
    u32 j = tid;
    for (; j + 3 * 256 < nq; j += 4 * 256) {
        const uint4 v0 = ld16(j), v1 = ld16(j + 256), v2 = ld16(j + 512), v3 = ld16(j + 768);
        d16[j] = v0;
        d16[j + 256] = v1;
        d16[j + 512] = v2;
        d16[j + 768] = v3;
    }
    for (; j < nq; j += 256) d16[j] = ld16(j);
    const u32 done = head + 16 * nq;
    if (done + tid < size) dst[done + tid] = src[done + tid];
}

// ------------------------------------------------------------------------------------------
// K8: decoder (ans_fold.hpp:179-228, 283-311).  One workgroup per block:
//   1. lane 0 parses the prelude (vbyte, log2 M, interpolative code — inherently serial bit
//      parsing, ans_util.hpp:25-42, interp.hpp:47-63,81-97) into the cumulative table,
//   2. all lanes build the slot -> symbol table (the reference's table[M], ans_fold.hpp:190-204,
//      in 2 bytes per slot),
//   3. each quad of lanes decodes one segment forward from its restart point: lane q carries
//      state 3-q and produces out[4g+q]; bytes consumed per step (renorm word, then exception
//      bytes) are located by a quad prefix sum, one unaligned 8-byte load per lane and step.
// Tables live in LDS (LDS_TAB) or, for very large frames, in a global workspace.
// ------------------------------------------------------------------------------------------
// ---- K7: prelude parse.  Bit parsing of the interpolative code is serial per block, so blocks
// are spread over LANES: every lane of a wave walks one block's prelude (interp.hpp:47-63,81-97;
// vbyte.hpp:82-95; ans_util.hpp:25-42).  The bit stream is read through a 128-bit register
// window refilled by unaligned 8-byte global loads issued one refill ahead; the traversal stack
// lives in LDS with one 16-byte entry per pending right subtree.  Output per block: inc[] in
// the global table row (cum[s+1] = inc[s], un-prefix-summed by k_decode) and
// {nsyms, log2 M, rfold flag, error}.
// Per-lane header of a block stream: ANSrfold flag + most-frequent table (ans_reorder_fold.hpp:
// 238-254), vbyte(max_sym) (vbyte.hpp:82-95), log2 M (ans_util.hpp:27-31).
struct parse_hdr {
    u32 err, ns, logM, flag, pos, sbytes;
    const u8* stream;
};
// pa_info (per-block alphabet compaction, ansx_pa.h): {sigma, header bytes, error}; the codec stream starts
// behind the alphabet header, and a block with one distinct value (or a bad header) has none: err = 2
// ("nothing to parse", not a format error of its own).
template <bool RFOLD>
__device__ __forceinline__ parse_hdr parse_header(const u8* __restrict__ cont, const ansx_geo& g, u32 NSP,
    const u64* __restrict__ block_off, u64 payload_off, u32 max_ns, u32 maxM, u32 b,
    const uint4* __restrict__ pa_info = nullptr)
{
    const u32 T = fold_T(g.f);
    u32 err = 0, ns = 0, logM = 0, flag = 0;
    const u64 boff = block_off[b], boff1 = block_off[b + 1];
    if (!g.trusted_index && !index_entry_ok(g, b, boff, boff1)) {  // (single-stream mode: the host wrote the two entries itself)
        parse_hdr Hb;
        Hb.err = 1, Hb.ns = 1, Hb.logM = 0, Hb.flag = 0, Hb.pos = 0, Hb.sbytes = 0, Hb.stream = cont;
        return Hb;
    }
    const u8* stream = cont + payload_off + boff;
    const u32 sbytes = (u32)(boff1 - boff);
    u32 pos = 0;
    if (pa_info != nullptr) {
        const uint4 pi = pa_info[b];
        pos = pi.y;
        if (pi.x == 1 || pi.z) {
            parse_hdr H0;
            H0.err = 2, H0.ns = 1, H0.logM = 0, H0.flag = 0, H0.pos = pos, H0.sbytes = sbytes, H0.stream = stream;
            return H0;
        }
    }
    if (RFOLD) {  // ans_reorder_fold.hpp:238-254
        flag = ld_u32_unaligned(stream);
        pos = 4 + (flag == 1 ? 4 * T : 0);
        if (flag > 1) err = 1;
    }
    u32 ms = 0, shv = 0;  // vbyte (vbyte.hpp:82-95)
    if (!err && sbytes >= pos + 38) {
        for (int i = 0; i < 5; i++) {
            u8 cbyte = stream[pos++];
            ms += (u32)(cbyte & 127) << shv;
            if (!(cbyte & 128)) break;
            shv += 7;
        }
        logM = stream[pos++];
    } else {
        err = 1;
    }
    ns = ms + 1;
    if (ns > max_ns || ns > NSP || logM > 30 || ((u64)1 << logM) > maxM || sbytes < pos + 32) err = 1;
    parse_hdr H;
    H.err = err, H.ns = ns, H.logM = logM, H.flag = flag, H.pos = pos, H.sbytes = sbytes, H.stream = stream;
    return H;
}

// Generic item loop (any alphabet / frame size): 128-bit register bit window refilled from global
// memory one load ahead, traversal stack of (a, n, low, high) in LDS.  stk: [20][64] uint4.
__device__ __forceinline__ u32 parse_items_generic(const parse_hdr& H, u32 NSP, u32 b, u32 lane,
    uint4 (*stk)[64], u32* __restrict__ g_cum)
{
    u32 err = 0;
    const u32 ns = H.ns, logM = H.logM, pos = H.pos, sbytes = H.sbytes;
    const u8* stream = H.stream;
    {
        const u8* bp = stream + pos;  // interpolative words start here
        u32* cum = g_cum + (u64)b * (NSP + 8);
        // 128-bit window: w0 = bits [base, base+64), w1 = next 64 bits (already in flight)
        u64 w0 = ld_u64_unaligned(bp), w1 = ld_u64_unaligned(bp + 8);
        u32 consumed = 0;          // bits of w0 already used
        u32 next_byte = 16;        // offset of the next refill
        u32 total_bits = 0;
        const u32 maxbits = (sbytes - pos) * 8;
        auto getbits = [&](u32 nbits) -> u32 {  // nbits in [0, 32]
            if (nbits == 0) return 0u;
            u64 v = w0 >> consumed;
            if (consumed + nbits > 64) v |= w1 << (64 - consumed);
            consumed += nbits;
            total_bits += nbits;
            if (consumed >= 64) {
                consumed -= 64;
                w0 = w1;
                // the stream continues behind the prelude (payload + 32 bytes of states); the refill never
                // reads past the block's own bytes, whatever a malformed prelude claims
                const u32 nb_ = next_byte + 8 <= sbytes - pos ? next_byte : sbytes - pos - 8;
                w1 = ld_u64_unaligned(bp + nb_);
                next_byte += 8;
            }
            return (u32)(v & ((nbits >= 32) ? 0xFFFFFFFFull : ((1ull << nbits) - 1ull)));
        };
        // One item per iteration for every lane (no nested pop/descend loops, so lanes stay
        // convergent): if the current range is empty, pop the pending right subtree; decode the
        // range's middle item; push its right subtree; continue into the left one.
        // All quantities fit 32 bits: u + 1 = M + nsyms + 2 < 2^31 (M <= 2^30 enforced above).
        const u32 u = (1u << logM) + ns + 1;
        u32 sp = 0;
        u32 a = 0, n = ns, low = 1, high = u + 1;
        for (u32 it = 0; it < ns; it++) {
            if (n == 0) {  // a right subtree is always pending here (ns items in total)
                if (sp == 0) {
                    err = 1;
                    break;
                }
                sp--;
                const uint4 e = stk[sp][lane];
                a = e.x;
                n = e.y;
                low = e.z;
                high = e.w;
            }
            const u32 h = (n + 1) >> 1;
            const u32 n1 = h - 1, n2 = n - h;
            const u32 U = high - n2 - low - n1 + 1;
            if (U == 0 || U > u + 1 || total_bits > maxbits) {
                err = 1;
                break;
            }
            u32 val = 1;  // read_center_mid (interp.hpp:47-63)
            if (U != 1) {
                const u32 bb = 32 - __clz(U - 1);       // hi(U-1)+1
                const u32 m = (u32)((1ull << bb) - U);
                const u32 dh = U - (1u << (bb - 1));    // (2U - 2^bb) >> 1
                val = getbits(bb - 1) + 1;
                if (val > m) val = (2 * val + getbits(1)) - m - 1;
                val += dh;
                if (val > U) val -= U;
            }
            const u32 v = low + n1 - 1 + val;
            cum[a + h] = v - 1;  // inc[a+h-1]
            if (n2) {
                stk[sp][lane] = make_uint4(a + h, n2, v + 1, high);
                sp++;
            }
            n = n1;
            high = v - 1;
        }
    }
    return err;
}

template <bool RFOLD>
__global__ __launch_bounds__(64) void k_parse_prelude(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, u64 payload_off, u32 max_ns, u32 maxM,
    u32* __restrict__ g_cum, uint4* __restrict__ binfo, u32* __restrict__ gflags, const uint4* __restrict__ pa_info)
{
    __shared__ uint4 stk[20][64];  // [depth][lane]: conflict-free 16-byte accesses
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x * 64 + lane;
    if (b >= g.nblocks) return;
    const parse_hdr H = parse_header<RFOLD>(cont, g, NSP, block_off, payload_off, max_ns, maxM, b, pa_info);
    u32 err = H.err;
    if (!err) err = parse_items_generic(H, NSP, b, lane, stk, g_cum);
    binfo[b] = make_uint4(H.ns, H.logM, H.flag, err);
    if (err == 1) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
}

// Fast item loop for the common shapes (frames + alphabets whose interpolative values fit 16 bits,
// alphabets that fit LDS).  Three observations make the per-item chain short:
//  * the traversal (which element is decoded when, interp.hpp:81-97) depends on ns only, so the
//    next node is worked out while the current item's bits are being decoded, and its stack holds
//    just (a, n);
//  * a node's bounds are its neighbours' decoded values: low = E[a] + 1, high = E[a + n + 1] - 1
//    with sentinels E[0] = 0, E[ns + 1] = u + 2; E lives in LDS as u16, [element][lane];
//  * the first ANSX_PF_SW words of the prelude are staged in LDS, [word][lane]: an item reads its
//    <= 31 bits through one 32-bit window (two words + v_alignbit), no refill logic, no 64-bit
//    shifts, no loads behind the cum[] stores.
// A lane whose prelude outgrows the staged words falls back to parse_items_generic afterwards.
#define ANSX_PF_SW 128u
template <bool RFOLD>
__global__ __launch_bounds__(64) void k_parse_prelude_fast(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, u64 payload_off, u32 max_ns, u32 maxM,
    u32* __restrict__ g_cum, uint4* __restrict__ binfo, u32* __restrict__ gflags, u32 stage_words,
    const uint4* __restrict__ pa_info)
{
    extern __shared__ __attribute__((aligned(16))) u8 pf_smem[];
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x * 64 + lane;
    if (b >= g.nblocks) return;
    u32 ebytes = (max_ns + 2) * 128;
    ebytes = ebytes < 20480u ? 20480u : ((ebytes + 15u) & ~15u);  // also hosts the generic stack
    u16* E = (u16*)pf_smem;                                         // [max_ns + 2][64]
    u32* stage = (u32*)(pf_smem + ebytes);                          // [ANSX_PF_SW][64]
    u32* stack = stage + ANSX_PF_SW * 64;                           // [21][64], row 20 = dump
    const parse_hdr H = parse_header<RFOLD>(cont, g, NSP, block_off, payload_off, max_ns, maxM, b, pa_info);
    u32 err = H.err, slow = 0;
    if (!err) {
        const u32 ns = H.ns;
        const u8* bp = H.stream + H.pos;
        const u32 avail = H.sbytes - H.pos;  // >= 32
        // stage_words <= ANSX_PF_SW (the LDS allocation); smaller values only exercise the fallback
        const u32 nd = (avail >> 3) < stage_words / 2 ? (avail >> 3) : stage_words / 2;  // whole 8-byte pieces
#pragma unroll 8
        for (u32 j = 0; j < ANSX_PF_SW / 2; j++) {
            const u64 v = j < nd ? ld_u64_unaligned(bp + 8 * j) : 0ull;
            stage[(2 * j) * 64 + lane] = (u32)v;
            stage[(2 * j + 1) * 64 + lane] = (u32)(v >> 32);
        }
        const u32 staged_bits = nd * 64;
        const u32 maxbits = avail * 8;
        const u32 u = (1u << H.logM) + ns + 1;
        E[lane] = 0;
        E[(ns + 1) * 64 + lane] = (u16)(u + 2);
        stack[lane] = 0;  // row 0 is read speculatively while the stack is empty
        u32* cum = g_cum + (u64)b * (NSP + 8);
        // The loop body is one basic block: a lane that hits an error or runs out of staged bits
        // only sets `stop` (1 = malformed, 2 = continue in the generic loop) and keeps going with
        // frozen bit position -- every LDS / global index below is data independent or clamped, so
        // that is harmless, and without exits the speculative stack and bounds reads at the top
        // overlap the item's decode instead of following it.
        u32 a = 0, n = ns, low = 1, high = u + 1, bitpos = 0, sp = 0, stop = 0;
        for (u32 it = 0; it < ns; it++) {
            const u32 h = (n + 1) >> 1;
            const u32 n1 = h - 1, n2 = n - h, pe = a + h;
            // next node (data independent): left child, else right child, else the pending one
            const bool caseA = n1 != 0, caseB = !caseA && n2 != 0, caseC = !caseA && !caseB;
            const u32 pc = stack[(sp ? sp - 1 : 0) * 64 + lane];
            u32 wi = bitpos >> 5;
            wi = wi < ANSX_PF_SW - 2 ? wi : ANSX_PF_SW - 2;
            const u32 w0 = stage[wi * 64 + lane], w1 = stage[(wi + 1) * 64 + lane];
            const u32 pa = pc >> 16, pn = pc & 0xFFFFu;
            const u32 Ea = E[pa * 64 + lane], Eb = E[(pa + pn + 1) * 64 + lane];
            const bool push = caseA && n2 != 0;
            stack[(push ? sp : 20u) * 64 + lane] = (pe << 16) | n2;
            // this item (read_center_mid, interp.hpp:47-63)
            const u32 U = high - n2 - low - n1 + 1;
            const u32 win = __builtin_amdgcn_alignbit(w1, w0, bitpos & 31u);
            const u32 Um1 = U - 1;
            const u32 bb = 32 - __clz(Um1 | 1u) - (Um1 == 0 ? 1u : 0u);  // hi(U-1)+1; 0 for U == 1
            const u32 lb = bb ? bb - 1 : 0;                               // bits of the first read
            const u32 m = (u32)((1ull << bb) - U);
            const u32 dh = U - ((1u << lb) & (bb ? ~0u : 0u));
            u32 val = (win & ((1u << lb) - 1u)) + 1;
            const bool big = (U != 1) && (val > m);
            val = big ? (2 * val + ((win >> lb) & 1u)) - m - 1 : val;
            val += dh;
            if (val > U) val -= U;
            if (U == 1) val = 1;
            const u32 v = low + n1 - 1 + val;
            // v <= u < 2^16 for a well-formed prelude
            const bool bad = (U == 0) || (U > u + 1) || (bitpos > maxbits) || (v > 0xFFFFu);
            const u32 now = (bitpos + 64 > staged_bits) ? 2u : (bad ? 1u : 0u);
            stop = stop ? stop : now;
            bitpos += (stop || U == 1) ? 0u : lb + (big ? 1u : 0u);
            E[pe * 64 + lane] = (u16)v;
            cum[pe] = v - 1;  // inc[pe - 1] (rewritten by the generic loop / ignored when stop != 0)
            // descend
            const u32 na = caseA ? a : (caseB ? pe : pa);
            const u32 nn = caseA ? n1 : (caseB ? n2 : pn);
            const u32 nlow = caseA ? low : (caseB ? v + 1 : Ea + 1);
            const u32 nhigh = caseA ? v - 1 : (caseB ? high : Eb - 1);
            sp = sp + (push ? 1u : 0u) - ((caseC && sp) ? 1u : 0u);
            a = na, n = nn, low = nlow, high = nhigh;
        }
        err = (stop == 1) ? 1u : 0u;
        slow = (stop == 2) ? 1u : 0u;
        if (slow) err = parse_items_generic(H, NSP, b, lane, (uint4(*)[64])pf_smem, g_cum);
    }
    binfo[b] = make_uint4(H.ns, H.logM, H.flag, err);
    if (err == 1) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
}

// ---- K7, windowed form: any alphabet / frame size.
// One lane per (sub)tree, no per-symbol LDS array: a node's bounds travel WITH the node (the pending right
// subtrees sit on an LDS stack as (start | size, low, high) in three lane-major arrays, and the top entry
// is mirrored in registers, so a pop takes its node from registers and only refreshes the mirror), and the
// bit stream is read through three consecutive 32-bit words held in registers plus one prefetched word: an
// item takes its <= 31 + 1 bits from one v_alignbit of two of them, and at most one word boundary is
// crossed per item.  The words come from an LDS window of SW words per lane, staged from the stream and
// re-staged FOR EVERY LANE as soon as one lane gets near its end (lanes progress at about the same rate,
// so a wave re-stages once per ~SW words, not once per lane), so a code may be arbitrarily long.  The item
// body is straight-line code under a per-lane predicate: finished or malformed lanes keep their state
// frozen.  parse_subtree_win decodes the `n` items of the subtree rooted at items [a, a + n) with value
// bounds [low, high], whose code starts at bit `bitpos` of the interpolative words at bp; every lane of
// the wave must call it (lanes with nothing to do pass n = 0).  Returns the lane's error flag.
template <u32 SW, u32 STK>
__device__ __forceinline__ u32 parse_subtree_win(u32 (*stage)[64], u32 (*stkA)[64], u32 (*stkL)[64], u32 (*stkH)[64],
    u32 lane, const u8* __restrict__ bp, u32 avail_words, u32 u, u32 a, u32 n, u32 low, u32 high, u32 bitpos,
    u32 err, u32* __restrict__ cum)
{
    const u32 cnt = err ? 0u : n;  // items of this lane
    const u32 maxbits = avail_words * 32;
    u32 ta = 0, tl = 0, th = 0;     // register mirror of the stack's top entry
    u32 sp = 0, done = 0;
    u32 base_w = 0;                 // stream word held in stage[0][lane]
    u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // words wi, wi + 1, wi + 2 (wi = bitpos >> 5), prefetched wi + 3
    bool trigger = true;
    for (;;) {
        const bool pending = !err && done < cnt;
        if (__builtin_amdgcn_ballot_w64(pending) == 0) break;
        if (trigger) {
            // (re)stage every unfinished lane's next SW words from its current word on
            if (pending) base_w = bitpos >> 5;
            wave_lds_sync();
#pragma unroll 8
            for (u32 j = 0; j < SW / 2; j++) {
                const u32 w = base_w + 2 * j;
                u64 v = 0;
                if (pending && w + 2 <= avail_words) v = ld_u64_unaligned(bp + 4 * (u64)w);
                else if (pending && w < avail_words) v = ld_u32_unaligned(bp + 4 * (u64)w);
                stage[2 * j][lane] = (u32)v;
                stage[2 * j + 1][lane] = (u32)(v >> 32);
            }
            wave_lds_sync();
            w0 = stage[0][lane], w1 = stage[1][lane], w2 = stage[2][lane], w3 = stage[3][lane];
            trigger = false;
        }
        for (;;) {
            const bool act = !err && done < cnt;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            // node shape (a function of the subtree size alone)
            const u32 h = (n + 1) >> 1;
            const u32 n1 = h - 1, n2 = n - h, pe = a + h;
            // this item (read_center_mid, interp.hpp:47-63)
            const u32 U = high - n2 - low - n1 + 1;
            const u32 win = __builtin_amdgcn_alignbit(w1, w0, bitpos & 31u);  // the next 32 bits
            const u32 Um1 = U - 1;
            const u32 bb = 32 - __clz(Um1 | 1u) - (Um1 == 0 ? 1u : 0u);      // hi(U-1)+1; 0 for U == 1
            const u32 lb = bb ? bb - 1 : 0;                                   // bits of the first read (<= 31)
            const u32 m = (u32)((1ull << bb) - U);
            const u32 dh = U - ((1u << lb) & (bb ? ~0u : 0u));
            u32 val = (win & ((1u << lb) - 1u)) + 1;
            const bool big = (U != 1) && (val > m);
            val = big ? (2 * val + ((win >> lb) & 1u)) - m - 1 : val;
            val += dh;
            if (val > U) val -= U;
            if (U == 1) val = 1;
            const u32 v = low + n1 - 1 + val;
            const u32 len = (U == 1) ? 0u : lb + (big ? 1u : 0u);
            const bool bad = (U == 0) || (U > u + 1) || (bitpos + len > maxbits);
            if (act && bad) err = 1;
            const bool go = act && !bad;
            // next node: left child, else right child, else the pending one
            const bool caseA = n1 != 0, caseB = !caseA && n2 != 0;
            const bool push = go && caseA && n2 != 0;
            const bool pop = go && !caseA && !caseB;
            if (go) cum[pe] = v - 1;  // inc[pe - 1]
            const u32 srow = push ? sp : STK - 1;
            stkA[srow][lane] = pe | (n2 << 16);  // start, size < 2^15 (alphabets <= 16384 slots)
            stkL[srow][lane] = v + 1;
            stkH[srow][lane] = high;
            const u32 na = caseA ? a : (caseB ? pe : (ta & 0xFFFFu));
            const u32 nn = caseA ? n1 : (caseB ? n2 : (ta >> 16));
            const u32 nlow = caseA ? low : (caseB ? v + 1 : tl);
            const u32 nhigh = caseA ? v - 1 : (caseB ? high : th);
            if (push) {  // (before `high` moves on to the child)
                ta = pe | (n2 << 16), tl = v + 1, th = high;
            }
            if (go) {
                a = na, n = nn, low = nlow, high = nhigh;
                done++;
            }
            sp = sp + (push ? 1u : 0u) - ((pop && sp) ? 1u : 0u);
            // after a pop the entry below becomes the top: fetch it for the mirror (used at the NEXT pop
            // at the earliest)
            const u32 below = sp ? sp - 1 : STK - 1;
            const u32 ra = stkA[below][lane], rl = stkL[below][lane], rh = stkH[below][lane];
            if (pop) {
                ta = ra, tl = rl, th = rh;
            }
            // bit window: at most one word boundary per item
            const u32 nbit = go ? bitpos + len : bitpos;
            if ((nbit >> 5) != (bitpos >> 5)) {
                w0 = w1, w1 = w2, w2 = w3;
            }
            bitpos = nbit;
            const u32 rel = (nbit >> 5) - base_w;
            w3 = stage[rel + 3 < SW ? rel + 3 : SW - 1][lane];
            // near the end of the staged words: every lane re-stages before the next item
            if (__builtin_amdgcn_ballot_w64(go && done < cnt && rel + 5 >= SW) != 0) {
                trigger = true;
                break;
            }
        }
        if (!trigger) break;
    }
    return err;
}

// one lane per block (ANSX_PARSE_WIN; default for containers without usable parse hints)
template <bool RFOLD, u32 SW>
__global__ __launch_bounds__(64) void k_parse_prelude_win(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, u64 payload_off, u32 max_ns, u32 maxM,
    u32* __restrict__ g_cum, uint4* __restrict__ binfo, u32* __restrict__ gflags, const uint4* __restrict__ pa_info)
{
    extern __shared__ u32 pw_lds[];
    u32(*stage)[64] = (u32(*)[64])pw_lds;                       // [SW][64]
    u32(*stkA)[64] = (u32(*)[64])(pw_lds + SW * 64);            // [24][64] start | size << 16
    u32(*stkL)[64] = stkA + 24;                                 // low bound of the pending subtree
    u32(*stkH)[64] = stkL + 24;                                 // high bound; row 23 of each = dump
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x * 64 + lane;
    const bool live = b < g.nblocks;
    parse_hdr H;
    H.err = 1, H.ns = 1, H.logM = 0, H.flag = 0, H.pos = 0, H.sbytes = 0, H.stream = cont;
    if (live) H = parse_header<RFOLD>(cont, g, NSP, block_off, payload_off, max_ns, maxM, b, pa_info);
    const u32 herr = H.err;  // 2: compaction, no codec stream to parse (not an error of this kernel)
    const u32 ns = herr ? 0u : H.ns;
    const u8* bp = H.stream + H.pos;                                  // interpolative words start here
    const u32 avail_words = herr ? 0u : (H.sbytes - H.pos) >> 2;      // whole words inside the block stream
    const u32 u = (1u << H.logM) + ns + 1;                            // universe (ans_util.hpp:60), < 2^31
    u32* cum = g_cum + (u64)(live ? b : 0) * (NSP + 8);
    const u32 err = parse_subtree_win<SW, 24>(stage, stkA, stkL, stkH, lane, bp, avail_words, u, 0, ns, 1, u + 1, 0, herr, cum);
    if (live) {
        binfo[b] = make_uint4(H.ns, H.logM, H.flag, err);
        if (err && herr != 2) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
    }
}

// ---- K7, parallel form (the default): the container index carries, per block, the bit offsets at which
// the right subtrees of the code's top seven nodes begin (written by the prelude writer, which has every
// item's offset anyway).  Eight lanes per block: the top three levels are decoded
// level by level (a node needs its parent's value for its bounds; left children follow their parent in
// the stream, right children sit at the hinted offsets), then every lane runs parse_subtree_win on one of
// the eight depth-3 subtrees -- an eighth of the serial chain.  Hints are untrusted input like the
// payload: offsets are bounds-checked, and a wrong one yields a table that fails the decoder's
// consistency checks or decodes to garbage, never an out-of-range access.
template <bool RFOLD, u32 SW>
__global__ __launch_bounds__(256) void k_parse_prelude_par(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, u64 payload_off, u32 max_ns, u32 maxM, const u32* __restrict__ hints,
    u32* __restrict__ g_cum, uint4* __restrict__ binfo, u32* __restrict__ gflags, const uint4* __restrict__ pa_info)
{
    // SW staged words per lane (a subtree's share of a prelude is ~35 bytes at 530 symbols, ~130 at 2300); the host
    // launches SW = 32.  The kernel is VALU-issue bound machine-wide (~125 instructions per item).
    // Up to four waves per workgroup, each on its own slice of the dynamic LDS and never synchronised with the
    // others: like the encoder's, single-wave workgroups pile up unevenly on a CU's SIMDs once a CU holds more
    // than two of them (0.036 ms up to 2 per CU, 0.103 at 4, 0.168 at 8); see the launch site for the sweep.
    extern __shared__ u32 par_lds[];
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 (*stage)[64] = (u32 (*)[64])(par_lds + wv * ((SW + 48) * 64));
    u32 (*stkA)[64] = stage + SW;   // subtrees of <= 2048 items: depth <= 12; row 15 = dump
    u32 (*stkL)[64] = stkA + 16;
    u32 (*stkH)[64] = stkL + 16;
    const u32 j = lane & 7;                      // lane within the block's group
    const u32 b = (blockIdx.x * (blockDim.x >> 6) + wv) * 8 + (lane >> 3);
    const bool live = b < g.nblocks;
    parse_hdr H;
    H.err = 1, H.ns = 1, H.logM = 0, H.flag = 0, H.pos = 0, H.sbytes = 0, H.stream = cont;
    if (live) H = parse_header<RFOLD>(cont, g, NSP, block_off, payload_off, max_ns, maxM, b, pa_info);
    const u32 herr = H.err;
    const u32 ns = herr ? 0u : H.ns;
    const u8* bp = H.stream + H.pos;
    const u32 avail_words = herr ? 0u : (H.sbytes - H.pos) >> 2;
    const u32 maxbits = avail_words * 32;
    const u32 u = (1u << H.logM) + ns + 1;
    u32* cum = g_cum + (u64)(live ? b : 0) * (NSP + 8);
    const u32 hv = live ? hints[(u64)b * 8 + j] : 0u;   // lane j holds hint word j of its block
    // ---- top three levels, one level per round; lane j < 2^k owns node j of level k
    u32 a = 0, n = (j == 0) ? ns : 0u, low = 1, high = u + 1, bit = 0, err = 0;
    for (u32 k = 0; k < 3; k++) {
        u32 v = 0, len = 0;
        const bool mine = j < (1u << k) && n != 0;
        if (mine) {  // decode this node's item at `bit` (read_center_mid, interp.hpp:47-63)
            const u32 h = (n + 1) >> 1;
            const u32 n1 = h - 1, n2 = n - h;
            const u32 U = high - n2 - low - n1 + 1;
            u64 w = 0;
            const u32 byte = bit >> 3, avail = avail_words * 4;
            if (byte + 8 <= avail) w = ld_u64_unaligned(bp + byte);
            else
                for (u32 i = 0; byte + i < avail && i < 8; i++) w |= (u64)bp[byte + i] << (8 * i);
            const u32 win = (u32)(w >> (bit & 7u));
            const u32 Um1 = U - 1;
            const u32 bb = 32 - __clz(Um1 | 1u) - (Um1 == 0 ? 1u : 0u);
            const u32 lb = bb ? bb - 1 : 0;
            const u32 m = (u32)((1ull << bb) - U);
            const u32 dh = U - ((1u << lb) & (bb ? ~0u : 0u));
            u32 val = (win & ((1u << lb) - 1u)) + 1;
            const bool big = (U != 1) && (val > m);
            val = big ? (2 * val + ((win >> lb) & 1u)) - m - 1 : val;
            val += dh;
            if (val > U) val -= U;
            if (U == 1) val = 1;
            v = low + n1 - 1 + val;
            len = (U == 1) ? 0u : lb + (big ? 1u : 0u);
            if (U == 0 || U > u + 1 || bit + len > maxbits) err = 1;
            else cum[a + h] = v - 1;
        }
        // children: lane j' < 2^(k+1) takes child (j' & 1) of the node held by lane j' >> 1 of its group
        const int src = (int)((lane & ~7u) | (j >> 1));
        const u32 pa_ = (u32)__shfl((int)a, src), pn = (u32)__shfl((int)n, src), plow = (u32)__shfl((int)low, src),
                  phigh = (u32)__shfl((int)high, src), pbit = (u32)__shfl((int)bit, src), pv = (u32)__shfl((int)v, src),
                  plen = (u32)__shfl((int)len, src), perr = (u32)__shfl((int)err, src);
        // hint word of the parent's right subtree: parent = node (2^k - 1) + (j >> 1) in breadth-first order
        const u32 hword = (u32)__shfl((int)hv, (int)((lane & ~7u) | (((1u << k) + (j >> 1)) & 7u)));
        if (j < (2u << k)) {
            const u32 ph = (pn + 1) >> 1;
            const bool right = (j & 1u) != 0;
            const bool dead = pn == 0 || perr != 0;
            a = right ? pa_ + ph : pa_;
            n = dead ? 0u : (right ? pn - ph : ph - 1);
            low = right ? pv + 1 : plow;
            high = right ? phigh : pv - 1;
            bit = right ? hword : pbit + plen;
            err = perr;
            if (n != 0 && bit > maxbits) err = 1;
        }
    }
    // ---- the eight depth-3 subtrees
    err = parse_subtree_win<SW, 16>(stage, stkA, stkL, stkH, lane, bp, avail_words, u, a, n, low, high, bit,
        herr ? herr : err, cum);
    // block verdict: any lane of the group
    u32 e = err;
    e |= (u32)__shfl_xor((int)e, 1);
    e |= (u32)__shfl_xor((int)e, 2);
    e |= (u32)__shfl_xor((int)e, 4);
    if (live && j == 0) {
        binfo[b] = make_uint4(H.ns, H.logM, H.flag, herr == 2 ? 2u : (e ? 1u : 0u));
        if (e && herr != 2) atomicOr(&gflags[ANSX_G_ERR], 1u << 3 /* FORMAT */);
    }
}

// 8 stream bytes ending at byte offset `end` (exclusive): from the LDS-staged copy (aligned
// words + v_alignbyte) or straight from global memory (one unaligned 8-byte load)
// per-quad stream ring of the block decoder: ANSX_RING_CHK steps per refill check, 32 bytes per lane
// and step of check interval per refill (>= the 28 bytes a quad can consume per step)
#ifndef ANSX_RING_CHK
#define ANSX_RING_CHK 4
#endif
#define ANSX_DEC_SCRATCH 96u  // k_decode_rank: scan scratch + error flag between the tables and the stream area
#define ANSX_RING_BYTES (128 * ANSX_RING_CHK)
#define ANSX_RING_STRIDE (ANSX_RING_BYTES + 16)  // + 8 mirror bytes (ring bytes 0..7 once more) and padding to 16
#define ANSX_SRING_BYTES 256                       // the speculative small ring (round 4), see dec_segments_ring_small
#define ANSX_SRING_STRIDE (ANSX_SRING_BYTES + 16)
// end8 = end - 8: the decoder keeps its cursor biased by 8, so this is what its prefix sum yields
template <int MODE>
__device__ __forceinline__ u64 dec_fetch8(const u8* __restrict__ stream, const u32* lds_stream, int end8)
{
    if (MODE == 1) {
        // lds_stream word 0 holds stream bytes [-8,-4): byte a of the stream is at lds byte a+8
        const int w = end8 >> 2;  // >= -2: the two guard words
        const u32 sh = (u32)end8 & 3u;
        const u32 w0 = lds_stream[w + 2], w1 = lds_stream[w + 3], w2 = lds_stream[w + 4];
        const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
        const u32 hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
        return ((u64)hi << 32) | lo;
    } else if (MODE == 2) {
        // lds_stream = this quad's ring: stream byte s lives at ring byte s & (RB - 1), and ring bytes 0..7 are
        // kept once more behind the ring, so the three words that cover the 8 bytes are contiguous wherever
        // they start: one address, ds_read2_b32 + ds_read_b32
        typedef __attribute__((address_space(3))) const u32 lds_cu32;
        const u32 rb = (u32)(size_t)(__attribute__((address_space(3))) const void*)lds_stream;
        const u32 a = (u32)end8;
        lds_cu32* wp = (lds_cu32*)(size_t)(rb + (a & (ANSX_RING_BYTES - 4)));
        const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2];
        const u32 sh = a & 3;
        const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
        const u32 hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
        return ((u64)hi << 32) | lo;
    } else if (MODE == 3) {
        // the 256-byte speculative ring (dec_segments_ring_small): as MODE 2 with that ring's mask
        typedef __attribute__((address_space(3))) const u32 lds_cu32;
        const u32 rb = (u32)(size_t)(__attribute__((address_space(3))) const void*)lds_stream;
        const u32 a = (u32)end8;
        lds_cu32* wp = (lds_cu32*)(size_t)(rb + (a & (ANSX_SRING_BYTES - 4)));
        const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2];
        const u32 sh = a & 3;
        const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
        const u32 hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
        return ((u64)hi << 32) | lo;
    } else {
        return ld_u64_unaligned(stream + end8);
    }
}

// slot -> (symbol, freq, base) lookups of the decoder.
//  dec_lut_table: the reference's layout idea (ans_fold.hpp:190-204) in 2 bytes per slot: a
//                 slot -> symbol array plus the cumulative table (any frame size; arrays may
//                 live in LDS or HBM).
//  dec_lut_rank : rank/select form, ~(M/4 + 6 nsyms) bytes instead of 2M + 4 nsyms: a bitmap with
//                 one bit per slot that starts a symbol, interleaved with the running popcount
//                 before each 32-bit word; rank(slot) = v_bcnt(word & mask, prefix) - 1 indexes
//                 compact per-present-symbol entries (base << 16 | freq) and value words.  Still
//                 two dependent LDS reads per step, but a fraction of the LDS: frames up to 2^16
//                 keep their tables (and the stream rings / staged stream) in LDS.
// Both return freq, base and pv = k << 30 | value-without-exception-bytes for the slot's symbol
// (k = number of exception bytes, ans_fold.hpp:150-175; for ANSrfold with the reorder flag set the
// value is the remapped one, ans_reorder_fold.hpp:207-219,300-301 -- with flag 0 plain fold,
// SURVEY F3).
#define ANSX_PV_MASK 0x3FFFFFFFu
// mfv: the block's most-frequent-table entry of sym (only read when rf && sym < T)
__device__ __forceinline__ u32 dec_make_pv(const ansx_map& f, u32 sym, bool rf, u32 T, u32 mfv)
{
    const u32 k = unmap_nbytes(f, sym);
    u32 v0 = unmap_value(f, sym, k);
    if (rf) v0 = (sym < T) ? mfv : (v0 - T);
    return (k << 30) | (v0 & ANSX_PV_MASK);
}
struct dec_lut_table {
    static constexpr bool WIDE = true;  // any frame size: 32-bit frequencies, full 64-bit state arithmetic
    const u32* cum;
    const u16* s2s;
    ansx_map f;
    u32 T;
    const u32* mf;  // non-null: ANSrfold block with the reorder flag set
    __device__ __forceinline__ void get(u32 slot, u32& fr, u32& base, u32& pv) const
    {
        const u32 sym = s2s[slot];
        const u32 c0 = cum[sym], c1 = cum[sym + 1];
        fr = c1 - c0;
        base = c0;
        pv = dec_make_pv(f, sym, mf != nullptr, T, (mf != nullptr && sym < T) ? mf[sym] : 0u);
    }
};
struct dec_lut_rank {
    static constexpr bool WIDE = false;
    const uint2* bwp;  // {bitmap word, (set bits before it) - 1 + (LDS byte address of ep) / 8}
    const uint2* ep;   // per present symbol: {base << 16 | freq, k << 30 | value without its exception bytes}
    // the second word of a bitmap entry, see above (ep is 8-byte aligned)
    __device__ __forceinline__ u32 bias() const { return (u32)(size_t)(__attribute__((address_space(3))) const void*)ep / 8u - 1u; }
    __device__ __forceinline__ void get(u32 slot, u32& fr, u32& base, u32& pv) const
    {
        // The table starts at LDS address 0 (k_decode_rank has no static LDS and checks it): the entry's address is
        // its byte offset, built from an integer so that no link-time base is added to it.
        const u64 wpw = *(__attribute__((address_space(3))) const u64*)(size_t)((slot >> 2) & 0x3FF8u);
        const uint2 wp = make_uint2((u32)wpw, (u32)(wpw >> 32));
        // bits [0, slot & 31] of the word: shift the rest out at the top (the shifter uses the low
        // five bits of ~slot = 31 - (slot & 31)); v_bcnt adds the biased prefix: the entry's address / 8
        const u32 r8 = (u32)__builtin_popcount(wp.x << (~slot & 31u)) + wp.y;
        const u64 e = *(__attribute__((address_space(3))) const u64*)(size_t)(r8 << 3);  // one 8-byte LDS read
        pv = (u32)(e >> 32);
        fr = (u32)e & 0xFFFFu;
        base = (u32)e >> 16;
    }
};

// One decoder step of one state (ans_fold.hpp:216-228,135-147).  q = -(byte cursor): the bytes a
// quad consumes in a step (per lane: renorm word, exception bytes below it) are located by the
// packed-byte quad sum of enc_update_n: lane ql puts its count into byte ql, two DPP adds give
// every lane the quad's counts S, v_sad_u8 adds the bytes of S & lomask (lanes before this one) or
// of S (all four) onto q.
struct dec_quad_const {
    u32 ql8;       // 8 * ql
    u32 lomask;    // (1 << 8 ql) - 1
    u32 four_pos;  // 4 << 8 ql
};
__device__ __forceinline__ dec_quad_const dec_make_qc(u32 ql) { return dec_quad_const{ 8 * ql, (1u << (8 * ql)) - 1u, 4u << (8 * ql) }; }
// q = 8 - (byte cursor): the sums below then yield (position - 8), where the 8 bytes a lane needs start
#define ANSX_DEC_Q(p) ((u32)(8 - (int)(p)))
#define ANSX_DEC_P(q) (8 - (int)(q))
template <int STREAM_LDS, typename LUT>
__device__ __forceinline__ u32 dec_step(u64& st, u32& q, bool active, const dec_quad_const qc, u32 logM, u32 mask,
    u64 Lb, const LUT& lut, const u8* __restrict__ stream, const u32* lds_stream)
{
    const u32 slot = (u32)st & mask;
    u32 fr, base, pv;
    lut.get(slot, fr, base, pv);
    // ans_fold.hpp:218-220: fr * (st >> logM) + (slot - base).  st < 2^52 for frames up to 2^16, so
    // the high word of the quotient is small: one 32x32->64 mad, and a 24-bit mad into its high word
    const u64 qs = st >> logM;
    u64 ns_;
    bool rn;
    u32 n_lo = 0, n_hi = 0;  // (rank form: the new state as two words -- packing them into a u64 and taking it apart
                             // again below made hipcc copy the 64-bit product with a second, quarter-rate v_mad_u64_u32)
    if constexpr (LUT::WIDE) {
        ns_ = (u64)fr * qs + (u64)(slot - base);  // frames above 2^16: freq up to 2^27, quotient below 2^36
        rn = active && (ns_ < Lb);
    } else {
        const u64 t = (u64)fr * (u32)qs + (u64)(slot - base);
        n_lo = (u32)t;
        n_hi = (u32)(t >> 32) + __umul24(fr, (u32)(qs >> 32));
        rn = active && n_hi == 0 && n_lo < (u32)Lb;  // Lb = 16 M <= 2^20 here
        ns_ = 0;
    }
    const u32 k = pv >> 30;
    const u32 cq = active ? ((k << qc.ql8) + (rn ? qc.four_pos : 0u)) : 0u;
    const u32 s1 = quad_add_dpp<0xB1>(cq);
    const u32 S = quad_add_dpp<0x4E>(s1);
    int myp8 = -(int)__builtin_amdgcn_sad_u8(S & qc.lomask, 0u, q);
    q = __builtin_amdgcn_sad_u8(S, 0u, q);
    // a corrupt stream can drive the cursor below 0: the ring wraps every address into itself, the
    // other sources have 8 guard bytes in front and clamp
    if (STREAM_LDS != 2 && STREAM_LDS != 3) myp8 = myp8 < -8 ? -8 : myp8;
    const u64 v = dec_fetch8<STREAM_LDS>(stream, lds_stream, myp8);
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    if constexpr (LUT::WIDE) {
        if (rn) ns_ = (ns_ << 32) | hi;  // ans_fold.hpp:221-225
        if (active) st = ns_;
    } else {
        const u32 s_lo = rn ? hi : n_lo, s_hi = rn ? n_lo : n_hi;
        if (active) st = ((u64)s_hi << 32) | s_lo;
    }
    // the k exception bytes sit just below the renorm word (or at the top when there is none): the top k
    // bytes of lo resp. hi = that word, zero-extended to 64 bits, shifted right by 32 - 8k (0 for k = 0).
    // (Added, not OR-ed: an ANSrfold value has T subtracted and its low 8k bits are no longer zero.)
    const u32 e = (u32)((u64)(rn ? lo : hi) >> (32u - 8u * k));
    return (pv & ANSX_PV_MASK) + e;
}

// 4 x 4 transpose inside a quad: lane l (0..3 within its quad) holds v[u] = ITS value of step u and ends with
// v[c] = lane c's value of step l -- the four neighbouring output ints of step l -- so that an interval's outputs leave as
// ONE 16-byte store per lane (a quad writes 64 contiguous bytes) instead of four 4-byte stores 16 bytes apart per step.
// (The decoder's scattered dword stores -- 16 partial 16-byte writes per wave and step -- were a third of its time: 0.72 ms
// with them, 0.48 with the same stores aimed at one address)  Butterfly: elements u and
// u ^ 1 between lanes l and l ^ 1 where bit 0 of u and l differ, then u and u ^ 2 between l and l ^ 2 on bit 1.
__device__ __forceinline__ void quad_transpose4(u32 (&v)[4], u32 ql)
{
    const bool o1 = (ql & 1u) != 0, o2 = (ql & 2u) != 0;
    u32 a[4];
    {
        const u32 x0 = (u32)__builtin_amdgcn_update_dpp(0, (int)v[0], 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
        const u32 x1 = (u32)__builtin_amdgcn_update_dpp(0, (int)v[1], 0xB1, 0xF, 0xF, true);
        const u32 x2 = (u32)__builtin_amdgcn_update_dpp(0, (int)v[2], 0xB1, 0xF, 0xF, true);
        const u32 x3 = (u32)__builtin_amdgcn_update_dpp(0, (int)v[3], 0xB1, 0xF, 0xF, true);
        a[0] = o1 ? x1 : v[0];
        a[1] = o1 ? v[1] : x0;
        a[2] = o1 ? x3 : v[2];
        a[3] = o1 ? v[3] : x2;
    }
    {
        const u32 x0 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[0], 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
        const u32 x1 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[1], 0x4E, 0xF, 0xF, true);
        const u32 x2 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[2], 0x4E, 0xF, 0xF, true);
        const u32 x3 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[3], 0x4E, 0xF, 0xF, true);
        v[0] = o2 ? x2 : a[0];
        v[1] = o2 ? x3 : a[1];
        v[2] = o2 ? a[2] : x0;
        v[3] = o2 ? a[3] : x1;
    }
}

// decode every segment of one block (one quad of lanes per segment)
template <bool STREAM_LDS, typename LUT>
__device__ __forceinline__ void dec_segments(const ansx_geo& g, u32 b, u32 nb, u32 sbytes, u32 tid,
    u32 nt, u32 logM, const LUT& lut, const u8* __restrict__ stream, const u32* lds_stream,
    const u64* __restrict__ ckpt_state, const u32* __restrict__ ckpt_off, u32* __restrict__ o)
{
    const u32 nseg = geo_nseg(nb, g.ckpt);
    const u32 nq = nt >> 2, quad = tid >> 2, ql = tid & 3;
    const dec_quad_const qc = dec_make_qc(ql);
    const u64 Lb = (u64)16 << logM;
    const u32 mask = (1u << logM) - 1;
    const u32 rtail = nb & 3, nfull = nb - rtail;
    for (u32 seg = quad; seg < nseg; seg += nq) {
        u64 st;
        int p;
        if (seg == 0) {  // ans_fold.hpp:289-295: states 3,2,1,0 from the end
            st = ld_u64_unaligned(stream + sbytes - 32 + 8 * (3 - ql)) + Lb;
            p = (int)sbytes - 32;
        } else {
            const u64 idx = (u64)b * g.nckf + (seg - 1);
            u32 po;
            ckpt_load(g, ckpt_state, ckpt_off, idx, 3 - ql, &st, &po);
            p = (int)(po < sbytes ? po : sbytes);
        }
        u32 q = ANSX_DEC_Q(p);
        const u32 start = seg * g.ckpt;
        u32 end = (seg == nseg - 1) ? nfull : (start + g.ckpt);
        end = end < nfull ? end : nfull;
        if (STREAM_LDS) {
            u32* op = o + start + ql;
            const u32 steps = (end - start) >> 2;
            const u32 steps0 = (u32)__builtin_amdgcn_readfirstlane((int)steps);
            if (__builtin_amdgcn_ballot_w64(steps != steps0) == 0) {
                // every quad of the wave decodes the same number of groups (all but the wave that
                // holds a short last segment): scalar trip count, unrolled
                u32 i = 0;
                u32* ob = o + start;  // (16-byte aligned: restart intervals and block lengths are multiples of 4 ints)
                for (; i + 4 <= steps0; i += 4) {
                    u32 v4[4];
#pragma unroll
                    for (u32 u = 0; u < 4; u++) v4[u] = dec_step<1>(st, q, true, qc, logM, mask, Lb, lut, stream, lds_stream);
                    quad_transpose4(v4, ql);  // one 16-byte store per lane instead of four dwords 16 bytes apart (see quad_transpose4)
                    *(ansx_u32x4*)(ob + 4 * (i + ql)) = ansx_u32x4{ v4[0], v4[1], v4[2], v4[3] };
                }
                for (; i < steps0; i++)
                    op[4 * i] = dec_step<1>(st, q, true, qc, logM, mask, Lb, lut, stream, lds_stream);
            } else {
                for (u32 i = 0; i < steps; i++)
                    op[4 * i] = dec_step<1>(st, q, true, qc, logM, mask, Lb, lut, stream, lds_stream);
            }
        } else {
            // The cursor walks down the stream ~4.5 bytes per step; pull the next 512 bytes
            // (one 128-byte line per lane) towards the CU whenever it gets within 256 bytes of
            // the prefetched frontier, so that the dependent 8-byte loads hit in cache.
            int pf_front = p;
            u32 pf = 0;
            for (u32 i = start; i < end; i += 4) {
                if (ANSX_DEC_P(q) - 256 < pf_front) {
                    asm volatile("" ::"v"(pf));
                    int a = pf_front - 128 * (int)(ql + 1);
                    a = a < 0 ? 0 : a;
                    pf = ld_u32_unaligned(stream + (a & ~3));
                    pf_front -= 512;
                }
                o[i + ql] = dec_step<0>(st, q, true, qc, logM, mask, Lb, lut, stream, lds_stream);
            }
            asm volatile("" ::"v"(pf));
        }
        if (seg == nseg - 1) {  // tail symbols come from state 0 = lane 3 (ans_fold.hpp:307-310)
            for (u32 i = nfull; i < nb; i++) {
                u32 val = dec_step<STREAM_LDS ? 1 : 0>(st, q, ql == 3, qc, logM, mask, Lb, lut, stream, lds_stream);
                if (ql == 3) o[i] = val;
            }
        }
    }
}

// ---- per-quad stream ring ------------------------------------------------------------------
// Instead of the whole block stream (~20 KB at the defaults, which caps a CU at 5 blocks in
// flight), every quad keeps a window of RB = ANSX_RING_BYTES bytes of ITS segment in LDS; stream
// byte s lives at ring byte s & (RB-1).  The cursor p walks down at most 28 bytes per step and a
// step reads down to p - 32.  Every CHK = ANSX_RING_CHK steps ("check") the R = 32*CHK bytes
// requested at the previous check are written into the ring (lo -= R) and, if fewer than T bytes
// remain below p, another R are requested (R/64 16-byte buffer loads per lane; a lane with
// nothing to fetch gets an out-of-range offset).  With c = 28*CHK the most an interval consumes:
//   L = 32 + c   bytes must remain below p when an interval starts
//   T = RB - R - 16 >= L + c : after a landing p - lo >= L always (requested: >= L - c + R, not
//   requested: >= T - c), and before a landing p - lo < T, so the R new bytes replace consumed ones
//   and the window [lo, p + 4) never exceeds RB.
// The loads are inline asm, invisible to hipcc's waitcnt pass, and are consumed behind
// s_waitcnt vmcnt(CHK): vmcnt is in-order and exactly the interval's CHK output stores are
// younger, so the wait never touches them (letting the compiler wait for these loads costs a
// drain of the output stores per refill -- the reason an earlier ring attempt was slower than
// whole-stream staging).  Used when every segment of the block has the same length.
struct dec_ring_desc {
    ansx_u32x4 rsrc;  // buffer descriptor over [stream - backoff, stream + sbytes + slack) inside the container
    int backoff;      // stream offset s is at buffer offset s + backoff
};
__device__ __forceinline__ ansx_u32x4 ring_load16(const dec_ring_desc& D, int s, bool want)
{
    const int off = s + D.backoff;
    const u32 voff = (want && off >= 0) ? (u32)off : ANSX_BUF_OOB;
    ansx_u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(D.rsrc) : "memory");
    return r;
}

// What a quad needs before it can decode its first segment -- restart state, cursor and the initial ring window --
// requested BEFORE the block's tables are built, so that the two dependent round trips (restart point, then the
// stream bytes it points at) overlap the table build instead of following it.
struct dec_ring_pre {
    u64 st;
    int p, lo;
    ansx_u32x4 r[ANSX_RING_BYTES / 64];
};
__device__ __forceinline__ void dec_ring_prefetch(dec_ring_pre& P, const ansx_geo& g, u32 b, u32 sbytes, u32 tid,
    u32 logM, const u8* __restrict__ stream, const dec_ring_desc& D, const u64* __restrict__ ckpt_state,
    const u32* __restrict__ ckpt_off)
{
    constexpr int RB = ANSX_RING_BYTES, R = 32 * ANSX_RING_CHK, T = RB - R - 16;
    const u32 seg = tid >> 2, ql = tid & 3;
    const u32 nseg = g.block_ints / g.ckpt;
    P.st = 0;
    P.p = 0;
    if (seg == 0) {  // ans_fold.hpp:289-295: states 3,2,1,0 from the end
        P.st = ld_u64_unaligned(stream + sbytes - 32 + 8 * (3 - ql)) + ((u64)16 << logM);
        P.p = (int)sbytes - 32;
    } else if (seg < nseg) {
        const u64 idx = (u64)b * g.nckf + (seg - 1);
        u32 po;
        ckpt_load(g, ckpt_state, ckpt_off, idx, 3 - ql, &P.st, &po);
        P.p = (int)(po < sbytes ? po : sbytes);
    }
    P.lo = (P.p - T) & ~(R / 4 - 1);
#pragma unroll
    for (int j = 0; j < RB / 64; j++) P.r[j] = ring_load16(D, P.lo + (RB / 4) * (int)ql + 16 * j, seg < nseg);
}

template <typename LUT>
__device__ __forceinline__ void dec_segments_ring(const ansx_geo& g, u32 b, u32 sbytes, u32 tid, u32 nt,
    u32 logM, const LUT& lut, const u8* __restrict__ stream, u32* rings, const dec_ring_desc& D,
    const u64* __restrict__ ckpt_state, const u32* __restrict__ ckpt_off, u32* __restrict__ o, dec_ring_pre& P)
{
    constexpr int RB = ANSX_RING_BYTES, CHK = ANSX_RING_CHK;
    constexpr int R = 32 * CHK;        // bytes per refill, 8 * CHK per lane
    constexpr int NP = CHK / 2;        // 16-byte pieces per lane and refill
    constexpr int T = RB - R - 16;     // request threshold
    static_assert(CHK == 2 || CHK == 4, "lane pieces are 16 or 32 bytes");
    static_assert(T >= 32 + 2 * 28 * CHK, "ring too small for the worst-case consumption");
    const u32 nseg = g.block_ints / g.ckpt;  // uniform segments (checked by the caller)
    const u32 nq = nt >> 2, quad = tid >> 2, ql = tid & 3;
    const dec_quad_const qc = dec_make_qc(ql);
    const u64 Lb = (u64)16 << logM;
    const u32 mask = (1u << logM) - 1;
    u32* ring = rings + quad * (ANSX_RING_STRIDE / 4);
    u8* ring8 = (u8*)ring;
    const u32 steps = g.ckpt >> 2;
    const int lane_off = (R / 4) * (int)ql;  // this lane's share of a refill
    for (u32 seg = quad; seg < nseg; seg += nq) {
        u64 st;
        int p;
        int lo;
        // initial window [lo, lo + RB) with p - lo in [T, T + R/4): RB/64 pieces per lane.  lo is a
        // multiple of the lane share, so a lane's pieces never straddle the ring's end.
        ansx_u32x4 r[RB / 64];
        if (seg == quad) {  // the quad's first segment: requested by dec_ring_prefetch before the table build
            st = P.st;
            p = P.p;
            lo = P.lo;
#pragma unroll
            for (int j = 0; j < RB / 64; j++) r[j] = P.r[j];
        } else {
            const u64 idx = (u64)b * g.nckf + (seg - 1);  // (seg >= nq > 0)
            u32 po;
            ckpt_load(g, ckpt_state, ckpt_off, idx, 3 - ql, &st, &po);
            p = (int)(po < sbytes ? po : sbytes);
            lo = (p - T) & ~(R / 4 - 1);
#pragma unroll
            for (int j = 0; j < RB / 64; j++) r[j] = ring_load16(D, lo + (RB / 4) * (int)ql + 16 * j, true);
        }
        u32 q = ANSX_DEC_Q(p);
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < RB / 64; j++) {
                asm volatile("" : "+v"(r[j]));  // keep the dependence on the asm loads behind the wait
                const u32 d = (u32)(lo + (RB / 4) * (int)ql + 16 * j) & (RB - 1);
                *(ansx_u32x4*)(ring8 + d) = r[j];
                if (d == 0) *(uint2*)(ring8 + RB) = make_uint2(r[j].x, r[j].y);  // second copy of ring bytes 0..7
            }
        }
        u32* op = o + seg * g.ckpt + ql;
        u32* ob = o + seg * g.ckpt;  // (16-byte aligned: the output buffer is, and block_ints and the interval are multiples of 4)
        bool pending = false;
        ansx_u32x4 rr[NP];
#pragma unroll
        for (int j = 0; j < NP; j++) rr[j] = ansx_u32x4{ 0u, 0u, 0u, 0u };
        auto land = [&]() {
            if (pending) {
                lo -= R;
                const u32 d0 = (u32)(lo + lane_off) & (RB - 1);
#pragma unroll
                for (int j = 0; j < NP; j++) *(ansx_u32x4*)(ring8 + d0 + 16 * j) = rr[j];  // aligned share: no wrap inside
                if (d0 == 0) *(uint2*)(ring8 + RB) = make_uint2(rr[0].x, rr[0].y);  // second copy of ring bytes 0..7
            }
        };
        u32 i = 0;
        for (; i + CHK <= steps; i += CHK) {
            // ---- check: land the previous request, decide the next one
            // (exactly ONE operation is younger than the request: the previous interval's 16-byte output store)
            if constexpr (CHK == 4) asm volatile("s_waitcnt vmcnt(1)" : "+v"(rr[0]), "+v"(rr[NP - 1])::"memory");
            else asm volatile("s_waitcnt vmcnt(2)" : "+v"(rr[0])::"memory");
            land();
            const int cur = ANSX_DEC_P(q);
            pending = (cur - lo) < T;
#pragma unroll
            for (int j = 0; j < NP; j++) rr[j] = ring_load16(D, lo - R + lane_off + 16 * j, pending);
            if constexpr (CHK == 4) {
                u32 v4[4];
#pragma unroll
                for (u32 u = 0; u < 4; u++) v4[u] = dec_step<2>(st, q, true, qc, logM, mask, Lb, lut, stream, ring);
                quad_transpose4(v4, ql);  // lane ql: the four ints of step i + ql
                *(ansx_u32x4*)(ob + 4 * (i + ql)) = ansx_u32x4{ v4[0], v4[1], v4[2], v4[3] };
            } else {
#pragma unroll
                for (u32 u = 0; u < (u32)CHK; u++)
                    op[4 * (i + u)] = dec_step<2>(st, q, true, qc, logM, mask, Lb, lut, stream, ring);
            }
        }
        // leftover steps (restart interval not a multiple of 4*CHK): land what is in flight first
        if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0]), "+v"(rr[1])::"memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0])::"memory");
        land();
        for (; i < steps; i++) op[4 * i] = dec_step<2>(st, q, true, qc, logM, mask, Lb, lut, stream, ring);
    }
}

// ---- per-quad stream ring, 256 bytes, speculative (round 4) -------------------------------------------------------
// The 512-byte ring above is sized for the format's worst case -- 28 stream bytes per step, every step -- and is 8.4 of the
// 15 KB of LDS a block of the headline workload needs: ten blocks per CU, and the decoder is bound by how many chains a CU
// holds.  Lists whose streams average a few bytes per step (1.1 bytes per int on the headline
// workload: 4.4 per step) never come near that.  This form keeps a 256-byte window (sixteen blocks per CU) and does NOT
// guarantee that an interval's reads stay inside it: it CHECKS, after every interval of CHK = 4 steps, that the cursor is
// still at least 16 bytes above the window's low end -- every byte an interval reads lies at most 11 below its final cursor --
// and otherwise takes the interval back: states and cursor as saved at its start, a full window loaded synchronously below
// that cursor (at least 225 bytes: more than the 112 + 16 an interval can need), the four steps again (their output stores
// are idempotent).  R = 128 more bytes are requested when at most 128 remain below the cursor -- they land one check later,
// when at most 128 remain for certain, so they only replace consumed bytes -- and an interval can only fail if the one before
// it and itself consumed more than 112 bytes together: never on such lists, every interval on a list of 30-bit values with
// three exception bytes each (the host chooses this form by the container's bytes per int; correctness does not depend on it).
struct dec_sring_pre {
    u64 st;
    int p, lo;
    ansx_u32x4 r[ANSX_SRING_BYTES / 64];
};
__device__ __forceinline__ int dec_sring_lo(int p) { return (p - 225) & ~31; }  // p - lo in [225, 256]
__device__ __forceinline__ void dec_sring_prefetch(dec_sring_pre& P, const ansx_geo& g, u32 b, u32 sbytes, u32 tid,
    u32 logM, const u8* __restrict__ stream, const dec_ring_desc& D, const u64* __restrict__ ckpt_state,
    const u32* __restrict__ ckpt_off)
{
    constexpr int RB = ANSX_SRING_BYTES;
    const u32 seg = tid >> 2, ql = tid & 3;
    const u32 nseg = g.block_ints / g.ckpt;
    P.st = 0;
    P.p = 0;
    if (seg == 0) {  // ans_fold.hpp:289-295: states 3,2,1,0 from the end
        P.st = ld_u64_unaligned(stream + sbytes - 32 + 8 * (3 - ql)) + ((u64)16 << logM);
        P.p = (int)sbytes - 32;
    } else if (seg < nseg) {
        const u64 idx = (u64)b * g.nckf + (seg - 1);
        u32 po;
        ckpt_load(g, ckpt_state, ckpt_off, idx, 3 - ql, &P.st, &po);
        P.p = (int)(po < sbytes ? po : sbytes);
    }
    P.lo = dec_sring_lo(P.p);
#pragma unroll
    for (int j = 0; j < RB / 64; j++) P.r[j] = ring_load16(D, P.lo + (RB / 4) * (int)ql + 16 * j, seg < nseg);
}

template <typename LUT>
__device__ __forceinline__ void dec_segments_ring_small(const ansx_geo& g, u32 b, u32 sbytes, u32 tid, u32 nt,
    u32 logM, const LUT& lut, const u8* __restrict__ stream, u32* rings, const dec_ring_desc& D,
    const u64* __restrict__ ckpt_state, const u32* __restrict__ ckpt_off, u32* __restrict__ o, dec_sring_pre& P)
{
    constexpr int RB = ANSX_SRING_BYTES, CHK = 4;
    constexpr int R = 128;             // bytes per refill, 32 per lane
    constexpr int NP = 2;              // 16-byte pieces per lane and refill
    constexpr int T = 128;             // request when at most this much remains below the cursor
    constexpr int MARGIN = 16;         // an interval that ends with less than this below its cursor is taken back
    const u32 nseg = g.block_ints / g.ckpt;  // uniform segments (checked by the caller)
    const u32 nq = nt >> 2, quad = tid >> 2, ql = tid & 3;
    const dec_quad_const qc = dec_make_qc(ql);
    const u64 Lb = (u64)16 << logM;
    const u32 mask = (1u << logM) - 1;
    u32* ring = rings + quad * (ANSX_SRING_STRIDE / 4);
    u8* ring8 = (u8*)ring;
    const u32 steps = g.ckpt >> 2;
    const int lane_off = (R / 4) * (int)ql;  // this lane's share of a refill
    for (u32 seg = quad; seg < nseg; seg += nq) {
        u64 st;
        int p, lo;
        ansx_u32x4 r[RB / 64];
        if (seg == quad) {  // the quad's first segment: requested by dec_sring_prefetch before the table build
            st = P.st;
            p = P.p;
            lo = P.lo;
#pragma unroll
            for (int j = 0; j < RB / 64; j++) r[j] = P.r[j];
        } else {
            const u64 idx = (u64)b * g.nckf + (seg - 1);  // (seg >= nq > 0)
            u32 po;
            ckpt_load(g, ckpt_state, ckpt_off, idx, 3 - ql, &st, &po);
            p = (int)(po < sbytes ? po : sbytes);
            lo = dec_sring_lo(p);
#pragma unroll
            for (int j = 0; j < RB / 64; j++) r[j] = ring_load16(D, lo + (RB / 4) * (int)ql + 16 * j, true);
        }
        u32 q = ANSX_DEC_Q(p);
        // the whole window [lo, lo + RB): RB / 64 pieces per lane, written once they have landed
        auto fill = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < RB / 64; j++) {
                asm volatile("" : "+v"(r[j]));  // keep the dependence on the asm loads behind the wait
                const u32 d = (u32)(lo + (RB / 4) * (int)ql + 16 * j) & (RB - 1);
                *(ansx_u32x4*)(ring8 + d) = r[j];
                if (d == 0) *(uint2*)(ring8 + RB) = make_uint2(r[j].x, r[j].y);  // second copy of ring bytes 0..7
            }
        };
        fill();
        u32* op = o + seg * g.ckpt + ql;
        u32* ob = o + seg * g.ckpt;
        bool pending = false;
        ansx_u32x4 rr[NP];
#pragma unroll
        for (int j = 0; j < NP; j++) rr[j] = ansx_u32x4{ 0u, 0u, 0u, 0u };
        auto land = [&]() {
            if (pending) {
                lo -= R;
                const u32 d0 = (u32)(lo + lane_off) & (RB - 1);
#pragma unroll
                for (int j = 0; j < NP; j++) *(ansx_u32x4*)(ring8 + d0 + 16 * j) = rr[j];  // aligned share: no wrap inside
                if (d0 == 0) *(uint2*)(ring8 + RB) = make_uint2(rr[0].x, rr[0].y);
            }
        };
        // a full window below cursor position p0, synchronously (the rare paths)
        auto refill_at = [&](int p0) {
            lo = dec_sring_lo(p0);
#pragma unroll
            for (int j = 0; j < RB / 64; j++) r[j] = ring_load16(D, lo + (RB / 4) * (int)ql + 16 * j, true);
            fill();
            pending = false;
        };
        u32 i = 0;
        for (; i + CHK <= steps; i += CHK) {
            // ---- check: land the previous request (exactly the interval's CHK output stores are younger), decide the next
            asm volatile("s_waitcnt vmcnt(1)" : "+v"(rr[0]), "+v"(rr[NP - 1])::"memory");  // (younger: the previous interval's one store)
            land();
            pending = (ANSX_DEC_P(q) - lo) <= T;
#pragma unroll
            for (int j = 0; j < NP; j++) rr[j] = ring_load16(D, lo - R + lane_off + 16 * j, pending);
            const u64 st0 = st;
            const u32 q0 = q;
            u32 v4[4];
#pragma unroll
            for (u32 u = 0; u < (u32)CHK; u++) v4[u] = dec_step<3>(st, q, true, qc, logM, mask, Lb, lut, stream, ring);
            if (__builtin_amdgcn_ballot_w64((ANSX_DEC_P(q) - lo) < MARGIN) != 0) {
                // some quad of the wave may have read below its window: the whole wave takes the interval back (a quad that
                // was fine repeats the same four steps from the same state)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0]), "+v"(rr[NP - 1])::"memory");
                st = st0;
                q = q0;
                refill_at(ANSX_DEC_P(q0));
#pragma unroll
                for (int j = 0; j < NP; j++) rr[j] = ansx_u32x4{ 0u, 0u, 0u, 0u };
#pragma unroll
                for (u32 u = 0; u < (u32)CHK; u++) v4[u] = dec_step<3>(st, q, true, qc, logM, mask, Lb, lut, stream, ring);
            }
            quad_transpose4(v4, ql);  // lane ql: the four ints of step i + ql, one 16-byte store (see quad_transpose4)
            *(ansx_u32x4*)(ob + 4 * (i + ql)) = ansx_u32x4{ v4[0], v4[1], v4[2], v4[3] };
        }
        // leftover steps (restart interval not a multiple of 16 ints): behind a full window, whatever is in flight dropped
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0]), "+v"(rr[NP - 1])::"memory");
        if (i < steps) {
            refill_at(ANSX_DEC_P(q));
            for (; i < steps; i++) op[4 * i] = dec_step<3>(st, q, true, qc, logM, mask, Lb, lut, stream, ring);
        }
    }
}

// stage the block's stream in LDS: byte a of the stream -> LDS byte a + 8 (8 guard bytes in front)
__device__ __forceinline__ void dec_stage_stream(u32* lds_stream, const u8* __restrict__ stream, u32 sbytes,
    u32 tid, u32 nt)
{
    const u32 nw = (sbytes + 3) >> 2;
    for (u32 w = tid; w < nw; w += nt) lds_stream[w + 2] = ld_u32_unaligned(stream + 4 * (u64)w);
    if (tid < 2) lds_stream[tid] = 0;
    if (tid < 2) lds_stream[nw + 2 + tid] = 0;
}

// The rank/select decoder tables of one block, built by the whole workgroup (nt threads) from its parsed inc[] row.
// Frequencies from the parsed inc[] (ans_util.hpp:33-41: nfreq[s] = inc[s]-inc[s-1]-1), compaction of the
// present symbols, running base, start-of-symbol bitmap: one packed exclusive scan per nt symbols (frequency in the low
// word, presence in the high word) gives every present symbol its base and its rank.  The inc[] values of four rounds
// are requested together, so a block pays one global round trip per 4 nt symbols (one wave doing 64 symbols per round
// with the loads inside the round was most of this kernel's time on 2300-symbol alphabets); the first four rounds'
// values arrive in cur4 / prv4 (requested by the caller before anything else was waited for).
// Bitmap entry w = {bitmap word, biased prefix} sits at bw[w * WSTRIDE + WOFF] (.x) and + 1 (.y): WSTRIDE 2 is the
// one-block layout (uint2 array), WSTRIDE 4 interleaves the entries of TWO blocks (WOFF 0 / 2) so that both are
// addressed by the same register with an immediate offset (k_decode_rank2).  The caller has zeroed the .x words and
// sh_bad and synchronised.  Returns false (after a workgroup barrier) if the row is not a valid table.
template <bool RFOLD, u32 WSTRIDE, u32 WOFF>
__device__ __forceinline__ bool dec_build_rank_tables(const ansx_geo& g, u32* bw, uint2* ep, u64* sh_scan, u32* sh_bad,
    const u32* __restrict__ gc, u32 (&cur4)[4], u32 (&prv4)[4], u32 ns, u32 M, u32 W, u32 max_ns, u32 rflag,
    const u8* __restrict__ stream, u32 tid, u32 nt, u32* __restrict__ gflags)
{
    const ansx_map f = g.map;
    const u32 T = fold_T(g.f);
    {
        u64 carry = 0;
        u32 bad = 0;
        for (u32 c0 = 0; c0 < ns; c0 += 4 * nt) {
            // this chunk's values were requested one chunk ago (the first: at kernel start); request the next
            u32 c4[4], p4[4], nx4[4] = {}, np4[4] = {};
#pragma unroll
            for (u32 q = 0; q < 4; q++) {
                c4[q] = cur4[q];
                p4[q] = (c0 + q * nt + tid) ? prv4[q] + 1u : 0u;
            }
            if (c0 + 4 * nt < ns) {
#pragma unroll
                for (u32 q = 0; q < 4; q++) {
                    const u32 s = c0 + 4 * nt + q * nt + tid;
                    nx4[q] = s < ns ? gc[s + 1] : 0u;
                    np4[q] = s < ns ? gc[s] : 0u;
                }
            }
#pragma unroll
            for (u32 q = 0; q < 4; q++) {
                const u32 s = c0 + q * nt + tid;
                if (c0 + q * nt >= ns) break;  // uniform
                u32 fr = 0;
                if (s < ns) {
                    fr = c4[q] - p4[q];
                    if (c4[q] < p4[q] || fr > M || fr > 0xFFFFu) {  // entries hold 16-bit freq and base
                        bad = 1;
                        fr = 0;
                    }
                }
                const u64 packed = (u64)fr | ((u64)(fr ? 1u : 0u) << 32);
                u64 total;
                const u64 ex = carry + block_excl_scan<u64>(packed, sh_scan, tid, nt, &total);
                carry += total;
                if (fr) {
                    const u64 base64 = ex & 0xFFFFFFFFull;
                    const u32 r = (u32)(ex >> 32);
                    if (base64 < M && base64 + fr <= M && r < max_ns) {  // (max_ns here: the header's bound on symbols PRESENT in a block, i.e. the entries ep[] holds; untrusted like the rest)
                        const u32 base = (u32)base64;
                        // ANSrfold: the most-frequent values follow the 4-byte flag word
                        // (ans_reorder_fold.hpp:132-154)
                        u32 mfv = 0;
                        if (RFOLD && rflag && s < T) mfv = ld_u32_unaligned(stream + 4 + 4 * (u64)s);
                        ep[r] = make_uint2((base << 16) | fr, dec_make_pv(f, s, RFOLD && rflag, T, mfv));
                        atomicOr(&bw[(base >> 5) * WSTRIDE + WOFF], 1u << (base & 31));
                    } else {
                        bad = 1;
                    }
                }
            }
#pragma unroll
            for (u32 q = 0; q < 4; q++) {
                cur4[q] = nx4[q];
                prv4[q] = np4[q];
            }
        }
        if ((carry & 0xFFFFFFFFull) != M || (carry >> 32) > max_ns) bad = 1;
        if (bad) {
            *sh_bad = 1;
            atomicOr(&gflags[ANSX_G_ERR], 1u << 3);
        }
    }
    __syncthreads();
    if (*sh_bad) return false;
    if (tid < 64) {  // running popcount before every bitmap word
        const u32 per = (W + 63) / 64;
        const u32 lo = tid * per;
        u32 loc = 0;
        for (u32 i = 0; i < per; i++)
            if (lo + i < W) loc += (u32)__builtin_popcount(bw[(lo + i) * WSTRIDE + WOFF]);
        const u32 incl = wave_incl_scan(loc);
        // (the second word of a bitmap entry: set bits before it - 1 + (LDS byte address of ep) / 8, see dec_lut_rank)
        u32 run = incl - loc + ((u32)(size_t)(__attribute__((address_space(3))) const void*)ep / 8u - 1u);
        for (u32 i = 0; i < per; i++)
            if (lo + i < W) {
                bw[(lo + i) * WSTRIDE + WOFF + 1] = run;
                run += (u32)__builtin_popcount(bw[(lo + i) * WSTRIDE + WOFF]);
            }
    }
    __syncthreads();
    return true;
}

// ---- K8: one workgroup per block.  Builds the decoder tables, then one quad of lanes per segment
// decodes forward from its restart point, reading the stream through per-quad LDS rings (RING),
// a staged copy of the whole block stream, or straight from HBM (partial block / no room).
// (History on MI355X, 256 Mi ints, with the original slot->symbol tables: no staging 1.56 ms;
// whole-stream staging 1.38 ms; a ring with compiler-managed waits 1.62 ms -- every refill drained
// the output stores.  With rank/select tables and the hand-counted vmcnt(4) ring: 0.73 ms.)
//
// k_decode_rank: frames up to 2^16, rank/select tables in LDS (the normal path).
// RING: per-quad stream rings instead of the staged stream (stream_cap is then the container size).
// RING: 0 = staged stream (or straight from HBM), 1 = per-quad 512-byte rings, 2 = per-quad 256-byte speculative rings
template <bool RFOLD, int RING>
__global__ void k_decode_rank(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, const u64* __restrict__ ckpt_state,
    const u32* __restrict__ ckpt_off, u64 payload_off, u32* __restrict__ outp, u32 maxM,
    u32 max_ns, u64 stream_cap, const u32* __restrict__ g_cum, const uint4* __restrict__ binfo,
    u32* __restrict__ gflags)
{
    // (no static LDS in this kernel: the bitmap table sits at LDS address 0, so its reads need no base added)
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const u32 b = blockIdx.x;
    // Everything the table build needs from HBM is requested before anything is waited for: the block's parse
    // results, its stream bounds and the first 4 nt parsed inc[] values (a row of g_cum has NSP + 8 entries
    // whatever the block's alphabet; values beyond it are masked below) -- one round trip instead of three.
    const u32* gc = g_cum + (u64)b * (NSP + 8);  // gc[s+1] = inc[s]
    u32 cur4[4], prv4[4];
#pragma unroll
    for (u32 q = 0; q < 4; q++) {
        const u32 s = q * nt + tid;
        cur4[q] = s < NSP ? gc[s + 1] : 0u;
        prv4[q] = (s < NSP && s) ? gc[s] : 0u;
    }
    const uint4 bi = binfo[b];
    const u64 boff = block_off[b], boff1 = block_off[b + 1];
    if (bi.w) return;  // parse error already flagged
    const u32 ns = bi.x, logM = bi.y, rflag = bi.z;
    const u32 nb = geo_block_n(g, b);
    const u8* stream = cont + payload_off + boff;
    const u32 sbytes = (u32)(boff1 - boff);
    const u32 M = 1u << logM;
    // LDS carve: [bitmap+prefix][entries][symbol ids][most-frequent table][staged stream]
    const u32 wmax = maxM >= 32 ? maxM / 32 : 1;
    u32 off = 0;
    uint2* bwp = (uint2*)(smem + off);
    off += (wmax * 8 + 15) & ~15u;
    uint2* ep = (uint2*)(smem + off);
    off += 2 * ((max_ns * 4 + 15) & ~15u);  // (same bytes as the host's 2 x rup(4 max_ns, 16))
    u64* sh_scan = (u64*)(smem + off);        // 9 words of scan scratch + the block's error flag (ANSX_DEC_SCRATCH bytes)
    u32& sh_bad = *(u32*)(smem + off + 80);
    off += ANSX_DEC_SCRATCH;
    u32* lds_stream = (u32*)(smem + off);
    const u32 W = M >= 32 ? M / 32 : 1;
    // full block: all segments have g.ckpt ints (host-checked), each quad's first one is requested now
    const bool use_ring = RING && nb == g.block_ints;
    dec_ring_desc D;
    typename std::conditional<RING == 2, dec_sring_pre, dec_ring_pre>::type RP;
    if (use_ring) {
        // buffer view of this block's stream with up to 1 KB in front of it (the initial window and
        // the guard bytes reach below offset 0) and 64 bytes behind, clipped to the container
        const u64 sabs = payload_off + boff;  // stream offset inside the container
        D.backoff = (int)(sabs < 1024 ? sabs : 1024);
        const u8* base = cont + (sabs - (u64)D.backoff);
        u64 span = (u64)D.backoff + sbytes + 64;
        const u64 room = stream_cap - (sabs - (u64)D.backoff);  // container bytes from base on
        if (span > room) span = room;
        const u64 ba = (u64)(uintptr_t)base;
        D.rsrc = ansx_u32x4{ (u32)ba, (u32)(ba >> 32) & 0xFFFFu, (u32)span, 0x00020000u };
        if constexpr (RING == 2) dec_sring_prefetch(RP, g, b, sbytes, tid, logM, stream, D, ckpt_state, ckpt_off);
        else if constexpr (RING == 1) dec_ring_prefetch(RP, g, b, sbytes, tid, logM, stream, D, ckpt_state, ckpt_off);
    }
    for (u32 w = tid; w < W; w += nt) bwp[w] = make_uint2(0u, 0u);
    if (tid == 0) {
        sh_bad = 0;
        // dec_lut_rank::get addresses the bitmap table from LDS address 0
        if ((u32)(size_t)(__attribute__((address_space(3))) void*)bwp != 0u) atomicOr(&gflags[ANSX_G_ERR], 1u << 3);
    }
    const bool st_lds = !RING && (sbytes + 24 <= stream_cap);
    if (st_lds) dec_stage_stream(lds_stream, stream, sbytes, tid, nt);
    __syncthreads();
    if (!dec_build_rank_tables<RFOLD, 2, 0>(g, (u32*)bwp, ep, sh_scan, &sh_bad, gc, cur4, prv4, ns, M, W, max_ns, rflag, stream, tid, nt, gflags)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the prefetched window is still on its way)
        return;
    }
    dec_lut_rank lut;
    lut.bwp = bwp;
    lut.ep = ep;
    u32* o = outp + (u64)b * g.block_ints;
    if (use_ring) {
        // rings start at the next 16-byte boundary of the LDS address space (the host adds the slack)
        const u32 labs = (u32)(size_t)(__attribute__((address_space(3))) void*)lds_stream;
        u32* rings = lds_stream + ((((labs + 15u) & ~15u) - labs) >> 2);
        if constexpr (RING == 2) dec_segments_ring_small(g, b, sbytes, tid, nt, logM, lut, stream, rings, D, ckpt_state, ckpt_off, o, RP);
        else if constexpr (RING == 1) dec_segments_ring(g, b, sbytes, tid, nt, logM, lut, stream, rings, D, ckpt_state, ckpt_off, o, RP);
    } else if (st_lds)
        dec_segments<true>(g, b, nb, sbytes, tid, nt, logM, lut, stream, lds_stream, ckpt_state, ckpt_off, o);
    else
        dec_segments<false>(g, b, nb, sbytes, tid, nt, logM, lut, stream, lds_stream, ckpt_state, ckpt_off, o);
}

// ---- k_decode_rank2: TWO blocks per workgroup, their segments decoded in ONE instruction stream ---------------
// A decoder step is a dependent chain (slot -> bitmap word -> entry -> multiply -> renormalise? -> quad sum -> stream
// window -> next slot: ~22 vector instructions and three LDS round trips, ~500 cycles for a wave alone), and what a wave
// waits for inside it is not free for the other waves of its SIMD: an instruction that depends on the one before it
// holds the vector pipe for ~8.4 cycles instead of ~4.3 (tests/tools/ubench_valu2.hip, "dependent f64 chain": the wave
// beside it gets one issue per ~8 cycles).  k_decode_rank runs 10 such single-chain waves per CU and saturates at about
// half the issue rate.  Here every lane carries the states of two segments -- segment i of block 2w and segment i of
// block 2w + 1 -- and the two steps are written into one basic block, so that consecutive instructions belong to
// different chains; the LDS per block is what it was (two tables, two sets of rings: five workgroups per CU instead of
// ten, the same blocks in flight).  The bitmap entries of the two blocks are interleaved (16 bytes per 32 slots:
// {word A, prefix A, word B, prefix B}) so that both tables are addressed from LDS address 0 by the same shift, block B
// through the instruction's immediate offset.
template <u32 OFF>
struct dec_lut_rank2 {
    static constexpr bool WIDE = false;
    static constexpr u32 OFFSET = OFF;
    __device__ __forceinline__ void get(u32 slot, u32& fr, u32& base, u32& pv) const
    {
        const u64 wpw = *(__attribute__((address_space(3))) const u64*)(size_t)(((slot >> 1) & 0x7FF0u) + OFF);
        const uint2 wp = make_uint2((u32)wpw, (u32)(wpw >> 32));
        const u32 r8 = (u32)__builtin_popcount(wp.x << (~slot & 31u)) + wp.y;
        const u64 e = *(__attribute__((address_space(3))) const u64*)(size_t)(r8 << 3);  // one 8-byte LDS read
        pv = (u32)(e >> 32);
        fr = (u32)e & 0xFFFFu;
        base = (u32)e >> 16;
    }
};

// one segment's decoder state with its stream ring (the body of dec_segments_ring, as an object, so that two of them
// can be advanced in turn)
template <typename LUT>
struct dec_ring_chain {
    static constexpr int RB = ANSX_RING_BYTES, CHK = ANSX_RING_CHK, R = 32 * CHK, NP = CHK / 2, T = RB - R - 16;
    u64 st, Lb;
    u32 q, logM, mask;
    int lo, lane_off;
    bool pending;
    ansx_u32x4 rr[NP];
    u32* ring;
    u8* ring8;
    u32* op;
    dec_ring_desc D;
    LUT lut;
    __device__ __forceinline__ void setup(u32* rings, u32 quad, u32 ql, u32 logM_, const dec_ring_desc& D_)
    {
        ring = rings + quad * (ANSX_RING_STRIDE / 4);
        ring8 = (u8*)ring;
        logM = logM_;
        mask = (1u << logM_) - 1;
        Lb = (u64)16 << logM_;
        lane_off = (R / 4) * (int)ql;
        D = D_;
    }
    // the quad's restart point and initial window as requested by dec_ring_prefetch (first segment) ...
    __device__ __forceinline__ void begin(const dec_ring_pre& P, ansx_u32x4 (&r)[RB / 64])
    {
        st = P.st;
        lo = P.lo;
        q = ANSX_DEC_Q(P.p);
#pragma unroll
        for (int j = 0; j < RB / 64; j++) r[j] = P.r[j];
    }
    // ... or requested now (further segments of the same quad)
    __device__ __forceinline__ void begin(const ansx_geo& g, u32 b, u32 seg, u32 ql, u32 sbytes, const u64* __restrict__ ckpt_state,
        const u32* __restrict__ ckpt_off, ansx_u32x4 (&r)[RB / 64])
    {
        const u64 idx = (u64)b * g.nckf + (seg - 1);
        u32 po;
        ckpt_load(g, ckpt_state, ckpt_off, idx, 3 - ql, &st, &po);
        const int p = (int)(po < sbytes ? po : sbytes);
        lo = (p - T) & ~(R / 4 - 1);
        q = ANSX_DEC_Q(p);
#pragma unroll
        for (int j = 0; j < RB / 64; j++) r[j] = ring_load16(D, lo + (RB / 4) * (int)ql + 16 * j, true);
    }
    __device__ __forceinline__ void fill(ansx_u32x4 (&r)[RB / 64], u32 ql)  // (behind the caller's vmcnt(0))
    {
#pragma unroll
        for (int j = 0; j < RB / 64; j++) {
            asm volatile("" : "+v"(r[j]));
            const u32 d = (u32)(lo + (RB / 4) * (int)ql + 16 * j) & (RB - 1);
            *(ansx_u32x4*)(ring8 + d) = r[j];
            if (d == 0) *(uint2*)(ring8 + RB) = make_uint2(r[j].x, r[j].y);  // second copy of ring bytes 0..7
        }
        pending = false;
#pragma unroll
        for (int j = 0; j < NP; j++) rr[j] = ansx_u32x4{ 0u, 0u, 0u, 0u };
    }
    __device__ __forceinline__ void land()
    {
        if (pending) {
            lo -= R;
            const u32 d0 = (u32)(lo + lane_off) & (RB - 1);
#pragma unroll
            for (int j = 0; j < NP; j++) *(ansx_u32x4*)(ring8 + d0 + 16 * j) = rr[j];
            if (d0 == 0) *(uint2*)(ring8 + RB) = make_uint2(rr[0].x, rr[0].y);
        }
    }
    __device__ __forceinline__ void request()
    {
        const int cur = ANSX_DEC_P(q);
        pending = (cur - lo) < T;
#pragma unroll
        for (int j = 0; j < NP; j++) rr[j] = ring_load16(D, lo - R + lane_off + 16 * j, pending);
    }
    __device__ __forceinline__ void keep()  // the request's registers stay live behind the caller's wait
    {
        if constexpr (NP == 2) asm volatile("" : "+v"(rr[0]), "+v"(rr[1]));
        else asm volatile("" : "+v"(rr[0]));
    }
    __device__ __forceinline__ u32 step(const dec_quad_const& qc)
    {
        return dec_step<2>(st, q, true, qc, logM, mask, Lb, lut, (const u8*)nullptr, ring);
    }
    // The same step (dec_step, rank/select table, ring stream) cut at its three LDS round trips, so that the caller can
    // alternate the stages of two chains: every stage ends by ISSUING its LDS reads, the next one starts by using them.
    u32 t_slot, t_sh, t_pv, t_nlo, t_nhi;
    u64 t_wpw, t_e, t_qs;
    u32 t_w0, t_w1, t_w2;
    bool t_rn;
    __device__ __forceinline__ void s1()  // slot -> bitmap entry
    {
        t_slot = (u32)st & mask;
        t_wpw = *(__attribute__((address_space(3))) const u64*)(size_t)(((t_slot >> 1) & 0x7FF0u) + LUT::OFFSET);
        t_qs = st >> logM;
    }
    __device__ __forceinline__ void s2()  // rank -> the symbol's entry
    {
        const u32 r8 = (u32)__builtin_popcount((u32)t_wpw << (~t_slot & 31u)) + (u32)(t_wpw >> 32);
        t_e = *(__attribute__((address_space(3))) const u64*)(size_t)(r8 << 3);
    }
    __device__ __forceinline__ void s3(const dec_quad_const& qc)  // state update, renormalise?, byte cursor -> stream window
    {
        const u32 fr = (u32)t_e & 0xFFFFu, base = (u32)t_e >> 16;
        t_pv = (u32)(t_e >> 32);
        const u64 t = (u64)fr * (u32)t_qs + (u64)(t_slot - base);  // ans_fold.hpp:218-220
        t_nlo = (u32)t;
        t_nhi = (u32)(t >> 32) + __umul24(fr, (u32)(t_qs >> 32));
        t_rn = t_nhi == 0 && t_nlo < (u32)Lb;
        const u32 k = t_pv >> 30;
        const u32 cq = (k << qc.ql8) + (t_rn ? qc.four_pos : 0u);
        const u32 s1_ = quad_add_dpp<0xB1>(cq);
        const u32 S = quad_add_dpp<0x4E>(s1_);
        const u32 a = 0u - __builtin_amdgcn_sad_u8(S & qc.lomask, 0u, q);
        q = __builtin_amdgcn_sad_u8(S, 0u, q);
        typedef __attribute__((address_space(3))) const u32 lds_cu32;
        const u32 rb = (u32)(size_t)(__attribute__((address_space(3))) const void*)ring;
        lds_cu32* wp = (lds_cu32*)(size_t)(rb + (a & (ANSX_RING_BYTES - 4)));
        t_w0 = wp[0], t_w1 = wp[1], t_w2 = wp[2];
        t_sh = a & 3;
    }
    __device__ __forceinline__ u32 s4()  // the renormalisation word / exception bytes, the value
    {
        const u32 lo = __builtin_amdgcn_alignbyte(t_w1, t_w0, t_sh);
        const u32 hi = __builtin_amdgcn_alignbyte(t_w2, t_w1, t_sh);
        const u32 s_lo = t_rn ? hi : t_nlo, s_hi = t_rn ? t_nlo : t_nhi;  // ans_fold.hpp:221-225
        st = ((u64)s_hi << 32) | s_lo;
        const u32 k = t_pv >> 30;
        const u32 e = (u32)((u64)(t_rn ? lo : hi) >> (32u - 8u * k));
        return (t_pv & ANSX_PV_MASK) + e;
    }
};

template <bool RFOLD>
__global__ __launch_bounds__(256) void k_decode_rank2(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, const u64* __restrict__ ckpt_state,
    const u32* __restrict__ ckpt_off, u64 payload_off, u32* __restrict__ outp, u32 maxM,
    u32 max_ns, u64 cont_bytes, const u32* __restrict__ g_cum, const uint4* __restrict__ binfo,
    u32* __restrict__ gflags)
{
    // (no static LDS in this kernel: the interleaved bitmap table sits at LDS address 0)
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    constexpr int RB = ANSX_RING_BYTES, CHK = ANSX_RING_CHK;
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const u32 bA = 2 * blockIdx.x, bB = bA + 1;
    const bool hasB = bB < g.nblocks;
    const u32 bBs = hasB ? bB : bA;  // (a valid index for the loads of an absent second block)
    const u32* gcA = g_cum + (u64)bA * (NSP + 8);
    const u32* gcB = g_cum + (u64)bBs * (NSP + 8);
    u32 curA[4], prvA[4], curB[4], prvB[4];
#pragma unroll
    for (u32 q = 0; q < 4; q++) {
        const u32 s = q * nt + tid;
        curA[q] = s < NSP ? gcA[s + 1] : 0u;
        prvA[q] = (s < NSP && s) ? gcA[s] : 0u;
        curB[q] = s < NSP ? gcB[s + 1] : 0u;
        prvB[q] = (s < NSP && s) ? gcB[s] : 0u;
    }
    const uint4 biA = binfo[bA], biB = binfo[bBs];
    const u64 boffA = block_off[bA], boffA1 = block_off[bA + 1], boffB1 = block_off[bBs + 1];
    const u64 boffB = hasB ? boffA1 : boffA;
    bool okA = biA.w == 0, okB = hasB && biB.w == 0;  // (a parse error is already flagged)
    const u32 logMA = biA.y, logMB = biB.y;
    const u32 nbA = geo_block_n(g, bA), nbB = geo_block_n(g, bBs);
    const u8* streamA = cont + payload_off + boffA;
    const u8* streamB = cont + payload_off + boffB;
    const u32 sbA = (u32)(boffA1 - boffA), sbB = (u32)(boffB1 - boffB);
    const u32 MA = 1u << logMA, MB = 1u << logMB;
    // LDS carve: [interleaved bitmap + prefix entries][entries A][entries B][scratch][rings A][rings B]
    const u32 wmax = maxM >= 32 ? maxM / 32 : 1;
    u32 off = 0;
    u32* bw = (u32*)(smem + off);
    off += wmax * 16;
    const u32 epb = 2 * ((max_ns * 4 + 15) & ~15u);
    uint2* epA = (uint2*)(smem + off);
    off += epb;
    uint2* epB = (uint2*)(smem + off);
    off += epb;
    u64* sh_scan = (u64*)(smem + off);
    u32& sh_bad = *(u32*)(smem + off + 80);
    off += ANSX_DEC_SCRATCH;
    u32* lds_rest = (u32*)(smem + off);
    const u32 labs = (u32)(size_t)(__attribute__((address_space(3))) void*)lds_rest;
    u32* ringsA = lds_rest + ((((labs + 15u) & ~15u) - labs) >> 2);
    u32* ringsB = ringsA + (nt >> 2) * (ANSX_RING_STRIDE / 4);
    const u32 WA = MA >= 32 ? MA / 32 : 1, WB = MB >= 32 ? MB / 32 : 1;
    // full blocks decode through rings (all segments have g.ckpt ints, host-checked); the container's one partial block
    // reads its stream straight from HBM
    const bool ringA = nbA == g.block_ints, ringB = hasB && nbB == g.block_ints;
    auto make_desc = [&](u64 boff_, u32 sbytes_) {
        dec_ring_desc D;
        const u64 sabs = payload_off + boff_;
        D.backoff = (int)(sabs < 1024 ? sabs : 1024);
        const u8* base = cont + (sabs - (u64)D.backoff);
        u64 span = (u64)D.backoff + sbytes_ + 64;
        const u64 room = cont_bytes - (sabs - (u64)D.backoff);
        if (span > room) span = room;
        const u64 ba = (u64)(uintptr_t)base;
        D.rsrc = ansx_u32x4{ (u32)ba, (u32)(ba >> 32) & 0xFFFFu, (u32)span, 0x00020000u };
        return D;
    };
    dec_ring_desc DA = make_desc(boffA, sbA), DB = make_desc(boffB, sbB);
    dec_ring_pre RPA, RPB;
#pragma unroll
    for (int j = 0; j < RB / 64; j++) RPA.r[j] = RPB.r[j] = ansx_u32x4{ 0u, 0u, 0u, 0u };
    RPA.st = RPB.st = 0;
    RPA.p = RPB.p = RPA.lo = RPB.lo = 0;
    if (okA && ringA) dec_ring_prefetch(RPA, g, bA, sbA, tid, logMA, streamA, DA, ckpt_state, ckpt_off);
    if (okB && ringB) dec_ring_prefetch(RPB, g, bB, sbB, tid, logMB, streamB, DB, ckpt_state, ckpt_off);
    {
        const u32 Wz = (WA > WB ? WA : WB) * 4;
        for (u32 w = tid; w < Wz; w += nt) bw[w] = 0u;
    }
    if (tid == 0) {
        sh_bad = 0;
        if ((u32)(size_t)(__attribute__((address_space(3))) void*)bw != 0u) atomicOr(&gflags[ANSX_G_ERR], 1u << 3);
    }
    __syncthreads();
    if (okA)
        okA = dec_build_rank_tables<RFOLD, 4, 0>(g, bw, epA, sh_scan, &sh_bad, gcA, curA, prvA, biA.x, MA, WA, max_ns, biA.z, streamA, tid, nt, gflags);
    if (okB) {
        if (tid == 0) sh_bad = 0;
        __syncthreads();
        okB = dec_build_rank_tables<RFOLD, 4, 2>(g, bw, epB, sh_scan, &sh_bad, gcB, curB, prvB, biB.x, MB, WB, max_ns, biB.z, streamB, tid, nt, gflags);
    }
    u32* oA = outp + (u64)bA * g.block_ints;
    u32* oB = outp + (u64)bBs * g.block_ints;
    dec_lut_rank2<0> lutA;
    dec_lut_rank2<8> lutB;
    if (okA && ringA && okB && ringB) {
        const u32 nseg = g.block_ints / g.ckpt;
        const u32 nq = nt >> 2, quad = tid >> 2, ql = tid & 3;
        const dec_quad_const qc = dec_make_qc(ql);
        const u32 steps = g.ckpt >> 2;
        dec_ring_chain<dec_lut_rank2<0>> A;
        dec_ring_chain<dec_lut_rank2<8>> B;
        A.setup(ringsA, quad, ql, logMA, DA);
        B.setup(ringsB, quad, ql, logMB, DB);
        for (u32 seg = quad; seg < nseg; seg += nq) {
            ansx_u32x4 ra[RB / 64], rb[RB / 64];
            if (seg == quad) {
                A.begin(RPA, ra);
                B.begin(RPB, rb);
            } else {
                A.begin(g, bA, seg, ql, sbA, ckpt_state, ckpt_off, ra);
                B.begin(g, bB, seg, ql, sbB, ckpt_state, ckpt_off, rb);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            A.fill(ra, ql);
            B.fill(rb, ql);
            A.op = oA + seg * g.ckpt + ql;
            B.op = oB + seg * g.ckpt + ql;
            u32 i = 0;
            for (; i + CHK <= steps; i += CHK) {
                // the two requests of the previous interval are older than its 2 CHK output stores
                if constexpr (CHK == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                A.keep();
                B.keep();
                A.land();
                B.land();
                A.request();
                B.request();
#pragma unroll
                for (u32 u = 0; u < (u32)CHK; u++) {
                    // stage by stage, the two chains in turn (the scheduling barriers keep hipcc from putting each chain's
                    // step back together: left alone it emits A's whole step, then B's, with a full LDS wait between)
                    A.s1();
                    B.s1();
                    __builtin_amdgcn_sched_barrier(0);
                    A.s2();
                    B.s2();
                    __builtin_amdgcn_sched_barrier(0);
                    A.s3(qc);
                    B.s3(qc);
                    __builtin_amdgcn_sched_barrier(0);
                    const u32 va = A.s4();
                    const u32 vb = B.s4();
                    A.op[4 * (i + u)] = va;
                    B.op[4 * (i + u)] = vb;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            A.keep();
            B.keep();
            A.land();
            B.land();
            for (; i < steps; i++) {
                const u32 va = A.step(qc);
                const u32 vb = B.step(qc);
                A.op[4 * i] = va;
                B.op[4 * i] = vb;
            }
        }
        return;
    }
    // anything else (the last workgroup of the grid: an odd block count, a partial last block; or a block that failed
    // its checks): whatever can be decoded is decoded one block at a time
    if (okA) {
        if (ringA) dec_segments_ring(g, bA, sbA, tid, nt, logMA, lutA, streamA, ringsA, DA, ckpt_state, ckpt_off, oA, RPA);
        else dec_segments<false>(g, bA, nbA, sbA, tid, nt, logMA, lutA, streamA, (const u32*)nullptr, ckpt_state, ckpt_off, oA);
    }
    if (okB) {
        if (ringB) dec_segments_ring(g, bB, sbB, tid, nt, logMB, lutB, streamB, ringsB, DB, ckpt_state, ckpt_off, oB, RPB);
        else dec_segments<false>(g, bB, nbB, sbB, tid, nt, logMB, lutB, streamB, (const u32*)nullptr, ckpt_state, ckpt_off, oB);
    }
    // nothing requested by dec_ring_prefetch may still be on its way into a register when the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < RB / 64; j++) asm volatile("" : "+v"(RPA.r[j]), "+v"(RPB.r[j]));
}

// k_decode: slot -> symbol table form, any frame size; tables in LDS (LDS_TAB) or in HBM.
template <bool LDS_TAB, bool RFOLD>
__global__ void k_decode(const u8* __restrict__ cont, ansx_geo g, u32 NSP,
    const u64* __restrict__ block_off, const u64* __restrict__ ckpt_state,
    const u32* __restrict__ ckpt_off, u64 payload_off, u32* __restrict__ outp, u32 maxM,
    u32 max_ns, u32 stream_cap, u16* __restrict__ g_s2s, u32* __restrict__ g_cum,
    const uint4* __restrict__ binfo, u32* __restrict__ gflags)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    __shared__ u32 sh_bad;
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const u32 b = blockIdx.x;
    const uint4 bi = binfo[b];
    if (bi.w) return;  // parse error already flagged
    const u32 ns = bi.x, logM = bi.y, rflag = bi.z;
    const u32 nb = geo_block_n(g, b);
    const u64 boff = block_off[b];
    const u8* stream = cont + payload_off + boff;
    const u32 sbytes = (u32)(block_off[b + 1] - boff);
    const ansx_map f = g.map;
    const u32 T = fold_T(g.f);
    const u32 M = 1u << logM;
    // LDS carve: [cum][s2s][most-frequent table][staged stream]
    const u32 cb = ((max_ns + 2) * 4 + 15) & ~15u;
    const u32 s2s_bytes = (u32)(((u64)maxM * 2 + 15) & ~15ull);
    u32* gc = g_cum + (u64)b * (NSP + 8);
    u32* cum;
    u16* s2s;
    u32* mfl = nullptr;
    u32* lds_stream;
    if (LDS_TAB) {
        cum = (u32*)smem;
        s2s = (u16*)(smem + cb);
        u32 off = cb + s2s_bytes;
        if (RFOLD) {
            mfl = (u32*)(smem + off);
            off += 4 * T;
        }
        lds_stream = (u32*)(smem + off);
    } else {
        cum = gc;
        s2s = g_s2s + (u64)b * maxM;
        u32 off = 0;
        if (RFOLD) {
            mfl = (u32*)smem;
            off = 4 * T;
        }
        lds_stream = (u32*)(smem + off);
    }
    if (tid == 0) sh_bad = 0;
    __syncthreads();
    // inc[s] = sum_{t<=s} nfreq[t] + s  (ans_util.hpp:33-41)  ->  cum[s+1] = inc[s] - s
    for (u32 s = tid; s < ns; s += nt) cum[s + 1] = gc[s + 1] - s;
    if (tid == 0) cum[0] = 0;
    if (RFOLD && rflag) {
        for (u32 i = tid; i < T; i += nt) mfl[i] = ld_u32_unaligned(stream + 4 + 4 * (u64)i);
    }
    const bool st_lds = (sbytes + 24 <= stream_cap);
    if (st_lds) dec_stage_stream(lds_stream, stream, sbytes, tid, nt);
    __threadfence_block();
    __syncthreads();
    {  // validation: monotone and sums to M
        u32 bad = 0;
        for (u32 s = tid; s < ns; s += nt) bad |= (cum[s + 1] < cum[s]) ? 1u : 0u;
        if (tid == 0 && cum[ns] != M) bad = 1;
        if (bad) {
            sh_bad = 1;
            atomicOr(&gflags[ANSX_G_ERR], 1u << 3);
        }
    }
    __syncthreads();
    if (sh_bad) return;
    {  // slot -> symbol table (the reference's table[M], ans_fold.hpp:190-204, 2 bytes per slot)
        const u32 per = (M + nt - 1) / nt;
        const u32 lo = tid * per;
        u32 hi = lo + per;
        hi = hi < M ? hi : M;
        if (lo < M) {
            u32 a = 0, z = ns;  // invariant cum[a] <= lo < cum[z]
            while (z - a > 1) {
                u32 mid = (a + z) >> 1;
                if (cum[mid] <= lo) a = mid;
                else z = mid;
            }
            u32 s = a;
            for (u32 slot = lo; slot < hi; slot++) {
                while (cum[s + 1] <= slot) s++;
                s2s[slot] = (u16)s;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    dec_lut_table lut;
    lut.cum = cum;
    lut.s2s = s2s;
    lut.f = f;
    lut.T = T;
    lut.mf = (RFOLD && rflag) ? mfl : nullptr;
    u32* o = outp + (u64)b * g.block_ints;
    if (st_lds)
        dec_segments<true>(g, b, nb, sbytes, tid, nt, logM, lut, stream, lds_stream, ckpt_state, ckpt_off, o);
    else
        dec_segments<false>(g, b, nb, sbytes, tid, nt, logM, lut, stream, lds_stream, ckpt_state, ckpt_off, o);
}

// ------------------------------------------------------------------------------------------
// Multi-GPU concatenation: the rank containers of contiguous whole-block ranges become ONE
// container.  Everything but the block index is a plain copy to a new offset; index entries are rebased
// by the payload bytes of the parts in front.  One launch: blockIdx.y = part, blockIdx.x = 64 KiB piece.
// ------------------------------------------------------------------------------------------
#define ANSX_MERGE_MAX_PARTS 64
struct ansx_merge_part {
    const u8* src;       // the part container
    u64 first_block;     // index of its first block in the merged container
    u64 pay_base;        // payload bytes of the parts in front of it
    u64 payload_bytes;
    u32 nblocks;
    u32 payload_off;     // inside the part
};
struct ansx_merge_desc {
    ansx_merge_part part[ANSX_MERGE_MAX_PARTS];
    u64 ckoff_off, ckstate_off, hint_off, payload_off;  // merged layout (index at 64)
    u32 nparts, nckf;
    u32 ckw;  // restart-point format of every part and of the result (ansx_dev.h)
};

// dst[0..n) = src[0..n), any alignment: dword stores on the aligned body of dst, unaligned dword loads
__device__ __forceinline__ void merge_copy(u8* __restrict__ dst, const u8* __restrict__ src, u64 n, u32 tid, u32 nt)
{
    u64 head = (u64)((4 - ((uintptr_t)dst & 3)) & 3);
    if (head > n) head = n;
    if (tid < head) dst[tid] = src[tid];
    const u64 nd = (n - head) >> 2;
    u32* d4 = (u32*)(dst + head);
    const u8* s1 = src + head;
    for (u64 j = tid; j < nd; j += nt) d4[j] = ld_u32_unaligned(s1 + 4 * j);
    const u64 done = head + 4 * nd;
    if (done + tid < n) dst[done + tid] = src[done + tid];
}

__global__ __launch_bounds__(256) void k_merge_containers(ansx_merge_desc D, u8* __restrict__ out)
{
    const u32 tid = threadIdx.x;
    const ansx_merge_part P = D.part[blockIdx.y];
    const u64 PIECE = 65536;
    // section sizes of this part, in the order they are walked by blockIdx.x
    // (packed restart points: one array of 29-byte records where the wide form has its cursors, no state array)
    const u64 cko_unit = D.ckw ? 4ull : (u64)ANSX_CK_RECORD;
    const u64 idx_bytes = 8ull * P.nblocks, cko_bytes = cko_unit * P.nblocks * D.nckf, cks_bytes = D.ckw ? 32ull * P.nblocks * D.nckf : 0ull;
    const u64 hint_bytes = 32ull * P.nblocks;
    const u64 n_idx = (idx_bytes + PIECE - 1) / PIECE, n_cko = (cko_bytes + PIECE - 1) / PIECE,
              n_cks = (cks_bytes + PIECE - 1) / PIECE, n_hint = (hint_bytes + PIECE - 1) / PIECE,
              n_pay = (P.payload_bytes + PIECE - 1) / PIECE;
    u64 piece = blockIdx.x;
    // part layout (make_plan): index at 64, restart offsets behind the nblocks + 1 index entries
    const u64 p_cko = 64 + 8ull * (P.nblocks + 1);
    const u64 p_cks = D.ckw ? (p_cko + cko_bytes + 7) / 8 * 8 : p_cko + cko_bytes;
    if (piece < n_idx) {  // block index, rebased
        const u64* src = (const u64*)(P.src + 64);
        u64* dst = (u64*)(out + 64) + P.first_block;
        const u64 lo = piece * (PIECE / 8);
        const u64 hi = lo + PIECE / 8 < P.nblocks ? lo + PIECE / 8 : P.nblocks;
        for (u64 i = lo + tid; i < hi; i += 256) dst[i] = src[i] + P.pay_base;
        return;
    }
    piece -= n_idx;
    if (piece < n_cko) {
        const u64 lo = piece * PIECE, len = cko_bytes - lo < PIECE ? cko_bytes - lo : PIECE;
        merge_copy(out + D.ckoff_off + cko_unit * P.first_block * D.nckf + lo, P.src + p_cko + lo, len, tid, 256);
        return;
    }
    piece -= n_cko;
    if (piece < n_cks) {
        const u64 lo = piece * PIECE, len = cks_bytes - lo < PIECE ? cks_bytes - lo : PIECE;
        merge_copy(out + D.ckstate_off + 32ull * P.first_block * D.nckf + lo, P.src + p_cks + lo, len, tid, 256);
        return;
    }
    piece -= n_cks;
    if (piece < n_hint) {
        const u64 p_hint = (p_cks + cks_bytes + 15) / 16 * 16;
        const u64 lo = piece * PIECE, len = hint_bytes - lo < PIECE ? hint_bytes - lo : PIECE;
        merge_copy(out + D.hint_off + 32ull * P.first_block + lo, P.src + p_hint + lo, len, tid, 256);
        return;
    }
    piece -= n_hint;
    if (piece < n_pay) {
        const u64 lo = piece * PIECE, len = P.payload_bytes - lo < PIECE ? P.payload_bytes - lo : PIECE;
        merge_copy(out + D.payload_off + P.pay_base + lo, P.src + P.payload_off + lo, len, tid, 256);
    }
}

// ansx -- the fast form of the per-block model (frame-size search + encoder table + prelude) used by calls whose
// geometry the context has seen before.  Same results as k_scale_attempts / k_select_model / k_write_prelude
// (ansx_kernels.h), which remain the exact path every deviation falls back to:
//
//   k_candidates     scale_freqs (ans_util.hpp:77-95) for NT frame sizes per block, one lane each
//   k_model_finish   cross entropy of every candidate (util.hpp:284-298), stop rule (ans_util.hpp:127-153), encoder
//                    table (ans_fold.hpp:82-91) and prelude (ans_util.hpp:46-63) of the chosen frame, one workgroup
//                    per block
//
// What makes it fast, and why the bytes cannot differ:
//  * The normalised frequencies S are integers produced by exactly the reference's double operations
//    (correctly rounded M_rem / fs_rem, * freq, + 0.5, truncation): the recurrence here is the same one, only
//    leaner -- the reciprocal of fs_rem is prepared per symbol once for all candidates of the block (fs_rem does
//    not depend on the frame size), and a candidate whose remaining frame goes negative simply runs on (the
//    remainder only decreases, so "M_rem != 0" at the end is the reference's failure test, ans_util.hpp:90-94).
//  * The stop rule compares XH = -sum p log2(S/M) with 1.001 H.  The reference sums both left to right in symbol
//    order; here XH is a tree sum in rank order, log2 M - (sum F log2 S) / n with log2 of the integers from a
//    table, and H is a tree sum too (k_fold_hist).  Both agree with the reference-order sums to ~1e-13 relative.
//    Every comparison must therefore clear the threshold by ANSX_FAST_GUARD = 1e-9 relative; a block that does
//    not (none was ever seen: candidates differ by >= 1e-4) raises the violation flag and the whole call is
//    repeated on the exact path, which is also where the 1e-12 "near threshold" accounting lives.
#pragma once

#include "ansx_kernels.h"

#ifdef ANSX_STAMPS_RF  // (development: the stamp rows belong to the ANSrfold remap kernel in this build)
#undef STAMP
#define STAMP(i) do { } while (0)
#endif

#define ANSX_FAST_GUARD 1e-9
#define ANSX_CAND_SL 64u         // symbols per LDS stage (NT = 5: 12.5 KB per wave, three 4-wave workgroups per CU)
#define ANSX_CAND_ROW (ANSX_CAND_SL + 1u)  // entries per block row (odd: rows start in different banks)
#define ANSX_CAND_MAXBPW 16u     // blocks per wave (NT >= 4 lanes per block)
#define ANSX_FIN_LUT 512u        // entries of the log2 table k_model_finish keeps in LDS (larger values: the table in HBM)

// log2 of the integers 0 .. 65535 (entry 0 = 0) with the portable log2: one table per context (512 KB)
__global__ void k_build_log2i_lut(double* __restrict__ lut)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < 65536u) lut[i] = i ? ansx_log2_portable((double)i) : 0.0;
}

// 1 / b exactly as ansx_div_int31 refines it (two Newton steps on the hardware seed)
__device__ __forceinline__ double ansx_rcp_int31(double b)
{
    const double y0 = __builtin_amdgcn_rcp(b);
    const double y1 = __builtin_fma(__builtin_fma(-b, y0, 1.0), y0, y0);
    return __builtin_fma(__builtin_fma(-b, y1, 1.0), y1, y1);
}

// Candidate frequencies in RANK order, 8 ranks per 16-byte chunk: srank[block][rank / 8][t][8] (u16), so that a
// lane stores one full chunk per 8 steps and the NT lanes of a block fill NT * 16 contiguous bytes.  (Stored per
// symbol -- [block][symbol][t], 2-byte stores -- the same values cost the CU's vector-memory pipe one cache line per
// block and STEP: 0.066 of this kernel's 0.18 ms on the headline workload.)
ANSX_HD u64 srank_chunk(u32 NSP, u32 NT, u32 b, u32 batch, u32 t) { return ((u64)b * (NSP / 8u) + batch) * NT + t; }

// NT lanes per block group (candidate frame sizes M0 * 2^t, t < NT).  A lane carries NCH independent recurrences --
// the same t of NCH blocks.  NCH = 1 while one chain per lane gives at most one wave per SIMD; beyond that NCH = 2,
// interleaved step by step: the step is a chain of 8 dependent f64 operations, and the 16384 x 5 chains of the
// headline workload are 1366 waves, more than the chip's 1024 SIMDs but not enough for two on each; two chains per
// lane halve the waves and fill each other's latency.  A wave covers NCH * (64 / NT) consecutive blocks; ANSX_CAND_WAVES independent waves per
// workgroup, each with its own LDS slice and never synchronised (single-wave workgroups pile up unevenly on a CU's
// SIMDs; waves of one workgroup land one per SIMD).
// pairs: k_sort_entropy's packed output.  attMeta: {ok, maxS} per (block, t).
#define ANSX_CAND_WAVES 4u
template <u32 NT, u32 NCH>  // compile-time: every staging load of a stage must be in flight at once (with a run-time trip count
                   // hipcc gives each load its own basic block and a full wait: 7 us per stage instead of one round trip)
__global__ __launch_bounds__(64 * ANSX_CAND_WAVES) void k_candidates(ansx_geo g, u32 NSP,
    const uint2* __restrict__ pairs, const ansx_blk* __restrict__ blk, uint4* __restrict__ srank,
    u32* __restrict__ attMeta)
{
    extern __shared__ double2 cand_lds_all[];  // per wave: [2 BPW][ANSX_CAND_ROW] { freq, reciprocal of fs_rem }
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    constexpr u32 BPW = 64u / NT;
    constexpr u32 rows = NCH * BPW;
    double2* const cand_lds = cand_lds_all + wv * rows * ANSX_CAND_ROW;
    const u32 bl = lane / NT, t = lane - bl * NT;
    const u32 wb0 = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * ANSX_CAND_WAVES + wv) * rows));
    if (wb0 >= g.nblocks) return;
    u32 bb[NCH], sigma[NCH], sh[NCH];
    bool live[NCH];
    double Md[NCH], fsd[NCH], mx[NCH];
    uint4* sp[NCH];
    const double2* row[NCH];
    u32 wsig = 0;
#pragma unroll
    for (u32 c = 0; c < NCH; c++) {
        bb[c] = wb0 + c * BPW + bl;
        live[c] = bl < BPW && bb[c] < g.nblocks;
        sigma[c] = 0, sh[c] = 0;
        double nd = 1.0;
        if (live[c]) {
            const ansx_blk* B = &blk[bb[c]];
            sigma[c] = B->sigma;
            sh[c] = B->m0_log2 + t;
            nd = (double)B->n;
        }
        wsig = sigma[c] > wsig ? sigma[c] : wsig;
        const bool dead = sh[c] > 31;  // frame sizes beyond 2^31 are unreachable for valid inputs
        Md[c] = dead ? -1.0 : (double)((u64)1 << (dead ? 0u : sh[c]));  // (a dead candidate fails: M_rem stays negative)
        fsd[c] = nd;
        mx[c] = 0.0;  // largest S so far
        sp[c] = srank + srank_chunk(NSP, NT, live[c] ? bb[c] : wb0, 0u, t);  // + batch * NT
        row[c] = cand_lds + (c * BPW + (bl < BPW ? bl : BPW - 1u)) * ANSX_CAND_ROW;
    }
    wsig = wave_max(wsig);  // the wave runs as long as its longest block
    // staging: the wave's 64 lanes fetch SL pairs of each of its blocks (coalesced) one stage ahead of the
    // recurrence and turn fs_rem into its reciprocal on the way into LDS
    constexpr u32 NITER = rows * (ANSX_CAND_SL / 64u);
    uint2 nxt[NITER];
    const u32 last_b = g.nblocks - 1u;
    auto fetch = [&](u32 c0) {
#pragma unroll
        for (u32 k = 0; k < NITER; k++) {
            const u32 i = k * 64u + lane;
            u32 b2 = wb0 + i / ANSX_CAND_SL;
            b2 = b2 < last_b ? b2 : last_b;  // (rows past the last block: any valid address, never consumed)
            u32 j = c0 + (i % ANSX_CAND_SL);
            j = j < NSP ? j : NSP - 1u;  // rows are NSP entries long
            nxt[k] = pairs[(u64)b2 * NSP + j];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (u32 k = 0; k < NITER; k++) {
            const u32 i = k * 64u + lane;
            const u32 rem = nxt[k].y ? nxt[k].y : 1u;  // (entries past a block's sigma are never consumed)
            cand_lds[(i / ANSX_CAND_SL) * ANSX_CAND_ROW + (i % ANSX_CAND_SL)] =
                double2{ (double)(nxt[k].x & 0xFFFFu), ansx_rcp_int31((double)rem) };
        }
    };
    // one step of scale_freqs (ans_util.hpp:80-92) of chain c
    auto step = [&](u32 c, const double2 e) -> u32 {
        // RN(M_rem / fs_rem): ansx_div_int31 with the prepared reciprocal
        const double q = Md[c] * e.y;
        const double r = __builtin_fma(-q, fsd[c], Md[c]);
        const double a = __builtin_fma(r, e.y, q);
        double v = a * e.x;
        v = 0.5 + v;
        v = __builtin_fmax(v, 1.0);  // (u32)v == 0 -> 1 (ans_util.hpp:86); also what a failed candidate keeps subtracting
        const double sd = __builtin_trunc(v);
        Md[c] = Md[c] - sd;
        fsd[c] = fsd[c] - e.x;
        mx[c] = __builtin_fmax(mx[c], sd);
        return (u32)sd;
    };
#ifdef ANSX_STAMPS
    unsigned long long tc0 = wall_clock64(), t_commit = 0, t_loop = 0, t_a = 0;
#endif
    fetch(0);
    for (u32 c0 = 0; c0 < wsig; c0 += ANSX_CAND_SL) {
#ifdef ANSX_STAMPS
        t_a = wall_clock64();
#endif
        wave_lds_sync();  // the previous stage has been consumed (one wave: LDS operations are in order)
        commit();
        if (c0 + ANSX_CAND_SL < wsig) fetch(c0 + ANSX_CAND_SL);
        wave_lds_sync();
        u32 lim[NCH];
#pragma unroll
        for (u32 c = 0; c < NCH; c++) lim[c] = sigma[c] > c0 ? (sigma[c] - c0 < ANSX_CAND_SL ? sigma[c] - c0 : ANSX_CAND_SL) : 0u;
        const u32 wlim = wsig - c0 < ANSX_CAND_SL ? wsig - c0 : ANSX_CAND_SL;
#ifdef ANSX_STAMPS
        { const unsigned long long tb = wall_clock64(); t_commit += tb - t_a; t_a = tb; }
#endif
        for (u32 j0 = 0; j0 < wlim; j0 += 8) {
            double2 e8[NCH][8];
#pragma unroll
            for (u32 c = 0; c < NCH; c++)
#pragma unroll
                for (int u = 0; u < 8; u++) e8[c][u] = row[c][j0 + u];  // (rows have SL + 1 entries, SL % 8 == 0: no overrun)
            u32 nS[NCH][8];
#pragma unroll
            for (u32 c = 0; c < NCH; c++)
#pragma unroll
                for (int u = 0; u < 8; u++) nS[c][u] = 0u;
            bool all_full = true;
#pragma unroll
            for (u32 c = 0; c < NCH; c++) all_full = all_full && (j0 + 8 <= lim[c]);
            if (all_full) {
                // The two chains operation by operation, in the written order (a scheduling barrier after every
                // line): hipcc's own schedule runs six dependent operations of one chain back to back, and with one
                // wave per SIMD nothing else fills their latency.  Every chain operation sits three slots behind
                // the one it depends on (the other chain's twin and one of the off-chain operations in between).
#define SB __builtin_amdgcn_sched_barrier(0)
                if constexpr (NCH == 1) {
#pragma unroll
                    for (int u = 0; u < 8; u++) nS[0][u] = step(0, e8[0][u]);
                } else {
                static_assert(NCH <= 2, "written out for two chains");
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const double2 eA = e8[0][u], eB = e8[NCH - 1][u];
                    const double qA = Md[0] * eA.y; SB;
                    const double qB = Md[NCH - 1] * eB.y; SB;
                    const double fA = fsd[0] - eA.x; SB;              // (off chain) next fs_rem
                    const double rA = __builtin_fma(-qA, fsd[0], Md[0]); SB;
                    const double rB = __builtin_fma(-qB, fsd[NCH - 1], Md[NCH - 1]); SB;
                    const double fB = fsd[NCH - 1] - eB.x; SB;
                    const double aA = __builtin_fma(rA, eA.y, qA); SB;
                    const double aB = __builtin_fma(rB, eB.y, qB); SB;
                    fsd[0] = fA;
                    fsd[NCH - 1] = fB;
                    double vA = aA * eA.x; SB;
                    double vB = aB * eB.x; SB;
                    vA = 0.5 + vA; SB;
                    vB = 0.5 + vB; SB;
                    vA = __builtin_fmax(vA, 1.0); SB;  // (u32)v == 0 -> 1 (ans_util.hpp:86)
                    vB = __builtin_fmax(vB, 1.0); SB;
                    const double sA = __builtin_trunc(vA); SB;
                    const double sB = __builtin_trunc(vB); SB;
                    Md[0] = Md[0] - sA; SB;
                    Md[NCH - 1] = Md[NCH - 1] - sB; SB;
                    mx[0] = __builtin_fmax(mx[0], sA); SB;
                    mx[NCH - 1] = __builtin_fmax(mx[NCH - 1], sB); SB;
                    nS[0][u] = (u32)sA; SB;
                    nS[NCH - 1][u] = (u32)sB; SB;
                }
                }
#undef SB
            } else {
#pragma unroll
                for (u32 c = 0; c < NCH; c++)
                    if (j0 < lim[c]) {
#pragma unroll
                        for (int u = 0; u < 8; u++)
                            if (j0 + u < lim[c]) nS[c][u] = step(c, e8[c][u]);
                    }
            }
#pragma unroll
            for (u32 c = 0; c < NCH; c++)
                if (j0 < lim[c])  // (values above 65535 end in the u16 exit and are never read: the low halves are stored)
                    sp[c][(u64)((c0 + j0) >> 3) * NT] = make_uint4((nS[c][0] & 0xFFFFu) | (nS[c][1] << 16), (nS[c][2] & 0xFFFFu) | (nS[c][3] << 16),
                        (nS[c][4] & 0xFFFFu) | (nS[c][5] << 16), (nS[c][6] & 0xFFFFu) | (nS[c][7] << 16));
        }
#ifdef ANSX_STAMPS
        t_loop += wall_clock64() - t_a;
#endif
    }
#ifdef ANSX_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        g_stamps[blockIdx.x * 16 + 12] = t_commit;
        g_stamps[blockIdx.x * 16 + 13] = t_loop;
        g_stamps[blockIdx.x * 16 + 14] = wall_clock64() - tc0;
    }
#endif
    // {ok, maxS}: the cross entropy of a successful candidate is k_model_finish's
#pragma unroll
    for (u32 c = 0; c < NCH; c++)
        if (live[c])
            *(uint2*)(attMeta + ((u64)bb[c] * ANSX_ATTEMPTS + t) * 4) =
                make_uint2((Md[c] == 0.0) ? 1u : 0u, mx[c] >= 4294967295.0 ? 0xFFFFFFFFu : (u32)mx[c]);
}

// Cross entropy of the NT candidates of k_candidates and the stop rule over them (guard band, see the header of this
// file), then the chosen frequencies by symbol, their exclusive scan (encoder table, compact 4-byte form; thread i
// owns IPT consecutive symbols) and the prelude (prelude_emit of ansx_kernels.h).  One workgroup of 256 threads per
// block.  XH_t = log2 M_t - (sum_j F_j log2 S_t,j) / n: thread i takes the 8-rank chunk i (and i + 256, ...) for ALL
// candidates -- their frequencies of a chunk are neighbours in memory -- so the chunk, the candidates' {ok, maxS} words
// and a thread's share of the log2 table's first ANSX_FIN_LUT entries (for LDS) are requested together with the block's
// fields, before anything is known about the block: one round trip for alphabets up to 2048 symbols, and the chosen
// candidate is later scattered -- rank order -> symbol order -- into LDS from those same registers.  (Until the end of
// round 3 a wave took one or two candidates and looped over the chunks with a load per iteration: a dependent round
// trip per 512 symbols in the sum and again in the scatter.)  NTC: compile-time bound of NT (5 or 8 registers sets).
// Anything this path does not cover raises the violation flag and leaves the block without a stream; the host
// repeats the call on the exact path: undecided after NT candidates, the u16 exit with no earlier success, a
// frame above 2^16, a comparison inside the guard band (`guard`: ANSX_FAST_GUARD; tests widen it to force the repeat).
template <int IPT, int NTC, int NTH = 256>
__global__ __launch_bounds__(NTH) void k_model_finish(ansx_geo g, u32 NSP, u32 NT, const uint2* __restrict__ pairs,
    const uint4* __restrict__ srank, const u32* __restrict__ attMeta, ansx_blk* __restrict__ blk,
    u32* __restrict__ tab32, u8* __restrict__ scratch, u64 scr_stride, const u32* __restrict__ mostfreq,
    u32* __restrict__ hints, u32* __restrict__ gflags, u32 cap, double guard, const double* __restrict__ lg2i,
    const uint2* __restrict__ geo, u32* __restrict__ incbuf = nullptr)
{
    static_assert(IPT % 4 == 0, "table rows are written 16 bytes at a time");
    static_assert(ANSX_FIN_LUT == 512 && (NTH == 256 || NTH == 64), "two table entries per thread (eight in the one-wave form)");
    static_assert(NTH == 256 || (IPT > 4), "the one-wave form takes every candidate in every lane");
    static_assert(NTC >= 4 && NTC <= (int)ANSX_ATTEMPTS, "candidates per block");
    extern __shared__ u32 lds32[];
    __shared__ u32 sh_part[8];
    __shared__ double lut[ANSX_FIN_LUT];
    __shared__ double wpart[ANSX_ATTEMPTS][4];  // per candidate and wave: partial sum of F log2 S (one wave: column 0)
    __shared__ double wsum[ANSX_ATTEMPTS];      // XH of every candidate
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const u32 b = blockIdx.x;
    ansx_blk* B = &blk[b];
    u32* off = lds32;             // [cap]
    u32* bits = lds32 + cap;      // bit buffer; until the prelude is written: the chosen frequencies by symbol
    // incbuf (IPT == 0): inc[] in HBM (the block's histogram row, free by now) instead of a third LDS array -- 8 instead
    // of 12 bytes of LDS per symbol, i.e. two workgroups per CU on 8000-symbol alphabets
    const bool inc_hbm = IPT == 0 && incbuf != nullptr;
    u32* inc = inc_hbm ? incbuf + (u64)b * NSP : lds32 + 2 * cap;   // [cap]
    const u32 dump = inc_hbm ? cap + 8u : 2u * cap + 8u;  // scratch word (relative to frq) behind the LDS arrays
    u32* frq = bits;
    STAMP(0);
    // ---- requests: everything the cross entropy of ALL candidates needs of 8-rank chunk `tid` -- its (F | sym << 16)
    // words and the NT candidates' frequencies, which lie next to each other (srank[block][chunk][t]) -- together with
    // the block's fields, before anything is known about the block: one round trip, whatever the alphabet up to 2048
    // symbols.  (NT <= NTC; candidates at or above NT are loaded from candidate 0's address and ignored.)
    const u32 ns = B->max_sym + 1;
    const u32 sigma = B->sigma;
    const double thr = B->thr;
    const u32 m0 = B->m0_log2;
    const double nd = (double)B->n;
    uint2 mt[ANSX_ATTEMPTS];
#pragma unroll
    for (u32 t = 0; t < ANSX_ATTEMPTS; t++) mt[t] = t < NT ? *(const uint2*)(attMeta + ((u64)b * ANSX_ATTEMPTS + t) * 4) : make_uint2(0u, 0u);
    double2 l2[256 / NTH];
#pragma unroll
    for (int i = 0; i < 256 / NTH; i++) l2[i] = *(const double2*)(lg2i + 2 * (tid + NTH * i));
    const uint2* prow = pairs + (u64)b * NSP;
    auto load_fs = [&](u32 c, u32 (&fs)[8]) {
        const uint2* pp = prow + (u64)c * 8u;
        const uint4 a0 = *(const uint4*)pp, a1 = *(const uint4*)(pp + 2), a2 = *(const uint4*)(pp + 4), a3 = *(const uint4*)(pp + 6);
        fs[0] = a0.x, fs[1] = a0.z, fs[2] = a1.x, fs[3] = a1.z, fs[4] = a2.x, fs[5] = a2.z, fs[6] = a3.x, fs[7] = a3.z;
    };
    auto load_s = [&](u32 c, uint4 (&sv)[NTC]) {
#pragma unroll
        for (int t = 0; t < NTC; t++) sv[t] = srank[srank_chunk(NSP, NT, b, c, (u32)t < NT ? (u32)t : 0u)];
    };
    // Alphabets up to 1024 slots (IPT == 4) keep the round-3 form -- wave w takes the candidates w and w + 4, lane l
    // the chunks l, l + 64 -- because the all-candidates form needs 96 registers instead of 66 (5 instead of 7 waves per
    // SIMD) and measured slower there (0.175 vs 0.155 ms); on 4096-slot alphabets it is the faster one (0.38 -> 0.34).
    constexpr bool ALLT = IPT > 4 || IPT == 0;  // (IPT == 0: any alphabet the LDS holds -- loops instead of register arrays)
    u32 fs0[8] = {};
    uint4 sv0[NTC] = {};
    if constexpr (ALLT) {
        if (tid < (NSP >> 3)) {  // (wave-uniform: NSP / 8 is a multiple of 64)
            load_fs(tid, fs0);
            load_s(tid, sv0);
        }
    }
    struct chunk {
        uint4 fa, fb;  // (freq | sym << 16) of the chunk's 8 ranks
        uint4 s;       // one candidate's 8 frequencies
    };
    auto load_chunk = [&](u32 c, u32 t) -> chunk {
        const uint2* pp = prow + (u64)c * 8u;
        const uint4 a0 = *(const uint4*)pp, a1 = *(const uint4*)(pp + 2), a2 = *(const uint4*)(pp + 4), a3 = *(const uint4*)(pp + 6);
        chunk k;
        k.fa = make_uint4(a0.x, a0.z, a1.x, a1.z);
        k.fb = make_uint4(a2.x, a2.z, a3.x, a3.z);
        k.s = srank[srank_chunk(NSP, NT, b, c, t)];
        return k;
    };
    const u32 t0 = wv, t1 = wv + 4u;
    chunk k0 = {}, k1 = {};
    if constexpr (!ALLT) {
        // first chunk of this lane for the wave's one or two candidates (chunk index lane < NSP / 8 always: NSP >= 512)
        k0 = load_chunk(lane, t0 < NT ? t0 : 0u);
        k1 = k0;  // (the same ranks: only the candidate's frequencies differ)
        if (t1 < NT) k1.s = srank[srank_chunk(NSP, NT, b, lane, t1)];
    }
    if (B->status) {  // (k_sort_entropy: alphabet above the hint -- violation already raised)
        if (tid == 0) B->prelude_bytes = 0;
        return;
    }
    for (u32 s = tid; s < cap; s += NTH) frq[s] = 0;  // absent symbols have frequency 0
#pragma unroll
    for (int i = 0; i < 256 / NTH; i++) *(double2*)(lut + 2 * (tid + NTH * i)) = l2[i];
    STAMP(1);
    __syncthreads();
    STAMP(2);
    // ---- cross entropy: XH_t = log2 M_t - (sum_j F_j log2 S_t,j) / n, every thread its chunk for all candidates
    const u32 nchunks = (sigma + 7u) >> 3;
    auto xh8 = [&](const u32 (&fs)[8], const uint4& sq) -> double {
        const u32 sv8[8] = { sq.x & 0xFFFFu, sq.x >> 16, sq.y & 0xFFFFu, sq.y >> 16, sq.z & 0xFFFFu, sq.z >> 16, sq.w & 0xFFFFu, sq.w >> 16 };
        // (a value that was above 65535 is seen truncated: such a candidate ends in the u16 exit before its XH is looked at)
        u32 big = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) big |= sv8[i];
        // (ranks past sigma in a block's last chunk were stored as S = 0, whose table entry is 0: no masking needed)
        double w = 0.0;
        if (big < ANSX_FIN_LUT) {
#pragma unroll
            for (int i = 0; i < 8; i++) w = __builtin_fma((double)(fs[i] & 0xFFFFu), lut[sv8[i]], w);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) w = __builtin_fma((double)(fs[i] & 0xFFFFu), lg2i[sv8[i]], w);
        }
        return w;
    };
    auto scatter = [&](const u32 (&fs)[8], const uint4& sq, u32 c) {
        const u32 sv8[8] = { sq.x & 0xFFFFu, sq.x >> 16, sq.y & 0xFFFFu, sq.y >> 16, sq.z & 0xFFFFu, sq.z >> 16, sq.w & 0xFFFFu, sq.w >> 16 };
        // (symbols of a block are below ns <= cap; ranks past sigma go to a scratch word behind the three arrays)
#pragma unroll
        for (int i = 0; i < 8; i++) frq[c * 8u + i < sigma ? fs[i] >> 16 : dump] = sv8[i];
    };
    if constexpr (ALLT) {
        double w[NTC];
#pragma unroll
        for (int t = 0; t < NTC; t++) w[t] = 0.0;
        if (tid < nchunks) {  // (nchunks <= NSP / 8: the chunk was loaded)
#pragma unroll
            for (int t = 0; t < NTC; t++)
                if ((u32)t < NT) w[t] = xh8(fs0, sv0[t]);
        }
        for (u32 c = tid + NTH; c < nchunks; c += NTH) {  // alphabets above 8 NTH symbols: a second round trip
            u32 fs1[8];
            uint4 s1[NTC];
            load_fs(c, fs1);
            load_s(c, s1);
#pragma unroll
            for (int t = 0; t < NTC; t++)
                if ((u32)t < NT) w[t] = w[t] + xh8(fs1, s1[t]);
        }
        if (wv * 64u < nchunks || nchunks > (u32)NTH) {  // (waves without a chunk have nothing to add)
#pragma unroll
            for (int t = 0; t < NTC; t++) {
                if ((u32)t >= NT) continue;
                const double ws = wave_sum(w[t]);
                if (lane == 0) wpart[t][wv] = ws;
            }
        } else if (lane < NTC) wpart[lane][wv] = 0.0;
        STAMP(3);
        __syncthreads();
        if (tid < NT) wsum[tid] = (double)(m0 + tid) - (NTH == 64 ? wpart[tid][0] : (wpart[tid][0] + wpart[tid][1]) + (wpart[tid][2] + wpart[tid][3])) / nd;  // one division per candidate
        __syncthreads();
    } else {
        auto xh_chunk = [&](const chunk& k) -> double {
            const u32 fs8[8] = { k.fa.x, k.fa.y, k.fa.z, k.fa.w, k.fb.x, k.fb.y, k.fb.z, k.fb.w };
            return xh8(fs8, k.s);
        };
        double w0 = 0.0, w1 = 0.0;
        if (t0 < NT) {
            if (lane < nchunks) w0 = xh_chunk(k0);
            for (u32 c = lane + 64u; c < nchunks; c += 64u) w0 = w0 + xh_chunk(load_chunk(c, t0));
            w0 = wave_sum(w0);
            if (lane == 0) wsum[t0] = (double)(m0 + t0) - w0 / nd;  // XH of candidate t0 (one division per wave, not per thread and candidate)
        }
        if (t1 < NT) {
            if (lane < nchunks) w1 = xh_chunk(k1);
            for (u32 c = lane + 64u; c < nchunks; c += 64u) w1 = w1 + xh_chunk(load_chunk(c, t1));
            w1 = wave_sum(w1);
            if (lane == 0) wsum[t1] = (double)(m0 + t1) - w1 / nd;
        }
        STAMP(3);
        __syncthreads();
    }
    STAMP(4);
    // ---- the (workgroup-uniform) rule, on the same NT results in every thread
    int chosen = -2, prev = -1;
    bool unsure = false;
#pragma unroll
    for (u32 t = 0; t < ANSX_ATTEMPTS; t++) {
        if (t >= NT || chosen != -2 || unsure) continue;
        if (!mt[t].x) continue;  // scale_freqs failed: M *= 2 (ans_util.hpp:131-135)
        if (mt[t].y >= ANSX_U16_LIMIT) {  // ans_util.hpp:141-145
            chosen = prev;
            continue;
        }
        const double XH = wsum[t];
        const double d = XH - thr;
        if ((d < 0 ? -d : d) <= guard * thr || !(thr > 0.0)) unsure = true;  // (H == 0: one-symbol block, exact path)
        else if (XH < thr) chosen = (int)t;  // ans_util.hpp:149
        else prev = (int)t;
    }
    const u32 logM = m0 + (u32)(chosen < 0 ? 0 : chosen);
    if (unsure || chosen < 0 || logM > 16 || ns > cap) {
        if (tid == 0) {
            B->prelude_bytes = 0;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
        }
        return;
    }
    // ---- chosen frequencies: rank order -> symbol order through LDS, from the registers that hold them
    if constexpr (ALLT) {
        if (tid < nchunks) {
            uint4 sc = sv0[0];
#pragma unroll
            for (int t = 1; t < NTC; t++)
                if (chosen == t) sc = sv0[t];
            scatter(fs0, sc, tid);
        }
        for (u32 c = tid + NTH; c < nchunks; c += NTH) {
            u32 fs1[8];
            load_fs(c, fs1);
            scatter(fs1, srank[srank_chunk(NSP, NT, b, c, (u32)chosen)], c);
        }
    } else if ((u32)chosen == t0 || (u32)chosen == t1) {  // ... by the wave that holds them
        auto scatter_chunk = [&](const chunk& k, u32 c) {
            const u32 fs8[8] = { k.fa.x, k.fa.y, k.fa.z, k.fa.w, k.fb.x, k.fb.y, k.fb.z, k.fb.w };
            scatter(fs8, k.s, c);
        };
        if (lane < nchunks) scatter_chunk((u32)chosen == t0 ? k0 : k1, lane);
        for (u32 c = lane + 64u; c < nchunks; c += 64u) scatter_chunk(load_chunk(c, (u32)chosen), c);
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
    u32 total;
    if constexpr (IPT > 0) {
        const u32 s0 = tid * IPT;
        u32 fr[IPT > 0 ? IPT : 4];
        u32 sum = 0;
#pragma unroll
        for (int i = 0; i < IPT; i += 4) {
            uint4 f4 = make_uint4(0u, 0u, 0u, 0u);
            if (s0 + i < cap) f4 = *(const uint4*)(frq + s0 + i);  // cap is a multiple of 16
            fr[i] = f4.x, fr[i + 1] = f4.y, fr[i + 2] = f4.z, fr[i + 3] = f4.w;
            sum += f4.x + f4.y + f4.z + f4.w;
        }
        u32 base = block_excl_scan<u32>(sum, sh_part, tid, NTH, &total);
        STAMP(7);
        if (s0 < ns) {
            u32* t32 = tab32 + (u64)b * NSP + s0;
#pragma unroll
            for (int i = 0; i < IPT; i += 4) {
                u32 w4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 s = s0 + i + k;
                    w4[k] = (base << 16) | fr[i + k];  // valid while M <= 65536 (base < 2^16, freq < 65535)
                    if (s < ns) inc[s] = base + fr[i + k] + s;  // ans_util.hpp:54-58: inc[s] = inc[s-1] + nfreq[s] + 1
                    base += fr[i + k];
                }
                *(uint4*)(t32 + i) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
    } else {
        // any alphabet: thread i owns `per` consecutive symbols (a multiple of 4) and walks them twice
        const u32 per = ((cap + 255u) / 256u + 3u) & ~3u;
        const u32 s0 = tid * per;
        u32 sum = 0;
        for (u32 i = 0; i < per && s0 + i < cap; i += 4) {
            const uint4 f4 = *(const uint4*)(frq + s0 + i);  // cap is a multiple of 16
            sum += f4.x + f4.y + f4.z + f4.w;
        }
        u32 base = block_excl_scan<u32>(sum, sh_part, tid, 256, &total);
        STAMP(7);
        u32* t32 = tab32 + (u64)b * NSP + s0;
        for (u32 i = 0; i < per && s0 + i < cap && s0 + i < ((ns + 3u) & ~3u); i += 4) {
            const uint4 f4 = *(const uint4*)(frq + s0 + i);
            const u32 f[4] = { f4.x, f4.y, f4.z, f4.w };
            u32 w4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 s = s0 + i + k;
                w4[k] = (base << 16) | f[k];
                if (s < ns) inc[s] = base + f[k] + s;
                base += f[k];
            }
            *(uint4*)(t32 + i) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    if (tid == 0) {
        B->logM = logM;
        B->resolved = 1;
        if (total != (1u << logM)) atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);  // (cannot happen: the candidate summed to M)
        // same-address atomics serialise in L2 (16 K blocks): only the few blocks that raise a running maximum issue one
        if (__hip_atomic_load(&gflags[ANSX_G_MAXLOGM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < logM)
            atomicMax(&gflags[ANSX_G_MAXLOGM], logM);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXNSYMS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ns)
            atomicMax(&gflags[ANSX_G_MAXNSYMS], ns);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXSIGMA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sigma)
            atomicMax(&gflags[ANSX_G_MAXSIGMA], sigma);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)chosen)
            atomicMax(&gflags[ANSX_G_MAXT], (u32)chosen);
    }
    STAMP(8);
    if (inc_hbm) __threadfence_block();  // (inc[] went to HBM: visible to the workgroup behind the barrier)
    __syncthreads();  // frq (= bits) has been read by everyone; inc[] is complete
    STAMP(9);
#ifndef FIN_NO_PRELUDE
    // (IPT == 0 serves alphabets up to 16384 slots here: 64 items per thread at most, their codes kept in registers)
    prelude_emit<IPT, false, true, IPT == 0 ? 8 : 0, NTH>(g, B, ns, logM, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, mostfreq, b, tid, 0, hints, geo);  // (logM <= 16)
#endif
    STAMP(10);
}

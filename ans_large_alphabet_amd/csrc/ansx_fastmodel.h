// ansx -- the fast form of the per-block model (frame-size search + encoder table + prelude) used by calls whose
// geometry the context has seen before.  Same results as k_scale_attempts / k_select_model / k_write_prelude
// (ansx_kernels.h), which remain the exact path every deviation falls back to:
//
//   k_candidates     scale_freqs (ans_util.hpp:77-95) for NT frame sizes per block, one lane each, and the cross
//                    entropy of every candidate (util.hpp:284-298) accumulated in the same pass
//   k_model_finish   stop rule (ans_util.hpp:127-153), encoder table (ans_fold.hpp:82-91) and prelude
//                    (ans_util.hpp:46-63) of the chosen frame in one workgroup per block
//
// What makes it fast, and why the bytes cannot differ:
//  * The normalised frequencies S are integers produced by exactly the reference's double operations
//    (correctly rounded M_rem / fs_rem, * freq, + 0.5, truncation): the recurrence here is the same one, only
//    leaner -- the reciprocal of fs_rem is prepared per symbol once for all candidates of the block (fs_rem does
//    not depend on the frame size), and a candidate whose remaining frame goes negative simply runs on (the
//    remainder only decreases, so "M_rem != 0" at the end is the reference's failure test, ans_util.hpp:90-94).
//  * The stop rule compares XH = -sum p log2(S/M) with 1.001 H.  The reference sums both left to right in symbol
//    order; here XH is accumulated in rank order as log2 M - (sum F log2 S) / n from a table of log2 of the
//    integers, and H is a tree sum (k_fold_hist).  Both agree with the reference-order sums to ~1e-13 relative.
//    Every comparison must therefore clear the threshold by ANSX_FAST_GUARD = 1e-9 relative; a block that does
//    not (none was ever seen: candidates differ by >= 1e-4) raises the violation flag and the whole call is
//    repeated on the exact path, which is also where the 1e-12 "near threshold" accounting lives.
#pragma once

#include "ansx_kernels.h"

#define ANSX_FAST_GUARD 1e-9
#define ANSX_CAND_SL 64u         // symbols per LDS stage (NT = 5: 12.5 KB per wave, three 4-wave workgroups per CU)
#define ANSX_CAND_ROW (ANSX_CAND_SL + 1u)  // entries per block row (odd: rows start in different banks)
#define ANSX_CAND_MAXBPW 16u     // blocks per wave (NT >= 4 lanes per block)

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

// NT lanes per block (candidate frame sizes M0 * 2^t, t < NT), 64 / NT blocks per wave; ANSX_CAND_WAVES independent
// waves per workgroup, each with its own blocks and LDS slice and never synchronised (single-wave workgroups pile up
// unevenly on a CU's SIMDs; waves of one workgroup land one per SIMD).
// pairs: k_sort_entropy's packed output.  attMeta: {ok, maxS, XH bits} per (block, t) as k_scale_attempts writes it.
#define ANSX_CAND_WAVES 4u
__global__ __launch_bounds__(64 * ANSX_CAND_WAVES) void k_candidates(ansx_geo g, u32 NSP, u32 NT,
    const uint2* __restrict__ pairs, const ansx_blk* __restrict__ blk, uint4* __restrict__ srank,
    u32* __restrict__ attMeta, const double* __restrict__ lg2i)
{
    extern __shared__ uint4 cand_lds_all[];  // per wave: [BPW][ANSX_CAND_ROW] { freq | sym << 16, -, reciprocal of fs_rem }
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 BPW = 64u / NT;
    uint4* const cand_lds = cand_lds_all + wv * BPW * ANSX_CAND_ROW;
    const u32 bl = lane / NT, t = lane - bl * NT;
    const u32 wb0 = (u32)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * ANSX_CAND_WAVES + wv) * BPW));
    if (wb0 >= g.nblocks) return;
    const u32 b = wb0 + bl;
    const bool live = bl < BPW && b < g.nblocks;
    u32 sigma = 0, sh = 0;
    double nd = 1.0;
    if (live) {
        const ansx_blk* B = &blk[b];
        sigma = B->sigma;
        sh = B->m0_log2 + t;
        nd = (double)B->n;
    }
    u32 wsig = sigma;  // the wave runs as long as its longest block
    for (int o = 32; o > 0; o >>= 1) {
        const u32 x = (u32)__shfl_xor((int)wsig, o);
        wsig = x > wsig ? x : wsig;
    }
    const bool dead = sh > 31;  // frame sizes beyond 2^31 are unreachable for valid inputs
    double Md = dead ? -1.0 : (double)((u64)1 << (dead ? 0u : sh));  // (a dead candidate fails: M_rem stays negative)
    double fsd = nd;
    double mx = 0.0;   // largest S so far
    double W = 0.0;    // sum F * log2(S)
    uint4* const sp = srank + srank_chunk(NSP, NT, live ? b : wb0, 0u, t);  // + batch * NT
    const uint4* const row = cand_lds + (bl < BPW ? bl : BPW - 1u) * ANSX_CAND_ROW;

    // staging: the wave's 64 lanes fetch SL pairs of each of its blocks (coalesced) one stage ahead of the
    // recurrence and turn fs_rem into its reciprocal on the way into LDS
    const u32 niter = BPW * (ANSX_CAND_SL / 64u);  // <= 32
    uint2 nxt[ANSX_CAND_MAXBPW * (ANSX_CAND_SL / 64u)];
    auto fetch = [&](u32 c0) {
#pragma unroll
        for (u32 k = 0; k < ANSX_CAND_MAXBPW * (ANSX_CAND_SL / 64u); k++) {
            if (k < niter) {
                const u32 i = k * 64u + lane;
                const u32 bb = wb0 + i / ANSX_CAND_SL;
                u32 j = c0 + (i % ANSX_CAND_SL);
                j = j < NSP ? j : NSP - 1u;  // rows are NSP entries long
                nxt[k] = bb < g.nblocks ? pairs[(u64)bb * NSP + j] : make_uint2(0u, 1u);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (u32 k = 0; k < ANSX_CAND_MAXBPW * (ANSX_CAND_SL / 64u); k++) {
            if (k < niter) {
                const u32 i = k * 64u + lane;
                const u32 rem = nxt[k].y ? nxt[k].y : 1u;  // (entries past a block's sigma are never consumed)
                const double y = ansx_rcp_int31((double)rem);
                const u64 yb = ansx_f64_to_bits(y);
                cand_lds[(i / ANSX_CAND_SL) * ANSX_CAND_ROW + (i % ANSX_CAND_SL)] = make_uint4(nxt[k].x, 0u, (u32)yb, (u32)(yb >> 32));
            }
        }
    };
    // one step of scale_freqs (ans_util.hpp:80-92) + this symbol's cross-entropy weight
    double pF[8], pL[8];  // pending: frequency and log2(S) of the previous 8 steps (table loads in flight)
#pragma unroll
    for (int u = 0; u < 8; u++) pF[u] = 0.0, pL[u] = 0.0;
    auto step = [&](const uint4 e, double& Fd_out, u32& sc_out) {
        const double Fd = (double)(e.x & 0xFFFFu);
        const double y = ansx_bits_to_f64((u64)e.z | ((u64)e.w << 32));
        // RN(M_rem / fs_rem): ansx_div_int31 with the prepared reciprocal
        const double q = Md * y;
        const double r = __builtin_fma(-q, fsd, Md);
        const double a = __builtin_fma(r, y, q);
        double v = a * Fd;
        v = 0.5 + v;
        v = __builtin_fmax(v, 1.0);  // (u32)v == 0 -> 1 (ans_util.hpp:86); also what a failed candidate keeps subtracting
        const double sd = __builtin_trunc(v);
        Md = Md - sd;
        fsd = fsd - Fd;
        mx = __builtin_fmax(mx, sd);
        Fd_out = Fd;
        sc_out = (u32)sd;
    };
    fetch(0);
    for (u32 c0 = 0; c0 < wsig; c0 += ANSX_CAND_SL) {
        wave_lds_sync();  // the previous stage has been consumed (one wave: LDS operations are in order)
        commit();
        if (c0 + ANSX_CAND_SL < wsig) fetch(c0 + ANSX_CAND_SL);
        wave_lds_sync();
        const u32 lim = sigma > c0 ? (sigma - c0 < ANSX_CAND_SL ? sigma - c0 : ANSX_CAND_SL) : 0u;
        const u32 wlim = wsig - c0 < ANSX_CAND_SL ? wsig - c0 : ANSX_CAND_SL;
        for (u32 j0 = 0; j0 < wlim; j0 += 8) {
            uint4 e8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) e8[u] = row[j0 + u];  // (rows have SL + 1 entries, SL % 8 == 0: no overrun)
            double nF[8];
            u32 nS[8];
#pragma unroll
            for (int u = 0; u < 8; u++) nF[u] = 0.0, nS[u] = 0u;
            if (j0 + 8 <= lim) {
#pragma unroll
                for (int u = 0; u < 8; u++) step(e8[u], nF[u], nS[u]);
            } else if (j0 < lim) {
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (j0 + u < lim) step(e8[u], nF[u], nS[u]);
            }
            // No vector-memory operation is issued inside the steps: the table loads of the previous batch are the
            // youngest ones outstanding here (vmcnt is one in-order counter for loads and stores -- a wait placed
            // behind this batch's store would drain it), and they have had the whole batch to arrive.
#pragma unroll
            for (int u = 0; u < 8; u++) W = __builtin_fma(pF[u], pL[u], W);
#ifndef CAND_NO_STORE
            if (j0 < lim)  // (values above 65535 end in the u16 exit and are never read: the low halves are stored)
                sp[(u64)((c0 + j0) >> 3) * NT] = make_uint4((nS[0] & 0xFFFFu) | (nS[1] << 16), (nS[2] & 0xFFFFu) | (nS[3] << 16),
                    (nS[4] & 0xFFFFu) | (nS[5] << 16), (nS[6] & 0xFFFFu) | (nS[7] << 16));
#endif
#pragma unroll
            for (int u = 0; u < 8; u++) {
                pF[u] = nF[u];
#ifndef CAND_NO_LUT
                pL[u] = lg2i[nS[u] < 65535u ? nS[u] : 65535u];
#else
                pL[u] = (double)nS[u];
#endif
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) W = __builtin_fma(pF[u], pL[u], W);
    if (!live) return;
    u32* meta = attMeta + ((u64)b * ANSX_ATTEMPTS + t) * 4;
    const u32 ok = (Md == 0.0) ? 1u : 0u;
    // XH = -sum (F/n) log2(S / 2^sh) = sh - (sum F log2 S) / n   (sum F = n: every counted symbol has S >= 1)
    const double XH = (double)sh - W / nd;
    const u64 xb = ansx_f64_to_bits(XH);
    *(uint4*)meta = make_uint4(ok, mx >= 4294967295.0 ? 0xFFFFFFFFu : (u32)mx, (u32)xb, (u32)(xb >> 32));
}

// Stop rule over the NT candidates of k_candidates (guard band, see the header of this file), then -- one workgroup
// of 256 threads per block -- the chosen frequencies by symbol, their exclusive scan (encoder table, compact 4-byte
// form; thread i owns IPT consecutive symbols) and the prelude (prelude_emit of ansx_kernels.h).
// Everything the kernel needs is requested before anything is decided (the block's fields, all candidates' results,
// the rank -> symbol pairs and EVERY candidate's chunk of a thread's 8 ranks): one round trip, then registers.
// Anything this path does not cover raises the violation flag and leaves the block without a stream; the host
// repeats the call on the exact path: undecided after NT candidates, the u16 exit with no earlier success, a
// frame above 2^16, a comparison inside the guard band (`guard`: ANSX_FAST_GUARD; tests widen it to force the repeat).
template <int IPT>
__global__ __launch_bounds__(256) void k_model_finish(ansx_geo g, u32 NSP, u32 NT, const uint2* __restrict__ pairs,
    const uint4* __restrict__ srank, const u32* __restrict__ attMeta, ansx_blk* __restrict__ blk,
    u32* __restrict__ tab32, u8* __restrict__ scratch, u64 scr_stride, const u32* __restrict__ mostfreq,
    u32* __restrict__ hints, u32* __restrict__ gflags, u32 cap, double guard)
{
    static_assert(IPT % 4 == 0, "table rows are written 16 bytes at a time");
    constexpr int RPT = (IPT + 7) / 8;  // 8-rank chunks per thread: 256 * 8 * RPT ranks >= 256 * IPT symbols
    extern __shared__ u32 lds32[];
    __shared__ u32 sh_part[8];
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    ansx_blk* B = &blk[b];
    u32* off = lds32;             // [cap]
    u32* bits = lds32 + cap;      // bit buffer; until the prelude is written: the chosen frequencies by symbol
    u32* inc = lds32 + 2 * cap;   // [cap]
    u32* frq = bits;
    // ---- requests
    const u32 ns = B->max_sym + 1;
    const u32 sigma = B->sigma;
    const double thr = B->thr;
    const u32 m0 = B->m0_log2;
    uint4 mt[ANSX_ATTEMPTS];
#pragma unroll
    for (u32 t = 0; t < ANSX_ATTEMPTS; t++) mt[t] = t < NT ? *(const uint4*)(attMeta + ((u64)b * ANSX_ATTEMPTS + t) * 4) : make_uint4(0u, 0u, 0u, 0u);
    uint4 py[RPT][2];            // (freq | sym << 16) of this thread's ranks
    uint4 ch[RPT][ANSX_ATTEMPTS];  // every candidate's chunk of them
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const u32 batch = tid + 256u * r;  // ranks 8 * batch .. + 7
        const bool in = batch * 8u < NSP;
        const uint2* pp = pairs + (u64)b * NSP + (in ? batch * 8u : 0u);
        const uint4 a0 = *(const uint4*)pp, a1 = *(const uint4*)(pp + 2), a2 = *(const uint4*)(pp + 4), a3 = *(const uint4*)(pp + 6);
        py[r][0] = make_uint4(a0.x, a0.z, a1.x, a1.z);
        py[r][1] = make_uint4(a2.x, a2.z, a3.x, a3.z);
#pragma unroll
        for (u32 t = 0; t < ANSX_ATTEMPTS; t++)
            ch[r][t] = (t < NT && in) ? srank[srank_chunk(NSP, NT, b, batch, t)] : make_uint4(0u, 0u, 0u, 0u);
    }
    for (u32 s = tid; s < cap; s += 256) frq[s] = 0;  // absent symbols have frequency 0
    // ---- the (wave-uniform) rule, on the same NT results in every thread
    int chosen = -2, prev = -1;
    bool unsure = false;
#pragma unroll
    for (u32 t = 0; t < ANSX_ATTEMPTS; t++) {
        if (t >= NT || chosen != -2 || unsure) continue;
        if (!mt[t].x) continue;  // scale_freqs failed: M *= 2 (ans_util.hpp:131-135)
        if (mt[t].y >= ANSX_U16_LIMIT) {  // ans_util.hpp:141-145
            chosen = prev;
            if (chosen == -2) chosen = -1;
            continue;
        }
        const double XH = ansx_bits_to_f64((u64)mt[t].z | ((u64)mt[t].w << 32));
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
    __syncthreads();
    // ---- chosen frequencies: rank order -> symbol order through LDS
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        uint4 c = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (u32 t = 0; t < ANSX_ATTEMPTS; t++)
            if ((int)t == chosen) c = ch[r][t];
        const u32 j0 = (tid + 256u * r) * 8u;
        const u32 fs8[8] = { py[r][0].x, py[r][0].y, py[r][0].z, py[r][0].w, py[r][1].x, py[r][1].y, py[r][1].z, py[r][1].w };
        const u32 sv8[8] = { c.x & 0xFFFFu, c.x >> 16, c.y & 0xFFFFu, c.y >> 16, c.z & 0xFFFFu, c.z >> 16, c.w & 0xFFFFu, c.w >> 16 };
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (j0 + k < sigma) frq[fs8[k] >> 16] = sv8[k];  // (symbols of a block are below ns <= cap)
    }
    __syncthreads();
    const u32 s0 = tid * IPT;
    u32 fr[IPT];
    u32 sum = 0;
#pragma unroll
    for (int i = 0; i < IPT; i += 4) {
        uint4 f4 = make_uint4(0u, 0u, 0u, 0u);
        if (s0 + i < cap) f4 = *(const uint4*)(frq + s0 + i);  // cap is a multiple of 16
        fr[i] = f4.x, fr[i + 1] = f4.y, fr[i + 2] = f4.z, fr[i + 3] = f4.w;
        sum += f4.x + f4.y + f4.z + f4.w;
    }
    u32 total;
    u32 base = block_excl_scan<u32>(sum, sh_part, tid, 256, &total);
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
    if (tid == 0) {
        B->logM = logM;
        B->resolved = 1;
        if (total != (1u << logM)) atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);  // (cannot happen: the candidate summed to M)
        // same-address atomics serialise in L2 (16 K blocks): only the few blocks that raise a running maximum issue one
        if (__hip_atomic_load(&gflags[ANSX_G_MAXLOGM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < logM)
            atomicMax(&gflags[ANSX_G_MAXLOGM], logM);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXNSYMS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ns)
            atomicMax(&gflags[ANSX_G_MAXNSYMS], ns);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)chosen)
            atomicMax(&gflags[ANSX_G_MAXT], (u32)chosen);
    }
    __syncthreads();  // frq (= bits) has been read by everyone; inc[] is complete
    prelude_emit<IPT>(g, B, ns, logM, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, mostfreq, b, tid, 0, hints);
}

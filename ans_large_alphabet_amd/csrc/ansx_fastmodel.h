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
#define ANSX_CAND_SL 128u        // symbols per LDS stage
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

// One wave per workgroup; NT lanes per block (candidate frame sizes M0 * 2^t, t < NT), 64 / NT blocks per wave.
// pairs: k_sort_entropy's packed output.  attS / attMeta: as k_scale_attempts writes them ([block][symbol][8],
// {ok, maxS, XH bits}), so that the exact path's readers and tests see one format.
__global__ __launch_bounds__(64) void k_candidates(ansx_geo g, u32 NSP, u32 NT, const uint2* __restrict__ pairs,
    const ansx_blk* __restrict__ blk, u16* __restrict__ attS, u32* __restrict__ attMeta,
    const double* __restrict__ lg2i)
{
    extern __shared__ uint4 cand_lds[];  // [BPW][ANSX_CAND_ROW] { freq | sym << 16, -, reciprocal of fs_rem }
    const u32 lane = threadIdx.x;
    const u32 BPW = 64u / NT;
    const u32 bl = lane / NT, t = lane - bl * NT;
    const u32 wb0 = blockIdx.x * BPW;
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
    // candidate frequencies go to attS[block][symbol][t] through a wave-uniform base and 32-bit lane offsets; a lane
    // without a symbol in some step stores to its row's last slot NSP - 1, which is never a symbol (nsyms < NSP)
    u8* const sbase = (u8*)(attS + (u64)wb0 * NSP * ANSX_ATTEMPTS);
    const u32 loff = ((live ? bl : 0u) * NSP * ANSX_ATTEMPTS + t) * 2u;  // (lanes without a block: the wave's first row)
    const u32 dummy = loff + (NSP - 1u) * ANSX_ATTEMPTS * 2u;
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
    auto step = [&](const uint4 e, double& Fd_out, u32& sc_out, u32& off_out) {
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
        off_out = loff + (e.x >> 16) * (ANSX_ATTEMPTS * 2u);
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
            u32 nS[8], nO[8];
#pragma unroll
            for (int u = 0; u < 8; u++) nF[u] = 0.0, nS[u] = 0u, nO[u] = dummy;
            if (j0 + 8 <= lim) {
#pragma unroll
                for (int u = 0; u < 8; u++) step(e8[u], nF[u], nS[u], nO[u]);
            } else if (j0 < lim) {
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (j0 + u < lim) step(e8[u], nF[u], nS[u], nO[u]);
            }
            // No vector-memory operation is issued inside the steps: the table loads of the previous batch are the
            // youngest ones outstanding here (vmcnt is one in-order counter for loads and stores -- a wait placed
            // behind this batch's stores would drain them), and they have had the whole batch to arrive.
#pragma unroll
            for (int u = 0; u < 8; u++) W = __builtin_fma(pF[u], pL[u], W);
#pragma unroll
            for (int u = 0; u < 8; u++) *(u16*)(sbase + nO[u]) = (u16)nS[u];  // (values above 65535 end in the u16 exit: never read)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                pF[u] = nF[u];
                pL[u] = lg2i[nS[u] < 65535u ? nS[u] : 65535u];
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
// of 256 threads per block, thread i owning IPT consecutive symbols -- the chosen frequencies, their exclusive
// scan (encoder table, compact 4-byte form) and the prelude (prelude_emit of ansx_kernels.h).
// Anything this path does not cover raises the violation flag and leaves the block without a stream; the host
// repeats the call on the exact path: undecided after NT candidates, the u16 exit with no earlier success, a
// frame above 2^16, a comparison inside the guard band (`guard`: ANSX_FAST_GUARD; tests widen it to force the repeat).
template <int IPT>
__global__ __launch_bounds__(256) void k_model_finish(ansx_geo g, u32 NSP, u32 NT, const u32* __restrict__ hist,
    const u16* __restrict__ attS, const u32* __restrict__ attMeta, ansx_blk* __restrict__ blk,
    u32* __restrict__ tab32, u8* __restrict__ scratch, u64 scr_stride, const u32* __restrict__ mostfreq,
    u32* __restrict__ hints, u32* __restrict__ gflags, u32 cap, double guard)
{
    static_assert(IPT % 4 == 0, "rows are read 16 bytes at a time");
    extern __shared__ u32 lds32[];
    __shared__ u32 sh_part[8];
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    ansx_blk* B = &blk[b];
    const u32 ns = B->max_sym + 1;
    const double thr = B->thr;
    const u32 m0 = B->m0_log2;
    // every thread applies the (wave-uniform) rule to the same NT results
    int chosen = -2, prev = -1;
    bool unsure = false;
    for (u32 t = 0; t < NT; t++) {
        const uint4 mt = *(const uint4*)(attMeta + ((u64)b * ANSX_ATTEMPTS + t) * 4);
        if (!mt.x) continue;  // scale_freqs failed: M *= 2 (ans_util.hpp:131-135)
        if (mt.y >= ANSX_U16_LIMIT) {  // ans_util.hpp:141-145
            chosen = prev;
            break;
        }
        const double XH = ansx_bits_to_f64((u64)mt.z | ((u64)mt.w << 32));
        const double d = XH - thr;
        if ((d < 0 ? -d : d) <= guard * thr || !(thr > 0.0)) {  // (H == 0: one-symbol block, exact path)
            unsure = true;
            break;
        }
        if (XH < thr) {  // ans_util.hpp:149
            chosen = (int)t;
            break;
        }
        prev = (int)t;
    }
    const u32 logM = m0 + (u32)(chosen < 0 ? 0 : chosen);
    if (unsure || chosen < 0 || logM > 16 || ns > cap) {
        if (tid == 0) {
            B->prelude_bytes = 0;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
        }
        return;
    }
    // chosen frequencies and their exclusive scan; symbols beyond ns (and absent ones) have frequency 0
    const u32 s0 = tid * IPT;
    const u32* h = hist + (u64)b * NSP;
    const u16* S = attS + (u64)b * NSP * ANSX_ATTEMPTS + (u32)chosen;
    u32 fr[IPT];
    u32 sum = 0;
    if (s0 < ns) {
#pragma unroll
        for (int i = 0; i < IPT; i += 4) {
            const uint4 hv = *(const uint4*)(h + s0 + i);  // rows are NSP (a multiple of IPT * 256 / ... >= ns + 8) long
            const u32 h4[4] = { hv.x, hv.y, hv.z, hv.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 s = s0 + i + k;
                const u32 sv = (u32)S[(u64)s * ANSX_ATTEMPTS];
                fr[i + k] = (s < ns && h4[k]) ? sv : 0u;
                sum += fr[i + k];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < IPT; i++) fr[i] = 0;
    }
    u32 total;
    u32 base = block_excl_scan<u32>(sum, sh_part, tid, 256, &total);
    u32* off = lds32;             // [cap]
    u32* bits = lds32 + cap;      // bit buffer
    u32* inc = lds32 + 2 * cap;   // [cap]
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
    __syncthreads();
    prelude_emit<IPT>(g, B, ns, logM, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, mostfreq, b, tid, 0, hints);
}

// k_model_fused: the whole per-block model (ans_fold.hpp:68-98 create() + serialize(), ans_util.hpp:
// 100-157 adjust_freqs) in ONE workgroup per block with every intermediate in LDS:
//
//   P1 histogram of folded symbols (LDS atomics, several bank-offset copies)            ans_fold.hpp:74-78
//   P2 copies -> hist[], entropy terms p*log2 p                                         util.hpp:271-282
//   P3 wave 0: stable counting sort by (freq, sym); wave 1 lane 0: H in index order     ans_util.hpp:114-124
//   P4 remaining-frequency prefix sums, per-position reciprocal * F (see "recurrence")
//   P5 scale_freqs for `natt` frame sizes at once, one lane each                        ans_util.hpp:77-95
//   P6 cross entropy of every candidate: terms by all threads, in-order sums by lanes   util.hpp:284-298
//   P7 stop rule                                                                        ans_util.hpp:127-153
//   P8 exclusive scan -> compact encoder table (the only model state that leaves the CU)
//   P9 interpolative prelude                                                            ans_util.hpp:46-63
//
// It replaces k_fold_hist + k_sort_entropy + k_scale_attempts + k_select_model + k_write_prelude (which
// hand ~2 GB per 256 Mi ints through HBM and need a host round trip for the alphabet size) whenever the
// host has an alphabet-size hint from an earlier call on the context: the LDS is sized for `cap`
// symbols, a block whose alphabet exceeds it (or whose frame exceeds 2^16) raises ANSX_G_VIOL and the
// host repeats the call on the general path.  Arithmetic is the general path's, operation for
// operation, except the recurrence:
//
// Recurrence.  S_j = max(1, trunc(0.5 + RN(RN(M_j / fs_j) * F_j))), M_{j+1} = M_j - S_j, over the
// symbols in ascending (freq, sym) order.  fs_j (frequency mass not yet visited) does not depend on S,
// so the refined reciprocal y_j of fs_j -- the slow part of the correctly rounded division
// ansx_div_int31 -- is computed for all j in parallel, and a serial step is the reference's operation
// sequence with that reciprocal plugged in (bit-identical to ansx_div_int31, which computes the same y):
//   q = M y;  r = fma(-q, fs, M);  a = fma(r, y, q);  w = 0.5 + a F;  S = max(trunc(w), 1);  M -= S
// 8 dependent f64 operations, no branch.  (An approximate w = fma(M, y F, 0.5) with an exact redo near
// integers was tried first: blocks of 2^k ints meet frames of 2^j, so EXACT ties M F / fs + 0.5 = integer
// are routine -- 81 of 200 steps of one candidate in a Zipf block -- and the redo path dominated.)
// While every S so far was 1, M_j = M - j is known in advance: the longest prefix whose steps all give 1
// is found by evaluating the step for every j in parallel, and the serial loop starts behind it (the tail
// of a skewed block is hundreds of symbols of frequency 1 and 2).
#pragma once

#include "ansx_kernels.h"


struct ansx_model_lds {  // byte offsets into the kernel's dynamic LDS (host: model_layout in ansx.hip)
    u32 cap;        // symbol capacity, a multiple of 8
    u32 nc;         // histogram copies (power of two)
    u32 natt;       // frame sizes tried per batch (8 or 4)
    u32 off_pairs;  // u32 [cap]  sorted (freq | sym << 16); later inc[] | off[] | bits[] of the prelude (12 cap bytes with yF)
    u32 off_yF;     // f64 [cap] refined reciprocal of fs_j
    u32 off_S;      // u16 [natt][cap] candidate frequencies in SORTED order; before P5: f64 [cap] entropy terms
    u32 off_E;      // sort scratch (ANSX_MODEL_SORT_BYTES)
    u32 off_pos;    // u16 [cap] symbol -> sorted position
    u32 off_ffs;    // f32 [cap + 4] fs_j, 0 from j = sigma on (F_j = fs_j - fs_{j+1})
    u32 off_X;      // f64 [256] cross-entropy terms of one chunk, double buffered with
    u32 off_X1;     // f64 [256] (inside the pairs area, dead by then, when that is large enough)
    u32 total;
};

#define ANSX_MS_VMAX 256u  // frequencies below this are binned; the (<= n/256) larger ones ranked separately
#define ANSX_MS_NBIG 68u   // capacity of the "big" list: blocks of at most 16384 ints (ANSX_MODEL_MAX_BLOCK)
#define ANSX_MODEL_MAX_BLOCK 16384u
#define ANSX_MODEL_SORT_BYTES (ANSX_MS_VMAX * 4 + ANSX_MS_VMAX * 8 + ANSX_MS_NBIG * 8)

// workgroup barrier for data exchanged through LDS only: unlike __syncthreads() it does not wait for
// outstanding global loads (vmcnt), so prefetched loads stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one step of scale_freqs (ans_util.hpp:80-92) given y = the refined reciprocal of fs that
// ansx_div_int31(M, fs) would compute itself: returns max(1, trunc(0.5 + RN(RN(M / fs) * F)))
__device__ __forceinline__ double model_step(double Md, double y, double fsd, double frd)
{
    const double q = Md * y;
    const double r = __builtin_fma(-q, fsd, Md);
    const double aratio = __builtin_fma(r, y, q);
    double v = aratio * frd;
    v = 0.5 + v;
    return __builtin_fmax(__builtin_trunc(v), 1.0);
}

template <int IPT>
__global__ __launch_bounds__(256, 6) void k_model_fused(const u32* __restrict__ in, ansx_geo g, u32 NSP,
    ansx_model_lds ML, const ansx_log2_ent* __restrict__ l2lut, ansx_blk* __restrict__ blk,
    u32* __restrict__ tab32, u8* __restrict__ scratch, u64 scr_stride, const u32* __restrict__ mostfreq,
    u32* __restrict__ gflags, u32 value_limit, u32* __restrict__ hints)
{
    extern __shared__ __attribute__((aligned(16))) u8 msm[];
    __shared__ u32 sh_u[16];      // 0 max_sym, 1 sigma, 2 nbig, 3 xmax, 4 jmin, 5 sat mask, 6 chosen+2 (0 = undecided), 7 prev+1
    __shared__ u32 sh_j0[8];
    __shared__ u32 sh_meta[8][4];  // per candidate: ok, -, XH bits lo, hi
    __shared__ double sh_H[2];     // H, thr
    __shared__ u32 sh_part[8];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 b = blockIdx.x;
    const u32 nb = geo_block_n(g, b);
    const u32 cap = ML.cap, nc = ML.nc, natt = ML.natt;
    const u32 cstride = cap + 8;
    u32* const hcopy = (u32*)msm;                 // [nc][cstride]; copy 0 becomes hist[]
    u32* const pairs = (u32*)(msm + ML.off_pairs);
    double* const yF = (double*)(msm + ML.off_yF);
    u16* const Ssort = (u16*)(msm + ML.off_S);
    double* const hterm = (double*)(msm + ML.off_S);
    u16* const pos = (u16*)(msm + ML.off_pos);
    float* const fsf = (float*)(msm + ML.off_ffs);
    double* const Xb0 = (double*)(msm + ML.off_X);
    double* const Xb1 = (double*)(msm + ML.off_X1);
    // last successful-but-rejected candidate of an earlier batch (only blocks that need more than natt
    // frame sizes, e.g. constant ones, ever touch it): parked in the block's not yet written table row
    u32* const prevS = tab32 + (u64)b * NSP;
    ansx_blk* const B = &blk[b];

    // ---- P0/P1: histogram
    for (u32 s = tid; s < nc * cstride; s += 256) hcopy[s] = 0;
    if (tid < 16) sh_u[tid] = 0;
    __syncthreads();
    {
        u32* const my_hist = hcopy + (tid & (nc - 1)) * cstride;
        const ansx_map mp = g.map;
        const u32 capm1 = cap - 1;
        u32 lmax = 0, xmax = 0;
        auto take = [&](u32 x) {
            xmax = x > xmax ? x : xmax;
            const u32 k = map_nbytes(mp, x);
            const u32 s = map_sym(mp, x, k);
            lmax = s > lmax ? s : lmax;
            atomicAdd(&my_hist[s < capm1 ? s : capm1], 1u);  // (an alphabet beyond cap is caught through lmax)
        };
        const u32* src = in + (u64)b * g.block_ints;
        u32 done = 0;
        if ((((uintptr_t)src) & 15u) == 0) {
            const uint4* v4 = (const uint4*)src;
            const u32 nvec = nb >> 2;
            u32 v = tid;
            // rounds of 4 x 16 bytes per thread; the next round is requested before this one is counted
            if (v + 3 * 256 < nvec) {
                uint4 q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = v4[v + j * 256];
                for (;;) {
                    const u32 vn = v + 4 * 256;
                    const bool more = vn + 3 * 256 < nvec;  // uniform up to the last partial round
                    uint4 qn[4];
                    if (more) {
#pragma unroll
                        for (int j = 0; j < 4; j++) qn[j] = v4[vn + j * 256];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        take(q[j].x);
                        take(q[j].y);
                        take(q[j].z);
                        take(q[j].w);
                    }
                    v = vn;
                    if (!more) break;
#pragma unroll
                    for (int j = 0; j < 4; j++) q[j] = qn[j];
                }
            }
            for (; v < nvec; v += 256) {
                const uint4 q = v4[v];
                take(q.x);
                take(q.y);
                take(q.z);
                take(q.w);
            }
            done = nvec << 2;
        }
        for (u32 i = done + tid; i < nb; i += 256) take(src[i]);
        lmax = wave_max(lmax);
        xmax = wave_max(xmax);
        if (lane == 0) {
            atomicMax(&sh_u[0], lmax);
            atomicMax(&sh_u[3], xmax);
        }
    }
    __syncthreads();
    const u32 ns = sh_u[0] + 1;
    if (tid == 0 && sh_u[3] >= value_limit) atomicOr(&gflags[ANSX_G_ERR], 1u << 6 /* ANSX_ERR_DOMAIN */);
    auto violation = [&]() {  // uniform exit: this block needs the general path
        if (tid == 0) {
            B->n = nb;
            B->max_sym = ns - 1;
            B->status = 9;
            B->resolved = 1;
            B->prelude_bytes = 0;
            B->stream_bytes = 0;
            atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);
        }
    };
    if (ns > cap) {
        violation();
        return;
    }
    // ---- P2: hist[] = sum of the copies; entropy terms (util.hpp:276-279; an absent symbol adds +0.0,
    // which leaves the in-order sum -- never -0.0 -- unchanged)
    {
        const double nd = (double)nb;
        u32 nz = 0;
        for (u32 s = tid; s < cap; s += 256) {
            u32 fr = hcopy[s];
            for (u32 c = 1; c < nc; c++) fr += hcopy[c * cstride + s];
            double t = 0.0;
            if (fr) {
                nz++;
                const double p = ansx_div_int31((double)fr, nd);
                t = p * ansx_log2_portable(p);
            }
            hterm[s] = t;
            hcopy[s] = fr;  // (only this thread touches column s)
        }
        nz = wave_sum(nz);
        if (lane == 0) atomicAdd(&sh_u[1], nz);
    }
    __syncthreads();
    const u32* const hist = hcopy;
    const u32 sigma = sh_u[1];
    // ---- P3: wave 1 lane 0 sums H left to right (util.hpp:271-282), wave 0 sorts
    // (the single-wave serial phases run at raised priority: they are the block's critical path and
    // otherwise queue behind the bulk phases of the other workgroups sharing the SIMD)
    if (wave == 1) {
        if (lane == 0) {
            __builtin_amdgcn_s_setprio(3);
            double acc = 0.0;
            u32 i = 0;
            for (; i + 8 <= ns; i += 8) {
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = hterm[i + u];
#pragma unroll
                for (int u = 0; u < 8; u++) acc = acc + t8[u];
            }
            for (; i < ns; i++) acc = acc + hterm[i];
            const double H = -acc;
            sh_H[0] = H;
            sh_H[1] = H * (1.0 + (double)1 / (double)1000);  // ans_util.hpp:124
            __builtin_amdgcn_s_setprio(0);
        }
    } else if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        // stable counting sort by frequency of symbols already in index order (k_sort_entropy's
        // scheme): frequencies below VMAX are binned, lanes of a 64-symbol pass that share a
        // frequency rank themselves through a 64-bit mask per value; the few larger ones are ranked
        // among themselves
        u32* const cnt = (u32*)(msm + ML.off_E);
        unsigned long long* const vmask = (unsigned long long*)(msm + ML.off_E + ANSX_MS_VMAX * 4);
        u64* const big = (u64*)(msm + ML.off_E + ANSX_MS_VMAX * 12);
        for (u32 v = lane; v < ANSX_MS_VMAX; v += 64) {
            cnt[v] = 0;
            vmask[v] = 0;
        }
        wave_lds_sync();
        for (u32 s0 = 0; s0 < ns; s0 += 64) {
            const u32 s = s0 + lane;
            const u32 fr = s < ns ? hist[s] : 0u;
            if (fr) {
                if (fr < ANSX_MS_VMAX) atomicAdd(&cnt[fr], 1u);
                else {
                    const u32 slot = atomicAdd(&sh_u[2], 1u);
                    if (slot < ANSX_MS_NBIG) big[slot] = ((u64)fr << 16) | s;
                }
            }
        }
        wave_lds_sync();
        u32 nsmall;
        {
            const u32 per = ANSX_MS_VMAX / 64;
            u32 loc = 0;
            for (u32 i = 0; i < per; i++) loc += cnt[lane * per + i];
            const u32 incl = wave_incl_scan(loc);
            nsmall = wave_last(incl);
            u32 run = incl - loc;
            for (u32 i = 0; i < per; i++) {
                const u32 t = cnt[lane * per + i];
                cnt[lane * per + i] = run;
                run += t;
            }
        }
        wave_lds_sync();
        for (u32 s0 = 0; s0 < ns; s0 += 64) {
            const u32 s = s0 + lane;
            const u32 fr = s < ns ? hist[s] : 0u;
            const bool small = fr != 0 && fr < ANSX_MS_VMAX;
            if (small) atomicOr(&vmask[fr], 1ull << lane);
            wave_lds_sync();
            if (small) {
                const unsigned long long m = vmask[fr];
                const unsigned long long below = m & ((1ull << lane) - 1ull);
                const u32 p = cnt[fr] + (u32)__popcll(below);
                pairs[p] = fr | (s << 16);
                pos[s] = (u16)p;
                if (below == 0) {  // lowest lane of the value: advance its cursor, clear the mask
                    cnt[fr] += (u32)__popcll(m);
                    vmask[fr] = 0;
                }
            }
            wave_lds_sync();
        }
        const u32 nbig = sh_u[2] < ANSX_MS_NBIG ? sh_u[2] : ANSX_MS_NBIG;
        for (u32 i = lane; i < nbig; i += 64) {
            const u64 key = big[i];
            u32 rank = 0;
            for (u32 j = 0; j < nbig; j++) rank += (big[j] < key) ? 1u : 0u;
            const u32 s = (u32)(key & 0xFFFFu);
            pairs[nsmall + rank] = (u32)(key >> 16) | (s << 16);
            pos[s] = (u16)(nsmall + rank);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    // ---- P4: fs_j = n - sum_{i<j} F_i, yF_j
    {
        const u32 per = (sigma + 255) / 256;
        const u32 lo = tid * per, hi = (lo + per) < sigma ? (lo + per) : sigma;
        u32 sum = 0;
        for (u32 j = lo; j < hi; j++) sum += pairs[j] & 0xFFFFu;
        u32 total;
        u32 run = block_excl_scan<u32>(sum, sh_part, tid, 256, &total);
        for (u32 j = lo; j < hi; j++) {
            const u32 F = pairs[j] & 0xFFFFu;
            const u32 fs = nb - run;
            run += F;
            const double fsd = (double)fs;
            const double y0 = __builtin_amdgcn_rcp(fsd);
            const double y1 = __builtin_fma(__builtin_fma(-fsd, y0, 1.0), y0, y0);
            yF[j] = __builtin_fma(__builtin_fma(-fsd, y1, 1.0), y1, y1);  // ansx_div_int31's y2
            fsf[j] = (float)fs;  // <= 16384: exact
        }
        for (u32 j = sigma + tid; j < cap + 4; j += 256) fsf[j] = 0.0f;
        for (u32 j = sigma + tid; j < cap; j += 256) yF[j] = 0.0;
    }
    u32 m0 = 0;  // ans_util.hpp:109-112
    if (sigma != 0 && (sigma & (sigma - 1)) == 0) m0 = 31 - __clz(sigma);
    else m0 = sigma == 0 ? 0 : (32 - __clz(sigma));
    __syncthreads();
    const double thr = sh_H[1];
    const double ndi = (double)(int)nb;  // util.hpp:288: the sums are taken in an int
    int prev = -1;     // last successful-but-rejected candidate (global index); uniform
    int chosen = -2;   // uniform
    u32 crow = 0;      // row of Ssort that holds the chosen candidate
    for (u32 batch = 0; batch * natt < 24; batch++) {
        // ---- P5a: longest all-ones prefix per candidate, in parallel
        if (tid < 8) sh_j0[tid] = sigma;
        if (tid == 0) sh_u[5] = 0;
        __syncthreads();
        {
            // thread -> candidate t = tid % natt, positions j = tid / natt, + 256 / natt, ...: stop at the
            // first position that is not provably 1
            const u32 t = tid % natt, per = 256 / natt;
            const u32 sh = m0 + batch * natt + t;
            u32 jf = sigma;
            if (sh <= 31) {
                const i64 M0 = (i64)1 << sh;
                for (u32 j = tid / natt; j < sigma && jf == sigma; j += 4 * per) {
                    double y4[4];
                    float f4[4], g4[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const u32 jj = j + q * per < cap ? j + q * per : cap - 1;
                        y4[q] = yF[jj];
                        f4[q] = fsf[jj];
                        g4[q] = fsf[jj + 1];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const u32 jj = j + q * per;
                        const double Mj = (double)(M0 - (i64)jj);  // M - j >= 1 (j < sigma <= M0)
                        const double fsd = (double)f4[q];
                        const double sv = model_step(Mj, y4[q], fsd, fsd - (double)g4[q]);
                        if (jj < sigma && jf == sigma && sv != 1.0) jf = jj;
                    }
                }
            }
            // lanes t, t + natt, t + 2 natt, ... of a wave hold the same candidate
            for (u32 o = natt; o < 64; o <<= 1) {
                const u32 v = __shfl_xor(jf, (int)o);
                jf = v < jf ? v : jf;
            }
            if (lane < natt && jf < sigma) atomicMin(&sh_j0[lane], jf);
        }
        __syncthreads();
        u32 jmin = sigma;
        for (u32 t = 0; t < natt; t++) {
            const u32 sh = m0 + batch * natt + t;
            if (sh <= 31) jmin = sh_j0[t] < jmin ? sh_j0[t] : jmin;
        }
        const u32 jstart = jmin & ~3u;
        for (u32 idx = tid; idx < jstart * natt; idx += 256) Ssort[(idx / jstart) * cap + (idx % jstart)] = 1;
        __syncthreads();
        // ---- P5b: the serial part, one lane per candidate (ans_util.hpp:80-92)
        if (wave == 0 && lane < natt) {
            __builtin_amdgcn_s_setprio(3);
            const u32 t = lane;
            const u32 sh = m0 + batch * natt + t;
            double Md = sh > 31 ? -1.0 : (double)(i64)(((i64)1 << sh) - (i64)jstart);
            u16* const St = Ssort + t * cap;
            // frames below 2^16 cannot hold a value of 65535 or more: no saturation needed when storing
            const bool may_sat = sh >= 16;
            double mx = 1.0;  // largest value so far
            auto step = [&](double y, double fsd, double fsn) -> u32 {
                const double tr = model_step(Md, y, fsd, fsd - fsn);
                Md = Md - tr;
                // (a failed candidate keeps running with M < 0; its values are never used)
                mx = __builtin_fmax(mx, tr);
                u32 sv = (u32)tr;
                if (may_sat) sv = sv > 65535u ? 65535u : sv;
                return sv;
            };
            // operands of 4 steps per load group, requested two groups ahead (LDS latency under the other
            // workgroups' histogram atomics is several hundred cycles)
            struct grp {
                double y[4];
                float4 f;
            };
            auto ld = [&](u32 j0) -> grp {
                const u32 jj = j0 + 4 <= cap ? j0 : cap - 4;
                grp r;
                r.y[0] = yF[jj], r.y[1] = yF[jj + 1], r.y[2] = yF[jj + 2], r.y[3] = yF[jj + 3];
                r.f = *(const float4*)(fsf + jj);
                return r;
            };
            auto run4 = [&](u32 j0, const grp& r, float fnext) {
                const double f0 = (double)r.f.x, f1 = (double)r.f.y, f2 = (double)r.f.z, f3 = (double)r.f.w;
                const u32 s0 = step(r.y[0], f0, f1), s1 = step(r.y[1], f1, f2), s2 = step(r.y[2], f2, f3),
                          s3 = step(r.y[3], f3, (double)fnext);
                *(uint2*)(St + j0) = make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
            };
            auto tail3 = [&](u32 j0, const grp& r) {  // the last 0..3 steps (fs is 0 from sigma on)
                const double f0 = (double)r.f.x, f1 = (double)r.f.y, f2 = (double)r.f.z, f3 = (double)r.f.w;
                if (j0 < sigma) St[j0] = (u16)step(r.y[0], f0, f1);
                if (j0 + 1 < sigma) St[j0 + 1] = (u16)step(r.y[1], f1, f2);
                if (j0 + 2 < sigma) St[j0 + 2] = (u16)step(r.y[2], f2, f3);
            };
            u32 j = jstart;
            grp A = ld(j), B2 = ld(j + 4);
            for (;;) {
                if (j + 4 > sigma) {
                    tail3(j, A);
                    break;
                }
                const grp C2 = ld(j + 8);
                run4(j, A, B2.f.x);
                j += 4;
                if (j + 4 > sigma) {
                    tail3(j, B2);
                    break;
                }
                A = ld(j + 8);
                run4(j, B2, C2.f.x);
                j += 4;
                if (j + 4 > sigma) {
                    tail3(j, C2);
                    break;
                }
                B2 = ld(j + 8);
                run4(j, C2, A.f.x);
                j += 4;
            }
            sh_meta[t][0] = (Md == 0.0) ? 1u : 0u;
            sh_meta[t][1] = (mx >= 65535.0) ? 1u : 0u;  // u16 exit (ans_util.hpp:141-145)
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        // ---- P6: cross entropy in index order (util.hpp:284-298): all threads evaluate the terms of
        // 256 / natt symbols at a time, the candidates' lanes add them left to right
        {
            const u32 CH = 256 / natt;
            const u32 t = tid % natt, sl = tid / natt;
            const u32 sh = m0 + batch * natt + t;
            const u16* const St = Ssort + t * cap;
            double acc = 0.0;
            const u32 nch = (ns + CH - 1) / CH;
            // The table entry of a term hangs off two LDS reads and comes from L2: one round trip per
            // chunk would be the whole phase.  Terms are produced G chunks at a time (G entries in
            // flight) and the next group's entries are requested before this group is summed; the
            // barriers in between wait for LDS only (lds_barrier), never for those loads.
            constexpr int G = 2;  // chunks per group (VGPR budget: two groups of entries are live)
            struct ent4 {
                double y[G], ylo[G];
                int e[G];
                u32 h[G];
            };
            auto fetch = [&](u32 c0) -> ent4 {
                ent4 e;
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const u32 s = (c0 + u) * CH + sl;
                    u32 h = 0, sv = 1;
                    if (s < ns) {
                        h = hist[s];
                        if (h != 0) sv = St[pos[s]];
                    }
                    e.h[u] = h;
                    const ansx_log2_ent* le = l2lut + sv;
                    e.y[u] = le->y;
                    e.ylo[u] = le->ylo;
                    e.e[u] = ansx_log2_e_of_int(sv);
                }
                return e;
            };
            ent4 cur = fetch(0);
            for (u32 c0 = 0; c0 < nch; c0 += G) {
                double tm[G];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    tm[u] = 0.0;  // absent: p * log2(1) = +0.0
                    if (cur.h[u] != 0) {
                        const double p = ansx_div_int31((double)cur.h[u], ndi);
                        const double lg = ansx_log2_stage2(cur.e[u] - (int)sh, cur.y[u], cur.ylo[u]);
                        tm[u] = p * lg;
                    }
                }
                ent4 nxt = cur;
                if (c0 + G < nch) nxt = fetch(c0 + G);
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const u32 c = c0 + u;
                    if (c < nch) {  // uniform
                        double* const Xb = (c & 1) ? Xb1 : Xb0;  // rewritten two barriers later: its sum is done by then
                        Xb[sl * natt + t] = tm[u];
                        lds_barrier();
                        if (wave == 0 && lane < natt) {
                            __builtin_amdgcn_s_setprio(3);
                            const double* x = Xb + lane;
                            const u32 lim = ns - c * CH < CH ? ns - c * CH : CH;
                            u32 v = 0;
                            for (; v + 8 <= lim; v += 8) {  // 8 reads in flight, then the 8 dependent adds
                                double x8[8];
#pragma unroll
                                for (int q = 0; q < 8; q++) x8[q] = x[(v + q) * natt];
#pragma unroll
                                for (int q = 0; q < 8; q++) acc = acc + x8[q];
                            }
                            for (; v < lim; v++) acc = acc + x[v * natt];
                            __builtin_amdgcn_s_setprio(0);
                        }
                    }
                }
                cur = nxt;
            }
            if (wave == 0 && lane < natt) {
                const u64 xb = ansx_f64_to_bits(-acc);
                sh_meta[lane][2] = (u32)xb;
                sh_meta[lane][3] = (u32)(xb >> 32);
            }
        }
        __syncthreads();
        // ---- P7: stop rule (ans_util.hpp:127-153), evaluated redundantly by every thread
        for (u32 t = 0; t < natt && chosen == -2; t++) {
            const u32 sh = m0 + batch * natt + t;
            if (sh > 31 || !sh_meta[t][0]) continue;  // scale_freqs failed: M *= 2 (ans_util.hpp:131-135)
            const int T = (int)(batch * natt + t);
            if (sh_meta[t][1]) {  // ans_util.hpp:141-145
                chosen = prev;
                break;
            }
            const double XH = ansx_bits_to_f64((u64)sh_meta[t][2] | ((u64)sh_meta[t][3] << 32));
            if (tid == 0 && ansx_near_threshold(XH, thr)) atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_VIOL_BIT);  // a close call: the exact path (and the host) decide
            if (XH < thr) {  // ans_util.hpp:149
                chosen = T;
                break;
            }
            prev = T;
        }
        if (chosen != -2) {
            // the chosen candidate may be the rejected one of an earlier batch: bring it back
            if (chosen >= 0 && chosen < (int)(batch * natt)) {
                __threadfence_block();
                for (u32 j = tid; j < sigma; j += 256) Ssort[j] = (u16)prevS[j];
                __syncthreads();
                crow = 0;
            } else if (chosen >= 0) {
                crow = (u32)chosen - batch * natt;
            }
            break;
        }
        if (prev >= (int)(batch * natt)) {
            const u16* const St = Ssort + (u32)(prev - (int)(batch * natt)) * cap;
            for (u32 j = tid; j < sigma; j += 256) prevS[j] = St[j];  // (thread j reads back what it wrote)
        }
        __syncthreads();
    }
    if (chosen < 0) {  // -1: "prev" is the all-zero vector, the reference's degenerate exit; -2: no frame fits
        if (tid == 0) {
            B->n = nb;
            B->max_sym = ns - 1;
            B->sigma = sigma;
            B->m0_log2 = m0;
            B->resolved = 1;
            B->status = 7;  // ANSX_ERR_MODEL
            B->logM = 0;
            B->prelude_bytes = 0;
            atomicOr(&gflags[ANSX_G_ERR], 1u << 7);
        }
        return;
    }
    const u32 logM = m0 + (u32)chosen;
    if (logM > 16) {  // frames above 2^16 need the 16-byte table entries / integer-state encoder
        violation();
        return;
    }
    // ---- P8: exclusive scan of the chosen frequencies -> compact encoder table (ans_fold.hpp:82-91)
    u32* const inc = (u32*)(msm + ML.off_pairs);  // pairs / yF are dead: inc[cap] | off[cap] | bits[cap]
    u32* const off = inc + cap;
    u32* const bits = off + cap;
    {
        const u16* const Sc = Ssort + crow * cap;
        const u32 per = (ns + 255) / 256;
        const u32 lo = tid * per, hi = (lo + per) < ns ? (lo + per) : ns;
        u32 frs[IPT];  // per <= cap / 256 <= IPT
        u32 sum = 0;
#pragma unroll
        for (u32 i = 0; i < (u32)IPT; i++) {
            const u32 s = lo + i;
            u32 fr = 0;
            if (i < per && s < hi && hist[s] != 0) fr = Sc[pos[s]];
            frs[i] = fr;
            sum += fr;
        }
        u32 total;
        u32 run = block_excl_scan<u32>(sum, sh_part, tid, 256, &total);
        // (pairs / yF were last read before the barriers inside the scan: inc[] may overwrite them)
        u32* const t32 = tab32 + (u64)b * NSP;
#pragma unroll
        for (u32 i = 0; i < (u32)IPT; i++) {
            const u32 s = lo + i;
            if (i < per && s < hi) {
                t32[s] = (run << 16) | frs[i];      // base < 2^16, freq < 65535 (M <= 2^16)
                inc[s] = run + frs[i] + s;          // ans_util.hpp:54-58: inc[s] = inc[s-1] + nfreq[s] + 1
                run += frs[i];
            }
        }
    }
    __syncthreads();
    // ---- P9: prelude (rfold header, vbyte(max_sym), log2 M, interpolative code)
    prelude_emit<IPT>(g, B, ns, logM, inc, off, bits, sh_part, scratch + (u64)b * scr_stride, mostfreq, b, tid, 0, hints);
    if (tid == 0) {
        B->n = nb;
        B->max_sym = ns - 1;
        B->sigma = sigma;
        B->m0_log2 = m0;
        B->H = sh_H[0];
        B->thr = thr;
        B->resolved = 1;
        B->prev = prev;
        B->logM = logM;
        B->status = 0;
        // same-address atomics serialise in L2 (16 K blocks): only the few blocks that raise a running
        // maximum issue one
        if (__hip_atomic_load(&gflags[ANSX_G_MAXLOGM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < logM)
            atomicMax(&gflags[ANSX_G_MAXLOGM], logM);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXNSYMS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ns)
            atomicMax(&gflags[ANSX_G_MAXNSYMS], ns);
        if (__hip_atomic_load(&gflags[ANSX_G_MAXSIGMA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sigma)
            atomicMax(&gflags[ANSX_G_MAXSIGMA], sigma);
    }
}

// Device/host helpers shared by every ansx kernel (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ANSX_HD __host__ __device__ __forceinline__
#define ANSX_D __device__ __forceinline__

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

// ---------------------------------------------------------------------------------------------
// Constants of the reference codec (include/ans_byte.hpp:24-31 as used by ans_fold.hpp)
// ---------------------------------------------------------------------------------------------
#define ANSX_K_LOG2 4          // K = 16: lower bound L = 16*M
#define ANSX_RADIX_LOG2 32     // renormalise in 32-bit words
#define ANSX_U16_LIMIT 65535u  // ans_util.hpp:126

// fold geometry (include/ans_fold.hpp:38-50): threshold T = 2^(f+7), per-byte offset D = 255*2^(f-1)
ANSX_HD u32 fold_T(u32 f) { return 1u << (f + 7); }
ANSX_HD u32 fold_D(u32 f) { return 255u << (f - 1); }
ANSX_HD u32 fold_NSP(u32 f) { return 1u << (f + 9); }  // symbol-array stride = reference MAX_SIGMA (ans_fold.hpp:70)

// Byte-stripping map value -> (symbol, k exception bytes), parameterised so that one set of
// kernels serves ANSfold<f>/ANSrfold<f> (ans_fold.hpp:38-65,150-175: thresholds T*256^j,
// k*D symbol offset) and ANSmsb (ans_msb.hpp:41-74,159-180: thresholds 256^j inclusive, 256*k).
//   k   = (x >= t1) + (x >= t2) + (x >= t3);        sym   = (x >> 8k) + k*D
//   k   = (sym >= u1) + (sym >= u2) + (sym >= u3);  value = (sym - k*D) << 8k
// This is the closed form of the reference's while-loop; valid for all 32-bit x.
struct ansx_map {
    u32 t1, t2, t3;
    u32 u1, u2, u3;
    u32 D;
};
ANSX_HD ansx_map map_fold(u32 f)
{
    const u32 T = fold_T(f), D = fold_D(f);
    ansx_map m = { T, T << 8, T << 16, T, T + D, T + 2 * D, D };
    return m;
}
// ANSint (ans_int.hpp:40-48): the symbol is the value, nothing is stripped
ANSX_HD ansx_map map_int()
{
    ansx_map m = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u };
    return m;
}
ANSX_HD ansx_map map_msb()
{
    ansx_map m = { 257u, 65537u, (1u << 24) + 1u, 257u, 513u, 769u, 256u };
    return m;
}
ANSX_HD u32 map_nbytes(const ansx_map& m, u32 x)
{
    return ((x >= m.t1) ? 1u : 0u) + ((x >= m.t2) ? 1u : 0u) + ((x >= m.t3) ? 1u : 0u);
}
ANSX_HD u32 map_sym(const ansx_map& m, u32 x, u32 k) { return (x >> (8 * k)) + k * m.D; }
ANSX_HD u32 unmap_nbytes(const ansx_map& m, u32 sym)
{
    return ((sym >= m.u1) ? 1u : 0u) + ((sym >= m.u2) ? 1u : 0u) + ((sym >= m.u3) ? 1u : 0u);
}
ANSX_HD u32 unmap_value(const ansx_map& m, u32 sym, u32 k) { return (sym - k * m.D) << (8 * k); }

// ---------------------------------------------------------------------------------------------
// Portable log2: only IEEE +,-,*,/ and fma, identical on host and device.  Replaces glibc's
// log2 in entropy()/cross_entropy() (include/util.hpp:271-298).  Accuracy ~1 ulp; exact for
// powers of two.  The normaliser compares sums of ~10^3 such terms against a 0.1 % margin
// (ans_util.hpp:127-128,149), so the last-ulp difference to glibc only matters when
// |XH - 1.001 H| < ~1e-15 H, where the reference's own decision depends on the libm build.
// ---------------------------------------------------------------------------------------------
ANSX_HD double ansx_bits_to_f64(u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b);
#else
    double d;
    __builtin_memcpy(&d, &b, 8);
    return d;
#endif
}
ANSX_HD u64 ansx_f64_to_bits(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (u64)__double_as_longlong(d);
#else
    u64 b;
    __builtin_memcpy(&b, &d, 8);
    return b;
#endif
}

#pragma clang fp contract(off)
// Stage 1: everything that depends on the mantissa of x only (plus the exponent it starts from);
// stage 2: the final e + y.  log2(x * 2^-k) = stage2(e - k, y, ylo) bit for bit, because scaling
// by a power of two changes nothing but the exponent field -- the normaliser uses this for
// q = S / M with integer S <= 65535 and M = 2^k (util.hpp:284-298), reading stage 1 from a table.
ANSX_HD void ansx_log2_stage1(double x, int* e_out, double* y_out, double* ylo_out)
{
    // x is a ratio of positive integers < 2^53: positive, finite, normal.
    u64 ix = ansx_f64_to_bits(x);
    int e = (int)(ix >> 52) - 1023;
    u64 m = (ix & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double z = ansx_bits_to_f64(m);  // [1,2)
    if (z > 1.4142135623730951) {
        z = z * 0.5;
        e += 1;
    }
    double f = z - 1.0;  // exact
    double s = f / (2.0 + f);
    double s2 = s * s;
    // log(z) = 2s * (1 + s2/3 + s2^2/5 + ...); |s| <= 0.1716 -> s2 <= 0.02944; 12 terms < 2^-60
    double p = 1.0 / 25.0;
    p = __builtin_fma(p, s2, 1.0 / 23.0);
    p = __builtin_fma(p, s2, 1.0 / 21.0);
    p = __builtin_fma(p, s2, 1.0 / 19.0);
    p = __builtin_fma(p, s2, 1.0 / 17.0);
    p = __builtin_fma(p, s2, 1.0 / 15.0);
    p = __builtin_fma(p, s2, 1.0 / 13.0);
    p = __builtin_fma(p, s2, 1.0 / 11.0);
    p = __builtin_fma(p, s2, 1.0 / 9.0);
    p = __builtin_fma(p, s2, 1.0 / 7.0);
    p = __builtin_fma(p, s2, 1.0 / 5.0);
    p = __builtin_fma(p, s2, 1.0 / 3.0);
    p = p * s2;  // s2/3 + s2^2/5 + ...
    double t = 2.0 * s;
    // residual of the division, including the rounding of den = 2 + f (Fast2Sum, |f| < 2):
    //   f = s*(den + dlo) + r   ->   s_exact = s + (r - s*dlo)/den
    double den = 2.0 + f;
    double dlo = f - (den - 2.0);
    double r = __builtin_fma(-s, den, f);
    r = __builtin_fma(-s, dlo, r);
    // 2/den = 1/(1 + f/2) ~ 1 - f/2 + f^2/4 (|f| <= 0.42: relative error < 1e-2, applied to a
    // term that is itself < 2^-52 of the result) -- avoids a second division
    double rc2 = __builtin_fma(__builtin_fma(0.25, f, -0.5), f, 1.0);
    double tlo = __builtin_fma(t, p, r * rc2);  // low part of log(z)
    // log2(z) = (t + tlo) / ln2, in two pieces
    const double IL2_HI = 1.4426950408889634;       // 1/ln2 rounded to double
    const double IL2_LO = 2.0355273740931033e-17;   // 1/ln2 - IL2_HI
    double y = t * IL2_HI;
    double ye = __builtin_fma(t, IL2_HI, -y);
    double ylo = __builtin_fma(tlo, IL2_HI, __builtin_fma(t, IL2_LO, ye));
    *e_out = e;
    *y_out = y;
    *ylo_out = ylo;
}
ANSX_HD double ansx_log2_stage2(int e, double y, double ylo)
{
    // e + y with Fast2Sum (e == 0 or |e| >= 1 > |y|)
    double ed = (double)e;
    double hi = ed + y;
    double herr = y - (hi - ed);
    return hi + (herr + ylo);
}
ANSX_HD double ansx_log2_portable(double x)
{
    int e;
    double y, ylo;
    ansx_log2_stage1(x, &e, &y, &ylo);
    return ansx_log2_stage2(e, y, ylo);
}

// ---------------------------------------------------------------------------------------------
// Correctly rounded a / b for integer-valued doubles 0 <= a, 0 < b, both < 2^31 -- the only
// divisions of the normaliser (M_rem / fs_rem, ans_util.hpp:83; count / n, util.hpp:276,291).
// hipcc expands an IEEE f64 division into v_div_scale x2, v_rcp, ~6 fma, v_div_fmas, v_div_fixup,
// several of them slow-rate; for these operands the Markstein sequence below gives the same bits:
//   y  = 1/b to ~1 ulp (hardware seed 2^-24.5, two Newton steps)
//   q  = RN(a*y);  r = a - q*b (exact in one fma: q is within 2 ulp of a/b);  q' = RN(q + r*y)
// q + r*y differs from a/b by |r|*|y - 1/b| <= 2^-105 relative, while a/b -- a ratio of integers
// below 2^31 -- is either representable or at least 2^-84 (relative) away from every rounding
// midpoint (b*m - a is a non-zero multiple of the midpoint's last place), so the final rounding
// cannot differ from the IEEE one.  Checked against true division on the device (ansx_selftest_div).
// Host builds (tests) use the plain operator.
ANSX_HD double ansx_div_int31(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double y0 = __builtin_amdgcn_rcp(b);
    const double y1 = __builtin_fma(__builtin_fma(-b, y0, 1.0), y0, y0);
    const double y2 = __builtin_fma(__builtin_fma(-b, y1, 1.0), y1, y1);
    const double q = a * y2;
    const double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, y2, q);
#else
    return a / b;
#endif
}

// ---------------------------------------------------------------------------------------------
// Quad (4-lane) cross-lane helpers on DPP quad_perm — the renorm byte-emit compaction of
// ans_fold.hpp:100-112 / :216-228: 4 interleaved states share one byte cursor.
// ---------------------------------------------------------------------------------------------
#define ANSX_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

// quad_perm through an explicit DPP mov.  Inline asm on purpose: with the builtin
// (__builtin_amdgcn_mov_dpp) hipcc -O3 (ROCm 7.2) folds the [3,3,3,3] broadcast into the
// cursor update as `v_subrev_u32_dpp p, incl, p`, and that folded form returned the lane's OWN
// value instead of lane 3's on MI355X (measured: decode cursors diverged inside a quad; -O1, a
// ds_bpermute shuffle, or this asm form are all correct).  The asm statement cannot be combined
// with its consumer; `s_nop 1` covers the 2 wait states a DPP read needs after a VALU write of
// the same VGPR (hipcc pads nothing inside asm).
template <int CTRL> ANSX_D u32 quad_perm(u32 v)
{
    u32 r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=v"(r)
                 : "v"(v), "n"(CTRL & 3), "n"((CTRL >> 2) & 3), "n"((CTRL >> 4) & 3), "n"((CTRL >> 6) & 3));
    return r;
}

// inclusive prefix sum over the 4 lanes of a quad; *total receives the quad sum
ANSX_D u32 quad_incl_scan(u32 c, u32 ql, u32* total)
{
    u32 a = quad_perm<ANSX_QP(0, 0, 1, 2)>(c);
    u32 s1 = c + ((ql >= 1) ? a : 0u);
    u32 b = quad_perm<ANSX_QP(0, 0, 0, 1)>(s1);
    u32 s2 = s1 + ((ql >= 2) ? b : 0u);
    *total = quad_perm<ANSX_QP(3, 3, 3, 3)>(s2);
    return s2;
}

// Wave-wide (64 lanes) inclusive scan and reductions over DPP row shifts and row broadcasts -- six VALU moves instead
// of six ds_bpermute round trips through the LDS pipe (each ~100 cycles of latency on a dependent chain, which is what
// the per-block kernels with one or two waves in flight spend their time on).  Same explicit-asm form as quad_perm
// above, for the same reason.  Lanes switched off by EXEC and lanes outside a row contribute the identity (all-zero
// bits: 0, +0.0), so the caller may be inside divergent code as long as the lanes that matter are on.
//   steps 0..3: row_shr:1,2,4,8 (scan inside each row of 16); 4: row_bcast:15 into rows 1,3; 5: row_bcast:31 into
//   rows 2,3 -- the sequence LLVM's own atomic optimiser emits for gfx9.
#define ANSX_DPP_ASM(CTRL, ROWM)                                                                         \
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 " CTRL " row_mask:" ROWM " bank_mask:0xf" : "+v"(r) : "v"(v))
template <int STEP> ANSX_D u32 dpp_scan_mov(u32 v)
{
    u32 r = 0;
    if constexpr (STEP == 0) ANSX_DPP_ASM("row_shr:1", "0xf");
    else if constexpr (STEP == 1) ANSX_DPP_ASM("row_shr:2", "0xf");
    else if constexpr (STEP == 2) ANSX_DPP_ASM("row_shr:4", "0xf");
    else if constexpr (STEP == 3) ANSX_DPP_ASM("row_shr:8", "0xf");
    else if constexpr (STEP == 4) ANSX_DPP_ASM("row_bcast:15", "0xa");
    else ANSX_DPP_ASM("row_bcast:31", "0xc");
    return r;
}
template <int STEP, typename T> ANSX_D T dpp_scan_mov_t(T v)
{
    if constexpr (sizeof(T) == 4) {
        const u32 r = dpp_scan_mov<STEP>(__builtin_bit_cast(u32, v));
        return __builtin_bit_cast(T, r);
    } else {
        static_assert(sizeof(T) == 8, "32- or 64-bit values");
        const u64 b = __builtin_bit_cast(u64, v);
        const u32 lo = dpp_scan_mov<STEP>((u32)b), hi = dpp_scan_mov<STEP>((u32)(b >> 32));
        const u64 r = (u64)lo | ((u64)hi << 32);
        return __builtin_bit_cast(T, r);
    }
}
struct ansx_op_add { template <typename T> ANSX_D T operator()(T a, T b) const { return a + b; } };
struct ansx_op_max { template <typename T> ANSX_D T operator()(T a, T b) const { return a > b ? a : b; } };
struct ansx_op_or { template <typename T> ANSX_D T operator()(T a, T b) const { return a | b; } };
template <typename T, typename Op = ansx_op_add> ANSX_D T wave_incl_scan(T v, Op op = Op())
{
    v = op(v, dpp_scan_mov_t<0>(v));
    v = op(v, dpp_scan_mov_t<1>(v));
    v = op(v, dpp_scan_mov_t<2>(v));
    v = op(v, dpp_scan_mov_t<3>(v));
    v = op(v, dpp_scan_mov_t<4>(v));
    v = op(v, dpp_scan_mov_t<5>(v));
    return v;
}
// lane 63's value in every lane (a scalar broadcast)
template <typename T> ANSX_D T wave_last(T v)
{
    if constexpr (sizeof(T) == 4) {
        const u32 r = (u32)__builtin_amdgcn_readlane((int)__builtin_bit_cast(u32, v), 63);
        return __builtin_bit_cast(T, r);
    } else {
        const u64 b = __builtin_bit_cast(u64, v);
        const u64 r = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)b, 63) |
                      ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(b >> 32), 63) << 32);
        return __builtin_bit_cast(T, r);
    }
}
// reductions over the (active lanes of the) wave, result in every lane.  The order of additions is the scan's, not a
// butterfly's: only for integers and for floating-point sums that sit behind a guard band.
template <typename T> ANSX_D T wave_sum(T v) { return wave_last(wave_incl_scan(v, ansx_op_add())); }
template <typename T> ANSX_D T wave_max(T v) { return wave_last(wave_incl_scan(v, ansx_op_max())); }
template <typename T> ANSX_D T wave_or(T v) { return wave_last(wave_incl_scan(v, ansx_op_or())); }

// byte-granular global accesses.  gfx950 global memory supports unaligned dword/dwordx2
// accesses (HSA unaligned access mode); the packed structs make hipcc emit single
// global_load/store instructions with align 1.
struct __attribute__((packed)) ansx_u64_u { u64 v; };
struct __attribute__((packed)) ansx_u32_u { u32 v; };
struct __attribute__((packed)) ansx_u16_u { u16 v; };
ANSX_D u64 ld_u64_unaligned(const u8* p) { return ((const ansx_u64_u*)p)->v; }
ANSX_D u32 ld_u32_unaligned(const u8* p) { return ((const ansx_u32_u*)p)->v; }
ANSX_D void st_u64_unaligned(u8* p, u64 v) { ((ansx_u64_u*)p)->v = v; }
ANSX_D void st_u32_unaligned(u8* p, u32 v) { ((ansx_u32_u*)p)->v = v; }
ANSX_D void st_u16_unaligned(u8* p, u16 v) { ((ansx_u16_u*)p)->v = v; }

// ---------------------------------------------------------------------------------------------
// Block geometry shared by host and kernels
// ---------------------------------------------------------------------------------------------
struct ansx_geo {
    u64 n;           // total ints
    u32 block_ints;  // ints per block (last block may be shorter)
    u32 nblocks;
    u32 ckpt;        // restart interval in ints (0 = none)
    u32 nckf;        // restart points stored per block (stride)
    u32 f;           // fidelity (0 for ANSmsb)
    u32 kind;        // 0 fold, 1 rfold, 2 msb, 3 int
    u32 pa;          // 1: per-block alphabet compaction (ansx_pa.h)
    u32 ckw;         // restart points: 0 = packed 29-byte records, 1 = wide (u32 cursor + 4 x u64 states), see below
    ansx_map map;    // value <-> symbol map of this codec
    u64 payload_bytes;  // decode: bytes of block streams behind the container's payload offset (0 on the encode side)
    u32 trusted_index;  // decode: 1 = the two index entries were written by the host itself (single-stream mode: the
                        // plain reference stream has no index); 0 = they come from the container and every parser checks
                        // the pair of its own block (index_entry_ok).  Never derived from a container field.
    u32 pad_;
};

// Restart points in the container index.  A state is below 2^36 M and a cursor below the block's
// stream length, so with frames up to 2^16 and streams below 16 MiB a restart point is 4 x 52 + 24 bits = 29 bytes
// (container v3, the default): states 0 and 1 as one 104-bit little-endian integer in bytes 0..12 (state 0 in the low
// 52 bits), states 2 and 3 likewise in bytes 13..25, the cursor in bytes 26..28.  Anything larger (ANSint frames,
// huge blocks) keeps the v2 arrays -- u32 cursors, then 4 x u64 states -- and says so in the header (kind | 0x200).
#define ANSX_CK_RECORD 29u
#define ANSX_CK_STATE_BITS 52u
#define ANSX_CK_CURSOR_BITS 24u
#define ANSX_KIND_WIDE_RESTART 0x200u
ANSX_D void ckpt_load(const ansx_geo& g, const u64* __restrict__ ckpt_state, const u32* __restrict__ ckpt_off, u64 idx,
    u32 j, u64* st, u32* po)
{
    if (g.ckw) {
        *st = ckpt_state[idx * 4 + j];
        *po = ckpt_off[idx];
    } else {
        const u8* rec = (const u8*)ckpt_state + idx * ANSX_CK_RECORD;
        const u8* p = rec + 13u * (j >> 1) + 6u * (j & 1u);  // (an odd state starts at bit 52 = byte 6, bit 4)
        *st = (((const ansx_u64_u*)p)->v >> (4u * (j & 1u))) & ((1ull << ANSX_CK_STATE_BITS) - 1ull);
        *po = ((const ansx_u32_u*)(rec + 25))->v >> 8;
    }
}

// One entry pair of a container's block index, as untrusted as the payload: the same conditions k_validate_index
// applies to the whole index, for the one block a parser is about to touch (a reference stream has at least 2
// prelude bytes + one interpolative word + 32 state bytes; with compaction a one-value block is just its header).
ANSX_HD bool index_entry_ok(const ansx_geo& g, u32 b, u64 a, u64 e)
{
    const u64 min_bytes = g.pa ? 8 : 38;
    bool bad = (e < a) || (e - a) < min_bytes || (e - a) >= (1ull << 31) || e > g.payload_bytes;
    if (b == 0 && a != 0) bad = true;
    if (b == g.nblocks - 1 && e != g.payload_bytes) bad = true;
    return !bad;
}

ANSX_HD u32 geo_block_n(const ansx_geo& g, u32 b)
{
    u64 start = (u64)b * g.block_ints;
    u64 rem = g.n - start;
    return (u32)(rem < g.block_ints ? rem : g.block_ints);
}
// decoder segments of a block with nb ints
ANSX_HD u32 geo_nseg(u32 nb, u32 ckpt)
{
    u32 nfull = nb - (nb & 3u);
    if (ckpt == 0 || nfull == 0) return 1;
    return (nfull + ckpt - 1) / ckpt;
}

// Input generators of the reference's benchmark harness as counter-based functions of (seed, index):
// the same code runs in a HIP kernel (k_generate) and on the host (ansx_generate_host), bit for bit.
//
//   uniform(lo..hi)      src/generate_inputs.cpp:94-101  (std::uniform_int_distribution)
//   geometric(p)         src/generate_inputs.cpp:103-118 (std::geometric_distribution: failures before a success)
//   zipf(n, q)           include/zipf_dist.hpp:49-59     (Hoermann / Derflinger rejection-inversion)
//
// The reference draws from std::mt19937(0) through libstdc++ distributions and libm, which cannot be
// reproduced bit for bit on a GPU (SURVEY 8d "Generators"); what is kept is the distribution.  Every
// element is a pure function of its index -- uniforms come from splitmix64(seed, index, draw) -- and
// the only transcendental pieces (log, exp) are built from IEEE +,-,*,/ and fma like ansx_log2_portable,
// so host and device agree exactly (tests/test_gpu_parity.py::test_generators_device_equals_host).
#pragma once

#include "../../include/ansx.h"
#include "ansx_dev.h"

// distributions: ansx_gen_dist (include/ansx.h)

#pragma clang fp contract(off)

ANSX_HD u64 gen_mix(u64 seed, u64 index, u32 draw)
{
    u64 z = seed + (index + 1) * 0x9E3779B97F4A7C15ull + (u64)draw * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;  // splitmix64 finaliser
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// uniform double in (0, 1]: 53 random bits
ANSX_HD double gen_u01(u64 x) { return (double)((x >> 11) + 1) * (1.0 / 9007199254740992.0); }

// natural log / exp from the portable log2 and a fma polynomial (relative error ~1e-15; the
// generators need the distribution, not libm's last bit)
ANSX_HD double gen_log(double x) { return ansx_log2_portable(x) * 0.6931471805599453; }
ANSX_HD double gen_ldexp(double m, int e)
{
    // m * 2^e for |e| < 2000 without libm: two exact scalings
    const int e1 = e / 2, e2 = e - e1;
    const double s1 = ansx_bits_to_f64((u64)(e1 + 1023) << 52), s2 = ansx_bits_to_f64((u64)(e2 + 1023) << 52);
    return m * s1 * s2;
}
ANSX_HD double gen_exp(double x)
{
    if (x > 700.0) x = 700.0;
    if (x < -700.0) return 0.0;
    const double kf = __builtin_floor(x * 1.4426950408889634 + 0.5);
    const double r = __builtin_fma(-kf, 1.9082149292705877e-10, __builtin_fma(-kf, 0.6931471803691238, x));  // ln2 hi/lo
    // exp(r), |r| <= 0.347: Taylor to r^13 (< 1e-17)
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return gen_ldexp(p, (int)kf);
}
// (exp(x) - 1) / x  (zipf_dist.hpp:67-72), series where the subtraction would cancel
ANSX_HD double gen_expxm1bx(double x)
{
    const double ax = x < 0 ? -x : x;
    if (ax > 0.25) return (gen_exp(x) - 1.0) / x;
    double p = 1.0 / 87178291200.0;  // 1/14!
    p = __builtin_fma(p, x, 1.0 / 6227020800.0);
    p = __builtin_fma(p, x, 1.0 / 479001600.0);
    p = __builtin_fma(p, x, 1.0 / 39916800.0);
    p = __builtin_fma(p, x, 1.0 / 3628800.0);
    p = __builtin_fma(p, x, 1.0 / 362880.0);
    p = __builtin_fma(p, x, 1.0 / 40320.0);
    p = __builtin_fma(p, x, 1.0 / 5040.0);
    p = __builtin_fma(p, x, 1.0 / 720.0);
    p = __builtin_fma(p, x, 1.0 / 120.0);
    p = __builtin_fma(p, x, 1.0 / 24.0);
    p = __builtin_fma(p, x, 1.0 / 6.0);
    p = __builtin_fma(p, x, 0.5);
    p = __builtin_fma(p, x, 1.0);
    return p;
}
// log(1 + x) / x  (zipf_dist.hpp:87-92), x >= -1
ANSX_HD double gen_log1pxbx(double x)
{
    const double ax = x < 0 ? -x : x;
    if (ax > 0.125) return gen_log(1.0 + x) / x;
    // 1 - x/2 + x^2/3 - ... to x^19 (|x| <= 1/8: < 1e-18)
    double p = 0.0;
    for (int k = 20; k >= 1; k--) p = __builtin_fma(-p, x, 1.0 / (double)k);
    return p;
}

struct ansx_gen_params {
    u32 dist;
    u64 seed;
    double a, b;      // uniform: lo, hi (inclusive); geometric: p, -; zipf: n, q
    double c0, c1, c2;  // derived: geometric 1/log2(1-p); zipf H(1.5) - 1, H(n + 0.5), 1 - q
};

ANSX_HD double gen_zipf_H(double x, double omq)  // zipf_dist.hpp:80-84
{
    const double lx = gen_log(x);
    return gen_expxm1bx(omq * lx) * lx;
}
ANSX_HD double gen_zipf_Hinv(double x, double omq)  // zipf_dist.hpp:95-99
{
    double t = x * omq;
    if (t < -1.0) t = -1.0;
    return gen_exp(gen_log1pxbx(t) * x);
}

ANSX_HD void gen_prepare(ansx_gen_params* P)
{
    P->c0 = P->c1 = P->c2 = 0.0;
    if (P->dist == ANSX_GEN_GEOMETRIC) P->c0 = 1.0 / ansx_log2_portable(1.0 - P->a);
    if (P->dist == ANSX_GEN_ZIPF) {
        P->c2 = 1.0 - P->b;
        P->c0 = gen_zipf_H(1.5, P->c2) - 1.0;  // zipf_dist.hpp:44
        P->c1 = gen_zipf_H(P->a + 0.5, P->c2); // zipf_dist.hpp:45
    }
}

// One pass of the rejection loop of zipf_dist.hpp:49-59 for the canonical uniform u01 (the reference maps it to
// [H(x1), H(n)) the same way: libstdc++'s uniform_real_distribution is u01 * (b - a) + a): candidate value and
// whether it is accepted.  tests/test_generators.py checks this map against the reference class itself, driven by
// the same uniforms (oracle/ref_shim.cpp::ref_zipf_trace, tests/golden/zipf_trace.json).
ANSX_HD bool gen_zipf_try(const ansx_gen_params& P, double u01, u32* k_out)
{
    const double n = P.a, q = P.b, omq = P.c2;
    const double u = P.c0 + u01 * (P.c1 - P.c0);
    const double x = gen_zipf_Hinv(u, omq);
    double kr = __builtin_floor(x + 0.5);  // std::round for positive x
    kr = kr < 1.0 ? 1.0 : (kr > n ? n : kr);
    *k_out = (u32)kr;
    const double hk = gen_exp(-q * gen_log(kr));  // h(k) = k^-q, zipf_dist.hpp:102
    return u >= gen_zipf_H(kr + 0.5, omq) - hk;
}

ANSX_HD u32 gen_value(const ansx_gen_params& P, u64 index)
{
    if (P.dist == ANSX_GEN_UNIFORM) {
        const u64 range = (u64)(P.b - P.a) + 1;  // <= 2^32
        const u64 r = gen_mix(P.seed, index, 0) >> 32;
        return (u32)P.a + (u32)((r * range) >> 32);
    }
    if (P.dist == ANSX_GEN_GEOMETRIC) {
        // number of failures before the first success: floor(log(u) / log(1 - p)), u in (0, 1]
        const double u = gen_u01(gen_mix(P.seed, index, 0));
        const double k = __builtin_floor(ansx_log2_portable(u) * P.c0);
        return k >= 1073741823.0 ? 1073741823u : (u32)k;
    }
    // zipf over {1..n} with exponent q: rejection-inversion (zipf_dist.hpp:49-59); the loop is bounded so
    // that a device lane always terminates (the acceptance rate is above 80 %)
    u32 k = 1;
    for (u32 draw = 0; draw < 64; draw++)
        if (gen_zipf_try(P, gen_u01(gen_mix(P.seed, index, draw)), &k)) break;
    return k;
}

#if defined(__HIPCC__)
__global__ __launch_bounds__(256) void k_generate(ansx_gen_params P, u32* __restrict__ out, u64 n, u64 first_index)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = gen_value(P, first_index + i);
}
#endif

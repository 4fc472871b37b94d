/*
 * TEST INFRASTRUCTURE ONLY — see ans_oracle.h.  Clean-room CPU restatement of the reference's
 * ANSfold<f>/ANSrfold<f> path.  Compile with -ffp-contract=off (oracle/Makefile) so that the
 * double arithmetic of adjust_freqs is evaluated operation by operation, as written.
 *
 * Citations are file:line under /root/reference.
 */
#include "ans_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* include/ans_byte.hpp:24-31 — the constants ans_fold.hpp actually uses */
#define ANS_K 16ull
#define ANS_RADIX_LOG2 32

/* kind 2 = ANSmsb (include/ans_msb.hpp): same skeleton, different byte-stripping map */
#define ANS_ORACLE_MSB 2
#define ANS_ORACLE_INT 3
#define MSB_MAX_SIGMA 1280u /* ans_msb.hpp:28 */

static inline uint32_t fold_T(uint32_t f) { return 1u << (f + 7); }          /* ans_fold.hpp:43 */
static inline uint32_t fold_D(uint32_t f) { return 255u << (f - 1); }        /* ans_fold.hpp:47 */
static inline uint32_t fold_max_sigma(uint32_t f) { return 1u << (f + 9); }  /* ans_fold.hpp:70 */

/* include/ans_fold.hpp:38-50 and :52-65 */
uint32_t ans_oracle_fold(uint32_t f, uint32_t x, uint32_t* nbytes)
{
    const uint32_t T = fold_T(f), D = fold_D(f);
    uint32_t k = 0, off = 0;
    while (x >= T) {
        x >>= 8;
        off += D;
        k++;
    }
    if (nbytes) *nbytes = k;
    return x + off;
}

/* include/ans_fold.hpp:150-161 and :165-175 */
uint32_t ans_oracle_unfold(uint32_t f, uint32_t sym, uint32_t* nbytes)
{
    const uint32_t T = fold_T(f), D = fold_D(f);
    if (sym < T) {
        if (nbytes) *nbytes = 0;
        return sym;
    }
    uint32_t k = (sym - T) / D + 1;
    if (nbytes) *nbytes = k;
    return (sym - D * k) << (8 * k);
}

/* include/ans_msb.hpp:41-74 */
static uint32_t msb_map(uint32_t x, uint32_t* nbytes)
{
    uint32_t k = (x > 256u) + (x > (1u << 16)) + (x > (1u << 24));
    if (nbytes) *nbytes = k;
    return (x >> (8 * k)) + 256u * k;
}
/* include/ans_msb.hpp:159-180 */
static uint32_t msb_unmap(uint32_t sym, uint32_t* nbytes)
{
    uint32_t k = (sym > 256u) + (sym > 512u) + (sym > 768u);
    if (nbytes) *nbytes = k;
    return (sym - 256u * k) << (8 * k);
}

/* ---------------------------------------------------------------- normalisation */

typedef struct {
    uint64_t freq;
    uint32_t sym;
} fs_pair;

static int cmp_fs(const void* a, const void* b)
{
    const fs_pair* x = (const fs_pair*)a;
    const fs_pair* y = (const fs_pair*)b;
    if (x->freq != y->freq) return x->freq < y->freq ? -1 : 1;
    if (x->sym != y->sym) return x->sym < y->sym ? -1 : 1;
    return 0;
}

/* include/ans_util.hpp:77-95 */
static int scale_freqs(uint32_t* S, const uint64_t* F, const uint32_t* order, int64_t M,
    size_t sigma, uint64_t freq_sum)
{
    for (size_t j = 0; j < sigma; j++) {
        uint32_t s = order[j];
        double aratio = (double)M / (double)freq_sum;
        double v = aratio * (double)F[s];
        v = 0.5 + v;
        S[s] = (uint32_t)v;
        if (S[s] == 0) S[s] = 1;
        M -= S[s];
        freq_sum -= F[s];
        if (M < 0) break;
    }
    return M != 0;
}

/* include/util.hpp:271-282 */
static double entropy_u64(const uint64_t* freqs, size_t nfreqs, uint64_t freq_sum)
{
    double H0 = 0.0;
    double n = (double)freq_sum;
    for (size_t i = 0; i < nfreqs; i++) {
        if (freqs[i] != 0) {
            double p = (double)freqs[i] / n;
            double t = p * log2(p);
            H0 += t;
        }
    }
    return -H0;
}

/* include/util.hpp:284-298 — note the int accumulators (std::accumulate with init 0) */
static double cross_entropy_u64_u32(const uint64_t* P, size_t np, const uint32_t* Q, size_t nq)
{
    int ni = 0, mi = 0;
    for (size_t i = 0; i < np; i++) ni = (int)((unsigned)ni + (unsigned)P[i]);
    for (size_t i = 0; i < nq; i++) mi = (int)((unsigned)mi + (unsigned)Q[i]);
    double n = (double)ni, m = (double)mi;
    double H0 = 0.0;
    for (size_t i = 0; i < np; i++) {
        if (P[i] != 0 && i < nq && Q[i] != 0) {
            double p = (double)P[i] / n;
            double q = (double)Q[i] / m;
            double t = p * log2(q);
            H0 += t;
        }
    }
    return -H0;
}

/* include/ans_util.hpp:100-157, H_approx = 1; require_u16 as passed (true for the fold codecs and ANSmsb,
 * false for ANSint, ans_int.hpp:50) */
uint64_t ans_oracle_adjust_freqs_ex(const uint64_t* freqs, size_t nfreqs, uint32_t largest_sym,
    uint32_t* scaled, int require_u16)
{
    size_t sigma = 0;
    uint64_t freq_sum = 0;
    for (size_t i = 0; i < nfreqs; i++) {
        freq_sum += freqs[i];
        sigma += (freqs[i] != 0);
    }
    uint64_t target = sigma; /* :109-112, next_power_of_two :65-72 */
    if (!(target != 0 && (target & (target - 1)) == 0)) {
        if (target == 0) target = 1;
        else {
            uint32_t r = 63 - (uint32_t)__builtin_clzll(target);
            target = 1ull << (r + 1);
        }
    }
    fs_pair* sorted = (fs_pair*)malloc(sizeof(fs_pair) * (sigma ? sigma : 1));
    size_t c = 0;
    for (size_t i = 0; i < nfreqs; i++)
        if (freqs[i] != 0) {
            sorted[c].freq = freqs[i];
            sorted[c].sym = (uint32_t)i;
            c++;
        }
    qsort(sorted, sigma, sizeof(fs_pair), cmp_fs); /* :114-122 */
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (sigma ? sigma : 1));
    for (size_t i = 0; i < sigma; i++) order[i] = sorted[i].sym;
    free(sorted);

    double H = entropy_u64(freqs, nfreqs, freq_sum);
    size_t ns = (size_t)largest_sym + 1;
    uint32_t* prev = (uint32_t*)calloc(ns, sizeof(uint32_t));
    memset(scaled, 0, ns * sizeof(uint32_t));
    double approx_factor = 1.0 + (double)1 / (double)1000;
    double threshold = H * approx_factor;
    const uint32_t u16_limit = 65535;
    for (;;) {
        if (scale_freqs(scaled, freqs, order, (int64_t)target, sigma, freq_sum)) {
            target *= 2;
            continue;
        }
        uint32_t maxf = 0;
        for (size_t i = 0; i < ns; i++)
            if (scaled[i] > maxf) maxf = scaled[i];
        double XH = cross_entropy_u64_u32(freqs, nfreqs, scaled, ns);
        if (require_u16 && maxf >= u16_limit) {
            memcpy(scaled, prev, ns * sizeof(uint32_t));
            break;
        }
        if (XH < threshold) break;
        target *= 2;
        memcpy(prev, scaled, ns * sizeof(uint32_t));
    }
    free(prev);
    free(order);
    uint64_t M = 0;
    for (size_t i = 0; i < ns; i++) M += scaled[i];
    return M;
}

uint64_t ans_oracle_adjust_freqs(const uint64_t* freqs, size_t nfreqs, uint32_t largest_sym,
    uint32_t* scaled)
{
    return ans_oracle_adjust_freqs_ex(freqs, nfreqs, largest_sym, scaled, 1);
}

/* ---------------------------------------------------------------- bit I/O + interpolative */

/* include/bits.hpp:84-105,146-218 restated as a plain LSB-first writer into LE u32 words.
 * The reference leaves the unused high bits of the final word indeterminate;
 * the canonical form written here has them zero. */
typedef struct {
    uint8_t* out;
    uint64_t acc;
    uint32_t nacc;
    size_t bytes;
    uint64_t total_bits;
} bitw;

static void bw_put(bitw* w, uint32_t val, uint32_t bits)
{
    if (bits == 0) return; /* bits.hpp:174-175 */
    if (bits < 32) val &= (1u << bits) - 1u;
    w->acc |= (uint64_t)val << w->nacc;
    w->nacc += bits;
    w->total_bits += bits;
    if (w->nacc >= 32) {
        uint32_t word = (uint32_t)w->acc;
        memcpy(w->out + w->bytes, &word, 4);
        w->bytes += 4;
        w->acc >>= 32;
        w->nacc -= 32;
    }
}

static size_t bw_flush(bitw* w)
{
    if (w->nacc != 0) { /* bits.hpp:208-216 */
        uint32_t word = (uint32_t)w->acc;
        memcpy(w->out + w->bytes, &word, 4);
        w->bytes += 4;
        w->acc = 0;
        w->nacc = 0;
    }
    return w->bytes;
}

typedef struct {
    const uint8_t* in;
    uint64_t bitpos;
} bitr;

static uint64_t br_get(bitr* r, uint32_t bits)
{
    if (bits == 0) return 0; /* bits.hpp:187-188 */
    uint64_t v = 0;
    /* read up to 32 bits starting at bitpos, LSB first, words are LE so this is a plain
     * little-endian bit string */
    size_t byte = (size_t)(r->bitpos >> 3);
    uint32_t sh = (uint32_t)(r->bitpos & 7);
    uint64_t window = 0;
    for (int i = 0; i < 5; i++) window |= (uint64_t)r->in[byte + i] << (8 * i);
    v = (window >> sh) & ((bits >= 32) ? 0xFFFFFFFFull : ((1ull << bits) - 1ull));
    r->bitpos += bits;
    return v;
}

static inline uint32_t hi_bit(uint64_t x) /* bits.hpp:33-39 */
{
    if (x == 0) return 0;
    return 63 - (uint32_t)__builtin_clzll(x);
}

/* include/interp.hpp:28-46 */
static void write_center_mid(bitw* os, uint64_t val, uint64_t u)
{
    if (u == 1) return;
    uint64_t b = hi_bit(u - 1) + 1ull;
    uint64_t d = 2ull * u - (1ull << b);
    val = val + (u - (d >> 1));
    if (val > u) val -= u;
    uint32_t m = (uint32_t)((1ull << b) - u);
    if (val <= m) {
        bw_put(os, (uint32_t)(val - 1ull), (uint32_t)(b - 1ull));
    } else {
        val += m;
        bw_put(os, (uint32_t)((val - 1ull) >> 1), (uint32_t)(b - 1ull));
        bw_put(os, (uint32_t)((val - 1ull) & 1ull), 1);
    }
}

/* include/interp.hpp:47-63 */
static uint64_t read_center_mid(bitr* is, uint64_t u)
{
    uint64_t b = (u == 1ull) ? 0ull : hi_bit(u - 1ull) + 1ull;
    uint64_t d = 2ull * u - (1ull << b);
    uint64_t val = 1ull;
    if (u != 1) {
        uint64_t m = (1ull << b) - u;
        val = br_get(is, (uint32_t)(b - 1)) + 1;
        if (val > m) val = (2ull * val + br_get(is, 1)) - m - 1ull;
    }
    val = val + (d >> 1);
    if (val > u) val -= u;
    return val;
}

/* include/interp.hpp:65-79 */
static void encode_interp(bitw* os, const uint32_t* in_buf, size_t n, uint64_t low, uint64_t high)
{
    if (n == 0) return;
    uint64_t h = (n + 1ull) >> 1;
    uint64_t n1 = h - 1ull;
    uint64_t n2 = n - h;
    uint64_t v = (uint64_t)in_buf[h - 1ull] + 1ull;
    write_center_mid(os, v - low - n1 + 1ull, high - n2 - low - n1 + 1ull);
    encode_interp(os, in_buf, (size_t)n1, low, v - 1ull);
    encode_interp(os, in_buf + h, (size_t)n2, v + 1ull, high);
}

/* include/interp.hpp:81-97 */
static void decode_interp(bitr* is, uint32_t* out_buf, size_t n, uint64_t low, uint64_t high)
{
    if (n == 0) return;
    uint64_t h = (n + 1ull) >> 1;
    uint64_t n1 = h - 1ull;
    uint64_t n2 = n - h;
    uint64_t v = low + n1 - 1ull + read_center_mid(is, high - n2 - low - n1 + 1ull);
    out_buf[h - 1] = (uint32_t)(v - 1);
    if (n1) decode_interp(is, out_buf, (size_t)n1, low, v - 1ull);
    if (n2) decode_interp(is, out_buf + h, (size_t)n2, v + 1ull, high);
}

/* include/vbyte.hpp:57-80 */
static uint8_t* vbyte_put(uint8_t* out, uint32_t x)
{
    while (x >= 128) {
        *out++ = (uint8_t)((x & 127) | 128);
        x >>= 7;
    }
    *out++ = (uint8_t)(x & 127);
    return out;
}

/* include/vbyte.hpp:82-95 */
static const uint8_t* vbyte_get(const uint8_t* in, uint32_t* x)
{
    uint32_t v = 0, shift = 0;
    for (;;) {
        uint8_t c = *in++;
        v += (uint32_t)(c & 127) << shift;
        if (!(c & 128)) break;
        shift += 7;
    }
    *x = v;
    return in;
}

/* include/ans_util.hpp:46-63 */
size_t ans_oracle_write_prelude(const uint32_t* nfreqs, size_t nsyms, uint64_t frame_size,
    uint8_t* out, uint32_t* valid_bits)
{
    uint8_t* p = out;
    uint32_t max_sym = (uint32_t)(nsyms - 1);
    p = vbyte_put(p, max_sym);
    *p++ = (uint8_t)hi_bit(frame_size); /* log2 of a power of two, :51 */
    uint32_t* inc = (uint32_t*)malloc(sizeof(uint32_t) * nsyms);
    inc[0] = nfreqs[0];
    for (size_t s = 1; s <= max_sym; s++) inc[s] = inc[s - 1] + nfreqs[s] + 1;
    bitw w;
    memset(&w, 0, sizeof(w));
    w.out = p;
    /* interp.hpp:100-108: low = 1, high = u + 1, u = frame_size + nsyms + 1 */
    encode_interp(&w, inc, nsyms, 1, frame_size + nsyms + 1 + 1);
    size_t wb = bw_flush(&w);
    if (valid_bits) *valid_bits = (uint32_t)w.total_bits;
    free(inc);
    return (size_t)(p - out) + wb;
}

/* include/ans_util.hpp:25-42 */
size_t ans_oracle_read_prelude(const uint8_t* in, uint32_t* nfreqs, uint32_t* frame_log2)
{
    uint32_t max_sym;
    const uint8_t* p = vbyte_get(in, &max_sym);
    uint32_t lg = *p++;
    uint64_t frame_size = 1ull << lg;
    size_t nsyms = (size_t)max_sym + 1;
    bitr r;
    r.in = p;
    r.bitpos = 0;
    decode_interp(&r, nfreqs, nsyms, 1, frame_size + nsyms + 1 + 1);
    uint32_t prev = nfreqs[0];
    for (size_t s = 1; s <= max_sym; s++) {
        uint32_t cur = nfreqs[s];
        nfreqs[s] = cur - prev - 1;
        prev = cur;
    }
    if (frame_log2) *frame_log2 = lg;
    return nsyms;
}

/* ---------------------------------------------------------------- encode */

typedef struct {
    uint32_t freq; /* reference keeps u16 (ans_fold.hpp:30-34); require_u16 guarantees the range */
    uint32_t base;
} enc_entry;

static void radix_sort_u32(uint32_t* a, uint32_t* tmp, size_t n)
{
    for (int pass = 0; pass < 4; pass++) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (size_t i = 0; i < n; i++) cnt[((a[i] >> sh) & 255) + 1]++;
        for (int i = 0; i < 256; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; i++) tmp[cnt[(a[i] >> sh) & 255]++] = a[i];
        uint32_t* t = a;
        a = tmp;
        tmp = t;
    }
}

typedef struct {
    uint64_t count;
    uint32_t value;
} cv_pair;

static int cmp_cv(const void* a, const void* b) /* (-count, value) ascending: ans_reorder_fold.hpp:79-85 */
{
    const cv_pair* x = (const cv_pair*)a;
    const cv_pair* y = (const cv_pair*)b;
    if (x->count != y->count) return x->count > y->count ? -1 : 1;
    if (x->value != y->value) return x->value < y->value ? -1 : 1;
    return 0;
}

size_t ans_oracle_bound(int kind, uint32_t f, size_t n)
{
    /* ANSint: the model spans every value up to the largest; this oracle covers inputs whose largest value
     * is below 16384 (the alphabet the GPU path takes: lists of small values, ans_int.hpp:40-48) or does not exceed
     * n + 1024 (the per-block dense remap of pseudo_adaptive.cpp) */
    if (kind == ANS_ORACLE_INT) return 16 + 8 * ((n > 16384 ? n : 16384) + 1027) + 4 * n + 32 + 64;
    size_t nsyms_max = kind == ANS_ORACLE_MSB ? MSB_MAX_SIGMA : (size_t)fold_T(f) + 3u * (size_t)fold_D(f);
    size_t hdr = kind == ANS_ORACLE_RFOLD ? 4 + 4 * (size_t)fold_T(f) : 0;
    return hdr + 8 + 4 * nsyms_max + 8 + 7 * n + 32;
}

/* ANSint on values beyond that range (the GPU path models them in rank space, ansx_intsparse.h; the restatement stays the
 * reference's dense one): the prelude codes max_value + 1 items, of which only nodes with a present symbol in reach carry
 * bits -- at most 62 nodes of at most 31 bits per distinct value. */
size_t ans_oracle_bound_int(size_t n, uint32_t max_value)
{
    size_t b = ans_oracle_bound(ANS_ORACLE_INT, 1, n);
    if (max_value >= 16384 && (size_t)max_value > n + 1024) b += 256 * n + 64;
    return b;
}

size_t ans_oracle_encode(int kind, uint32_t f, const uint32_t* in, size_t n, uint8_t* out,
    size_t cap, ans_oracle_info* info, size_t ckpt_interval, uint64_t* ckpt_states,
    uint32_t* ckpt_off, size_t* n_ckpt)
{
    if (kind == ANS_ORACLE_MSB || kind == ANS_ORACLE_INT) f = 1; /* fidelity is not a parameter of ANSmsb / ANSint */
    if (n == 0 || f < 1 || f > 7) return 0; /* n == 0 never terminates in the reference (F4) */
    uint32_t int_max = 0; /* ANSint (ans_int.hpp:40-48): symbols are the values, alphabet = max value + 1 */
    if (kind == ANS_ORACLE_INT) {
        for (size_t i = 0; i < n; i++)
            if (in[i] > int_max) int_max = in[i];
        if (int_max >= (1u << 30)) return 0; /* the code's universe M + max + 2 is kept below 2^31 */
        if (cap < ans_oracle_bound_int(n, int_max)) return 0;
    } else if (cap < ans_oracle_bound(kind, f, n)) return 0;
    const uint32_t T = kind == ANS_ORACLE_INT ? 0xFFFFFFFFu : fold_T(f); /* ANSint never strips bytes */
    const uint32_t MAX_SIGMA = kind == ANS_ORACLE_MSB ? MSB_MAX_SIGMA : (kind == ANS_ORACLE_INT ? int_max + 1 : fold_max_sigma(f));
    ans_oracle_info local;
    memset(&local, 0, sizeof(local));

    /* ---- rfold: ans_reorder_fold.hpp:70-106 */
    uint32_t* most_frequent = NULL; /* T entries when reorder */
    cv_pair* runs = NULL;
    size_t nruns = 0;
    int reorder = 0;
    if (kind == ANS_ORACLE_RFOLD) {
        uint32_t* sorted = (uint32_t*)malloc(sizeof(uint32_t) * n);
        uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * n);
        memcpy(sorted, in, sizeof(uint32_t) * n);
        radix_sort_u32(sorted, tmp, n);
        free(tmp);
        runs = (cv_pair*)malloc(sizeof(cv_pair) * n);
        for (size_t i = 0; i < n;) {
            size_t j = i;
            while (j < n && sorted[j] == sorted[i]) j++;
            runs[nruns].count = j - i;
            runs[nruns].value = sorted[i];
            nruns++;
            i = j;
        }
        free(sorted);
        local.sigma = nruns;
        if (nruns >= T) { /* :94-106 */
            reorder = 1;
            qsort(runs, nruns, sizeof(cv_pair), cmp_cv);
            most_frequent = (uint32_t*)malloc(sizeof(uint32_t) * T);
            for (uint32_t i = 0; i < T; i++) most_frequent[i] = runs[i].value;
        }
    }
    /* value -> mapped value: lookup structure for the top-T values, sorted by value
     * (count field reused as the rank 0..T-1) */
    cv_pair* top = NULL;
    if (reorder) {
        top = (cv_pair*)malloc(sizeof(cv_pair) * T);
        for (uint32_t i = 0; i < T; i++) {
            top[i].value = most_frequent[i];
            top[i].count = i;
        }
        for (uint32_t gap = T / 2; gap > 0; gap /= 2) /* shell sort by value */
            for (uint32_t i = gap; i < T; i++) {
                cv_pair t = top[i];
                uint32_t j = i;
                while (j >= gap && top[j - gap].value > t.value) {
                    top[j] = top[j - gap];
                    j -= gap;
                }
                top[j] = t;
            }
    }
#define MAP_VALUE(v, dst)                                                                        \
    do {                                                                                         \
        uint32_t _v = (v);                                                                       \
        if (!reorder) (dst) = _v;                                                                \
        else {                                                                                   \
            uint32_t lo = 0, hi = T;                                                             \
            while (lo < hi) {                                                                    \
                uint32_t mid = (lo + hi) >> 1;                                                   \
                if (top[mid].value < _v) lo = mid + 1;                                           \
                else hi = mid;                                                                   \
            }                                                                                    \
            if (lo < T && top[lo].value == _v) (dst) = (uint32_t)top[lo].count;                  \
            else (dst) = _v + T; /* :99-101 */                                                   \
        }                                                                                        \
    } while (0)

    /* ---- histogram of folded symbols: ans_fold.hpp:70-78 / ans_reorder_fold.hpp:107-115 */
    uint64_t* freqs = (uint64_t*)calloc(MAX_SIGMA, sizeof(uint64_t));
    uint32_t max_sym = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t mv;
        MAP_VALUE(in[i], mv);
        uint32_t s = kind == ANS_ORACLE_MSB ? msb_map(mv, NULL) : (kind == ANS_ORACLE_INT ? mv : ans_oracle_fold(f, mv, NULL));
        freqs[s]++;
        if (s > max_sym) max_sym = s;
    }
    if (kind != ANS_ORACLE_RFOLD) {
        for (uint32_t i = 0; i < MAX_SIGMA; i++) local.sigma += (freqs[i] != 0);
    }
    local.present_syms = 0;
    for (uint32_t i = 0; i < MAX_SIGMA; i++) local.present_syms += (freqs[i] != 0);
    size_t nsyms = (size_t)max_sym + 1;
    uint32_t* nfreqs = (uint32_t*)calloc(nsyms, sizeof(uint32_t));
    uint64_t M = ans_oracle_adjust_freqs_ex(freqs, MAX_SIGMA, max_sym, nfreqs, kind != ANS_ORACLE_INT); /* :79; ans_int.hpp:50 */
    free(freqs);
    if (M == 0 || (M & (M - 1)) != 0) { /* degenerate "prev is all zero" exit */
        free(nfreqs);
        free(most_frequent);
        free(runs);
        free(top);
        return 0;
    }
    enc_entry* table = (enc_entry*)malloc(sizeof(enc_entry) * nsyms); /* :82-91 */
    uint64_t cur_base = 0;
    for (size_t s = 0; s < nsyms; s++) {
        table[s].freq = nfreqs[s];
        table[s].base = (uint32_t)cur_base;
        cur_base += nfreqs[s];
    }
    const uint64_t lower_bound = ANS_K * M;
    uint32_t log2M = hi_bit(M);

    /* ---- serialize: ans_reorder_fold.hpp:132-154, ans_fold.hpp:95-98 */
    uint8_t* p = out;
    if (kind == ANS_ORACLE_RFOLD) {
        uint32_t flag = reorder ? 1u : 0u;
        memcpy(p, &flag, 4);
        p += 4;
        if (reorder) {
            memcpy(p, most_frequent, 4 * (size_t)T);
            p += 4 * (size_t)T;
        }
        local.header_bytes = (uint32_t)(p - out);
        local.reorder_flag = flag;
    }
    uint32_t vbits = 0;
    size_t pb = ans_oracle_write_prelude(nfreqs, nsyms, M, p, &vbits);
    p += pb;
    local.prelude_bytes = (uint32_t)pb;
    local.interp_bits = vbits;
    local.max_sym = max_sym;
    local.log2_frame = log2M;

    /* ---- 4-state encode, backwards: ans_fold.hpp:249-272 */
    uint64_t st[4] = { lower_bound, lower_bound, lower_bound, lower_bound };
    const uint32_t D = fold_D(f);
    size_t nck = 0;
    size_t r = n % 4;
    size_t nfull = n - r;
    size_t nseg = 1;
    if (ckpt_interval) {
        nseg = nfull ? (nfull + ckpt_interval - 1) / ckpt_interval : 1;
        nck = nseg - 1;
    }

#define ENC_SYM(state, value)                                                                    \
    do {                                                                                         \
        uint32_t x;                                                                              \
        MAP_VALUE((value), x);                                                                   \
        uint32_t off = 0;                                                                        \
        if (kind == ANS_ORACLE_MSB) { /* ans_msb.hpp:52-74: low bytes first, like fold */        \
            uint32_t kk;                                                                         \
            uint32_t sy = msb_map(x, &kk);                                                       \
            for (uint32_t bi = 0; bi < kk; bi++) *p++ = (uint8_t)((x >> (8 * bi)) & 0xFF);       \
            x = sy;                                                                              \
        } else                                                                                   \
            while (x >= T) { /* ans_fold.hpp:57-61 */                                            \
                *p++ = (uint8_t)(x & 0xFF);                                                      \
                x >>= 8;                                                                         \
                off += D;                                                                        \
            }                                                                                    \
        const enc_entry* e = &table[x + off];                                                    \
        uint64_t ub = ((uint64_t)e->freq) << (ANS_RADIX_LOG2 + 4); /* K*RADIX*freq, :89 */       \
        if ((state) >= ub) { /* :105-110 */                                                      \
            uint32_t w = (uint32_t)((state) & 0xFFFFFFFFull);                                    \
            memcpy(p, &w, 4);                                                                    \
            p += 4;                                                                              \
            (state) >>= ANS_RADIX_LOG2;                                                          \
        }                                                                                        \
        (state) = (((state) / e->freq) * M) + ((state) % e->freq) + e->base; /* :111 */          \
    } while (0)

    size_t cur = 0;
    while ((n - cur) % 4 != 0) { /* :257-261 */
        ENC_SYM(st[0], in[n - cur - 1]);
        cur++;
    }
    while (cur != n) { /* :262-272 */
        ENC_SYM(st[0], in[n - cur - 1]);
        ENC_SYM(st[1], in[n - cur - 2]);
        ENC_SYM(st[2], in[n - cur - 3]);
        ENC_SYM(st[3], in[n - cur - 4]);
        cur += 4;
        size_t idx = n - cur; /* every symbol with index >= idx has been encoded */
        if (ckpt_interval && idx > 0 && idx % ckpt_interval == 0) {
            size_t s = idx / ckpt_interval; /* decoder segment s starts here */
            if (ckpt_states)
                for (int j = 0; j < 4; j++) ckpt_states[4 * (s - 1) + j] = st[j];
            if (ckpt_off) ckpt_off[s - 1] = (uint32_t)(p - out);
        }
    }
    for (int j = 0; j < 4; j++) { /* :275-278, :115-120 */
        uint64_t v = st[j] - lower_bound;
        memcpy(p, &v, 8);
        p += 8;
        local.final_states[j] = st[j];
    }
    if (n_ckpt) *n_ckpt = nck;
    if (info) *info = local;
    free(table);
    free(nfreqs);
    free(most_frequent);
    free(runs);
    free(top);
    return (size_t)(p - out);
#undef ENC_SYM
#undef MAP_VALUE
}

/* ---------------------------------------------------------------- decode */

typedef struct {
    uint32_t freq;   /* u16 in the fold codecs; ANSint's LARGE table has 32 bits (ans_int.hpp:100-110) */
    uint32_t offset;
    uint32_t value;  /* unfolded high part (reference packs this with nbytes, ans_fold.hpp:198-200) */
    uint32_t nbytes;
} dec_entry;

int ans_oracle_decode(int kind, uint32_t f, const uint8_t* in, size_t nbytes_in, uint32_t* out,
    size_t n, int ref_f3_compat)
{
    if (kind == ANS_ORACLE_MSB || kind == ANS_ORACLE_INT) f = 1;
    if (f < 1 || f > 7) return -1;
    const uint32_t T = fold_T(f);
    const uint8_t* p = in;
    uint32_t flag = 0;
    const uint8_t* mf = NULL;
    if (kind == ANS_ORACLE_RFOLD) { /* ans_reorder_fold.hpp:238-254 */
        memcpy(&flag, p, 4);
        p += 4;
        if (flag == 1) {
            mf = p;
            p += 4 * (size_t)T;
        }
    }
    uint32_t max_sym_peek; /* guard against corrupt input: the alphabet has < 2^(f+9) symbols */
    vbyte_get(p, &max_sym_peek);
    if (kind == ANS_ORACLE_INT ? (max_sym_peek >= (1u << 30))
                               : max_sym_peek >= (kind == ANS_ORACLE_MSB ? MSB_MAX_SIGMA : fold_max_sigma(f)))
        return -2;
    uint32_t* nfreqs = (uint32_t*)calloc((size_t)max_sym_peek + 2, sizeof(uint32_t));
    uint32_t lg = 0;
    size_t nsyms = ans_oracle_read_prelude(p, nfreqs, &lg);
    uint64_t M = 0;
    for (size_t s = 0; s < nsyms; s++) M += nfreqs[s];
    if (M != (1ull << lg)) {
        free(nfreqs);
        return -3;
    }
    dec_entry* table = (dec_entry*)malloc(sizeof(dec_entry) * M); /* ans_fold.hpp:190-204 */
    uint64_t base = 0;
    for (size_t s = 0; s < nsyms; s++) {
        uint32_t k;
        uint32_t val;
        if (kind == ANS_ORACLE_INT) { /* ans_int.hpp:134-143: entry->sym = sym */
            val = (uint32_t)s;
            k = 0;
        } else
            val = kind == ANS_ORACLE_MSB ? msb_unmap((uint32_t)s, &k) : ans_oracle_unfold(f, (uint32_t)s, &k);
        if (kind == ANS_ORACLE_RFOLD) {
            /* ans_reorder_fold.hpp:207-219: s < T -> most_frequent[s] + T; final "- T" at :301 */
            if (flag == 1) {
                if (s < T) {
                    uint32_t v;
                    memcpy(&v, mf + 4 * s, 4);
                    val = v + T;
                }
            } else if (ref_f3_compat) {
                if (s < T) val = (uint32_t)s + T; /* most_frequent[i] = i, :251-253 */
            }
        }
        for (uint32_t k2 = 0; k2 < nfreqs[s]; k2++) {
            table[base + k2].freq = nfreqs[s];
            table[base + k2].offset = k2;
            table[base + k2].value = val;
            table[base + k2].nbytes = k;
        }
        base += nfreqs[s];
    }
    free(nfreqs);
    const uint64_t lower_bound = ANS_K * M;
    const uint64_t mask = M - 1;
    const uint32_t sub = (kind == ANS_ORACLE_RFOLD && (flag == 1 || ref_f3_compat)) ? T : 0;
    static const uint32_t emask[4] = { 0x0, 0xFF, 0xFFFF, 0xFFFFFF };

    const uint8_t* q = in + nbytes_in; /* ans_fold.hpp:289-295 */
    uint64_t st[4];
    for (int j = 3; j >= 0; j--) {
        uint64_t v;
        q -= 8;
        memcpy(&v, q, 8);
        st[j] = v + lower_bound;
    }

#define DEC_SYM(state, dst)                                                                      \
    do {                                                                                         \
        const dec_entry* e = &table[(state) & mask];                                             \
        (state) = (uint64_t)e->freq * ((state) >> lg) + (uint64_t)e->offset; /* :218-220 */      \
        if ((state) < lower_bound) { /* :221-225 */                                              \
            uint32_t w;                                                                          \
            q -= 4;                                                                              \
            memcpy(&w, q, 4);                                                                    \
            (state) = ((state) << ANS_RADIX_LOG2) | (uint64_t)w;                                 \
        }                                                                                        \
        uint32_t ex = 0; /* :135-147 */                                                          \
        q -= e->nbytes;                                                                          \
        if (e->nbytes) {                                                                         \
            uint32_t w = 0;                                                                      \
            memcpy(&w, q, e->nbytes);                                                            \
            ex = w & emask[e->nbytes];                                                           \
        }                                                                                        \
        (dst) = e->value + ex - sub;                                                             \
    } while (0)

    size_t i = 0;
    size_t fast = n - (n % 4);
    while (i != fast) { /* :300-306 */
        DEC_SYM(st[3], out[i]);
        DEC_SYM(st[2], out[i + 1]);
        DEC_SYM(st[1], out[i + 2]);
        DEC_SYM(st[0], out[i + 3]);
        i += 4;
    }
    while (i != n) { /* :307-310 */
        DEC_SYM(st[0], out[i]);
        i++;
    }
#undef DEC_SYM
    free(table);
    return 0;
}


/* ---------------------------------------------------------------- per-block alphabet compaction */

/* One block of src/pseudo_adaptive.cpp run<t_compressor>() (:85-130): u32 alphabet size, u32 universe
 * (sum of the block's distinct values + 1, u32 arithmetic), interpolative code of the running sums of the
 * ascending distinct values (:106-113; interp.hpp:99-108), then t_compressor::encode of the block remapped to
 * 1-based ranks (:91-103,115-123) -- nothing when the block has a single distinct value.  Restart points of
 * the codec stream are reported relative to the start of the WHOLE block stream. */
size_t ans_oracle_pa_encode(int kind, uint32_t f, const uint32_t* in, size_t n, uint8_t* out, size_t cap,
    ans_oracle_pa_info* pinfo, ans_oracle_info* info, size_t ckpt_interval, uint64_t* ckpt_states,
    uint32_t* ckpt_off, size_t* n_ckpt)
{
    if (n == 0 || kind == ANS_ORACLE_RFOLD) return 0;
    uint32_t* sorted = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * n);
    memcpy(sorted, in, sizeof(uint32_t) * n);
    radix_sort_u32(sorted, tmp, n);
    size_t sigma = 0;
    for (size_t i = 0; i < n; i++)
        if (i == 0 || sorted[i] != sorted[i - 1]) tmp[sigma++] = sorted[i]; /* distinct values, ascending */
    /* remap: rank (1-based) of every value; running sums of the alphabet in u32 */
    uint32_t* mapped = (uint32_t*)malloc(sizeof(uint32_t) * n);
    for (size_t i = 0; i < n; i++) {
        size_t lo = 0, hi = sigma;
        while (lo < hi) {
            size_t mid = (lo + hi) >> 1;
            if (tmp[mid] < in[i]) lo = mid + 1;
            else hi = mid;
        }
        mapped[i] = (uint32_t)lo + 1;
    }
    uint64_t exact = 0;
    for (size_t i = 0; i < sigma; i++) {
        exact += tmp[i];
        sorted[i] = (uint32_t)exact; /* block_alphabet[k] += block_alphabet[k - 1], :103-105 */
    }
    if (exact >= 0xFFFFFFFFull || cap < 8 + 4 * sigma + 8 + ans_oracle_bound(kind, f, n)) { /* u32 sums must not wrap */
        free(sorted);
        free(tmp);
        free(mapped);
        return 0;
    }
    uint32_t universe = sorted[sigma - 1] + 1;
    uint32_t sg = (uint32_t)sigma;
    memcpy(out, &sg, 4);
    memcpy(out + 4, &universe, 4);
    bitw w;
    memset(&w, 0, sizeof(w));
    w.out = out + 8;
    encode_interp(&w, sorted, sigma, 1, (uint64_t)universe + 1);
    size_t hb = 8 + bw_flush(&w);
    if (pinfo) {
        pinfo->sigma = sg;
        pinfo->universe = universe;
        pinfo->header_bytes = (uint32_t)hb;
        pinfo->interp_bits = (uint32_t)w.total_bits;
    }
    size_t total = hb;
    if (n_ckpt) *n_ckpt = 0;
    if (info) memset(info, 0, sizeof(*info));
    if (sigma != 1) {
        size_t nck = 0;
        size_t cb = ans_oracle_encode(kind, f, mapped, n, out + hb, cap - hb, info, ckpt_interval, ckpt_states, ckpt_off, &nck);
        if (cb == 0) total = 0;
        else {
            total += cb;
            if (ckpt_off)
                for (size_t i = 0; i < nck; i++) ckpt_off[i] += (uint32_t)hb;
            if (n_ckpt) *n_ckpt = nck;
        }
    }
    free(sorted);
    free(tmp);
    free(mapped);
    return total;
}

/* the inverse (this build's own: pseudo_adaptive.cpp never decodes) */
int ans_oracle_pa_decode(int kind, uint32_t f, const uint8_t* in, size_t nbytes, uint32_t* out, size_t n)
{
    if (nbytes < 8) return -1;
    uint32_t sigma, universe;
    memcpy(&sigma, in, 4);
    memcpy(&universe, in + 4, 4);
    if (sigma == 0 || sigma > n) return -2;
    uint32_t* alpha = (uint32_t*)malloc(sizeof(uint32_t) * sigma);
    bitr r;
    r.in = in + 8;
    r.bitpos = 0;
    decode_interp(&r, alpha, sigma, 1, (uint64_t)universe + 1);
    size_t hb = 8 + 4 * (size_t)((r.bitpos + 31) / 32);
    for (size_t i = sigma; i-- > 1;) alpha[i] -= alpha[i - 1]; /* running sums -> values */
    int rc = 0;
    if (sigma == 1) {
        for (size_t i = 0; i < n; i++) out[i] = alpha[0];
    } else {
        rc = ans_oracle_decode(kind, f, in + hb, nbytes - hb, out, n, 0);
        for (size_t i = 0; rc == 0 && i < n; i++) {
            if (out[i] < 1 || out[i] > sigma) rc = -4;
            else out[i] = alpha[out[i] - 1];
        }
    }
    free(alpha);
    return rc;
}


/* ---------------------------------------------------------------- container index: parse hints */

static void hints_walk(bitr* is, size_t n, uint64_t low, uint64_t high, uint32_t depth, uint32_t bfs, uint32_t* hints)
{
    if (n == 0) return;
    uint64_t h = (n + 1ull) >> 1;
    uint64_t n1 = h - 1ull;
    uint64_t n2 = n - h;
    uint64_t v = low + n1 - 1ull + read_center_mid(is, high - n2 - low - n1 + 1ull);
    hints_walk(is, (size_t)n1, low, v - 1ull, depth + 1, 2 * bfs + 1, hints);
    if (depth < 3 && n2) hints[1 + bfs] = (uint32_t)is->bitpos; /* the right subtree's first item starts here */
    hints_walk(is, (size_t)n2, v + 1ull, high, depth + 1, 2 * bfs + 2, hints);
}

/* This build's container carries, per block, where the interpolative code of the codec prelude can be
 * entered in parallel: hints[0] = valid bits of the code, hints[1 + i] = bit offset of
 * the first item of the RIGHT subtree of top node i (the 7 nodes of the first three levels, breadth-first:
 * children of i are 2i+1, 2i+2), 0 where that subtree is empty.  `prelude` points at the vbyte(max_sym). */
void ans_oracle_prelude_hints(const uint8_t* prelude, uint32_t* hints)
{
    uint32_t max_sym;
    const uint8_t* p = vbyte_get(prelude, &max_sym);
    uint32_t lg = *p++;
    size_t nsyms = (size_t)max_sym + 1;
    bitr r;
    r.in = p;
    r.bitpos = 0;
    for (int i = 0; i < 8; i++) hints[i] = 0;
    hints_walk(&r, nsyms, 1, (1ull << lg) + nsyms + 1 + 1, 0, 0, hints);
    hints[0] = (uint32_t)r.bitpos;
}


/* ---------------------------------------------------------------------------------------------
 * Whole lists, block by block, multi-threaded (test infrastructure for full-size parity).
 * ------------------------------------------------------------------------------------------- */
#include <pthread.h>

uint64_t ans_oracle_hash(const uint8_t* p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
    }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

typedef struct {
    int kind;
    uint32_t f;
    const uint32_t* in;
    size_t n, block_ints, ckpt, nblocks;
    uint32_t *sizes, max_lg, max_ns;
    uint64_t *stream_hash, *ckpt_digest;
    size_t next;
    int bad;
    pthread_mutex_t mu;
    /* hash_spans */
    const uint8_t* buf;
    const uint64_t* offs;
    uint64_t* out;
} blocks_job;

static void* blocks_worker(void* arg)
{
    blocks_job* J = (blocks_job*)arg;
    const size_t cap = ans_oracle_bound(J->kind, J->f, J->block_ints);
    const size_t nck_max = J->ckpt ? J->block_ints / J->ckpt + 1 : 1;
    uint8_t* out = (uint8_t*)malloc(cap);
    uint64_t* st = (uint64_t*)malloc(nck_max * 4 * sizeof(uint64_t));
    uint32_t* off = (uint32_t*)malloc(nck_max * sizeof(uint32_t));
    uint32_t my_lg = 0, my_ns = 0;
    int bad = (!out || !st || !off);
    for (;;) {
        pthread_mutex_lock(&J->mu);
        const size_t b = J->next;
        J->next += 16;
        pthread_mutex_unlock(&J->mu);
        if (b >= J->nblocks || bad) break;
        for (size_t bb = b; bb < b + 16 && bb < J->nblocks; bb++) {
            const size_t lo = bb * J->block_ints, cnt = J->n - lo < J->block_ints ? J->n - lo : J->block_ints;
            ans_oracle_info info;
            size_t nck = 0;
            const size_t nb = ans_oracle_encode(J->kind, J->f, J->in + lo, cnt, out, cap, &info, J->ckpt, st, off, &nck);
            if (!nb) {
                bad = 1;
                break;
            }
            J->sizes[bb] = (uint32_t)nb;
            J->stream_hash[bb] = ans_oracle_hash(out, nb);
            uint64_t dg = 0;
            for (size_t s = 0; s < nck; s++) {
                for (int j = 0; j < 4; j++) dg += st[4 * s + j] * (2654435761ull + 2ull * (4 * s + (size_t)j));
                dg += (uint64_t)off[s] * (40503ull + 2ull * s);
            }
            J->ckpt_digest[bb] = dg;
            if (info.log2_frame > my_lg) my_lg = info.log2_frame;
            if (info.max_sym + 1 > my_ns) my_ns = info.max_sym + 1;
        }
    }
    pthread_mutex_lock(&J->mu);
    if (bad) J->bad = 1;
    if (my_lg > J->max_lg) J->max_lg = my_lg;
    if (my_ns > J->max_ns) J->max_ns = my_ns;
    pthread_mutex_unlock(&J->mu);
    free(out);
    free(st);
    free(off);
    return NULL;
}

static void* spans_worker(void* arg)
{
    blocks_job* J = (blocks_job*)arg;
    for (;;) {
        pthread_mutex_lock(&J->mu);
        const size_t b = J->next;
        J->next += 64;
        pthread_mutex_unlock(&J->mu);
        if (b >= J->nblocks) break;
        for (size_t i = b; i < b + 64 && i < J->nblocks; i++)
            J->out[i] = ans_oracle_hash(J->buf + J->offs[i], (size_t)(J->offs[i + 1] - J->offs[i]));
    }
    return NULL;
}

static void run_threads(blocks_job* J, int threads, void* (*fn)(void*))
{
    pthread_t th[64];
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_mutex_init(&J->mu, NULL);
    int started = 0;
    for (int t = 0; t < threads; t++)
        if (pthread_create(&th[started], NULL, fn, J) == 0) started++;
    if (!started) fn(J);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&J->mu);
}

int ans_oracle_blocks_digest(int kind, uint32_t f, const uint32_t* in, size_t n, size_t block_ints, size_t ckpt_interval,
    int threads, uint32_t* sizes, uint64_t* stream_hash, uint64_t* ckpt_digest, uint32_t* max_log2_frame, uint32_t* max_nsyms)
{
    if (!n || !block_ints) return -1;
    blocks_job J;
    memset(&J, 0, sizeof(J));
    J.kind = kind, J.f = f, J.in = in, J.n = n, J.block_ints = block_ints;
    J.ckpt = ckpt_interval >= block_ints ? 0 : ckpt_interval;
    J.nblocks = (n + block_ints - 1) / block_ints;
    J.sizes = sizes, J.stream_hash = stream_hash, J.ckpt_digest = ckpt_digest;
    run_threads(&J, threads, blocks_worker);
    if (max_log2_frame) *max_log2_frame = J.max_lg;
    if (max_nsyms) *max_nsyms = J.max_ns;
    return J.bad ? -1 : 0;
}

void ans_oracle_hash_spans(const uint8_t* buf, const uint64_t* offs, size_t nspans, int threads, uint64_t* out)
{
    blocks_job J;
    memset(&J, 0, sizeof(J));
    J.buf = buf, J.offs = offs, J.out = out, J.nblocks = nspans;
    run_threads(&J, threads, spans_worker);
}

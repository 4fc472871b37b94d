// ansx — host side of the C-ABI (include/ansx.h): context, device workspace, launch sequences.
// Everything that computes runs in the HIP kernels of ansx_kernels.h / ansx_rfold.h; there is no
// CPU fallback: if no gfx950 device is usable, ansx_init fails with ANSX_ERR_NO_DEVICE.
#include "../../include/ansx.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is resolved at first use (ansx_gather_containers)

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "ansx_kernels.h"
#include "ansx_rfold.h"
#include "ansx_model.h"
#include "ansx_fastmodel.h"
#include "ansx_gen.h"
#include "ansx_pa.h"
#include "ansx_intsparse.h"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ProfRec {
    std::string name;
    hipEvent_t e0, e1;
};

struct Layout {  // container layout, a pure function of the geometry (restart-point format included)
    u64 index_off, ckoff_off, ckstate_off, hint_off, payload_off;
};

}  // namespace

struct ansx_ctx {
    u32 num_cus = 256;
    int device = 0;
    hipStream_t stream = nullptr;
    int last_hip = 0;
    bool profile = false;
    std::vector<ProfRec> recs;
    std::map<std::string, std::pair<double, u64>> acc;
    std::vector<std::string> order;
    DevBuf pre_work;  // f = 6, 7: off[] and bit buffer of the generic prelude writer
    DevBuf hist, hterm, sortF, sortSym, attS, prevS, attMeta, blk, table, tab32, scratch, misc, mapped, mostfreq,
        stage_in, stage_out, dec_s2s, dec_cum, dec_info, plain, rf_tmp, log2lut, pa_alpha, pa_info, pairs, lg2i, sizes, nearlist, force;
    u32* h_pin = nullptr;  // pinned: [0..3] gflags, [4..7] result (2 x u64), [8..] header scratch
    // Largest alphabet (max_sym + 1) seen per (kind, fidelity, block_ints): sizes the LDS of the fused
    // model kernel and of the LDS-table encoder without a mid-call round trip (see encode_dev).
    std::map<u64, u32> ns_hint;
    std::map<u64, u32> rf_hint;  // rfold: most distinct values per block seen per geometry (optimistic hash-table size)
    // header of the last container decoded per (kind, fidelity, n, bytes): the next decode of that shape is launched
    // on it without waiting for the header to come back, and a one-thread kernel compares it with the real one
    std::map<std::array<u64, 4>, ansx_container_header> hdr_cache;
    std::deque<std::array<u64, 4>> hdr_order;  // keys of hdr_cache, oldest first (an erased key may linger: erase is idempotent)
    std::map<u32, DevBuf> geo;   // tree nodes of the interpolative code per alphabet size, tabulated per symbol-array size (<= 4096)
    DevBuf geo_big;              // the same for alphabets up to geo_big_cap symbols (symbol arrays above 4096 slots, fast model path:
    u32 geo_big_cap = 0;         //   sized from the geometry's alphabet hint, cap^2 * 4 bytes -- 284 MB for fidelity 5 on 2^20-valued lists)
    int last_gather_ranks = 0;   // ranks of the communicator the last ansx_gather_containers call ran on (ncclCommCount)
    std::set<u64> wide_hint;     // geometries that met a frame above 2^16: wide restart points from the start
    std::map<u64, u32> t_hint;   // largest chosen candidate index t (frame M0 * 2^t) + 1 seen per geometry: lanes per block of k_candidates
    const u32* cur_force = nullptr;  // per-block frames decided by the host (resolve_near), device array, for the repeat of a call
    const u32* cur_src = nullptr;    // set by encode_general: the ints the model kernels saw (the input, or its remapped form)
    bool used_pc = false;        // set by launch_lds_encoder: the producer / consumer encoder kernel ran in the call
    bool used_fast = false;      // set by encode_general: the call's model came from k_candidates / k_model_finish
    u32 cur_nt = 0;              // set by encode_dev: candidates per block for the fast model path of this call (0 = exact path)
    u32 cur_rf_slots = 0;        // set by encode_dev for the optimistic attempt of the current call
    u32 cur_pa_distinct = 0;     // the same for the compaction layer's k_pa_remap (the hint itself)
    std::set<u64> int_sparse_hint;  // plain-ANSint geometries whose values outgrew the dense 16384-symbol model: rank space from the start
    bool cur_int_sparse = false;    // this call models its blocks in rank space (ansx_intsparse.h)
    u32 sp_retries = 0;             // (tests) calls repeated with the full-size arrays
    bool sp_full_lds = false;       // ... and its prelude writer runs with the full-size LDS arrays (a block's code outgrew the hint-sized ones)
    ansx_encode_stats last = {};
    // Path-selection overrides for tests and experiments (every path must give identical bytes).
    // Taken from the environment ONCE in ansx_init, changed afterwards only through ansx_debug_set;
    // the per-call hot path never looks at the environment.
    struct {
        bool table16_fixup = false;   // ANSX_TEST_TABLE16_FIXUP: integer-state encoder fed by k_table16_from32
        bool encode_gtab16 = false;   // ANSX_ENCODE_GTAB16: force the 16-byte-entry integer-state encoder
        bool parse_generic = false;   // ANSX_PARSE_GENERIC: generic prelude parser kernel
        bool parse_win = false;       // ANSX_PARSE_WIN: one lane per block, windowed parser (ignores the parse hints)
        bool parse_fast = false;      // ANSX_PARSE_FAST: one lane per block, E-array fast loop where it applies
        bool decode_table = false;    // ANSX_DECODE_TABLE: slot -> symbol decoder tables
        bool no_stream_lds = false;   // ANSX_NO_STREAM_LDS: staged decoder reads the stream from HBM
        int decode_mode = 0;          // ANSX_DECODE_MODE: 0 auto, 1 "ring", 2 "staged"
        u32 parse_stage_words = 0;    // ANSX_PARSE_STAGE_WORDS: 0 = default
        bool model_fused = false;     // ANSX_MODEL_FUSED: the single LDS-resident model kernel instead of the five tailored ones
        bool model_sync = false;      // ANSX_MODEL_SYNC: always discover the alphabet with the mid-call read-back
        bool use_pc = false;          // ANSX_USE_PC: k_encode_pc's chip-filling shape even under ANSX_NO_PC_AUTO
        u32 pc_b_pairs = 2;           // ANSX_PC_B_PAIRS: pairs per workgroup of shape B (2: one workgroup per CU -- 1.04 ms on BASELINE config 3;
                                      // 1: two workgroups per CU, whose waves the dispatcher does not spread as evenly -- 1.21 ms)
        u32 test_sp_bits = 0;         // ANSX_TEST_SP_BITS: words of the sparse ANSint prelude writer's bit buffer on the first attempt (tests: forces its repeat)
        bool fin_one_wave = false;    // ANSX_FIN_ONE_WAVE: k_model_finish with one wave per block (alphabets up to 1024 slots)
        bool no_big_geo = false;      // ANSX_NO_BIG_GEO: no tabulated tree geometry for alphabets above 4096 slots
        bool no_pc_auto = false;      // ANSX_NO_PC_AUTO: never choose the pair kernel by itself (shapes A, B, C of launch_f64_encoder)
        bool encode_mode2 = false;    // ANSX_ENCODE_MODE2: the compact-table encoder (k_encode<2>) even where the tables fit LDS (tests)
        bool force_pc = false;        // ANSX_FORCE_PC: the pair kernel for every workgroup of 64 full blocks, however few (tests)
        bool no_pc = false;           // ANSX_NO_PC: the LDS-table encoder as one wave per 16 blocks everywhere (k_encode<1>), no producer / consumer pairs
        int decode_small_ring = 0;    // ANSX_DECODE_SMALL_RING: "never" / "always" (default: by the container's bytes per int)
        int decode_pair = 0;          // ANSX_DECODE_PAIR: "0"/unset auto, "never", "always" (k_decode_rank2: two blocks per workgroup)
        u32 pair_lds_limit = 0;       // ANSX_DECODE_PAIR_LDS: auto uses the pair kernel up to this many bytes of LDS per workgroup (0: never --
                                      // measured SLOWER than one block per workgroup, 0.77-0.79 vs 0.72 ms on the headline workload)
        u32 wide_at = 16;             // ANSX_TEST_WIDE_AT: frames above 2^this need wide restart points (tests lower it to force the repeat)
        bool wide_restart = false;    // ANSX_WIDE_RESTART: 36-byte restart points (the v2 form) in every container
        u32 ns_hint = 0;              // ANSX_NS_HINT: alphabet hint for every call (0 = learn per geometry)
        u32 t_hint = 0;               // ANSX_T_HINT: candidates per block for every call (0 = learn per geometry)
        bool no_fast_model = false;   // ANSX_NO_FAST_MODEL: optimistic calls keep the exact model kernels
        double near_band = ANSX_NEAR_BAND;  // ANSX_NEAR_BAND: relative band around the stop-rule threshold inside which the host decides (tests widen it)
        bool near_flip = false;       // ANSX_TEST_NEAR_FLIP: the device decides close calls the wrong way (tests: the host must fix them)
        u32 cand_chains = 0;          // ANSX_CAND_CHAINS: 1 | 2 recurrences per lane in k_candidates (0 = by the call's size)
        double fast_guard = ANSX_FAST_GUARD;  // ANSX_FAST_GUARD: relative guard band of the fast model path's stop rule (tests widen it)
    } dbg;
};

namespace {

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t _e = (call);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            (ctx)->last_hip = (int)_e;                                                           \
            return ANSX_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

int ensure(ansx_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return ANSX_OK;
    if (b.p) {
        hipError_t e = hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
        if (e != hipSuccess) {
            c->last_hip = (int)e;
            return ANSX_ERR_HIP;
        }
    }
    size_t want = bytes + (bytes >> 3) + 4096;  // a little headroom against re-allocation
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        return ANSX_ERR_HIP;
    }
    b.cap = want;
    return ANSX_OK;
}

void prof_begin(ansx_ctx* c, const char* name, hipStream_t s)
{
    if (!c->profile) return;
    ProfRec r;
    r.name = name;
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, s);
    c->recs.push_back(r);
}
void prof_end(ansx_ctx* c, hipStream_t s)
{
    if (!c->profile) return;
    (void)hipEventRecord(c->recs.back().e1, s);
}

#define LAUNCH(ctx, name, kern, grid, block, shmem, strm, ...)                                   \
    do {                                                                                         \
        prof_begin(ctx, name, strm);                                                             \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (shmem), strm, __VA_ARGS__);           \
        prof_end(ctx, strm);                                                                     \
        HIPCHK(ctx, hipGetLastError());                                                          \
    } while (0)

inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

// worst-case bytes of one block's reference stream
size_t codec_nsp(int kind, u32 f) { return kind == ANSX_MSB ? 2048u : (kind == ANSX_INT ? 16384u : fold_NSP(f)); }

size_t block_bound(int kind, u32 f, size_t nb, bool pa = false)
{
    size_t hdr = kind == ANSX_RFOLD ? 4 + 4 * (size_t)fold_T(f) : 0;
    size_t nsp = codec_nsp(kind, f);
    if (pa) hdr += 8 + 4 * nb + 8;  // alphabet header: at most 32 bits per distinct value
    return hdr + 8 + 4 * nsp + 7 * nb + 32;
}

struct Plan {
    ansx_geo g;
    bool plain;
    Layout lay;
    u32 NSP;
};

Layout layout_of(const ansx_geo& g, bool plain)
{
    Layout L;
    const u64 nck = (u64)g.nblocks * g.nckf;
    L.index_off = sizeof(ansx_container_header);
    L.ckoff_off = L.index_off + 8 * ((u64)g.nblocks + 1);
    if (g.ckw) {
        L.ckstate_off = rup(L.ckoff_off + 4 * nck, 8);
        L.hint_off = rup(L.ckstate_off + 32 * nck, 16);  // 8 x u32 parse hints per block
    } else {
        L.ckstate_off = L.ckoff_off;  // (one array of records)
        L.hint_off = rup(L.ckoff_off + (u64)ANSX_CK_RECORD * nck, 16);
    }
    L.payload_off = L.hint_off + 32 * (u64)g.nblocks;
    if (plain) L.index_off = L.ckoff_off = L.ckstate_off = L.hint_off = L.payload_off = 0;
    return L;
}

int make_plan(int kind, int f, size_t n, const ansx_opts* opts, Plan* P)
{
    if (kind != ANSX_FOLD && kind != ANSX_RFOLD && kind != ANSX_MSB && kind != ANSX_INT) return ANSX_ERR_ARG;
    if (kind == ANSX_MSB || kind == ANSX_INT) {
        if (f != 0) return ANSX_ERR_ARG;  // ANSmsb / ANSint have no fidelity parameter (methods.hpp:484-515)
    } else if (f < 1 || f > ANSX_MAX_FIDELITY) return ANSX_ERR_ARG;  // see include/ansx.h
    if (n == 0) return ANSX_ERR_ARG;
    u32 bi = opts ? opts->block_ints : 0;
    u32 ck = opts ? opts->ckpt_interval : 0;
    const u32 flags = opts ? opts->flags : 0;
    if (flags & ~(u32)ANSX_FLAG_COMPACT_ALPHABET) return ANSX_ERR_ARG;
    const bool pa = (flags & ANSX_FLAG_COMPACT_ALPHABET) != 0;
    // ANSint models every value up to the largest (ans_int.hpp:40-48).  Without the compaction layer the values
    // themselves must fit the 16384-symbol model (ANSX_ERR_DOMAIN otherwise: checked on the device); with it the
    // codec runs on a block's dense ranks.  ANSrfold brings its own remap.
    if (kind == ANSX_RFOLD && pa) return ANSX_ERR_ARG;
    if (pa) {
        if (bi == ANSX_SINGLE_STREAM) return ANSX_ERR_ARG;
        // ranks are 1-based (pseudo_adaptive.cpp:91-103): ANSint's alphabet is sigma + 1 <= 16384 symbols
        const u32 lim = kind == ANSX_INT ? 16380u : ANSX_PA_MAX_BLOCK;
        if (bi == 0) bi = kind == ANSX_INT ? 8192u : ANSX_DEFAULT_BLOCK_INTS;
        if (bi > lim) return ANSX_ERR_ARG;
    }
    P->plain = (bi == ANSX_SINGLE_STREAM);
    if (bi == 0) bi = ANSX_DEFAULT_BLOCK_INTS;
    if (ck == 0) ck = ANSX_DEFAULT_CKPT_INTERVAL;
    if (ck == ANSX_NO_CHECKPOINTS) ck = 0;
    if (P->plain) {
        if (n >= ((size_t)1 << 31)) return ANSX_ERR_ARG;  // reference limit
        bi = (u32)n;
        ck = 0;
    } else {
        if (bi & 3u) return ANSX_ERR_ARG;
        if (bi >= (1u << 31)) return ANSX_ERR_ARG;
    }
    if (ck & 3u) return ANSX_ERR_ARG;
    if (ck >= bi) ck = 0;
    size_t nblocks = (n + bi - 1) / bi;
    if (nblocks > 0x7FFFFFFFull) return ANSX_ERR_ARG;
    ansx_geo g;
    g.n = n;
    g.block_ints = bi;
    g.nblocks = (u32)nblocks;
    g.ckpt = ck;
    g.nckf = geo_nseg(bi, ck) - 1;
    g.f = (u32)f;
    g.kind = (u32)kind;
    g.pa = pa ? 1u : 0u;
    g.ckw = 0;            // packed restart points unless set_restart_format() says otherwise
    g.payload_bytes = 0;  // (set by decode_dev from the container header)
    g.trusted_index = 0;  // (set by decode_dev on the single-stream path only, where the host writes the two entries)
    g.pad_ = 0;
    g.map = kind == ANSX_MSB ? map_msb() : (kind == ANSX_INT ? map_int() : map_fold((u32)f));
    P->g = g;
    // symbol-array stride: the reference's MAX_SIGMA (ans_fold.hpp:70; ans_msb.hpp:28 has 1280)
    P->NSP = (u32)codec_nsp(kind, (u32)f);
    P->lay = layout_of(g, P->plain);
    return ANSX_OK;
}

// Restart points: packed 29-byte records (container v3 default) or the wide form (ansx_dev.h).  Wide is needed when a
// state can exceed 52 bits (frames above 2^16: ANSint always may, the others only with large alphabets in large
// blocks) or a cursor 24 bits; the encoder finds out about frames on the device and repeats the call (encode_dev).
void set_restart_format(Plan* P, bool wide)
{
    P->g.ckw = wide ? 1u : 0u;
    P->lay = layout_of(P->g, P->plain);
}

// fold maps have power-of-two thresholds 2^(f+7), 2^(f+15), 2^(f+23): the encoder derives the exception-byte
// count from the bit length instead of three comparisons
bool map_is_pow2(const ansx_map& m)
{
    return m.t1 >= 2 && (m.t1 & (m.t1 - 1)) == 0 && m.t1 < (1u << 15) && m.t2 == m.t1 << 8 && m.t3 == m.t1 << 16;
}
int flags_to_status(u32 fl)
{
    if (fl & (1u << 6)) return ANSX_ERR_DOMAIN;
    if (fl & (1u << 7)) return ANSX_ERR_MODEL;
    if (fl & (1u << 2)) return ANSX_ERR_CAPACITY;
    if (fl & (1u << 3)) return ANSX_ERR_FORMAT;
    return ANSX_OK;
}

// --------------------------------------------------------------------------------- rfold
// ans_reorder_fold.hpp:70-106 on the device: LDS hash table per block for blocks <= 16384 ints,
// HBM hash table for longer blocks (incl. whole-list single-stream mode).
// Optimistic hash-table size for a geometry whose blocks had at most `distinct` different values so far: 1.5 x that,
// if two such tables (+ selection buffers) share a CU's LDS; 0 = use the full-size table.
u32 rf_opt_slots(u32 distinct, u32 T)
{
    if (distinct == 0) return 0;
    u32 slots = (2 * distinct + distinct / 2 + 64 + 255) & ~255u;
    if (slots < 1024) slots = 1024;
    const size_t lds = 6 * (size_t)slots + 8 * (size_t)(T < 512 ? 512 : T);
    return lds <= 78 * 1024 ? slots : 0u;
}
// opt_slots != 0: optimistic table size for the LDS form (see k_rfold_remap_hash), from rf_opt_slots()
int rfold_remap(ansx_ctx* c, const ansx_geo& g, const u32* d_in, u32* mapped, u32* mostfreq,
    ansx_blk* blk, u32* gflags, hipStream_t s, u32 opt_slots = 0)
{
    const u32 T = fold_T(g.f);
    if (g.block_ints > 16384u || T > 4096u) {  // (T > 4096: f = 6, 7 -- the selection buffer alone is 64 / 128 KB)
        // large blocks (incl. whole-list single-stream mode): hash table in HBM
        u32 slots = 2;
        while ((u64)slots < 2ull * g.block_ints && slots < (1u << 31)) slots <<= 1;
        int rc;
        if ((rc = ensure(c, c->rf_tmp, (size_t)g.nblocks * slots * 8 + (size_t)g.nblocks * 16))) return rc;
        u32* keys = (u32*)c->rf_tmp.p;
        u32* counts = keys + (size_t)g.nblocks * slots;
        u32* bstat = counts + (size_t)g.nblocks * slots;
        HIPCHK(c, hipMemsetAsync(keys, 0xFF, (size_t)g.nblocks * slots * 4, s));
        HIPCHK(c, hipMemsetAsync(counts, 0, (size_t)g.nblocks * slots * 4 + (size_t)g.nblocks * 16, s));
        const size_t grid = (g.n + 255) / 256;
        LAUNCH(c, "k_rfg_insert", k_rfg_insert, grid, 256, 0, s, d_in, g, slots, keys, counts, bstat);
        if ((size_t)T * 8 > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_rfg_select,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)T * 8)));
        LAUNCH(c, "k_rfg_select", k_rfg_select, g.nblocks, 256, (size_t)T * 8, s, g, slots, (const u32*)keys,
            counts, (const u32*)bstat, mostfreq, blk, gflags);
        LAUNCH(c, "k_rfg_map", k_rfg_map, grid, 256, 0, s, d_in, g, slots, (const u32*)keys, (const u32*)counts,
            (const ansx_blk*)blk, mapped);
        return ANSX_OK;
    }
    if (T <= 4096) {  // hash-table form
        const size_t sel_bytes = 8 * (size_t)(T < 512 ? 512 : T);  // also holds a 1024-bin histogram
        if (opt_slots != 0 && opt_slots < ANSX_RF_SLOTS) {
            const size_t lds = 6 * (size_t)opt_slots + sel_bytes;
            HIPCHK(c, hipFuncSetAttribute((const void*)k_rfold_remap_hash2,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef ANSX_STAMPS_RF
            { static unsigned long long z[3] = { 0, 0, 0 }; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, 24, 4101 * 8); }
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, s);
#endif
            LAUNCH(c, "k_rfold_remap", k_rfold_remap_hash2, g.nblocks, 1024, lds, s, d_in, g, opt_slots, mapped,
                mostfreq, blk, gflags);
#ifdef ANSX_STAMPS_RF
            {
                (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                unsigned long long z[3]; (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_stamps), 24, 4101 * 8);
                fprintf(stderr, "[stamps] k_rfold_remap_hash2: events %.3f ms; workgroup lifetimes: max %llu ticks, mean %.0f ticks, %llu above 40 us\n", ms, z[0], (double)z[1] / g.nblocks, z[2]);
            }
#endif
            return ANSX_OK;
        }
        const size_t lds = 6 * (size_t)ANSX_RF_SLOTS + sel_bytes;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_rfold_remap_hash,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_rfold_remap", k_rfold_remap_hash, g.nblocks, 1024, lds, s, d_in, g, (u32)ANSX_RF_SLOTS, mapped,
            mostfreq, blk, gflags);
        return ANSX_OK;
    }
    u32 N2 = 2;
    while (N2 < g.block_ints) N2 <<= 1;
    size_t lds = 6 * (size_t)N2 + 8 * (size_t)T + 16;
    if (lds > 48 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)k_rfold_remap,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LAUNCH(c, "k_rfold_remap_sort", k_rfold_remap, g.nblocks, 256, lds, s, d_in, g, N2, mapped, mostfreq,
        blk, gflags);
    return ANSX_OK;
}

// --------------------------------------------------------------------------------- encode
constexpr int ANSX_RETRY_GENERAL = -1;  // internal: an optimistic assumption did not hold, repeat without it
constexpr int ANSX_RETRY_WIDE = -2;     // internal: a frame above 2^16 in a call laid out for packed restart points, repeat with wide ones

// K5, f64-state forms (frames <= 2^16, every block's alphabet <= ns_entries).  Three kernels share the call's blocks:
//   k_encode_pc   producer / consumer wave pairs, 16 x pairs blocks per workgroup, where that form wins: (A) chip-filling calls
//                 whose tables fit 64 to a CU (four pairs, S = 8) -- BASELINE config 2; (B) alphabets too large for that whose
//                 tables fit at 32 (two pairs, S = 4: every entry in LDS, two rounds, each wave alone on its SIMD) -- BASELINE
//                 config 3; (C) short lists (one pair per workgroup, at most two per CU).  It takes ALL blocks of the call: the
//                 last workgroup pads itself with neutral steps (blocks that do not exist, the partial last block)
//   k_encode<1>   one wave per 16 blocks, 4-byte LDS entries: geometries the pair kernel does not take
//   k_encode<2>   compact tables in HBM with the hottest 1151 symbols per block in LDS: alphabets that fit neither
template <bool POW2, int S>
static int launch_pc(ansx_ctx* c, const ansx_geo& g, u32 NSP, const u32* src, u32 ns_entries, u32 rowwords, u32 pairs, u32 wgs,
    size_t lds, ansx_blk* blk, u64 scr_stride, u64* ck_state, u32* ck_off, u32* enc_sizes, unsigned long long* enc_gsums, hipStream_t s)
{
    HIPCHK(c, hipFuncSetAttribute((const void*)k_encode_pc<POW2, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LAUNCH(c, "k_encode", (k_encode_pc<POW2, S>), wgs, 128 * pairs, lds, s, src, g, NSP, (const u32*)c->tab32.p, ns_entries, rowwords, blk,
        (u8*)c->scratch.p, scr_stride, ck_state, ck_off, enc_sizes, enc_gsums);
    return ANSX_OK;
}
static int launch_f64_encoder(ansx_ctx* c, const ansx_geo& g, u32 NSP, const u32* src, u32 ns_entries, ansx_blk* blk, u64 scr_stride,
    u64* ck_state, u32* ck_off, u32* enc_sizes, unsigned long long* enc_gsums, u32 NB, hipStream_t s)
{
    c->used_pc = false;
    const u32 lds_stride = ns_entries | 1u;  // odd stride spreads the 16 tables over the banks
    const size_t enc_lds = (size_t)16 * lds_stride * 4;
    const bool mode1 = enc_lds <= 40 * 1024 && !c->dbg.encode_mode2;
    const bool pow2 = map_is_pow2(g.map);
    int rc;
    u32 first = 0;
    // ---- producer / consumer pairs
    const u32 rowwords = ((ns_entries + 2u) / 2u) | 1u;  // ns_entries + 1 running sums of 16 bits, an odd number of words per row
    const u32 full_blocks = (u32)(g.n / g.block_ints);
    const u32 enc_waves_all = (NB + 15) / 16;
    auto pc_lds = [&](u32 pairs, u32 S) { return (size_t)pairs * 16 * rowwords * 4 + (size_t)pairs * (2 * S * 1024 + (S == 8 ? 16 * 144 : 0)); };
    u32 pairs = 0, S = 0;
    // (block_ints <= 2^22: the stand-in for a neutral step of a 2^16 frame lets the state creep by 2^-16 per step, see the kernel)
    const bool pc_geo = !c->dbg.no_pc && g.block_ints % 128u == 0 && g.block_ints <= (1u << 22) && (u64)scr_stride * 16 < 0x40000000ull;
    if (pc_geo && (!c->dbg.no_pc_auto || c->dbg.use_pc || c->dbg.force_pc) && mode1 && pc_lds(4, 8) <= 160 * 1024
        && (((NB + 63) / 64) * 2 >= (u32)c->num_cus || c->dbg.force_pc))
        pairs = 4, S = 8;  // (A) the chip-filling form: 0.66 against k_encode<1>'s 0.71 ms on the headline workload since the producer
                           // loads its inputs 16 bytes at a time (equal before that)
    else if (pc_geo && !mode1 && pc_lds(2, 4) <= 160 * 1024 && (full_blocks >= 32 || c->dbg.force_pc) && !c->dbg.no_pc_auto)
        pairs = c->dbg.pc_b_pairs, S = 4;                                    // (B) every table entry in LDS, two rounds
    else if (pc_geo && mode1 && enc_waves_all <= 2u * (u32)c->num_cus && (full_blocks >= 16 || c->dbg.force_pc) && !c->dbg.no_pc_auto)
        pairs = 1, S = 8;  // (C) short lists: the call waits for one wave's state chain, and the consumer's is 20 % shorter.  (Lists of
                           // fewer than 16 full blocks stay with k_encode<1>, which walks a short block's own steps only.)
    if (pairs && (g.ckpt == 0 || g.ckpt % (4u * S) == 0)) {
        const u32 wgs = (NB + 16 * pairs - 1) / (16 * pairs);
        const size_t lds = pc_lds(pairs, S);
        if (pow2) rc = S == 8 ? launch_pc<true, 8>(c, g, NSP, src, ns_entries, rowwords, pairs, wgs, lds, blk, scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, s)
                              : launch_pc<true, 4>(c, g, NSP, src, ns_entries, rowwords, pairs, wgs, lds, blk, scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, s);
        else rc = S == 8 ? launch_pc<false, 8>(c, g, NSP, src, ns_entries, rowwords, pairs, wgs, lds, blk, scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, s)
                         : launch_pc<false, 4>(c, g, NSP, src, ns_entries, rowwords, pairs, wgs, lds, blk, scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, s);
        if (rc) return rc;
        first = wgs * 16 * pairs;
        c->used_pc = true;
        if (first >= NB) return ANSX_OK;
    }
    const u32 enc_waves = (NB - first + 15) / 16;
    u32 wpw = (enc_waves + c->num_cus - 1) / c->num_cus;
    wpw = wpw < 1 ? 1 : (wpw > 4 ? 4 : wpw);
    const size_t enc_grid = (enc_waves + wpw - 1) / wpw;
    if (mode1) {
        // ---- one wave per 16 blocks.  Waves of one workgroup run the main loop in step (a barrier per super-batch): up to four
        // waves per workgroup -- one per SIMD of a CU -- as soon as there are that many waves per CU (see k_encode)
        if (wpw * enc_lds > 48 * 1024) {
            HIPCHK(c, hipFuncSetAttribute((const void*)k_encode<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wpw * enc_lds)));
            HIPCHK(c, hipFuncSetAttribute((const void*)k_encode<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wpw * enc_lds)));
        }
        if (pow2)
            LAUNCH(c, first ? "k_encode_rest" : "k_encode", (k_encode<1, true>), enc_grid, 64 * wpw, wpw * enc_lds, s, src, g, NSP,
                (const ansx_enc_entry*)nullptr, (const u32*)c->tab32.p, lds_stride, blk, (u8*)c->scratch.p,
                (u64)scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, first);
        else
            LAUNCH(c, first ? "k_encode_rest" : "k_encode", (k_encode<1, false>), enc_grid, 64 * wpw, wpw * enc_lds, s, src, g, NSP,
                (const ansx_enc_entry*)nullptr, (const u32*)c->tab32.p, lds_stride, blk, (u8*)c->scratch.p,
                (u64)scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, first);
        return ANSX_OK;
    }
    // ---- alphabets too large for LDS: compact table entries from HBM, same branch-free f64 step
    const size_t lds2 = (size_t)wpw * 16 * ANSX_ENC_HOT * 4;
    if (lds2 > 48 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)k_encode<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    LAUNCH(c, first ? "k_encode_rest" : "k_encode_gtab", (k_encode<2>), enc_grid, 64 * wpw, lds2, s, src, g, NSP,
        (const ansx_enc_entry*)nullptr, (const u32*)c->tab32.p, (u32)ANSX_ENC_HOT, blk, (u8*)c->scratch.p,
        (u64)scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, first);
    return ANSX_OK;
}

// ns_cap == 0: discovery mode -- the largest alphabet / frame of the call are read back between the
// model kernels and the encoder launch (one host round trip per candidate batch).
// ns_cap != 0: optimistic mode -- the caller has seen this geometry before (alphabet hint): the first
// candidate batch is assumed to settle every block, frames are assumed to stay within 2^16 and alphabets
// within ns_cap, so everything is launched back to back; the assumptions are checked on the words that
// come back with the output size anyway, and a miss returns ANSX_RETRY_GENERAL.
int encode_general(ansx_ctx* c, const Plan& P, const u32* d_in, u8* d_out, size_t cap,
    size_t* out_bytes, hipStream_t s, u32* seen_ns, u32 ns_cap)
{
    const bool optimistic = ns_cap != 0;
    const ansx_geo& g = P.g;
    const u32 NB = g.nblocks, NSP = P.NSP, f = g.f;
    const size_t scr_stride = rup(block_bound(g.kind, f, g.block_ints, g.pa != 0) + 16, 256);
    if (!P.plain && cap < P.lay.payload_off) return ANSX_ERR_CAPACITY;
    int rc;
    if ((rc = ensure(c, c->hist, (size_t)NB * NSP * 4))) return rc;
    if ((rc = ensure(c, c->sortF, (size_t)NB * NSP * 4))) return rc;
    if ((rc = ensure(c, c->sortSym, (size_t)NB * NSP * 2))) return rc;
    const size_t fbytes = g.kind == ANSX_INT ? 4 : 2;  // candidate frequencies: u32 for ANSint (ans_int.hpp:30-34), u16 otherwise
    if ((rc = ensure(c, c->attS, (size_t)NB * ANSX_ATTEMPTS * NSP * fbytes))) return rc;
    if ((rc = ensure(c, c->prevS, (size_t)NB * NSP * fbytes))) return rc;
    if ((rc = ensure(c, c->attMeta, (size_t)NB * ANSX_ATTEMPTS * 16))) return rc;
    if (!c->log2lut.p) {  // stage-1 table of the portable log2, once per context (1.5 MB)
        if ((rc = ensure(c, c->log2lut, (size_t)65536 * sizeof(ansx_log2_ent)))) return rc;
        LAUNCH(c, "k_build_log2_lut", k_build_log2_lut, 256, 256, 0, s, (ansx_log2_ent*)c->log2lut.p);
    }
    const uint2* geo = nullptr;
    if (NSP <= 4096) {  // tree nodes of the prelude's interpolative code for every alphabet size up to NSP, once per context
        DevBuf& gb = c->geo[NSP];
        if (!gb.p) {
            if ((rc = ensure(c, gb, ((size_t)NSP * (NSP + 1) / 2 + 8) * 8))) return rc;
            LAUNCH(c, "k_build_interp_geo", k_build_interp_geo, NSP, 256, 0, s, NSP, (uint2*)gb.p);
        }
        geo = (const uint2*)gb.p;
    }
    if ((rc = ensure(c, c->blk, (size_t)NB * sizeof(ansx_blk)))) return rc;
    if ((rc = ensure(c, c->table, (size_t)NB * NSP * sizeof(ansx_enc_entry)))) return rc;
    if ((rc = ensure(c, c->tab32, (size_t)NB * NSP * 4))) return rc;
    if ((rc = ensure(c, c->scratch, (size_t)NB * scr_stride))) return rc;
    if ((rc = ensure(c, c->misc, 64 + 8 * ((size_t)NB + 1)))) return rc;
    // per-block stream sizes + their sums per 64 blocks, published by the encoder for k_assemble
    const size_t ngroups = (((size_t)NB + 63) / 64 + 1) & ~(size_t)1;  // (an even count: the sums end on a 16-byte boundary)
    if ((rc = ensure(c, c->sizes, ngroups * 8 + (size_t)NB * 4))) return rc;
    unsigned long long* enc_gsums = (unsigned long long*)c->sizes.p;
    u32* enc_sizes = (u32*)((u8*)c->sizes.p + ngroups * 8);
    u32* gflags = (u32*)c->misc.p;
    u64* result = (u64*)((u8*)c->misc.p + 16);
    u64* boff_ws = (u64*)((u8*)c->misc.p + 64);
    ansx_blk* blk = (ansx_blk*)c->blk.p;
    u32* hist = (u32*)c->hist.p;

    {
        // flag words, size sums, block metadata, and the container's header / index / restart-point area: unused
        // slots (short last block) and alignment padding are defined to be zero, so equal inputs give
        // byte-identical containers
        static_assert(sizeof(ansx_blk) % 16 == 0, "zeroed 16 bytes at a time");
        ansx_zero4 Z;
        Z.p[0] = (uint4*)c->misc.p, Z.n16[0] = 4;
        Z.p[1] = (uint4*)enc_gsums, Z.n16[1] = ngroups / 2;
        Z.p[2] = (uint4*)blk, Z.n16[2] = (u64)NB * (sizeof(ansx_blk) / 16);
        Z.p[3] = (uint4*)d_out, Z.n16[3] = P.plain ? 0 : (u64)P.lay.payload_off / 16;
        const u64 tot = Z.n16[0] + Z.n16[1] + Z.n16[2] + Z.n16[3];
        LAUNCH(c, "k_begin_encode", k_begin_encode, (u32)std::min<u64>(2048, (tot + 255) / 256), 256, 0, s, Z);
    }

    if ((rc = ensure(c, c->nearlist, (size_t)ANSX_NEAR_CAP * 4))) return rc;
    const u32* src = d_in;
    const u32* mostfreq = nullptr;
    if (g.kind == ANSX_RFOLD) {
        const u32 T = fold_T(f);
        if ((rc = ensure(c, c->mapped, (size_t)g.n * 4))) return rc;
        if ((rc = ensure(c, c->mostfreq, (size_t)NB * T * 4))) return rc;
        rc = rfold_remap(c, P.g, d_in, (u32*)c->mapped.p, (u32*)c->mostfreq.p, blk, gflags, s, optimistic ? c->cur_rf_slots : 0u);
        if (rc) return rc;
        src = (const u32*)c->mapped.p;
        mostfreq = (const u32*)c->mostfreq.p;
    }

    if (g.pa) {
        // per-block alphabet compaction (src/pseudo_adaptive.cpp:85-130): alphabet header into the block's
        // scratch slot, the codec then runs on the 1-based ranks
        if ((rc = ensure(c, c->mapped, (size_t)NB * g.block_ints * 4))) return rc;
        if ((rc = ensure(c, c->pa_alpha, (size_t)NB * g.block_ints * 4))) return rc;
        // sizes from the geometry's distinct-value hint (optimistic calls only): hash set 2.5 x, value list the next
        // power of two above 1.25 x; both workgroups of a CU must fit its LDS
        u32 pa_slots = ANSX_PA_SLOTS, pa_uqcap = ANSX_PA_MAX_BLOCK;
        bool pa_small = false;
        if (optimistic && c->cur_pa_distinct != 0) {
            const u32 d = c->cur_pa_distinct;
            const u32 sl = (2 * d + d / 2 + 64 + 255) & ~255u;
            u32 uc = 1024;
            while (uc < d + d / 4 + 16) uc <<= 1;
            if (((size_t)sl + uc) * 4 <= 78 * 1024 && uc <= ANSX_PA_MAX_BLOCK) pa_slots = sl, pa_uqcap = uc, pa_small = true;
        }
        const size_t lds1 = ((size_t)pa_slots + pa_uqcap) * 4;
        if (pa_small) {
            HIPCHK(c, hipFuncSetAttribute((const void*)k_pa_remap2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
            LAUNCH(c, "k_pa_remap", k_pa_remap2, NB, 1024, lds1, s, d_in, g, pa_slots, pa_uqcap, (u32*)c->mapped.p,
                (u32*)c->pa_alpha.p, blk, gflags, 1u << 30);
        } else {
            HIPCHK(c, hipFuncSetAttribute((const void*)k_pa_remap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
            LAUNCH(c, "k_pa_remap", k_pa_remap, NB, 1024, lds1, s, d_in, g, pa_slots, pa_uqcap, (u32*)c->mapped.p,
                (u32*)c->pa_alpha.p, blk, gflags, 1u << 30, 0u);
        }
        const size_t lds2 = pa_small ? ((size_t)3 * pa_uqcap + 32) * 4 : ((size_t)2 * ANSX_PA_MAX_BLOCK + 16) * 4;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_pa_header, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        LAUNCH(c, "k_pa_header", k_pa_header, NB, 256, lds2, s, g, (const u32*)c->pa_alpha.p, blk, (u8*)c->scratch.p,
            (u64)scr_stride, pa_small ? pa_uqcap : (u32)ANSX_PA_MAX_BLOCK);
        src = (const u32*)c->mapped.p;
    }
    const bool sparse = c->cur_int_sparse;
    if (sparse) {
        // plain ANSint on values beyond the dense model (ansx_intsparse.h): the codec runs on every block's 0-based ranks
        if ((rc = ensure(c, c->mapped, (size_t)NB * g.block_ints * 4))) return rc;
        if ((rc = ensure(c, c->pa_alpha, (size_t)NB * g.block_ints * 4))) return rc;
        const size_t lds1 = ((size_t)ANSX_PA_SLOTS + ANSX_PA_MAX_BLOCK) * 4;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_pa_remap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        LAUNCH(c, "k_pa_remap", k_pa_remap, NB, 1024, lds1, s, d_in, g, (u32)ANSX_PA_SLOTS, (u32)ANSX_PA_MAX_BLOCK, (u32*)c->mapped.p,
            (u32*)c->pa_alpha.p, blk, gflags, (u32)ANSX_SP_VALUE_LIMIT, 1u);
        src = (const u32*)c->mapped.p;
    }
    c->cur_src = src;
    // K1
    u32 chunk = g.block_ints < 16384u ? g.block_ints : 16384u;
    if (chunk & 3u) chunk = (chunk + 3u) & ~3u;
    const u32 cpb = (g.block_ints + chunk - 1) / chunk;
    if (cpb > 1) HIPCHK(c, hipMemsetAsync(hist, 0, (size_t)NB * NSP * 4, s));
    // blocks that fit one histogram workgroup (the normal case) get their entropy terms from K1
    // and the in-order sum from K2b; longer blocks keep both in K2a
    const bool h_deferred = (cpb == 1);
    // entropy terms are evaluated by the histogram kernel when it sees whole blocks; alphabets up
    // to 2048 slots are also summed there (terms in LDS), larger ones through HBM in K2b
    const bool h_in_hist = h_deferred && NSP <= 2048;
    // Fast model path (ansx_fastmodel.h): geometries seen before, whole-block histograms, 16-bit frequencies,
    // compact tables; every assumption is checked on the device and a miss repeats the call on the exact path.
    const u32 NT = optimistic ? c->cur_nt : 0u;
    // (alphabets above 4096 slots -- f = 4, 5 -- take the generic form of k_model_finish as long as its three LDS arrays,
    // sized from the alphabet hint, fit a CU; ANSint has no u16 rule and 32-bit frequencies: exact path)
    const u32 fcap_probe = std::min<u32>(NSP, std::max<u32>(64u, (ns_cap + 15u) & ~15u));
    const bool fast = NT != 0 && !g.pa && h_deferred && g.block_ints <= 65535u && NSP <= 16384 && g.kind != ANSX_INT
        && (NSP <= 4096 || (size_t)fcap_probe * 8 + 64 <= 150 * 1024) && !c->dbg.table16_fixup
        && !c->dbg.encode_gtab16 && (u64)scr_stride * 16 < 0x7FFFFF00ull;
    c->used_fast = fast;
    if (fast && NSP > 4096 && !c->dbg.no_big_geo) {
        // k_model_finish<0>: a 14-level tree descent per item was two thirds of its prelude writer; the nodes of every alphabet
        // size up to the hint come from a table here too (built once per context, rebuilt when the hint grows)
        if (c->geo_big_cap < fcap_probe) {
            if ((rc = ensure(c, c->geo_big, ((size_t)fcap_probe * (fcap_probe + 1) / 2 + 8) * 8))) return rc;
            LAUNCH(c, "k_build_interp_geo", k_build_interp_geo, fcap_probe, 256, 0, s, fcap_probe, (uint2*)c->geo_big.p);
            c->geo_big_cap = fcap_probe;
        }
        geo = (const uint2*)c->geo_big.p;
    }
    if (fast) {
        if ((rc = ensure(c, c->pairs, (size_t)NB * NSP * 8))) return rc;
        if (!c->lg2i.p) {  // log2 of the integers below 2^16, once per context (512 KB)
            if ((rc = ensure(c, c->lg2i, (size_t)65536 * 8))) return rc;
            LAUNCH(c, "k_build_log2i_lut", k_build_log2i_lut, 256, 256, 0, s, (double*)c->lg2i.p);
        }
    }
    double* hterm = nullptr;
    if (h_deferred && !h_in_hist && !fast) {
        if ((rc = ensure(c, c->hterm, (size_t)NB * NSP * 8))) return rc;
        hterm = (double*)c->hterm.p;
    }
    // (fast path: H is a tree sum in registers, no LDS row of terms)
    const size_t hist_lds = !h_in_hist ? (size_t)NSP * 4 : (size_t)4 * (NSP + ANSX_HCOPY_PAD) * 4 + (fast ? 0 : (size_t)NSP * 8 + 80);
    if (NSP >= 8192u) {  // f >= 4: 16-bit counters, two per LDS word (a chunk holds at most 16384 values): half the LDS, twice the workgroups per CU
        const size_t packed_lds = (size_t)NSP * 2;
        if (packed_lds > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_fold_hist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)packed_lds));
        // (sum_mode bit 1: on the fast model path H is the workgroup's tree sum and there is no hterm array to write)
        LAUNCH(c, "k_fold_hist", k_fold_hist<true>, (size_t)NB * cpb, 256, packed_lds, s, src, g, chunk, cpb, NSP, hist, hterm,
            fast ? 2u : 0u, blk, gflags, (g.kind == ANSX_INT && !g.pa) ? NSP : (1u << 30));
    } else {
        if (hist_lds > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_fold_hist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds));
        LAUNCH(c, "k_fold_hist", k_fold_hist<false>, (size_t)NB * cpb, 256, hist_lds, s, src, g, chunk, cpb, NSP, hist, hterm,
            (h_in_hist ? 1u : 0u) | (fast ? 2u : 0u), blk, gflags, (g.kind == ANSX_INT && !g.pa) ? NSP : (1u << 30));
    }
    // K2.  "big" symbols have freq >= ANSX_VMAX, so a block holds at most block_ints/ANSX_VMAX
    const u32 nbig_cap = (u32)std::min<size_t>(NSP, (size_t)g.block_ints / ANSX_VMAX + 2);
    // (optimistic calls with whole-block histograms: the staged row is as long as the alphabet hint, see the kernel)
    const u32 sort_cap = (optimistic && h_deferred) ? std::min<u32>(NSP, std::max<u32>(64u, (ns_cap + 7u) & ~7u)) : NSP;
    const bool sort16 = g.block_ints <= 65535u;  // (a count fits 16 bits: half the staged row)
    size_t k2a_lds = (size_t)nbig_cap * 8 + (size_t)sort_cap * (sort16 ? 2 : 4) + (h_deferred ? 0 : 512 * 8);
    const bool sort_staged = k2a_lds <= 150 * 1024;  // (f = 6, 7 with 32-bit counts: the row stays in HBM)
    if (!sort_staged) k2a_lds = (size_t)nbig_cap * 8 + (h_deferred ? 0 : 512 * 8);
    if (k2a_lds > 32 * 1024) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sort_entropy<u16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2a_lds));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_sort_entropy<u32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2a_lds));
    }
    if ((size_t)NSP * 8 + 64 > 48 * 1024 && (size_t)NSP * 8 + 64 <= 150 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)k_write_prelude<0>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)NSP * 8 + 64)));
    if (!sort_staged)
        LAUNCH(c, "k_sort_entropy", (k_sort_entropy<u32, false>), NB, 64, k2a_lds, s, g, NSP, nbig_cap, h_deferred ? 1u : 0u, hist,
            (u32*)c->sortF.p, (u16*)c->sortSym.p, blk, sort_cap, fast ? (uint2*)c->pairs.p : (uint2*)nullptr, gflags);
    else if (sort16)
        LAUNCH(c, "k_sort_entropy", k_sort_entropy<u16>, NB, 64, k2a_lds, s, g, NSP, nbig_cap, h_deferred ? 1u : 0u, hist,
            (u32*)c->sortF.p, (u16*)c->sortSym.p, blk, sort_cap, fast ? (uint2*)c->pairs.p : (uint2*)nullptr, gflags);
    else
        LAUNCH(c, "k_sort_entropy", k_sort_entropy<u32>, NB, 64, k2a_lds, s, g, NSP, nbig_cap, h_deferred ? 1u : 0u, hist,
            (u32*)c->sortF.p, (u16*)c->sortSym.p, blk, sort_cap, fast ? (uint2*)c->pairs.p : (uint2*)nullptr, gflags);
    // Frame sizes M0*2^t are tried ANSX_ATTEMPTS at a time.  Almost every block settles in the
    // first batch; the count of undecided blocks comes back with the words the encoder launch
    // needs anyway (largest alphabet / frame), so further batches are launched only on demand.
    const u32 nbatch = 24 / ANSX_ATTEMPTS;  // t < 24 (t <= 16 suffices)
    u32 max_logM = 0, max_ns = 0;
    // consumers of the 16-byte table entries that are certain before the frames are known: the
    // generic prelude writer (alphabets above 4096 slots) and the integer-state encoder (forced, or
    // scratch slots too far apart for the f64 encoder's 31-bit buffer offsets)
    const bool test_fixup = c->dbg.table16_fixup;  // tests: integer-state encoder fed by k_table16_from32
    const u32 always16 = (!test_fixup && (NSP > 4096 || (u64)scr_stride * 16 >= 0x7FFFFF00ull || c->dbg.encode_gtab16)) ? 1u : 0u;
    // (plain ANSint: no parse hints -- its decoder walks the value-range prelude sparsely, and the container must not depend on
    // which of the two models, dense or rank space, wrote it)
    u32* hints = (P.plain || (g.kind == ANSX_INT && !g.pa)) ? nullptr : (u32*)(d_out + P.lay.hint_off);
    if (fast) {
        const u32 bpw = 64u / NT;
        // chains per lane: one while that leaves at most one wave per SIMD, else two (see k_candidates)
        const u32 nch = c->dbg.cand_chains ? c->dbg.cand_chains : (((NB + bpw - 1) / bpw <= 4u * c->num_cus) ? 1u : 2u);
        const size_t cl = (size_t)ANSX_CAND_WAVES * nch * bpw * ANSX_CAND_ROW * 16;
        const u32 cwaves = (NB + nch * bpw - 1) / (nch * bpw);
#define ANSX_LAUNCH_CAND2(NT_, NCH_)                                                                                    \
    do {                                                                                                            \
        if (cl > 48 * 1024)                                                                                         \
            HIPCHK(c, hipFuncSetAttribute((const void*)k_candidates<NT_, NCH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cl)); \
        LAUNCH(c, "k_candidates", (k_candidates<NT_, NCH_>), (cwaves + ANSX_CAND_WAVES - 1) / ANSX_CAND_WAVES, 64 * ANSX_CAND_WAVES, cl, s, g, NSP, \
            (const uint2*)c->pairs.p, (const ansx_blk*)blk, (uint4*)c->attS.p, (u32*)c->attMeta.p);               \
    } while (0)
#define ANSX_LAUNCH_CAND(NT_)                                                                                           \
    do {                                                                                                            \
        if (nch == 1) ANSX_LAUNCH_CAND2(NT_, 1);                                                                    \
        else ANSX_LAUNCH_CAND2(NT_, 2);                                                                             \
    } while (0)
        switch (NT) {
        case 4: ANSX_LAUNCH_CAND(4); break;
        case 5: ANSX_LAUNCH_CAND(5); break;
        case 6: ANSX_LAUNCH_CAND(6); break;
        case 7: ANSX_LAUNCH_CAND(7); break;
        default: ANSX_LAUNCH_CAND(8); break;
        }
#undef ANSX_LAUNCH_CAND
#undef ANSX_LAUNCH_CAND2
        const u32 fcap = std::min<u32>(NSP, std::max<u32>(64u, (ns_cap + 15u) & ~15u));
        const bool fin_generic = NSP > 4096;  // (loop-based form: inc[] in the block's histogram row, two LDS arrays)
        const size_t fl = (size_t)fcap * (fin_generic ? 8 : 12) + 64;
#define ANSX_LAUNCH_FIN(IPT_, NTC_)                                                                                    \
    do {                                                                                                            \
        if (fl > 48 * 1024)                                                                                         \
            HIPCHK(c, hipFuncSetAttribute((const void*)k_model_finish<IPT_, NTC_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl)); \
        LAUNCH(c, "k_model_finish", (k_model_finish<IPT_, NTC_>), NB, 256, fl, s, g, NSP, NT, (const uint2*)c->pairs.p, (const uint4*)c->attS.p, \
            (const u32*)c->attMeta.p, blk, (u32*)c->tab32.p, (u8*)c->scratch.p, (u64)scr_stride, mostfreq, hints, gflags, fcap, c->dbg.fast_guard, (const double*)c->lg2i.p, geo, \
            fin_generic ? hist : (u32*)nullptr); \
    } while (0)
        if (NSP <= 1024 && c->dbg.fin_one_wave) {
            // one wave per block (16 slots per lane, every candidate in every lane, no workgroup barriers): four times the blocks in flight
            if (NT <= 5) {
                LAUNCH(c, "k_model_finish", (k_model_finish<16, 5, 64>), NB, 64, fl, s, g, NSP, NT, (const uint2*)c->pairs.p, (const uint4*)c->attS.p,
                    (const u32*)c->attMeta.p, blk, (u32*)c->tab32.p, (u8*)c->scratch.p, (u64)scr_stride, mostfreq, hints, gflags, fcap, c->dbg.fast_guard, (const double*)c->lg2i.p, geo,
                    (u32*)nullptr);
            } else {
                LAUNCH(c, "k_model_finish", (k_model_finish<16, 8, 64>), NB, 64, fl, s, g, NSP, NT, (const uint2*)c->pairs.p, (const uint4*)c->attS.p,
                    (const u32*)c->attMeta.p, blk, (u32*)c->tab32.p, (u8*)c->scratch.p, (u64)scr_stride, mostfreq, hints, gflags, fcap, c->dbg.fast_guard, (const double*)c->lg2i.p, geo,
                    (u32*)nullptr);
            }
        } else if (NSP <= 1024) {
            ANSX_LAUNCH_FIN(4, 8);  // (wave-per-candidate form: NTC is not used)
        } else if (NSP > 4096) {
            if (NT <= 5) ANSX_LAUNCH_FIN(0, 5);
            else ANSX_LAUNCH_FIN(0, 8);
        } else {
            if (NT <= 5) ANSX_LAUNCH_FIN(16, 5);
            else ANSX_LAUNCH_FIN(16, 8);
        }
#undef ANSX_LAUNCH_FIN
        max_logM = 16;
        max_ns = ns_cap;
    }
    for (u32 batch = 0; batch < (fast ? 0u : nbatch); batch++) {
        if (batch) HIPCHK(c, hipMemsetAsync(&gflags[ANSX_G_PAD], 0, 4, s));
        LAUNCH(c, "k_scale_attempts", k_scale_attempts, ((size_t)NB * ANSX_ATTEMPTS + 255) / 256, 256,
            0, s, g, NSP, batch, hist, (const u32*)c->sortF.p, (const u16*)c->sortSym.p, blk,
            (u16*)c->attS.p, (u32*)c->attMeta.p, (const double*)hterm, (const ansx_log2_ent*)c->log2lut.p,
            g.block_ints <= 65535u ? 1u : 0u);
        LAUNCH(c, "k_select_model", k_select_model, NB, 64, 0, s, g, NSP, batch, hist,
            (const u16*)c->attS.p, (const u32*)c->attMeta.p, (u16*)c->prevS.p, blk,
            (ansx_enc_entry*)c->table.p, (u32*)c->tab32.p, gflags, batch == nbatch - 1 ? 1u : 0u, always16,
            (u32*)c->nearlist.p, c->cur_force, c->dbg.near_band, c->dbg.near_flip ? 1u : 0u);
        if (optimistic) {  // checked after the fact (blocks left undecided carry no model and are skipped)
            max_logM = 16;
            max_ns = ns_cap;
            break;
        }
        HIPCHK(c, hipMemcpyAsync(c->h_pin, c->misc.p, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        int st0 = flags_to_status(c->h_pin[ANSX_G_ERR]);
        if (st0) return st0;
        max_logM = c->h_pin[ANSX_G_MAXLOGM];
        max_ns = c->h_pin[ANSX_G_MAXNSYMS];
        if (c->h_pin[ANSX_G_PAD] == 0) break;
    }
    if (!P.plain && !g.ckw && g.nckf != 0 && max_logM > c->dbg.wide_at) return ANSX_RETRY_WIDE;  // (discovery path: known before anything is encoded)
    if (!always16 && (max_logM > 16 || test_fixup))  // mixed call: a frame above 2^16 sends every block to the integer-state encoder
        LAUNCH(c, "k_table16_from32", k_table16_from32, NB, 256, 0, s, g, NSP, (const ansx_blk*)blk,
            (const u32*)c->tab32.p, (ansx_enc_entry*)c->table.p);
    // K3 (also fills the container's parse hints)
    // LDS arrays of the prelude writer: as long as the call's largest alphabet (known here on the discovery path,
    // assumed = the hint on the optimistic one), not as its slot count
    u32 pre_cap = optimistic ? ns_cap : max_ns;
    pre_cap = std::min<u32>(NSP, std::max<u32>(64u, (pre_cap + 7u) & ~7u));
    if (fast) {
        // (k_model_finish wrote the preludes)
    } else if (NSP <= 1024 && max_logM <= 16) {
        LAUNCH(c, "k_write_prelude", (k_write_prelude<4>), NB, 256, (size_t)pre_cap * 12 + 64, s, g, NSP,
            (const ansx_enc_entry*)c->table.p, (const u32*)c->tab32.p, hist, blk, (u8*)c->scratch.p,
            (u64)scr_stride, mostfreq, hints, pre_cap, geo);
    } else if (NSP <= 4096 && max_logM <= 16) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_write_prelude<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)((size_t)pre_cap * 12 + 64)));
        LAUNCH(c, "k_write_prelude", (k_write_prelude<16>), NB, 256, (size_t)pre_cap * 12 + 64, s, g, NSP,
            (const ansx_enc_entry*)c->table.p, (const u32*)c->tab32.p, hist, blk, (u8*)c->scratch.p,
            (u64)scr_stride, mostfreq, hints, pre_cap, geo);
    } else if (sparse) {
        // the reference's prelude over the VALUE range, from the rank-space model (16-byte entries: always16) and the block's values
        // LDS from the call's most distinct values per block (max_ns: read back above, the discovery path) unless a block's code
        // outgrew three words per value on the first attempt
        const u32 sp_cap = c->sp_full_lds ? (u32)ANSX_SP_MAX_SIGMA : std::min<u32>(ANSX_SP_MAX_SIGMA, (std::max<u32>(max_ns, 64u) + 63u) & ~63u);
        const u32 sp_bits = c->sp_full_lds ? (u32)ANSX_SP_MAX_SIGMA
                                           : (c->dbg.test_sp_bits ? c->dbg.test_sp_bits : std::min<u32>(ANSX_SP_MAX_SIGMA, 2u * sp_cap + 64u));
        const size_t sp_lds = (size_t)(sp_cap + sp_bits + 2) * 4;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_int_sparse_prelude, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_lds));
        LAUNCH(c, "k_int_sparse_prelude", k_int_sparse_prelude, NB, 256, sp_lds, s, g, NSP, (const u32*)c->pa_alpha.p,
            (const ansx_enc_entry*)c->table.p, blk, (u8*)c->scratch.p, (u64)scr_stride, sp_cap, sp_bits, (u32)(4 * NSP), gflags);
    } else {
        const size_t gen_lds = (size_t)pre_cap * 8 + 64;  // (as long as the call's largest alphabet, not as its slot count: workgroups per CU)
        if (gen_lds > 150 * 1024) {  // f = 6, 7: the writer's two arrays in HBM
            if ((rc = ensure(c, c->pre_work, (size_t)NB * (2 * (size_t)pre_cap + 16) * 4))) return rc;
            LAUNCH(c, "k_write_prelude", (k_write_prelude<0>), NB, 256, 64, s, g, NSP,
                (const ansx_enc_entry*)c->table.p, (const u32*)c->tab32.p, hist, blk, (u8*)c->scratch.p,
                (u64)scr_stride, mostfreq, hints, pre_cap, (const uint2*)nullptr, (u32*)c->pre_work.p);
        } else
        LAUNCH(c, "k_write_prelude", (k_write_prelude<0>), NB, 256, gen_lds, s, g, NSP,
            (const ansx_enc_entry*)c->table.p, (const u32*)c->tab32.p, hist, blk, (u8*)c->scratch.p,
            (u64)scr_stride, mostfreq, hints, pre_cap, (const uint2*)nullptr, (u32*)nullptr);
    }
    // K5.  The encoder keeps its 16 per-wave tables in LDS when they fit (sized from the largest
    // alphabet / frame actually produced, read back above).
    u64* ck_state = P.plain ? nullptr : (u64*)(d_out + P.lay.ckstate_off);
    u32* ck_off = P.plain ? nullptr : (u32*)(d_out + P.lay.ckoff_off);
    // (its emitted-byte stores go through a buffer descriptor spanning the wave's 16 scratch slots:
    // 31-bit offsets)
    const bool f64_ok = max_logM <= 16 && (u64)scr_stride * 16 < 0x7FFFFF00ull && !test_fixup;
    if (f64_ok && !c->dbg.encode_gtab16) {
        if ((rc = launch_f64_encoder(c, g, NSP, src, max_ns, blk, (u64)scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, NB, s))) return rc;
    } else {
        LAUNCH(c, "k_encode_gtab", (k_encode<0>), ((size_t)NB * 4 + 63) / 64, 64, 0, s, src, g, NSP,
            (const ansx_enc_entry*)c->table.p, (const u32*)c->tab32.p, 0u, blk, (u8*)c->scratch.p,
            (u64)scr_stride, ck_state, ck_off, enc_sizes, enc_gsums);
    }
    // K6
    u64* boff = P.plain ? boff_ws : (u64*)(d_out + P.lay.index_off);
    if (NB <= 65536u) {
        LAUNCH(c, "k_assemble", k_assemble, NB, 256, 0, s, g, (const u32*)enc_sizes, (const unsigned long long*)enc_gsums, boff, result,
            (const u8*)c->scratch.p, (u64)scr_stride, d_out, (u64)P.lay.payload_off, (u64)cap, gflags, P.plain ? 0u : 1u);
    } else {
        LAUNCH(c, "k_scan_sizes", k_scan_sizes, 1, 1024, 0, s, g, blk, boff, result, P.lay.payload_off,
            (u64)cap, gflags);
        LAUNCH(c, "k_compact", k_compact, NB, 256, 0, s, g, blk, boff, (const u8*)c->scratch.p,
            (u64)scr_stride, d_out + P.lay.payload_off, gflags);
        if (!P.plain)
            LAUNCH(c, "k_write_header", k_write_header, 1, 64, 0, s, g, d_out, gflags, result, P.lay.payload_off);
    }
    HIPCHK(c, hipMemcpyAsync(c->h_pin, c->misc.p, 64, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (optimistic) {
        if (c->h_pin[ANSX_G_ERR] & (1u << 6)) return ANSX_ERR_DOMAIN;
        if (c->h_pin[ANSX_G_ERR] & (1u << ANSX_G_VIOL_BIT)) return ANSX_RETRY_GENERAL;  // (rfold: optimistic hash table too small)
        if (c->h_pin[ANSX_G_PAD] != 0 || c->h_pin[ANSX_G_MAXLOGM] > 16 || c->h_pin[ANSX_G_MAXNSYMS] > ns_cap)
            return ANSX_RETRY_GENERAL;
        if (!g.ckw && g.nckf != 0 && c->h_pin[ANSX_G_MAXLOGM] > c->dbg.wide_at) return ANSX_RETRY_GENERAL;  // (tests only: wide_at < 16)
    }
    int st = flags_to_status(c->h_pin[ANSX_G_ERR]);
    if (st) return st;
    if (sparse && !c->sp_full_lds && (c->h_pin[ANSX_G_ERR] & (1u << ANSX_G_VIOL_BIT))) {
        c->sp_full_lds = true;  // (the value-range prelude of some block needs more than two words per distinct value)
        c->sp_retries++;
        const int rc2 = encode_general(c, P, d_in, d_out, cap, out_bytes, s, seen_ns, ns_cap);
        c->sp_full_lds = false;
        return rc2;
    }
    *seen_ns = c->h_pin[ANSX_G_MAXNSYMS];
    u64 payload;
    memcpy(&payload, (u8*)c->h_pin + 16, 8);
    *out_bytes = (size_t)(P.lay.payload_off + payload);
    return ANSX_OK;
}

// LDS carve of k_model_fused for `cap` symbols (a multiple of 8), see ansx_model.h
ansx_model_lds model_layout(u32 cap)
{
    ansx_model_lds L;
    L.cap = cap;
    L.nc = cap <= 1024 ? 4u : (cap <= 2560 ? 2u : 1u);
    L.natt = cap <= 1024 ? 8u : 4u;  // candidate rows are the largest region: fewer per batch for big alphabets
    const u32 A = L.nc * (cap + 8) * 4;
    u32 off;
    if (L.nc == 4) {  // pairs | yF (12 cap bytes) live in histogram copies 1..3, dead once copy 0 holds the sums
        L.off_pairs = (cap + 8) * 4;
        off = A;
    } else {
        L.off_pairs = A;
        off = A + 12 * cap;
    }
    L.off_yF = L.off_pairs + 4 * cap;
    L.off_S = off;
    const u32 sbytes = L.natt * cap * 2;  // >= 8 cap: the entropy terms come first
    if (sbytes >= 8 * cap + ANSX_MODEL_SORT_BYTES) {
        L.off_E = off + 8 * cap;  // sort scratch behind the entropy terms
        off += sbytes;
    } else {
        L.off_E = off + sbytes;
        off += sbytes + (u32)rup(ANSX_MODEL_SORT_BYTES, 16);
    }
    L.off_pos = off;
    off += (u32)rup(2 * cap, 16);
    L.off_ffs = off;
    off += (u32)rup(4 * (cap + 4), 16);
    L.off_X = off;
    off += 256 * 8;
    if (4 * cap >= 256 * 8) L.off_X1 = L.off_pairs;  // pairs are dead once yF / ffs exist
    else {
        L.off_X1 = off;
        off += 256 * 8;
    }
    L.total = off;
    return L;
}

// Optimistic encode, fused form: ONE model kernel per block with everything in LDS (sized from the context's
// alphabet hint), no host round trip before the final size read-back.  Any block that does not fit the
// assumptions (alphabet above the hint, frame above 2^16) raises the violation flag and the caller
// repeats the call on the general path.
int encode_fast(ansx_ctx* c, const Plan& P, const u32* d_in, u8* d_out, size_t cap, size_t* out_bytes,
    hipStream_t s, u32 ns_cap, u32* seen_ns)
{
    const ansx_geo& g = P.g;
    const u32 NB = g.nblocks, NSP = P.NSP, f = g.f;
    const size_t scr_stride = rup(block_bound(g.kind, f, g.block_ints) + 16, 256);
    if (cap < P.lay.payload_off) return ANSX_ERR_CAPACITY;
    if ((u64)scr_stride * 16 >= 0x7FFFFF00ull) return ANSX_RETRY_GENERAL;
    const ansx_model_lds ML = model_layout(ns_cap);
    if (ML.total > 150 * 1024) return ANSX_RETRY_GENERAL;
    int rc;
    if (!c->log2lut.p) {  // stage-1 table of the portable log2, once per context (1.5 MB)
        if ((rc = ensure(c, c->log2lut, (size_t)65536 * sizeof(ansx_log2_ent)))) return rc;
        LAUNCH(c, "k_build_log2_lut", k_build_log2_lut, 256, 256, 0, s, (ansx_log2_ent*)c->log2lut.p);
    }
    if ((rc = ensure(c, c->blk, (size_t)NB * sizeof(ansx_blk)))) return rc;
    if ((rc = ensure(c, c->tab32, (size_t)NB * NSP * 4))) return rc;
    if ((rc = ensure(c, c->scratch, (size_t)NB * scr_stride))) return rc;
    if ((rc = ensure(c, c->misc, 64 + 8 * ((size_t)NB + 1)))) return rc;
    const size_t ngroups = (((size_t)NB + 63) / 64 + 1) & ~(size_t)1;
    if ((rc = ensure(c, c->sizes, ngroups * 8 + (size_t)NB * 4))) return rc;
    unsigned long long* enc_gsums = (unsigned long long*)c->sizes.p;
    u32* enc_sizes = (u32*)((u8*)c->sizes.p + ngroups * 8);
    HIPCHK(c, hipMemsetAsync(enc_gsums, 0, ngroups * 8, s));
    u32* gflags = (u32*)c->misc.p;
    u64* result = (u64*)((u8*)c->misc.p + 16);
    ansx_blk* blk = (ansx_blk*)c->blk.p;
    HIPCHK(c, hipMemsetAsync(c->misc.p, 0, 64, s));
    HIPCHK(c, hipMemsetAsync(blk, 0, (size_t)NB * sizeof(ansx_blk), s));
    HIPCHK(c, hipMemsetAsync(d_out, 0, (size_t)P.lay.payload_off, s));
    const u32* src = d_in;
    const u32* mostfreq = nullptr;
    if (g.kind == ANSX_RFOLD) {
        const u32 T = fold_T(f);
        if ((rc = ensure(c, c->mapped, (size_t)g.n * 4))) return rc;
        if ((rc = ensure(c, c->mostfreq, (size_t)NB * T * 4))) return rc;
        rc = rfold_remap(c, P.g, d_in, (u32*)c->mapped.p, (u32*)c->mostfreq.p, blk, gflags, s, c->cur_rf_slots);
        if (rc) return rc;
        src = (const u32*)c->mapped.p;
        mostfreq = (const u32*)c->mostfreq.p;
    }
    if (ns_cap <= 1024) {
        if (ML.total > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_model_fused<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML.total));
        LAUNCH(c, "k_model_fused", (k_model_fused<4>), NB, 256, ML.total, s, src, g, NSP, ML,
            (const ansx_log2_ent*)c->log2lut.p, blk, (u32*)c->tab32.p, (u8*)c->scratch.p, (u64)scr_stride, mostfreq,
            gflags, 1u << 30, (u32*)(d_out + P.lay.hint_off));
    } else {
        if (ML.total > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_model_fused<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML.total));
        LAUNCH(c, "k_model_fused", (k_model_fused<16>), NB, 256, ML.total, s, src, g, NSP, ML,
            (const ansx_log2_ent*)c->log2lut.p, blk, (u32*)c->tab32.p, (u8*)c->scratch.p, (u64)scr_stride, mostfreq,
            gflags, 1u << 30, (u32*)(d_out + P.lay.hint_off));
    }
    u64* ck_state = (u64*)(d_out + P.lay.ckstate_off);
    u32* ck_off = (u32*)(d_out + P.lay.ckoff_off);
    {
        int rcl;
        if ((rcl = launch_f64_encoder(c, g, NSP, src, ns_cap, blk, (u64)scr_stride, ck_state, ck_off, enc_sizes, enc_gsums, NB, s))) return rcl;
    }
    u64* boff = (u64*)(d_out + P.lay.index_off);
    if (NB <= 65536u) {
        LAUNCH(c, "k_assemble", k_assemble, NB, 256, 0, s, g, (const u32*)enc_sizes, (const unsigned long long*)enc_gsums, boff, result,
            (const u8*)c->scratch.p, (u64)scr_stride, d_out, (u64)P.lay.payload_off, (u64)cap, gflags, 1u);
    } else {
        LAUNCH(c, "k_scan_sizes", k_scan_sizes, 1, 1024, 0, s, g, blk, boff, result, P.lay.payload_off, (u64)cap, gflags);
        LAUNCH(c, "k_compact", k_compact, NB, 256, 0, s, g, blk, boff, (const u8*)c->scratch.p, (u64)scr_stride,
            d_out + P.lay.payload_off, gflags);
        LAUNCH(c, "k_write_header", k_write_header, 1, 64, 0, s, g, d_out, gflags, result, P.lay.payload_off);
    }
    HIPCHK(c, hipMemcpyAsync(c->h_pin, c->misc.p, 64, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const u32 fl = c->h_pin[ANSX_G_ERR];
    if (fl & (1u << 6)) return ANSX_ERR_DOMAIN;
    if (fl & (1u << ANSX_G_VIOL_BIT)) return ANSX_RETRY_GENERAL;
    int st = flags_to_status(fl);
    if (st) return st;
    *seen_ns = c->h_pin[ANSX_G_MAXNSYMS];
    u64 payload;
    memcpy(&payload, (u8*)c->h_pin + 16, 8);
    *out_bytes = (size_t)(P.lay.payload_off + payload);
    return ANSX_OK;
}

// ---------------------------------------------------------------------------------------------
// Close calls of the frame-size stop rule, decided again on the host.  The reference compares
// XH = -sum p log2(S/M) with 1.001 H (ans_util.hpp:127-128,149) using libm's log2; the device uses its own
// portable log2 (<= 1 ulp apart), so a comparison whose two sides agree to ~1e-12 is the one place where the
// two could part.  k_select_model lists the blocks with such a comparison; for each of them the host reads the
// block back, rebuilds its histogram and runs adjust_freqs as written (ans_util.hpp:100-157, util.hpp:271-298;
// this TU is compiled with -ffp-contract=off, log2 is the host libm's -- the function the reference itself
// would call here).  If any decision differs from the device's, the call is repeated on the exact path with
// the host's frames forced for those blocks.
// Returns log2 of the frame size, or -1 for the reference's degenerate exit.
int host_adjust_freqs(const std::vector<u32>& freqs, u32 largest_sym, bool require_u16)
{
    const u32 ns = largest_sym + 1;
    u64 n = 0;
    u32 sigma = 0;
    for (u32 i = 0; i < ns; i++) {
        n += freqs[i];
        sigma += freqs[i] != 0;
    }
    if (sigma == 0) return -1;
    u32 lg = (sigma & (sigma - 1)) == 0 ? 31u - (u32)__builtin_clz(sigma) : 32u - (u32)__builtin_clz(sigma);  // :109-112
    std::vector<u32> order;
    order.reserve(sigma);
    for (u32 i = 0; i < ns; i++)
        if (freqs[i]) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return freqs[a] < freqs[b]; });  // (freq, sym) ascending, :114-122
    double H;
    {  // util.hpp:271-282
        double acc = 0.0;
        for (u32 i = 0; i < ns; i++)
            if (freqs[i]) {
                const double p = (double)freqs[i] / (double)n;
                acc = acc + p * std::log2(p);
            }
        H = -acc;
    }
    const double thr = H * (1.0 + (double)1 / (double)1000);  // :124,127-128
    std::vector<u32> scaled(ns, 0u);
    int prev = -1;
    for (; lg <= 31; lg++) {
        i64 Mr = (i64)1 << lg;
        u64 fr = n;
        for (u32 j = 0; j < sigma; j++) {  // :77-95
            const u32 sym = order[j];
            const double a = (double)Mr / (double)fr;
            double v = a * (double)freqs[sym];
            v = 0.5 + v;
            u32 S = (u32)v;
            if (S == 0) S = 1;
            scaled[sym] = S;
            Mr -= S;
            fr -= freqs[sym];
            if (Mr < 0) break;
        }
        if (Mr != 0) continue;  // :131-135
        u32 maxS = 0;
        for (u32 i = 0; i < ns; i++)
            if (freqs[i] && scaled[i] > maxS) maxS = scaled[i];
        if (require_u16 && maxS >= ANSX_U16_LIMIT) return prev;  // :141-145
        double XH;
        {  // util.hpp:284-298 (note the int accumulators there)
            double acc = 0.0;
            const double nd = (double)(int)n, md = (double)(int)((i64)1 << lg);
            for (u32 i = 0; i < ns; i++)
                if (freqs[i] && scaled[i]) {
                    const double p = (double)freqs[i] / nd;
                    const double q = (double)scaled[i] / md;
                    acc = acc + p * std::log2(q);
                }
            XH = -acc;
        }
        if (XH < thr) return (int)lg;  // :149
        prev = (int)lg;
    }
    return -1;
}

int encode_general(ansx_ctx* c, const Plan& P, const u32* d_in, u8* d_out, size_t cap, size_t* out_bytes, hipStream_t s,
    u32* seen_ns, u32 ns_cap);

int resolve_near(ansx_ctx* c, const Plan& P, const u32* d_in, u8* d_out, size_t cap, size_t* out_bytes, hipStream_t s,
    u32* seen_ns, u32* redecided)
{
    const ansx_geo& g = P.g;
    const u32 nn = c->h_pin[ANSX_G_NEAR];
    std::vector<u32> list;
    if (nn <= ANSX_NEAR_CAP) {
        list.resize(nn);
        HIPCHK(c, hipMemcpyAsync(list.data(), c->nearlist.p, (size_t)nn * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    } else {  // more close calls than the device lists: every block is looked at
        list.resize(g.nblocks);
        for (u32 b = 0; b < g.nblocks; b++) list[b] = b;
    }
    std::vector<u32> hin, hist;
    std::vector<u32> force;
    u32 nre = 0;
    for (const u32 b : list) {
        if (b >= g.nblocks) continue;
        const u32 nb = geo_block_n(g, b);
        ansx_blk hb;
        hin.resize(nb);
        HIPCHK(c, hipMemcpyAsync(hin.data(), c->cur_src + (u64)b * g.block_ints, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&hb, (const ansx_blk*)c->blk.p + b, sizeof(hb), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (hb.pa_sigma == 1) continue;  // (compaction: a one-value block has no model)
        hist.assign(P.NSP, 0u);
        u32 largest = 0;
        for (u32 i = 0; i < nb; i++) {
            const u32 x = hin[i];
            const u32 k = map_nbytes(g.map, x);
            const u32 sym = map_sym(g.map, x, k);
            if (sym >= P.NSP) return ANSX_ERR_DOMAIN;
            hist[sym]++;
            largest = sym > largest ? sym : largest;
        }
        const int lg = host_adjust_freqs(hist, largest, g.kind != ANSX_INT);
        const int dev = hb.status ? -1 : (int)hb.logM;
        if (lg != dev) {
            if (force.empty()) force.assign(g.nblocks, 0u);
            force[b] = lg < 0 ? 0xFFFFFFFFu : (u32)lg;
            nre++;
        }
    }
    *redecided = nre;
    if (nre == 0) return ANSX_OK;
    int rc;
    if ((rc = ensure(c, c->force, (size_t)g.nblocks * 4))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->force.p, force.data(), (size_t)g.nblocks * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));  // (force is a local)
    c->cur_force = (const u32*)c->force.p;
    c->cur_rf_slots = 0;
    c->cur_pa_distinct = 0;
    rc = encode_general(c, P, d_in, d_out, cap, out_bytes, s, seen_ns, 0);
    c->cur_force = nullptr;
    return rc;
}

int encode_dev_once(ansx_ctx* c, const Plan& P, const u32* d_in, u8* d_out, size_t cap, size_t* out_bytes,
    hipStream_t s)
{
    // The first call of a geometry discovers its alphabet size with a mid-call read-back; later calls
    // are launched back to back on that hint and repeat (rarely) if the input outgrew it.
    const u64 key = ((u64)P.g.pa << 48) | ((u64)P.g.kind << 40) | ((u64)P.g.f << 32) | P.g.block_ints;
    u32 seen = 0;
    int rc = ANSX_RETRY_GENERAL;
    const bool int_plain = P.g.kind == ANSX_INT && !P.g.pa;
    c->cur_int_sparse = int_plain && c->int_sparse_hint.count(key) != 0;
    const auto it = c->ns_hint.find(key);
    const u32 hint = c->dbg.ns_hint ? c->dbg.ns_hint : (it != c->ns_hint.end() ? it->second : 0u);
    const bool eligible = !P.plain && hint != 0 && (P.NSP <= 4096 || (P.NSP <= 16384 && P.g.kind != ANSX_INT && !P.g.pa)) && !c->dbg.encode_gtab16 && !c->dbg.table16_fixup
        && !c->dbg.model_sync;  // (with compaction too: the hint then describes the alphabets of the rank-remapped blocks)
    const auto rit = c->rf_hint.find(key);
    c->cur_rf_slots = (eligible && P.g.kind == ANSX_RFOLD && rit != c->rf_hint.end()) ? rf_opt_slots(rit->second, fold_T(P.g.f)) : 0u;
    c->cur_pa_distinct = (eligible && P.g.pa && rit != c->rf_hint.end()) ? rit->second : 0u;
    // candidates per block for the fast model path: one more than the largest index chosen so far, 4..8 lanes
    // (fewer than 4 would leave the wave's lanes to more blocks than its LDS rows); above 8 the exact path stays
    const auto tit = c->t_hint.find(key);
    const u32 tcount = c->dbg.t_hint ? c->dbg.t_hint : (tit != c->t_hint.end() ? tit->second : 0u);
    c->cur_nt = (eligible && !c->dbg.no_fast_model && tcount != 0 && tcount <= 8) ? std::max<u32>(4u, tcount) : 0u;
    if (eligible) {
        u32 ns_cap = (hint + 7u) & ~7u;
        if (ns_cap < 64) ns_cap = 64;
        if (ns_cap > P.NSP) ns_cap = P.NSP;
        // (k_model_fused: the LDS-resident single-kernel model, measured slower than the five tailored
        // kernels -- DESIGN.md section 6 -- and therefore opt-in)
        if (c->dbg.model_fused && P.g.block_ints <= ANSX_MODEL_MAX_BLOCK && !P.g.pa)
            rc = encode_fast(c, P, d_in, d_out, cap, out_bytes, s, ns_cap, &seen);
        else
            rc = encode_general(c, P, d_in, d_out, cap, out_bytes, s, &seen, ns_cap);
    }
    bool missed = false;
    u32 path = !eligible ? 0u : (c->dbg.model_fused && P.g.block_ints <= ANSX_MODEL_MAX_BLOCK && !P.g.pa ? 2u : (c->used_fast ? 5u : 1u));
    if (rc == ANSX_RETRY_GENERAL) {
        missed = eligible;
        c->cur_rf_slots = 0;
        c->cur_pa_distinct = 0;
        path = eligible ? path | 16u : 0u;
        rc = encode_general(c, P, d_in, d_out, cap, out_bytes, s, &seen, 0);
    }
    // Plain ANSint: the dense model takes values below 16384; a call with larger ones is repeated in rank space, and so is
    // every later call of the geometry from the start -- unless its values turn out small, where the dense form (whose
    // containers carry parse hints) is the one a fresh context would have written: equal inputs, equal bytes.
    if (int_plain && !c->cur_int_sparse && rc == ANSX_ERR_DOMAIN) {
        c->cur_int_sparse = true;
        c->int_sparse_hint.insert(key);
        rc = encode_general(c, P, d_in, d_out, cap, out_bytes, s, &seen, 0);
    } else if (int_plain && c->cur_int_sparse && rc == ANSX_OK && c->h_pin[ANSX_G_VMAX] < P.NSP) {
        c->cur_int_sparse = false;
        c->int_sparse_hint.erase(key);
        rc = encode_general(c, P, d_in, d_out, cap, out_bytes, s, &seen, 0);
    }
    if (c->cur_int_sparse) path |= 256u;  // plain ANSint modelled in rank space
    if (c->sp_retries) path |= 512u, c->sp_retries = 0;  // ... and repeated with the full-size prelude writer
    if (rc == ANSX_RETRY_WIDE) return rc;
    // close calls of the stop rule (counted by the exact kernels only; the fast path repeats on them): the host decides
    u32 redecided = 0;
    const u32 near_blocks = c->h_pin[ANSX_G_NEAR];  // (a forced repeat skips the rule for the blocks it forces)
    if (rc == ANSX_OK && near_blocks != 0) rc = resolve_near(c, P, d_in, d_out, cap, out_bytes, s, &seen, &redecided);
    if (rc == ANSX_RETRY_WIDE) return rc;
    c->last.host_redecided = redecided;
    c->last.path = path | (c->used_pc ? 128u : 0u);
    c->last.max_nsyms = c->h_pin[ANSX_G_MAXNSYMS];
    c->last.max_log2_frame = c->h_pin[ANSX_G_MAXLOGM];
    c->last.near_threshold_decisions = near_blocks;
    if (rc == ANSX_OK && !P.plain) {
        // a miss raises the hint past what was seen, so inputs whose alphabets creep upwards do not
        // miss on every call
        const u32 want = missed ? seen + seen / 8 + 8 : seen;
        u32& h = c->ns_hint[key];
        if (want > h) h = want;
        u32& th = c->t_hint[key];
        const u32 wantt = c->h_pin[ANSX_G_MAXT] + 1u + (missed ? 1u : 0u);
        if (wantt > th) th = wantt;
        if (P.g.kind == ANSX_RFOLD || P.g.pa) {
            const u32 d = c->h_pin[ANSX_G_RFDIST];
            u32& r = c->rf_hint[key];
            const u32 wantd = missed ? d + d / 8 + 8 : d;
            if (wantd > r) r = wantd;
        }
    }
    return rc;
}

// Restart-point format of the call (see set_restart_format): packed unless the geometry rules it out or an earlier
// call of the geometry met a frame above 2^16; a call that meets one is repeated once with the wide form.
int encode_dev(ansx_ctx* c, const Plan& P0, const u32* d_in, u8* d_out, size_t cap, size_t* out_bytes, hipStream_t s)
{
    Plan P = P0;
    if (!P.plain) {
        const u64 key = ((u64)P.g.pa << 48) | ((u64)P.g.kind << 40) | ((u64)P.g.f << 32) | P.g.block_ints;
        const size_t stream_bound = block_bound(P.g.kind, P.g.f, P.g.block_ints, P.g.pa != 0) + 16;
        const bool must = P.g.kind == ANSX_INT || stream_bound >= ((size_t)1 << ANSX_CK_CURSOR_BITS) || c->dbg.wide_restart;
        // The hint only picks which attempt runs FIRST; the format that is returned is a function of the input and the
        // options alone: wide if and only if this call's frames need it.
        const bool hinted = !must && c->wide_hint.count(key) != 0;
        set_restart_format(&P, must || hinted);
        int rc = encode_dev_once(c, P, d_in, d_out, cap, out_bytes, s);
        if (rc == ANSX_OK && hinted && c->last.max_log2_frame <= c->dbg.wide_at) {
            // an earlier call of this geometry needed wide restart points, this input does not: encode it again packed, so
            // that equal inputs give equal containers whatever the context encoded before (ranks of a multi-GPU job must
            // agree on the format, ansx_merge_containers_dev), and forget the hint
            c->wide_hint.erase(key);
            set_restart_format(&P, false);
            const u32 path0 = c->last.path;
            rc = encode_dev_once(c, P, d_in, d_out, cap, out_bytes, s);
            c->last.path |= (path0 & 16u) | 64u;  // 64: repeated with packed restart points
        }
        if (rc != ANSX_RETRY_WIDE) return rc;
        c->wide_hint.insert(key);
        set_restart_format(&P, true);
        const int rc2 = encode_dev_once(c, P, d_in, d_out, cap, out_bytes, s);
        c->last.path |= 32u;  // repeated with wide restart points
        return rc2;
    }
    return encode_dev_once(c, P, d_in, d_out, cap, out_bytes, s);
}

// --------------------------------------------------------------------------------- decode
template <bool RF>
int launch_decode(ansx_ctx* c, const ansx_geo& g, u32 NSP, const u8* cont, const u64* boff,
    const u64* ck_state, const u32* ck_off, u64 payload_off, u32* d_out, u32 maxM, u32 max_ns,
    u32 max_block_bytes, u64 cont_bytes, u32* gflags, hipStream_t s, const uint4* pa_info, const u32* hints, u32 max_ep,
    bool parsed = false)
{
    // max_ep: the header's bound on the symbols PRESENT in a block (<= max_ns, its bound on their indices): the
    // rank / select decoder keeps one 8-byte entry per present symbol, so this -- not max_ns -- sizes its LDS
    const u32 T = fold_T(g.f);
    int rc;
    if ((rc = ensure(c, c->dec_cum, (size_t)g.nblocks * (NSP + 8) * 4))) return rc;
    if ((rc = ensure(c, c->dec_info, (size_t)g.nblocks * 16))) return rc;
    // K7.  Containers carry parse hints (bit offsets of the top subtrees of every block's interpolative code):
    // eight lanes per block, k_parse_prelude_par.  Without hints (single-stream mode) or on request one lane
    // per block: the windowed parser (any alphabet / frame size; ANSX_PARSE_WIN), the older E-array fast loop
    // (ANSX_PARSE_FAST: 16-bit values, up to ~880 symbols) or the generic kernel (ANSX_PARSE_GENERIC) --
    // all four are cross-checked in the tests.
    const size_t pf_e = std::max<size_t>(20480, rup(((size_t)max_ns + 2) * 128, 16));
    const size_t pf_lds = pf_e + (size_t)ANSX_PF_SW * 64 * 4 + 21 * 64 * 4;
    u32 stage_words = ANSX_PF_SW;
    if (c->dbg.parse_stage_words >= 2 && c->dbg.parse_stage_words <= ANSX_PF_SW)  // tests: force the fast loop's in-kernel fallback
        stage_words = c->dbg.parse_stage_words & ~1u;
    if (parsed) {
        // (plain ANSint: k_int_sparse_parse has filled dec_cum / dec_info in rank space)
    } else if (c->dbg.parse_generic) {
        LAUNCH(c, "k_parse_prelude", (k_parse_prelude<RF>), (g.nblocks + 63) / 64, 64, 0, s, cont, g, NSP,
            boff, payload_off, max_ns, maxM, (u32*)c->dec_cum.p, (uint4*)c->dec_info.p, gflags, pa_info);
    } else if (hints != nullptr && !c->dbg.parse_win && !c->dbg.parse_fast) {
        // Waves per workgroup (each wave works alone on its own LDS slice): measured on MI355X at 16384 blocks,
        // 530-symbol tables 1/2/3/4 waves -> 0.161/0.162/0.109/0.122 ms, 2300-symbol tables 0.275/0.273/0.286/0.252.
        const u32 par_waves = (g.nblocks + 7) / 8;
        const u32 pw_max = max_ns <= 1024 ? 3u : 4u;
        const u32 pw = std::min<u32>(pw_max, std::max<u32>(1u, (par_waves + c->num_cus - 1) / c->num_cus));
        const u32 par_grid = (par_waves + pw - 1) / pw;
        const size_t lds = (size_t)pw * (32 + 48) * 64 * 4;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_parse_prelude_par<RF, 32>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_parse_prelude", (k_parse_prelude_par<RF, 32>), par_grid, 64 * pw, lds, s, cont, g, NSP, boff,
            payload_off, max_ns, maxM, hints, (u32*)c->dec_cum.p, (uint4*)c->dec_info.p, gflags, pa_info);
    } else if (c->dbg.parse_fast && (u64)maxM + max_ns + 3 <= 65535u && pf_lds <= 150 * 1024) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_parse_prelude_fast<RF>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_lds));
        LAUNCH(c, "k_parse_prelude", (k_parse_prelude_fast<RF>), (g.nblocks + 63) / 64, 64, pf_lds, s, cont, g,
            NSP, boff, payload_off, max_ns, maxM, (u32*)c->dec_cum.p, (uint4*)c->dec_info.p, gflags, stage_words, pa_info);
    } else if (max_ns <= 1024) {  // preludes of a few hundred bytes: 128 staged words per lane
        const size_t lds = (size_t)(128 + 72) * 64 * 4;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_parse_prelude_win<RF, 128>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_parse_prelude", (k_parse_prelude_win<RF, 128>), (g.nblocks + 63) / 64, 64, lds, s, cont, g,
            NSP, boff, payload_off, max_ns, maxM, (u32*)c->dec_cum.p, (uint4*)c->dec_info.p, gflags, pa_info);
    } else {
        const size_t lds = (size_t)(256 + 72) * 64 * 4;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_parse_prelude_win<RF, 256>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_parse_prelude", (k_parse_prelude_win<RF, 256>), (g.nblocks + 63) / 64, 64, lds, s, cont, g,
            NSP, boff, payload_off, max_ns, maxM, (u32*)c->dec_cum.p, (uint4*)c->dec_info.p, gflags, pa_info);
    }
    // K8
    const u32 nseg = geo_nseg(g.block_ints, g.ckpt);
    u32 threads = (u32)rup((size_t)nseg * 4, 64);
    if (threads > 256) threads = 256;
    const size_t mfb = RF ? (size_t)T * 4 : 0;
    const size_t want_stream = rup((size_t)max_block_bytes + 32, 16);
    const size_t LDS_LIMIT = 150 * 1024;
    // normal path: rank/select tables (frames up to 2^16), staged stream while >= 3 WGs/CU still fit
    const size_t rs_tables = rup((size_t)(maxM >= 32 ? maxM / 32 : 1) * 8, 16) + 2 * rup((size_t)max_ep * 4, 16) + ANSX_DEC_SCRATCH;
    if (maxM <= 65536u && rs_tables <= LDS_LIMIT && !c->dbg.decode_table) {
        // per-quad stream rings when every segment of a full block has the same length; the (at
        // most one) partial block of the container then reads its stream straight from HBM
        // (measured on MI355X, 256 Mi ints: rings 0.73 vs 0.76 ms at 16 Ki / 1024, 0.73 vs 1.85 ms at
        // 64 Ki / 1024 where the stream no longer fits; staged 0.60 vs 0.69 ms at 16 Ki / 512 -- the
        // ring bookkeeping costs ~6 VALU per step, so it only pays when it frees a lot of LDS)
        const size_t ring_lds = (size_t)(threads / 4) * ANSX_RING_STRIDE + 16;  // + alignment slack
        const bool staged_fits = rs_tables + want_stream <= 52 * 1024;
        const bool ring_ok = g.ckpt != 0 && g.block_ints % g.ckpt == 0 && g.ckpt % 4 == 0
            && rs_tables + ring_lds <= 60 * 1024;
        // large tables (alphabets of thousands of symbols: one wave per block, a handful of waves per CU either
        // way): rings, 1.33 vs 2.08 ms on 2300-symbol alphabets -- the staging pass is pure latency there
        const bool ring_pays = !staged_fits || rs_tables >= 12 * 1024
            || 10 * (rs_tables + want_stream) > 16 * (rs_tables + ring_lds);
        const int force = c->dbg.decode_mode;  // tests: 1 "ring" | 2 "staged"
        const bool use_ring = ring_ok && (force ? force == 1 : ring_pays);
        if (use_ring) {
            // two blocks per workgroup, decoded in one instruction stream (k_decode_rank2): as long as four such workgroups
            // still fit a CU's LDS; the container's last one or two blocks take the single-block code inside that kernel
            const size_t lds2 = rup((size_t)(maxM >= 32 ? maxM / 32 : 1) * 16, 16) + 4 * rup((size_t)max_ep * 4, 16) + ANSX_DEC_SCRATCH + 2 * ring_lds;
            const int pair = c->dbg.decode_pair;  // tests / experiments: 1 never, 2 always (LDS permitting)
            if (pair != 1 && g.nblocks >= 2 && (pair == 2 ? lds2 <= 150 * 1024 : lds2 <= c->dbg.pair_lds_limit)) {
                if (lds2 > 48 * 1024)
                    HIPCHK(c, hipFuncSetAttribute((const void*)k_decode_rank2<RF>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                LAUNCH(c, "k_decode", (k_decode_rank2<RF>), (g.nblocks + 1) / 2, threads, lds2, s, cont, g, NSP, boff,
                    ck_state, ck_off, payload_off, d_out, maxM, max_ep, (u64)cont_bytes, (const u32*)c->dec_cum.p,
                    (const uint4*)c->dec_info.p, gflags);
                return ANSX_OK;
            }
            // Streams of a few bytes per step (the container's bytes per int, header fields only): the 256-byte speculative
            // rings -- sixteen instead of ten blocks of the headline workload per CU; an interval that outran its window is
            // decoded again (dec_segments_ring_small), so a wrong guess here costs time, never correctness.
            const int small = c->dbg.decode_small_ring;  // tests: 1 never, 2 always
            const bool lean = g.n != 0 && (double)(cont_bytes - payload_off) / (double)g.n <= 2.0 && g.ckpt % 16 == 0;
            if (small != 1 && (small == 2 || lean)) {
                const size_t ldss = rs_tables + (size_t)(threads / 4) * ANSX_SRING_STRIDE + 16;
                if (ldss > 48 * 1024)
                    HIPCHK(c, hipFuncSetAttribute((const void*)k_decode_rank<RF, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldss));
                LAUNCH(c, "k_decode", (k_decode_rank<RF, 2>), g.nblocks, threads, ldss, s, cont, g, NSP, boff,
                    ck_state, ck_off, payload_off, d_out, maxM, max_ep, (u64)cont_bytes, (const u32*)c->dec_cum.p,
                    (const uint4*)c->dec_info.p, gflags);
                return ANSX_OK;
            }
            const size_t lds = rs_tables + ring_lds;
            if (lds > 48 * 1024)
                HIPCHK(c, hipFuncSetAttribute((const void*)k_decode_rank<RF, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LAUNCH(c, "k_decode", (k_decode_rank<RF, 1>), g.nblocks, threads, lds, s, cont, g, NSP, boff,
                ck_state, ck_off, payload_off, d_out, maxM, max_ep, (u64)cont_bytes, (const u32*)c->dec_cum.p,
                (const uint4*)c->dec_info.p, gflags);
            return ANSX_OK;
        }
        size_t lds = rs_tables;
        u32 stream_cap = 0;
        if (staged_fits && !c->dbg.no_stream_lds) {
            lds += want_stream;
            stream_cap = (u32)want_stream;
        }
        if (lds > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_decode_rank<RF, 0>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_decode", (k_decode_rank<RF, 0>), g.nblocks, threads, lds, s, cont, g, NSP, boff, ck_state,
            ck_off, payload_off, d_out, maxM, max_ep, (u64)stream_cap, (const u32*)c->dec_cum.p,
            (const uint4*)c->dec_info.p, gflags);
        return ANSX_OK;
    }
    // frames above 2^16 (or forced): slot -> symbol table form, in LDS if it fits, else in HBM
    const size_t cb = rup(((size_t)max_ns + 2) * 4, 16);
    const size_t s2sb = rup((size_t)maxM * 2, 16);
    const size_t tables = cb + s2sb + mfb;
    if (tables <= LDS_LIMIT) {
        size_t lds = tables;
        u32 stream_cap = 0;
        if (tables + want_stream <= 52 * 1024) {
            lds += want_stream;
            stream_cap = (u32)want_stream;
        }
        if (lds > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_decode<true, RF>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_decode_table", (k_decode<true, RF>), g.nblocks, threads, lds, s, cont, g, NSP, boff,
            ck_state, ck_off, payload_off, d_out, maxM, max_ns, stream_cap, (u16*)nullptr,
            (u32*)c->dec_cum.p, (const uint4*)c->dec_info.p, gflags);
    } else {
        if ((rc = ensure(c, c->dec_s2s, (size_t)g.nblocks * maxM * 2))) return rc;
        size_t lds = mfb;
        u32 stream_cap = 0;
        if (mfb + want_stream <= 52 * 1024) {
            lds += want_stream;
            stream_cap = (u32)want_stream;
        }
        if (lds < 16) lds = 16;
        if (lds > 48 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void*)k_decode<false, RF>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LAUNCH(c, "k_decode_gtab", (k_decode<false, RF>), g.nblocks, threads, lds, s, cont, g, NSP, boff,
            ck_state, ck_off, payload_off, d_out, maxM, max_ns, stream_cap, (u16*)c->dec_s2s.p,
            (u32*)c->dec_cum.p, (const uint4*)c->dec_info.p, gflags);
    }
    return ANSX_OK;
}

__global__ void k_selftest_div(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, u64 n)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ansx_div_int31(a[i], b[i]);
}

__global__ void k_selftest_log2(const double* __restrict__ in, double* __restrict__ out, u64 n)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ansx_log2_portable(in[i]);
}

#define ANSX_G_HDR_BIT 9u  // gflags[ANSX_G_ERR]: the container header is not the one this decode was launched on
__global__ void k_check_header(const u8* __restrict__ cont, ansx_container_header want, u32* __restrict__ gflags)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u32* a = (const u32*)cont;
    const u32* w = (const u32*)&want;
    u32 diff = 0;
    for (int i = 0; i < 16; i++) diff |= a[i] ^ w[i];
    if (diff) atomicOr(&gflags[ANSX_G_ERR], 1u << ANSX_G_HDR_BIT);
}

__global__ void k_validate_index(ansx_geo g, const u64* __restrict__ boff, u64 payload_bytes,
    u32* __restrict__ gflags)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.nblocks) return;
    u64 a = boff[i], b = boff[i + 1];
    // a reference stream has at least 2 prelude bytes + one interpolative word + 32 state bytes; with
    // compaction a one-value block is just its alphabet header (8 bytes + code)
    const u64 min_bytes = g.pa ? 8 : 38;
    bool bad = (b < a) || (b - a) < min_bytes || (b - a) >= (1ull << 31) || b > payload_bytes;
    if (i == 0 && a != 0) bad = true;
    if (i == g.nblocks - 1 && b != payload_bytes) bad = true;
    if (bad) atomicOr(&gflags[ANSX_G_ERR], 1u << 3);
    else if (__hip_atomic_load(&gflags[ANSX_G_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)(b - a))
        atomicMax(&gflags[ANSX_G_PAD], (u32)(b - a));  // largest block stream (sizes the LDS staging)
}

int parse_header(const u8* h, size_t bytes, ansx_container_header* out)
{
    if (bytes < sizeof(ansx_container_header)) return ANSX_ERR_FORMAT;
    ansx_container_header H;
    memcpy(&H, h, sizeof(H));
    static const char magic[6] = { 'A', 'N', 'S', 'X', 'v', '3' };
    if (memcmp(H.magic, magic, 6) != 0) return ANSX_ERR_FORMAT;
    const u32 k = H.kind & 0xFFu;  // bit 8: per-block alphabet compaction, bit 9: wide restart points
    if ((H.kind & ~0x3FFu) || k > 3 || H.n == 0 || H.block_ints == 0) return ANSX_ERR_FORMAT;
    if ((k == ANSX_MSB || k == ANSX_INT) ? H.fidelity != 0 : (H.fidelity < 1 || H.fidelity > ANSX_MAX_FIDELITY)) return ANSX_ERR_FORMAT;
    *out = H;
    return ANSX_OK;
}

int decode_dev(ansx_ctx* c, const Plan& Pin, const u8* d_in, size_t in_bytes, u32* d_out,
    hipStream_t s, bool allow_spec = true)
{
    Plan P = Pin;
    bool spec = false;
    ansx_container_header Hspec;
    const std::array<u64, 4> hkey = { (u64)Pin.g.kind, (u64)Pin.g.f, (u64)Pin.g.n, (u64)in_bytes };
    int rc;
    if ((rc = ensure(c, c->misc, 64 + 8 * ((size_t)P.g.nblocks + 1)))) return rc;
    u32* gflags = (u32*)c->misc.p;
    HIPCHK(c, hipMemsetAsync(c->misc.p, 0, 64, s));
    const u32 f = P.g.f;
    const u32 T = fold_T(f);
    u32 maxM, max_ns, max_block_bytes;
    u32 max_ep = 0;  // bound on the symbols present in a block (0: as many as max_ns)
    const u8* cont;
    const u64* boff;
    const u64* ck_state = nullptr;
    const u32* ck_off = nullptr;
    u64 payload_off;
    u64 in_bytes_payload = 0;  // bytes of block streams behind payload_off (bounds the ring decoder reads)
    if (P.plain) {
        // one reference stream: copy behind a 16-byte guard (the decoder reads 8 bytes below
        // its cursor) and peek max_sym / log2 M on the host
        if (in_bytes < 38) return ANSX_ERR_FORMAT;
        if ((rc = ensure(c, c->plain, in_bytes + 64))) return rc;
        HIPCHK(c, hipMemsetAsync(c->plain.p, 0, 16, s));
        HIPCHK(c, hipMemcpyAsync((u8*)c->plain.p + 16, d_in, in_bytes, hipMemcpyDeviceToDevice, s));
        // ONE host round trip: the first 16 bytes and, for ANSrfold, the 8 bytes behind a most-frequent table (where the
        // prelude starts if the block was reordered) are requested together
        size_t peek = in_bytes < 16 ? in_bytes : 16;
        u8* hp = (u8*)c->h_pin + 64;
        u8* hp2 = hp + 16;
        const size_t pos_rf = 4 + 4 * (size_t)T;
        const bool have_rf = P.g.kind == ANSX_RFOLD && pos_rf + 8 <= in_bytes;
        HIPCHK(c, hipMemcpyAsync(hp, d_in, peek, hipMemcpyDeviceToHost, s));
        if (have_rf) HIPCHK(c, hipMemcpyAsync(hp2, d_in + pos_rf, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        size_t pos = 0;
        if (P.g.kind == ANSX_RFOLD) {
            u32 flag;
            memcpy(&flag, hp, 4);
            if (flag > 1) return ANSX_ERR_FORMAT;
            pos = 4 + (flag ? 4 * (size_t)T : 0);
            if (pos + 8 > in_bytes) return ANSX_ERR_FORMAT;
            if (flag) {
                hp = hp2;
                pos = 0;
            }
        }
        u32 ms = 0, sh = 0;
        for (int i = 0; i < 5; i++) {
            u8 cb = hp[pos++];
            ms += (u32)(cb & 127) << sh;
            if (!(cb & 128)) break;
            sh += 7;
        }
        u32 lg = hp[pos];
        const bool int_sp = P.g.kind == ANSX_INT && !P.g.pa;  // (any max_sym below 2^30: the model is parsed in rank space)
        if ((int_sp ? ms >= ANSX_SP_VALUE_LIMIT : ms >= P.NSP) || lg > 31) return ANSX_ERR_FORMAT;
        maxM = 1u << lg;
        max_ns = ms + 1 < P.NSP ? ms + 1 : P.NSP;
        // (the two index entries of the one block go up from pinned memory: no wait -- the page is next written by
        // this call's final read-back, which the stream orders behind this copy)
        u64* hb = (u64*)((u8*)c->h_pin + 64 + 32);
        hb[0] = 0, hb[1] = (u64)in_bytes;
        P.g.trusted_index = 1;  // the index of this one block is the pair written here, not container bytes
        u64* boff_ws = (u64*)((u8*)c->misc.p + 64);
        HIPCHK(c, hipMemcpyAsync(boff_ws, hb, 16, hipMemcpyHostToDevice, s));
        cont = (const u8*)c->plain.p;
        boff = boff_ws;
        payload_off = 16;
        max_block_bytes = (u32)in_bytes;
        in_bytes_payload = in_bytes;
    } else {
        u8* hp = (u8*)c->h_pin + 64;
        if (in_bytes < sizeof(ansx_container_header)) return ANSX_ERR_FORMAT;
        ansx_container_header H;
        const auto hit = c->hdr_cache.find(hkey);
        if (allow_spec && hit != c->hdr_cache.end()) {
            // Same shape as a container decoded before: launch on that header now, verify it on the device
            // (k_check_header); a different header repeats the call the slow way.  Every kernel treats header-derived
            // sizes as untrusted anyway (index entries, restart points and hints are bounds-checked against them).
            H = hit->second;
            memcpy(hp, &H, sizeof(H));
            spec = true;
            Hspec = H;
        } else {
            HIPCHK(c, hipMemcpyAsync(hp, d_in, sizeof(ansx_container_header), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
        }
        if ((rc = parse_header(hp, in_bytes, &H))) return rc;
        if ((H.kind & 0xFFu) != P.g.kind || H.fidelity != f || H.n != P.g.n) return ANSX_ERR_FORMAT;
        // the container, not the caller's options, defines the geometry
        ansx_opts o;
        o.block_ints = H.block_ints;
        o.ckpt_interval = H.ckpt_interval ? H.ckpt_interval : ANSX_NO_CHECKPOINTS;
        o.flags = (H.kind & 0x100u) ? ANSX_FLAG_COMPACT_ALPHABET : 0;
        o.reserved = 0;
        if (H.block_ints == ANSX_SINGLE_STREAM) return ANSX_ERR_FORMAT;
        if ((rc = make_plan((int)(H.kind & 0xFFu), (int)f, (size_t)H.n, &o, &P))) return ANSX_ERR_FORMAT;
        set_restart_format(&P, (H.kind & ANSX_KIND_WIDE_RESTART) != 0);
        if (P.g.nblocks != H.nblocks || P.g.nckf != H.ckpts_per_block || P.g.ckpt != H.ckpt_interval
            || P.lay.payload_off != H.payload_offset)
            return ANSX_ERR_FORMAT;
        // (written so that a crafted payload_bytes near 2^64 cannot wrap the sum; payload_offset covers
        // the index and the restart-point area, so this also places those inside the input)
        if (H.payload_offset > in_bytes || H.payload_bytes > in_bytes - H.payload_offset) return ANSX_ERR_FORMAT;
        // every block stream has a minimum length (index_entry_ok): a payload shorter than that for all blocks is malformed
        // whatever the index says (in particular payload_bytes == 0, which no parser may take for "no index")
        if (H.payload_bytes < (u64)P.g.nblocks * (P.g.pa ? 8u : 38u)) return ANSX_ERR_FORMAT;
        // (with compaction a list whose blocks all hold a single distinct value has no model at all)
        if (H.max_log2_frame > 31 || (H.max_nsyms == 0 && !P.g.pa) || H.max_nsyms > P.NSP) return ANSX_ERR_FORMAT;
        maxM = 1u << H.max_log2_frame;
        max_ns = H.max_nsyms ? H.max_nsyms : 1u;
        max_ep = std::min<u32>(max_ns, (u32)H.max_present_m1 + 1u);
        cont = d_in;
        boff = (const u64*)(d_in + P.lay.index_off);
        ck_state = (const u64*)(d_in + P.lay.ckstate_off);
        ck_off = (const u32*)(d_in + P.lay.ckoff_off);
        payload_off = H.payload_offset;
        in_bytes_payload = H.payload_bytes;
        P.g.payload_bytes = H.payload_bytes;  // every parser validates the two index entries of its own block (index_entry_ok)
        // The ring decoder needs nothing else from the index: no validation kernel, no read-back.  The staged /
        // straight-from-HBM forms size their LDS from the largest block stream, which only the index knows.
        const size_t rs_probe = rup((size_t)(maxM >= 32 ? maxM / 32 : 1) * 8, 16) + 2 * rup((size_t)max_ep * 4, 16) + ANSX_DEC_SCRATCH;
        const bool ring_certain = !P.g.pa && maxM <= 65536u && P.g.ckpt != 0 && P.g.block_ints % P.g.ckpt == 0 && P.g.ckpt % 4 == 0
            && c->dbg.decode_mode != 2 && !c->dbg.decode_table
            && rs_probe + (size_t)(std::min<u32>(256u, (u32)rup((size_t)geo_nseg(P.g.block_ints, P.g.ckpt) * 4, 64)) / 4) * ANSX_RING_STRIDE + 16 <= 60 * 1024;
        if (spec && !ring_certain) {  // (a read-back follows anyway: nothing to gain from the cached header)
            c->hdr_cache.erase(hkey);
            return decode_dev(c, Pin, d_in, in_bytes, d_out, s, false);
        }
        if (ring_certain) {
            max_block_bytes = (u32)std::min<size_t>(0x7FFFFFFFu, block_bound((int)P.g.kind, f, P.g.block_ints, false));  // (=> never "staged fits")
        } else {
            LAUNCH(c, "k_validate_index", k_validate_index, (P.g.nblocks + 255) / 256, 256, 0, s, P.g, boff,
                H.payload_bytes, gflags);
            // the index must be sane before any block is touched
            HIPCHK(c, hipMemcpyAsync(c->h_pin, c->misc.p, 16, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->h_pin[ANSX_G_ERR]) return flags_to_status(c->h_pin[ANSX_G_ERR]);
            max_block_bytes = c->h_pin[ANSX_G_PAD];
        }
    }
    const u32* hints = P.plain ? nullptr : (const u32*)(d_in + P.lay.hint_off);
    const uint4* pa_info = nullptr;
    if (P.g.pa) {  // alphabet headers first: they tell where every block's codec stream starts
        if ((rc = ensure(c, c->pa_alpha, (size_t)P.g.nblocks * P.g.block_ints * 4))) return rc;
        if ((rc = ensure(c, c->pa_info, (size_t)P.g.nblocks * 16))) return rc;
        LAUNCH(c, "k_pa_parse", k_pa_parse, (P.g.nblocks + 63) / 64, 64, 0, s, cont, P.g, boff, payload_off,
            (u32*)c->pa_alpha.p, (uint4*)c->pa_info.p, gflags);
        pa_info = (const uint4*)c->pa_info.p;
    }
    const bool int_sparse = P.g.kind == ANSX_INT && !P.g.pa;
    if (int_sparse) {
        // plain ANSint: the prelude ranges over the VALUES (any max_sym); its present symbols become the block's ranks
        // (ansx_intsparse.h) -- whichever model the encoder ran, the stream is the reference's
        if ((rc = ensure(c, c->pa_alpha, (size_t)P.g.nblocks * P.g.block_ints * 4))) return rc;
        if ((rc = ensure(c, c->pa_info, (size_t)P.g.nblocks * 16))) return rc;
        if ((rc = ensure(c, c->dec_cum, (size_t)P.g.nblocks * (P.NSP + 8) * 4))) return rc;
        if ((rc = ensure(c, c->dec_info, (size_t)P.g.nblocks * 16))) return rc;
        LAUNCH(c, "k_parse_prelude", k_int_sparse_parse, (P.g.nblocks + 63) / 64, 64, 0, s, cont, P.g, P.NSP, boff, payload_off, maxM,
            (u32*)c->dec_cum.p, (uint4*)c->dec_info.p, (u32*)c->pa_alpha.p, (uint4*)c->pa_info.p, gflags);
    }
    if (P.g.kind == ANSX_RFOLD)
        rc = launch_decode<true>(c, P.g, P.NSP, cont, boff, ck_state, ck_off, payload_off, d_out, maxM,
            max_ns, max_block_bytes, (u64)payload_off + in_bytes_payload, gflags, s, pa_info, hints, max_ep ? max_ep : max_ns);
    else
        rc = launch_decode<false>(c, P.g, P.NSP, cont, boff, ck_state, ck_off, payload_off, d_out, maxM,
            max_ns, max_block_bytes, (u64)payload_off + in_bytes_payload, gflags, s, pa_info, hints, max_ep ? max_ep : max_ns, int_sparse);
    if (rc) return rc;
    if (int_sparse)
        LAUNCH(c, "k_int_unmap", k_int_unmap, P.g.nblocks, 256, 0, s, P.g, (const u32*)c->pa_alpha.p, (const uint4*)c->pa_info.p, d_out, gflags);
    if (P.g.pa)
        LAUNCH(c, "k_pa_unmap", k_pa_unmap, P.g.nblocks, 256, 0, s, P.g, (const u32*)c->pa_alpha.p, pa_info, d_out, gflags);
    if (spec) LAUNCH(c, "k_check_header", k_check_header, 1, 64, 0, s, d_in, Hspec, gflags);
    HIPCHK(c, hipMemcpyAsync(c->h_pin, c->misc.p, 64, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (spec && (c->h_pin[ANSX_G_ERR] & (1u << ANSX_G_HDR_BIT))) {
        c->hdr_cache.erase(hkey);
        return decode_dev(c, Pin, d_in, in_bytes, d_out, s, false);
    }
    const int st = flags_to_status(c->h_pin[ANSX_G_ERR]);
    if (!P.plain && !spec && st == ANSX_OK) {
        ansx_container_header Hc;
        memcpy(&Hc, (u8*)c->h_pin + 64, sizeof(Hc));
        if (c->hdr_cache.find(hkey) == c->hdr_cache.end()) {
            // evict the OLDEST remembered shape (insertion order), never the one being stored
            c->hdr_order.erase(std::remove(c->hdr_order.begin(), c->hdr_order.end(), hkey), c->hdr_order.end());  // (left behind by a mismatch)
            c->hdr_order.push_back(hkey);
            while (c->hdr_order.size() > 64) {
                c->hdr_cache.erase(c->hdr_order.front());
                c->hdr_order.pop_front();
            }
        }
        c->hdr_cache[hkey] = Hc;
    }
    return st;
}

}  // namespace

namespace {
struct RcclApi {
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;  // optional: only reported (ansx_last_gather_ranks)
    bool ok = false;
};
const RcclApi& rccl_api()
{
    static RcclApi api = [] {
        RcclApi a;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
        a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
        a.ok = a.AllGather && a.GroupStart && a.GroupEnd && a.Send && a.Recv;
        return a;
    }();
    return api;
}
}  // namespace

// ================================================================================= C ABI
extern "C" {

int ansx_init(int device, ansx_ctx** out)
{
    if (!out) return ANSX_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ANSX_ERR_NO_DEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return ANSX_ERR_NO_DEVICE;
    }
    if (device >= ndev) return ANSX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ANSX_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ANSX_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ANSX_ERR_NO_DEVICE;  // gfx950-only code object
    ansx_ctx* c = new ansx_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? (u32)prop.multiProcessorCount : 256u;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess) {
        delete c;
        return ANSX_ERR_HIP;
    }
    if (hipHostMalloc((void**)&c->h_pin, 4096, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return ANSX_ERR_HIP;
    }
    static const char* const names[] = { "ANSX_TEST_TABLE16_FIXUP", "ANSX_ENCODE_GTAB16", "ANSX_PARSE_GENERIC", "ANSX_PARSE_WIN", "ANSX_PARSE_FAST",
        "ANSX_DECODE_TABLE", "ANSX_NO_STREAM_LDS", "ANSX_DECODE_MODE", "ANSX_PARSE_STAGE_WORDS", "ANSX_MODEL_FUSED", "ANSX_MODEL_SYNC", "ANSX_NS_HINT", "ANSX_T_HINT", "ANSX_NO_FAST_MODEL", "ANSX_FAST_GUARD", "ANSX_CAND_CHAINS", "ANSX_NEAR_BAND", "ANSX_TEST_NEAR_FLIP", "ANSX_WIDE_RESTART", "ANSX_TEST_WIDE_AT" };
    for (const char* nm : names)
        if (const char* v = getenv(nm)) (void)ansx_debug_set(c, nm, v);
    *out = c;
    return ANSX_OK;
}

int ansx_last_encode_stats(const ansx_ctx* c, ansx_encode_stats* out)
{
    if (!c || !out) return ANSX_ERR_ARG;
    *out = c->last;
#ifdef ANSX_STAMPS
    {
        static unsigned long long h[16 * 256 + 16];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)) == hipSuccess) {
            double acc[16] = {};
            int nn = 0;
#ifdef ANSX_STAMPS_RF
            const int last = 8;
#else
            const int last = 10;
#endif
            for (int w = 0; w < 256; w++) {
                if (!h[w * 16] || !h[w * 16 + last]) continue;
                nn++;
                for (int i = 1; i <= last; i++) acc[i] += (double)(h[w * 16 + i] - h[w * 16 + i - 1]);
            }
            fprintf(stderr, "[stamps] %d workgroups, 100 MHz ticks per phase:", nn);
            for (int i = 1; i <= 10; i++) fprintf(stderr, " %d:%.0f", i, nn ? acc[i] / nn : 0.0);
#ifdef ANSX_STAMPS_RF
            { double a9 = 0, a10 = 0, a2 = 0; int m = 0; for (int w = 0; w < 256; w++) if (h[w*16+9] && h[w*16+10] && h[w*16+1]) { a9 += (double)(h[w*16+9]-h[w*16+1]); a10 += (double)(h[w*16+10]-h[w*16+9]); a2 += (double)(h[w*16+2]-h[w*16+10]); m++; }
              if (m) fprintf(stderr, "  [insert split: clear %.0f count %.0f merge %.0f]", a9/m, a10/m, a2/m); }
#endif
            fprintf(stderr, "\n");
#ifdef ANSX_STAMPS_RF
            {
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int w = 0; w < 256; w++) if (h[w * 16] && h[w * 16 + 8]) { t0 = std::min(t0, h[w * 16]); t1 = std::max(t1, h[w * 16 + 8]); }
                fprintf(stderr, "[stamps] rfold: first start .. last end = %llu ticks\n", t1 - t0);
                for (int w = 0; w < 256; w += 8)
                    fprintf(stderr, "[stamps]  wg %5d start %7llu dur %5llu xcc %llx\n", w * 61 + 7, h[w * 16] - t0, h[w * 16 + 8] - h[w * 16], h[w * 16 + 9]);
            }
#endif
            double cc = 0, cl = 0, ct = 0;
            int n2 = 0;
            for (int w = 0; w < 256; w++)
                if (h[w * 16 + 14]) cc += (double)h[w * 16 + 12], cl += (double)h[w * 16 + 13], ct += (double)h[w * 16 + 14], n2++;
            if (n2) fprintf(stderr, "[stamps] k_candidates (%d waves): commit %.0f loop %.0f total %.0f ticks\n", n2, cc / n2, cl / n2, ct / n2);
        }
    }
#endif
    return ANSX_OK;
}

int ansx_debug_set(ansx_ctx* c, const char* name, const char* value)
{
    if (!c || !name) return ANSX_ERR_ARG;
    const bool on = value != nullptr && value[0] != 0 && strcmp(value, "0") != 0;
    if (!strcmp(name, "ANSX_TEST_TABLE16_FIXUP")) c->dbg.table16_fixup = on;
    else if (!strcmp(name, "ANSX_ENCODE_GTAB16")) c->dbg.encode_gtab16 = on;
    else if (!strcmp(name, "ANSX_PARSE_GENERIC")) c->dbg.parse_generic = on;
    else if (!strcmp(name, "ANSX_PARSE_WIN")) c->dbg.parse_win = on;
    else if (!strcmp(name, "ANSX_PARSE_FAST")) c->dbg.parse_fast = on;
    else if (!strcmp(name, "ANSX_DECODE_TABLE")) c->dbg.decode_table = on;
    else if (!strcmp(name, "ANSX_NO_STREAM_LDS")) c->dbg.no_stream_lds = on;
    else if (!strcmp(name, "ANSX_MODEL_FUSED")) c->dbg.model_fused = on;
    else if (!strcmp(name, "ANSX_MODEL_SYNC")) c->dbg.model_sync = on;
    else if (!strcmp(name, "ANSX_DECODE_MODE"))
        c->dbg.decode_mode = !value ? 0 : !strcmp(value, "ring") ? 1 : !strcmp(value, "staged") ? 2 : 0;
    else if (!strcmp(name, "ANSX_PARSE_STAGE_WORDS")) c->dbg.parse_stage_words = value ? (u32)strtoul(value, nullptr, 10) : 0u;
    else if (!strcmp(name, "ANSX_NS_HINT")) c->dbg.ns_hint = value ? (u32)strtoul(value, nullptr, 10) : 0u;
    else if (!strcmp(name, "ANSX_T_HINT")) c->dbg.t_hint = value ? (u32)strtoul(value, nullptr, 10) : 0u;
    else if (!strcmp(name, "ANSX_NO_FAST_MODEL")) c->dbg.no_fast_model = on;
    else if (!strcmp(name, "ANSX_WIDE_RESTART")) c->dbg.wide_restart = on;
    else if (!strcmp(name, "ANSX_NO_PC")) c->dbg.no_pc = on;
    else if (!strcmp(name, "ANSX_FORCE_PC")) c->dbg.force_pc = on;
    else if (!strcmp(name, "ANSX_USE_PC")) c->dbg.use_pc = on;
    else if (!strcmp(name, "ANSX_NO_PC_AUTO")) c->dbg.no_pc_auto = on;
    else if (!strcmp(name, "ANSX_NO_BIG_GEO")) c->dbg.no_big_geo = on;
    else if (!strcmp(name, "ANSX_FIN_ONE_WAVE")) c->dbg.fin_one_wave = on;
    else if (!strcmp(name, "ANSX_TEST_SP_BITS")) c->dbg.test_sp_bits = value ? (u32)atoi(value) : 0u;
    else if (!strcmp(name, "ANSX_PC_B_PAIRS")) c->dbg.pc_b_pairs = (value && value[0] == '1') ? 1u : 2u;
    else if (!strcmp(name, "ANSX_ENCODE_MODE2")) c->dbg.encode_mode2 = on;
    else if (!strcmp(name, "ANSX_DECODE_SMALL_RING"))
        c->dbg.decode_small_ring = !value ? 0 : !strcmp(value, "never") ? 1 : !strcmp(value, "always") ? 2 : 0;
    else if (!strcmp(name, "ANSX_DECODE_PAIR"))
        c->dbg.decode_pair = !value ? 0 : !strcmp(value, "never") ? 1 : !strcmp(value, "always") ? 2 : 0;
    else if (!strcmp(name, "ANSX_DECODE_PAIR_LDS")) c->dbg.pair_lds_limit = (value && value[0]) ? (u32)strtoul(value, nullptr, 10) : 0u;
    else if (!strcmp(name, "ANSX_FORGET_HINTS")) {  // the next call of every geometry is a first call again (bench.py: first_call_ms)
        c->ns_hint.clear();
        c->rf_hint.clear();
        c->t_hint.clear();
        c->wide_hint.clear();
        c->hdr_cache.clear();
        c->hdr_order.clear();
    }
    else if (!strcmp(name, "ANSX_TEST_WIDE_AT")) {
        const u32 v = (value && value[0]) ? (u32)strtoul(value, nullptr, 10) : 16u;
        if (v > 16) return ANSX_ERR_ARG;
        c->dbg.wide_at = v;
    }
    else if (!strcmp(name, "ANSX_NEAR_BAND")) c->dbg.near_band = (value && value[0]) ? strtod(value, nullptr) : ANSX_NEAR_BAND;
    else if (!strcmp(name, "ANSX_TEST_NEAR_FLIP")) c->dbg.near_flip = on;
    else if (!strcmp(name, "ANSX_CAND_CHAINS")) {
        const u32 v = value ? (u32)strtoul(value, nullptr, 10) : 0u;
        if (v > 2) return ANSX_ERR_ARG;
        c->dbg.cand_chains = v;
    } else if (!strcmp(name, "ANSX_FAST_GUARD")) c->dbg.fast_guard = (value && value[0]) ? strtod(value, nullptr) : ANSX_FAST_GUARD;
    else return ANSX_ERR_ARG;
    return ANSX_OK;
}

void ansx_destroy(ansx_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = { &c->pre_work, &c->hist, &c->hterm, &c->sortF, &c->sortSym, &c->attS, &c->prevS, &c->attMeta, &c->blk,
        &c->table, &c->tab32, &c->scratch, &c->misc, &c->mapped, &c->mostfreq, &c->stage_in, &c->stage_out,
        &c->dec_s2s, &c->dec_cum, &c->dec_info, &c->plain, &c->rf_tmp, &c->log2lut, &c->pa_alpha, &c->pa_info, &c->pairs, &c->lg2i, &c->sizes, &c->nearlist, &c->force, &c->geo_big };
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& kv : c->geo)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& r : c->recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* ansx_strerror(int st)
{
    switch (st) {
    case ANSX_OK: return "ok";
    case ANSX_ERR_ARG: return "invalid argument";
    case ANSX_ERR_CAPACITY: return "output buffer too small";
    case ANSX_ERR_FORMAT: return "malformed container or stream";
    case ANSX_ERR_HIP: return "HIP runtime error";
    case ANSX_ERR_NO_DEVICE: return "no usable gfx950 device";
    case ANSX_ERR_DOMAIN: return "input value outside [0, 2^30)";
    case ANSX_ERR_MODEL: return "frequency normalisation hit the reference's degenerate exit";
    default: return "unknown status";
    }
}

int ansx_last_hip_error(const ansx_ctx* c) { return c ? c->last_hip : 0; }

int ansx_codec_name(int kind, int f, char* buf, size_t buflen)
{
    if (kind == ANSX_MSB) return snprintf(buf, buflen, "ANSmsb");
    if (kind == ANSX_INT) return snprintf(buf, buflen, "ANS");  // methods.hpp:485
    return snprintf(buf, buflen, "%s-%d", kind == ANSX_RFOLD ? "ANSrfold" : "ANSfold", f);
}

size_t ansx_bound(int kind, int f, size_t n, const ansx_opts* opts)
{
    Plan P;
    if (make_plan(kind, f, n, opts, &P)) return 0;
    if (!P.plain) set_restart_format(&P, true);  // (the larger of the two index forms)
    size_t per = block_bound(kind, (u32)f, 0, false);
    return (size_t)P.lay.payload_off + (size_t)P.g.nblocks * per + (P.g.pa ? 11 : 7) * n + 64;
}

int ansx_encode_dev(ansx_ctx* c, int kind, int f, const uint32_t* d_in, size_t n, uint8_t* d_out,
    size_t cap, size_t* out_bytes, const ansx_opts* opts, void* stream)
{
    if (!c || !d_in || !d_out || !out_bytes) return ANSX_ERR_ARG;
    if (((uintptr_t)d_out & 15u) || ((uintptr_t)d_in & 3u)) return ANSX_ERR_ARG;
    Plan P;
    int rc = make_plan(kind, f, n, opts, &P);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return encode_dev(c, P, d_in, d_out, cap, out_bytes, s);
}

int ansx_decode_dev(ansx_ctx* c, int kind, int f, const uint8_t* d_in, size_t in_bytes,
    uint32_t* d_out, size_t n, const ansx_opts* opts, void* stream)
{
    if (!c || !d_in || !d_out) return ANSX_ERR_ARG;
    if (((uintptr_t)d_in & 15u) || ((uintptr_t)d_out & 15u)) return ANSX_ERR_ARG;
    Plan P;
    int rc = make_plan(kind, f, n, opts, &P);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return decode_dev(c, P, d_in, in_bytes, d_out, s);
}

int ansx_encode(ansx_ctx* c, int kind, int f, const uint32_t* in, size_t n, uint8_t* out, size_t cap,
    size_t* out_bytes, const ansx_opts* opts)
{
    if (!c || !in || !out || !out_bytes) return ANSX_ERR_ARG;
    Plan P;
    int rc = make_plan(kind, f, n, opts, &P);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    size_t bound = ansx_bound(kind, f, n, opts);
    size_t dcap = cap < bound ? cap : bound;
    if ((rc = ensure(c, c->stage_in, n * 4 + 16))) return rc;
    if ((rc = ensure(c, c->stage_out, dcap + 64))) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->stage_in.p, in, n * 4, hipMemcpyHostToDevice, s));
    size_t nb = 0;
    rc = encode_dev(c, P, (const u32*)c->stage_in.p, (u8*)c->stage_out.p, dcap, &nb, s);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, nb, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *out_bytes = nb;
    return ANSX_OK;
}

int ansx_decode(ansx_ctx* c, int kind, int f, const uint8_t* in, size_t in_bytes, uint32_t* out,
    size_t n, const ansx_opts* opts)
{
    if (!c || !in || !out) return ANSX_ERR_ARG;
    Plan P;
    int rc = make_plan(kind, f, n, opts, &P);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = ensure(c, c->stage_out, in_bytes + 64))) return rc;
    if ((rc = ensure(c, c->stage_in, n * 4 + 16))) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->stage_out.p, in, in_bytes, hipMemcpyHostToDevice, s));
    rc = decode_dev(c, P, (const u8*)c->stage_out.p, in_bytes, (u32*)c->stage_in.p, s);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->stage_in.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return ANSX_OK;
}

int ansx_merge_containers_dev(ansx_ctx* c, const uint8_t* const* d_parts, const size_t* part_bytes, int nparts,
    uint8_t* d_out, size_t cap, size_t* out_bytes, void* stream)
{
    if (!c || !d_parts || !part_bytes || !d_out || !out_bytes || nparts < 1 || nparts > ANSX_MERGE_MAX_PARTS) return ANSX_ERR_ARG;
    if ((uintptr_t)d_out & 15u) return ANSX_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // the part headers decide the layout: one small read-back
    u8* hp = (u8*)c->h_pin + 64;  // 4096-byte pinned page: 63 headers fit behind the first 64 bytes
    std::vector<ansx_container_header> H((size_t)nparts);
    for (int i = 0; i < nparts; i += 63) {
        const int m = std::min(63, nparts - i);
        for (int j = 0; j < m; j++) {
            if (!d_parts[i + j] || ((uintptr_t)d_parts[i + j] & 7u) || part_bytes[i + j] < sizeof(ansx_container_header)) return ANSX_ERR_ARG;
            HIPCHK(c, hipMemcpyAsync(hp + 64 * j, d_parts[i + j], 64, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(c, hipStreamSynchronize(s));
        for (int j = 0; j < m; j++) {
            int rc = parse_header(hp + 64 * j, part_bytes[i + j], &H[(size_t)(i + j)]);
            if (rc) return rc;
        }
    }
    ansx_merge_desc D;
    memset(&D, 0, sizeof(D));
    u64 nblocks = 0, n = 0, pay = 0;
    u32 maxlg = 0, maxns = 0;
    for (int i = 0; i < nparts; i++) {
        const ansx_container_header& h = H[(size_t)i];
        if (h.kind != H[0].kind || h.fidelity != H[0].fidelity || h.block_ints != H[0].block_ints
            || h.ckpt_interval != H[0].ckpt_interval || h.ckpts_per_block != H[0].ckpts_per_block)
            return ANSX_ERR_FORMAT;
        if (i + 1 < nparts && h.n % h.block_ints != 0) return ANSX_ERR_FORMAT;  // only the last part may end in a partial block
        if (h.payload_offset > part_bytes[i] || h.payload_bytes > part_bytes[i] - h.payload_offset) return ANSX_ERR_FORMAT;
        if (h.payload_offset > 0xFFFFFFFFull) return ANSX_ERR_FORMAT;
        // the kernel derives every section of a part from its header's nblocks: the header must describe exactly the
        // layout its own geometry implies (as decode_dev checks), or a crafted part could send the copy past its buffer
        {
            if (h.n == 0 || h.block_ints == 0 || h.block_ints == ANSX_SINGLE_STREAM) return ANSX_ERR_FORMAT;
            ansx_opts po = { h.block_ints, h.ckpt_interval ? h.ckpt_interval : ANSX_NO_CHECKPOINTS,
                (h.kind & 0x100u) ? (u32)ANSX_FLAG_COMPACT_ALPHABET : 0u, 0 };
            Plan PP;
            if (make_plan((int)(h.kind & 0xFFu), (int)h.fidelity, (size_t)h.n, &po, &PP)) return ANSX_ERR_FORMAT;
            set_restart_format(&PP, (h.kind & ANSX_KIND_WIDE_RESTART) != 0);
            if (PP.g.nblocks != h.nblocks || PP.g.nckf != h.ckpts_per_block || PP.lay.payload_off != h.payload_offset) return ANSX_ERR_FORMAT;
        }
        D.part[i].src = d_parts[i];
        D.part[i].first_block = nblocks;
        D.part[i].pay_base = pay;
        D.part[i].payload_bytes = h.payload_bytes;
        D.part[i].nblocks = h.nblocks;
        D.part[i].payload_off = (u32)h.payload_offset;
        nblocks += h.nblocks;
        n += h.n;
        pay += h.payload_bytes;
        maxlg = std::max(maxlg, h.max_log2_frame);
        maxns = std::max(maxns, h.max_nsyms);
    }
    if (nblocks > 0x7FFFFFFFull) return ANSX_ERR_ARG;
    // (bit 8 of the kind word: per-block alphabet compaction -- a flag of the plan, kept in the merged header)
    ansx_opts o = { H[0].block_ints, H[0].ckpt_interval ? H[0].ckpt_interval : ANSX_NO_CHECKPOINTS,
        (H[0].kind & 0x100u) ? (u32)ANSX_FLAG_COMPACT_ALPHABET : 0u, 0 };
    Plan P;
    if (make_plan((int)(H[0].kind & 0xFFu), (int)H[0].fidelity, (size_t)n, &o, &P)) return ANSX_ERR_FORMAT;
    // (bit 9: the restart-point format.  The parts agree on it -- their kind words are equal -- and the result keeps it;
    // a part that needed wide restart points next to parts that did not is refused: re-encode those with ANSX_WIDE_RESTART)
    set_restart_format(&P, (H[0].kind & ANSX_KIND_WIDE_RESTART) != 0);
    if (P.g.nblocks != nblocks || P.g.nckf != H[0].ckpts_per_block) return ANSX_ERR_FORMAT;
    const u64 total = P.lay.payload_off + pay;
    if (total > cap) return ANSX_ERR_CAPACITY;
    D.nparts = (u32)nparts;
    D.nckf = P.g.nckf;
    D.ckw = P.g.ckw;
    D.ckoff_off = P.lay.ckoff_off;
    D.ckstate_off = P.lay.ckstate_off;
    D.hint_off = P.lay.hint_off;
    D.payload_off = P.lay.payload_off;
    // header, final index entry, alignment padding: written from the host image
    HIPCHK(c, hipMemsetAsync(d_out, 0, (size_t)P.lay.payload_off, s));
    ansx_container_header M = H[0];
    M.n = n;
    M.nblocks = (u32)nblocks;
    M.max_log2_frame = maxlg;
    M.max_nsyms = maxns;
    {
        u32 mp = 0;
        for (int i = 0; i < nparts; i++) mp = std::max<u32>(mp, H[(size_t)i].max_present_m1);
        M.max_present_m1 = (u16)mp;
    }
    M.payload_bytes = pay;
    M.payload_offset = P.lay.payload_off;
    memcpy(hp, &M, 64);
    memcpy(hp + 64, &pay, 8);
    HIPCHK(c, hipMemcpyAsync(d_out, hp, 64, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_out + 64 + 8 * nblocks, hp + 64, 8, hipMemcpyHostToDevice, s));
    u64 max_pieces = 1;
    for (int i = 0; i < nparts; i++) {
        const u64 nb_ = D.part[i].nblocks;
        const u64 cko_unit = D.ckw ? 4 : ANSX_CK_RECORD, cks_unit = D.ckw ? 32 : 0;  // (as k_merge_containers walks them)
        const u64 pieces = (8 * nb_ + 65535) / 65536 + (cko_unit * nb_ * D.nckf + 65535) / 65536 + (cks_unit * nb_ * D.nckf + 65535) / 65536
            + (32 * nb_ + 65535) / 65536 + (D.part[i].payload_bytes + 65535) / 65536;
        max_pieces = std::max(max_pieces, pieces);
    }
    prof_begin(c, "k_merge_containers", s);
    hipLaunchKernelGGL(k_merge_containers, dim3((u32)max_pieces, (u32)nparts), dim3(256), 0, s, D, d_out);
    prof_end(c, s);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));  // hp is reused by the next call
    *out_bytes = (size_t)total;
    return ANSX_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU gather over RCCL: ncclAllGather of the container sizes, then one grouped
// ncclSend / ncclRecv round -- every rank straight to the root, each over its own xGMI link -- and the merge
// kernel on the root.  librccl is not a link-time dependency of this library: the five entry points are
// resolved with dlopen at first use (a caller that holds an ncclComm_t has the library loaded already).
int ansx_gather_containers(ansx_ctx* c, void* nccl_comm, int rank, int nranks, int root, const uint8_t* d_container,
    size_t bytes, uint8_t* d_recv, size_t slot_bytes, uint8_t* d_merged, size_t merged_cap, size_t* merged_bytes,
    void* stream)
{
    if (!c || !nccl_comm || !d_container || !merged_bytes || nranks < 1 || nranks > ANSX_MERGE_MAX_PARTS || rank < 0
        || rank >= nranks || root < 0 || root >= nranks || bytes < sizeof(ansx_container_header))
        return ANSX_ERR_ARG;
    if (rank == root && (!d_recv || !d_merged || (slot_bytes & 15u) || ((uintptr_t)d_recv & 15u))) return ANSX_ERR_ARG;
    const RcclApi& R = rccl_api();
    if (!R.ok) return ANSX_ERR_NO_DEVICE;  // no RCCL library on this system
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    *merged_bytes = 0;
    // 1. sizes of all rank containers AND every rank's idea of slot_bytes, everywhere: the capacity check below uses the
    //    ROOT's slot_bytes on every rank, so that all ranks take the same branch (a rank that returned early on its own
    //    value would leave the root waiting in its ncclRecv)
    int rc;
    if ((rc = ensure(c, c->misc, 64 + 8 * (3 * (size_t)ANSX_MERGE_MAX_PARTS + 2)))) return rc;
    u64* d_sizes = (u64*)((u8*)c->misc.p + 64);
    u64* h_sizes = (u64*)((u8*)c->h_pin + 64);
    const u64 mine[2] = { (u64)bytes, (u64)slot_bytes };
    HIPCHK(c, hipMemcpyAsync(d_sizes + 2 * ANSX_MERGE_MAX_PARTS, mine, 16, hipMemcpyHostToDevice, s));
    if (R.AllGather(d_sizes + 2 * ANSX_MERGE_MAX_PARTS, d_sizes, 2, ncclUint64, comm, s) != ncclSuccess) return ANSX_ERR_HIP;
    HIPCHK(c, hipMemcpyAsync(h_sizes, d_sizes, 16 * (size_t)nranks, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int comm_ranks = 0;
    if (R.CommCount && R.CommCount(comm, &comm_ranks) == ncclSuccess) c->last_gather_ranks = comm_ranks;
    const u64 root_slot = h_sizes[2 * root + 1];
    std::vector<size_t> sizes((size_t)nranks);
    for (int r = 0; r < nranks; r++) {
        sizes[(size_t)r] = (size_t)h_sizes[2 * r];
        if (h_sizes[2 * r] > root_slot || h_sizes[2 * r] < sizeof(ansx_container_header)) return ANSX_ERR_CAPACITY;
    }
    // 2. every rank's container into its slot on the root
    if (R.GroupStart() != ncclSuccess) return ANSX_ERR_HIP;
    bool bad = false;
    if (rank == root) {
        for (int r = 0; r < nranks; r++)
            if (r != root) bad |= R.Recv(d_recv + (size_t)r * slot_bytes, sizes[(size_t)r], ncclUint8, r, comm, s) != ncclSuccess;
    } else {
        bad |= R.Send(d_container, bytes, ncclUint8, root, comm, s) != ncclSuccess;
    }
    if (R.GroupEnd() != ncclSuccess || bad) return ANSX_ERR_HIP;
    if (rank != root) return ANSX_OK;
    HIPCHK(c, hipMemcpyAsync(d_recv + (size_t)root * slot_bytes, d_container, bytes, hipMemcpyDeviceToDevice, s));
    // 3. one container
    std::vector<const uint8_t*> parts((size_t)nranks);
    for (int r = 0; r < nranks; r++) parts[(size_t)r] = d_recv + (size_t)r * slot_bytes;
    return ansx_merge_containers_dev(c, parts.data(), sizes.data(), nranks, d_merged, merged_cap, merged_bytes, s);
}

int ansx_last_gather_ranks(const ansx_ctx* c) { return c ? c->last_gather_ranks : 0; }

int ansx_container_info(const uint8_t* container, size_t bytes, ansx_container_header* out)
{
    if (!container || !out) return ANSX_ERR_ARG;
    return parse_header(container, bytes, out);
}

int ansx_profile_enable(ansx_ctx* c, int on)
{
    if (!c) return ANSX_ERR_ARG;
    c->profile = on != 0;
    return ANSX_OK;
}

static void prof_collect(ansx_ctx* c)
{
    for (auto& r : c->recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        auto it = c->acc.find(r.name);
        if (it == c->acc.end()) {
            c->order.push_back(r.name);
            c->acc[r.name] = std::make_pair((double)ms, (u64)1);
        } else {
            it->second.first += ms;
            it->second.second += 1;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    c->recs.clear();
}

int ansx_profile_reset(ansx_ctx* c)
{
    if (!c) return ANSX_ERR_ARG;
    prof_collect(c);
    c->acc.clear();
    c->order.clear();
    return ANSX_OK;
}

int ansx_profile_get(ansx_ctx* c, ansx_kernel_time* out, int max_entries, int* count)
{
    if (!c || !count) return ANSX_ERR_ARG;
    prof_collect(c);
    int k = 0;
    for (auto& name : c->order) {
        if (out && k < max_entries) {
            memset(&out[k], 0, sizeof(out[k]));
            strncpy(out[k].name, name.c_str(), sizeof(out[k].name) - 1);
            out[k].total_ms = c->acc[name].first;
            out[k].launches = c->acc[name].second;
        }
        k++;
    }
    *count = k;
    return ANSX_OK;
}

size_t ansx_workspace_bytes(const ansx_ctx* c)
{
    if (!c) return 0;
    const DevBuf* bufs[] = { &c->pre_work, &c->hist, &c->hterm, &c->sortF, &c->sortSym, &c->attS, &c->prevS, &c->attMeta, &c->blk,
        &c->table, &c->tab32, &c->scratch, &c->misc, &c->mapped, &c->mostfreq, &c->stage_in, &c->stage_out,
        &c->dec_s2s, &c->dec_cum, &c->dec_info, &c->plain, &c->rf_tmp, &c->pa_alpha, &c->pa_info, &c->pairs, &c->sizes };
    size_t t = 0;
    for (const DevBuf* b : bufs) t += b->cap;
    return t;
}

double ansx_host_log2(double x) { return ansx_log2_portable(x); }

static int gen_setup(int dist, double a, double b, uint64_t seed, ansx_gen_params* P)
{
    if (dist == ANSX_GEN_UNIFORM) {
        if (!(a >= 0.0 && a <= b && b <= 4294967295.0) || a != __builtin_floor(a) || b != __builtin_floor(b)) return ANSX_ERR_ARG;
    } else if (dist == ANSX_GEN_GEOMETRIC) {
        if (!(a > 0.0 && a < 1.0)) return ANSX_ERR_ARG;
    } else if (dist == ANSX_GEN_ZIPF) {
        if (!(a >= 1.0 && a <= 1073741823.0 && b > 0.0 && b < 64.0) || a != __builtin_floor(a)) return ANSX_ERR_ARG;
    } else return ANSX_ERR_ARG;
    P->dist = (u32)dist;
    P->seed = seed;
    P->a = a;
    P->b = b;
    gen_prepare(P);
    return ANSX_OK;
}

int ansx_generate_dev(ansx_ctx* c, int dist, double a, double b, uint64_t seed, uint64_t first_index,
    uint32_t* d_out, size_t n, void* stream)
{
    if (!c || !d_out || n == 0) return ANSX_ERR_ARG;
    ansx_gen_params P;
    int rc = gen_setup(dist, a, b, seed, &P);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    LAUNCH(c, "k_generate", k_generate, (n + 255) / 256, 256, 0, s, P, d_out, (u64)n, (u64)first_index);
    return ANSX_OK;
}

int ansx_generate_host(int dist, double a, double b, uint64_t seed, uint64_t first_index, uint32_t* out, size_t n)
{
    if (!out || n == 0) return ANSX_ERR_ARG;
    ansx_gen_params P;
    int rc = gen_setup(dist, a, b, seed, &P);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) out[i] = gen_value(P, first_index + i);
    return ANSX_OK;
}

int ansx_zipf_from_uniform(double n, double q, double u01, uint32_t* k, int* accepted)
{
    if (!k || !accepted || !(u01 >= 0.0 && u01 <= 1.0)) return ANSX_ERR_ARG;
    ansx_gen_params P;
    int rc = gen_setup(ANSX_GEN_ZIPF, n, q, 0, &P);
    if (rc) return rc;
    *accepted = gen_zipf_try(P, u01, k) ? 1 : 0;
    return ANSX_OK;
}

int ansx_selftest_log2(ansx_ctx* c, const double* in, double* out, size_t n)
{
    if (!c || !in || !out || n == 0) return ANSX_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->stage_in, n * 8))) return rc;
    if ((rc = ensure(c, c->stage_out, n * 8))) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->stage_in.p, in, n * 8, hipMemcpyHostToDevice, s));
    LAUNCH(c, "k_selftest_log2", k_selftest_log2, (n + 255) / 256, 256, 0, s, (const double*)c->stage_in.p,
        (double*)c->stage_out.p, (u64)n);
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return ANSX_OK;
}

int ansx_selftest_div(ansx_ctx* c, const double* a, const double* b, double* out, size_t n)
{
    if (!c || !a || !b || !out || n == 0) return ANSX_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->stage_in, 2 * n * 8))) return rc;
    if ((rc = ensure(c, c->stage_out, n * 8))) return rc;
    hipStream_t s = c->stream;
    double* da = (double*)c->stage_in.p;
    HIPCHK(c, hipMemcpyAsync(da, a, n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(da + n, b, n * 8, hipMemcpyHostToDevice, s));
    LAUNCH(c, "k_selftest_div", k_selftest_div, (n + 255) / 256, 256, 0, s, (const double*)da,
        (const double*)(da + n), (double*)c->stage_out.p, (u64)n);
    HIPCHK(c, hipMemcpyAsync(out, c->stage_out.p, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return ANSX_OK;
}

}  // extern "C"

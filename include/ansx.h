/*
 * ansx — MI355X-native ANSfold / ANSrfold codec: C-ABI boundary.
 *
 * This is the drop-in boundary for ONE hot path of mpetri/ans-large-alphabet: the
 * ANSfold<f> / ANSrfold<f> encode()/decode() pair.  Plain pointers and sizes only, so any host
 * language can bind it (the reference is header-only C++17 and has no FFI; the C++ mirror with
 * the reference's exact static signatures is ans_large_alphabet_amd/include/ansx_methods.hpp).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *
 *   ansx_encode / ansx_encode_dev   ANSfold<f>::encode   include/methods.hpp:535-540
 *                                   -> ans_fold_compress<f>           include/ans_fold.hpp:238-281
 *                                   ANSrfold<f>::encode  include/methods.hpp:555-560
 *                                   -> ans_reorder_fold_compress<f>   include/ans_reorder_fold.hpp:312-355
 *   ansx_decode / ansx_decode_dev   ANSfold<f>::decode   include/methods.hpp:541-546
 *                                   -> ans_fold_decompress<f>         include/ans_fold.hpp:283-311
 *                                   ANSrfold<f>::decode  include/methods.hpp:561-566
 *                                   -> ans_reorder_fold_decompress<f> include/ans_reorder_fold.hpp:357-385
 *   ansx_bound                      the harness's "n*8 bytes" output sizing, src/table_efficiency.cpp:73-74
 *                                   (the reference never checks dstCapacity, ans_fold.hpp:239-240)
 *   ansx_codec_name                 ANSfold<f>::name / ANSrfold<f>::name  include/methods.hpp:530-533,550-553
 *
 * Output format.  A reference encode() call is 4 serial rANS chains, so device
 * parallelism comes from independent blocks.  With opts.block_ints != ANSX_SINGLE_STREAM the
 * output is a *container*: a 64-byte header, a block index, decoder restart points, then the
 * concatenation of one UNMODIFIED reference stream per block — each block's bytes are
 * identical to what ANSfold<f>::encode(block) / ANSrfold<f>::encode(block) writes (modulo the
 * reference's own indeterminate padding bits, SURVEY F2, which are written as zero).
 * With opts.block_ints == ANSX_SINGLE_STREAM the output is exactly one reference stream for the
 * whole list (no header); ansx_decode with the same option accepts streams produced by the
 * reference CPU encoder.  Container layout: see DESIGN.md section 3 and ansx_container_header.
 *
 * Errors: the reference has none on this path (malformed input is UB, capacity is unchecked);
 * every function here returns an ansx_status instead.  n == 0 is an error (the reference never
 * terminates on it); values must be < 2^30 (rfold: value + 2^(f+7) < 2^30), the
 * reference's own decode limit (ans_fold.hpp:198-200).
 *
 * Threading: a context is bound to one device and must not be used from two host threads at
 * once; distinct contexts are independent (the reference codec is stateless and re-entrant).
 */
#ifndef ANSX_H
#define ANSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ansx_ctx ansx_ctx;

typedef enum {
    ANSX_FOLD = 0,  /* ANSfold<f>  */
    ANSX_RFOLD = 1, /* ANSrfold<f> */
    ANSX_MSB = 2,   /* ANSmsb (include/methods.hpp:499-515 -> include/ans_msb.hpp); fidelity must be 0 */
    ANSX_INT = 3    /* ANSint, name() == "ANS" (include/methods.hpp:484-497 -> include/ans_int.hpp); fidelity must be 0.
                       Its model spans every value up to the largest (ans_int.hpp:41-51).  Plain: any values below 2^30,
                       block container or ANSX_SINGLE_STREAM (= the bytes of ans_int_compress).  Lists whose values
                       stay below 16384 use a dense model; beyond that every block is modelled over its distinct
                       values (ranks) and only its prelude ranges over the values -- the same bytes -- which takes
                       blocks (single-stream: lists) of any length with at most 16384 DISTINCT values each:
                       ANSX_ERR_DOMAIN otherwise, and for a block
                       whose prelude would exceed 64 KiB (about 32 bits per distinct value).  A plain-ANSint
                       container carries no parse hints and its max_nsyms bounds a block's DISTINCT values, so its bytes
                       do not tell which of the two models wrote it.  With ANSX_FLAG_COMPACT_ALPHABET: the
                       harness's own layout, a block's dense ranks behind an alphabet header.  32-bit frequencies:
                       frames up to 2^27 (beyond, the reference's own 64-bit bound overflows) */
} ansx_kind;

typedef enum {
    ANSX_OK = 0,
    ANSX_ERR_ARG = 1,       /* bad kind / fidelity / n == 0 / null pointer / bad options      */
    ANSX_ERR_CAPACITY = 2,  /* output buffer too small (see ansx_bound)                        */
    ANSX_ERR_FORMAT = 3,    /* container/stream failed validation                              */
    ANSX_ERR_HIP = 4,       /* a HIP runtime call failed (ansx_last_hip_error has the code)    */
    ANSX_ERR_NO_DEVICE = 5, /* no usable gfx950 device                                         */
    ANSX_ERR_DOMAIN = 6,    /* an input value is outside the reference's decodable domain      */
    ANSX_ERR_MODEL = 7      /* normalisation hit the reference's degenerate exit   */
} ansx_status;

#define ANSX_SINGLE_STREAM 0xFFFFFFFFu /* opts.block_ints: one plain reference stream          */
#define ANSX_NO_CHECKPOINTS 0xFFFFFFFFu /* opts.ckpt_interval: no decoder restart points        */
/* Largest fidelity accepted.  The reference instantiates ANSfold<1..8> (methods.hpp:529-567) but is only
 * sound up to 7 (rfold<8> truncates symbols to u16, fold<8> can hit the u16 bail-out).  f <= 5
 * (up to 16384 symbol slots) keeps a block's model in a CU's LDS; f = 6, 7 (32 Ki / 64 Ki slots) run the same
 * stages with their per-block arrays in HBM -- correct and bit-identical, not tuned.  f = 8: ANSX_ERR_ARG. */
#define ANSX_MAX_FIDELITY 7
#define ANSX_DEFAULT_BLOCK_INTS 16384u
#define ANSX_DEFAULT_CKPT_INTERVAL 1024u

/* opts.flags.  ANSX_FLAG_COMPACT_ALPHABET: per-block alphabet compaction, the scheme of the reference's
 * src/pseudo_adaptive.cpp:85-130 -- every block is stored as u32 sigma | u32 universe | interpolative code of
 * the running sums of its sigma distinct values | the codec's stream of the block with each value replaced
 * by its 1-based rank among them (nothing when sigma == 1): byte for byte what that harness writes for the
 * block (it only measures sizes; decoding is this library's own).  For ANSX_FOLD, ANSX_MSB, ANSX_INT; blocks
 * of at most 16384 ints (ANSX_INT: 16380, default 8192); the sum of a block's distinct values must stay
 * below 2^32 - 1, as in the harness (ANSX_ERR_DOMAIN). */
#define ANSX_FLAG_COMPACT_ALPHABET 1u

typedef struct {
    uint32_t block_ints;    /* ints per independent reference stream; 0 = default             */
    uint32_t ckpt_interval; /* ints between decoder restart points (multiple of 4); 0 = default */
    uint32_t flags;         /* ANSX_FLAG_* bits                                                */
    uint32_t reserved;
} ansx_opts;

/* 64-byte container header (little endian), see DESIGN.md section 3. */
typedef struct {
    uint8_t magic[6];       /* "ANSXv3"                                                         */
    uint16_t max_present_m1; /* (max over blocks of the symbols PRESENT in the block) - 1: sizes the
                               decoder's per-present-symbol table (<= max_nsyms - 1; untrusted
                               like every other field: a block with more is a format error)     */
    uint32_t kind;          /* ansx_kind | 0x100 if ANSX_FLAG_COMPACT_ALPHABET | 0x200 if the restart
                               points are in the wide form (u32 cursor + 4 x u64 states: ANSint, frames
                               above 2^16, block streams of 16 MiB and more) instead of packed 29-byte
                               records (4 x 52-bit states + 24-bit cursor)                       */
    uint32_t fidelity;
    uint64_t n;             /* total ints                                                       */
    uint32_t block_ints;
    uint32_t ckpt_interval; /* 0 = none                                                         */
    uint32_t nblocks;
    uint32_t max_log2_frame; /* max over blocks of log2(M)                                      */
    uint32_t max_nsyms;      /* max over blocks of max_sym + 1                                  */
    uint32_t ckpts_per_block; /* restart points stored per block (fixed stride)                 */
    uint64_t payload_bytes;  /* sum of block stream sizes                                       */
    uint64_t payload_offset; /* byte offset of the first block stream                           */
} ansx_container_header;

typedef struct {
    char name[48];
    double total_ms;
    uint64_t launches;
} ansx_kernel_time;

/* Context: one per (process, device).  device < 0 -> current device. */
int ansx_init(int device, ansx_ctx** ctx);
void ansx_destroy(ansx_ctx* ctx);

const char* ansx_strerror(int status);
int ansx_last_hip_error(const ansx_ctx* ctx);
/* "ANSfold-<f>" / "ANSrfold-<f>" (methods.hpp:530-533,550-553); returns chars written. */
int ansx_codec_name(int kind, int fidelity, char* buf, size_t buflen);

/* Worst-case output bytes for n ints with these options (>= any actual output). */
size_t ansx_bound(int kind, int fidelity, size_t n, const ansx_opts* opts);

/* Host-buffer entry points (H2D + device path + D2H). */
int ansx_encode(ansx_ctx* ctx, int kind, int fidelity, const uint32_t* in, size_t n, uint8_t* out,
    size_t out_capacity, size_t* out_bytes, const ansx_opts* opts);
int ansx_decode(ansx_ctx* ctx, int kind, int fidelity, const uint8_t* in, size_t in_bytes,
    uint32_t* out, size_t n, const ansx_opts* opts);

/* Device-pointer entry points: in/out are HBM resident; `stream` is a hipStream_t (NULL = the
 * context's own stream, an ordinary blocking stream: it orders implicitly against work on the legacy
 * default stream -- e.g. PyTorch's default stream, whose handle is 0 -- but NOT against other
 * non-blocking streams; pass the stream the input was produced on if there is one).  They return
 * after the result size / status has been read back. */
int ansx_encode_dev(ansx_ctx* ctx, int kind, int fidelity, const uint32_t* d_in, size_t n,
    uint8_t* d_out, size_t out_capacity, size_t* out_bytes, const ansx_opts* opts, void* stream);
int ansx_decode_dev(ansx_ctx* ctx, int kind, int fidelity, const uint8_t* d_in, size_t in_bytes,
    uint32_t* d_out, size_t n, const ansx_opts* opts, void* stream);

/* Multi-GPU concatenation (the path shards by contiguous ranges of whole blocks, one container per
 * GPU; the reference is single-threaded and has no counterpart -- its per-block calls in
 * src/pseudo_adaptive.cpp:77-130 are the unit that is sharded).  d_parts[i] (8-byte aligned DEVICE
 * pointers, HOST array) are `nparts` <= 64 containers of the same codec / block_ints / ckpt_interval, in
 * list order, every one but the last holding whole blocks only; part_bytes[i] bounds each.  Writes ONE
 * container over all their blocks to d_out (16-byte aligned device memory, which must not overlap the
 * parts): index entries rebased, restart points and payload copied by one HIP kernel.  ansx_decode_dev
 * on the result returns the concatenated list.  The parts must agree on the restart-point format too (their
 * kind words are compared whole); a part that needed the wide form next to parts that did not is refused
 * (ANSX_ERR_FORMAT) -- encode such inputs with ANSX_WIDE_RESTART set on every rank. */
int ansx_merge_containers_dev(ansx_ctx* ctx, const uint8_t* const* d_parts, const size_t* part_bytes, int nparts,
    uint8_t* d_out, size_t out_capacity, size_t* out_bytes, void* stream);

/* The same over RCCL, for a C / C++ host with one process (or thread) per GPU: every rank calls this
 * with its own container; sizes are exchanged with ncclAllGather, every rank sends its container straight to the root
 * (grouped ncclSend / ncclRecv: one hop, each sender on its own xGMI link), and the root merges the slots with
 * ansx_merge_containers_dev.  nccl_comm: an ncclComm_t of the system's RCCL (librccl.so.1 is resolved at first use,
 * this library does not link it).  d_recv (root only): nranks * slot_bytes bytes, 16-byte aligned slots; a rank
 * container larger than slot_bytes fails the call on every rank (ANSX_ERR_CAPACITY).  *merged_bytes: size of the
 * merged container on the root, 0 elsewhere.  Blocks until the sizes are known; the transfers and the merge run on
 * `stream`.  slot_bytes is the ROOT's: it travels with the sizes, every rank checks every container against it and all
 * ranks return the same ANSX_ERR_CAPACITY together (a non-root rank's own argument is ignored).  A non-root rank returns
 * with its ncclSend queued on `stream`: d_container must stay untouched until that stream has been synchronised. */
int ansx_gather_containers(ansx_ctx* ctx, void* nccl_comm, int rank, int nranks, int root, const uint8_t* d_container,
    size_t bytes, uint8_t* d_recv, size_t slot_bytes, uint8_t* d_merged, size_t merged_cap, size_t* merged_bytes,
    void* stream);

/* Ranks of the communicator the context's most recent ansx_gather_containers call ran on, as RCCL itself reports them
 * (ncclCommCount); 0 before the first call.  For bench lines / logs that must show N GPUs really took part. */
int ansx_last_gather_ranks(const ansx_ctx* ctx);

/* Parse + validate a container header held in HOST memory. */
int ansx_container_info(const uint8_t* container, size_t bytes, ansx_container_header* out);

/* Per-kernel device timing (hipEvent pairs around every launch) for bench.py's roofline leg. */
int ansx_profile_enable(ansx_ctx* ctx, int on);
int ansx_profile_reset(ansx_ctx* ctx);
int ansx_profile_get(ansx_ctx* ctx, ansx_kernel_time* out, int max_entries, int* count);

/* Facts about the context's most recent ansx_encode / ansx_encode_dev call.
 *   near_threshold_decisions  frame-size stop-rule comparisons XH < 1.001 H (ans_util.hpp:149) whose two sides
 *                             agreed to 1e-12 relative.  The reference evaluates log2 with libm, this library
 *                             with its own portable log2 (<= 1 ulp apart): such a comparison is the only place
 *                             where the two could decide differently.  Expected to be 0, always.  Blocks with
 *                             such a comparison are decided again on the host with libm's log2 (the reference's
 *                             own arithmetic, ans_util.hpp:100-157), and if the host disagrees the call is
 *                             repeated with its decision forced: parity does not rest on this being 0.
 *   host_redecided            blocks whose frame size the host's re-decision changed (expected 0)
 *   path                      0 discovery (alphabet size read back mid-call), 1 launched back to back on the
 *                             context's hints for this geometry (largest alphabet; for ANSrfold and the
 *                             compaction layer also the most distinct values a block had, which sizes their
 *                             per-block hash tables), 2 the same with the fused model kernel; + 16: a hint
 *                             did not hold and the call was repeated on the discovery path; + 32: a frame above
 *                             2^16 turned up in a call laid out for packed restart points and the call was
 *                             repeated with wide ones (remembered per geometry, but only as the attempt to run
 *                             FIRST); + 64: the remembered wide form was not needed by this input and the call was
 *                             repeated with packed restart points; + 128: the producer / consumer encoder kernel
 *                             (k_encode_pc) ran; + 256: plain ANSint modelled in rank space (values of 16384 and more; remembered per geometry
 *                             as the attempt to run first, and given up again by a call whose values are small); + 512: that call was repeated with
 *                             the full-size arrays of the value-range prelude writer.  Either way the output bytes -- the restart-point
 *                             format included -- are a function of the input and the options only. */
typedef struct {
    uint32_t max_nsyms;
    uint32_t max_log2_frame;
    uint32_t near_threshold_decisions;
    uint32_t path;
    uint32_t host_redecided;
} ansx_encode_stats;
int ansx_last_encode_stats(const ansx_ctx* ctx, ansx_encode_stats* out);

/* Synthetic inputs of the reference's benchmark harness (src/generate_inputs.cpp:94-122, include/
 * zipf_dist.hpp:49-59) as counter-based generators: element i is a pure function of (seed, first_index + i),
 * so a list can be produced in pieces, on any number of GPUs, or on the host, with identical values.
 *   ANSX_GEN_UNIFORM    a = lo, b = hi (inclusive)            std::uniform_int_distribution
 *   ANSX_GEN_GEOMETRIC  a = p                                 std::geometric_distribution (failures before a success)
 *   ANSX_GEN_ZIPF       a = n (values 1..n), b = exponent q   zipf_distribution (rejection-inversion)
 * The distributions are the reference's; the random stream is not (std::mt19937 + libstdc++ + libm cannot be
 * reproduced bit for bit on a GPU).  ansx_generate_dev writes to device memory on `stream` (NULL = the
 * context's stream) and returns without synchronising; ansx_generate_host is the same function on the CPU. */
typedef enum { ANSX_GEN_UNIFORM = 0, ANSX_GEN_GEOMETRIC = 1, ANSX_GEN_ZIPF = 2 } ansx_gen_dist;
int ansx_generate_dev(ansx_ctx* ctx, int dist, double a, double b, uint64_t seed, uint64_t first_index,
    uint32_t* d_out, size_t n, void* stream);
int ansx_generate_host(int dist, double a, double b, uint64_t seed, uint64_t first_index, uint32_t* out, size_t n);

/* Test / experiment hook: select one of the equivalent internal code paths (all must produce identical
 * bytes).  Names are those of the environment variables read once by ansx_init: ANSX_DECODE_MODE
 * ("ring" | "staged" | ""), ANSX_DECODE_TABLE, ANSX_NO_STREAM_LDS, ANSX_PARSE_GENERIC, ANSX_PARSE_WIN, ANSX_PARSE_FAST,
 * ANSX_PARSE_STAGE_WORDS (number), ANSX_ENCODE_GTAB16, ANSX_TEST_TABLE16_FIXUP, ANSX_MODEL_FUSED, ANSX_MODEL_SYNC,
 * ANSX_NS_HINT (number: alphabet-size hint for every call instead of the per-geometry one the context
 * learns; too small a value only costs a repeat on the general path), ANSX_T_HINT (number: candidate frame sizes per
 * block of the fast model path), ANSX_NO_FAST_MODEL, ANSX_FAST_GUARD / ANSX_NEAR_BAND (numbers: relative bands around
 * the stop-rule threshold inside which the fast path repeats on the exact one / the host re-decides),
 * ANSX_TEST_NEAR_FLIP (the device decides close calls the wrong way), ANSX_CAND_CHAINS (1 | 2), ANSX_WIDE_RESTART
 * (wide restart points in every container -- the one switch here that changes the output: the index, not the block
 * streams), ANSX_TEST_WIDE_AT (number <= 16: frames above 2^this count as too large for packed restart points);
 * through this call only (round 4): ANSX_NO_PC / ANSX_FORCE_PC / ANSX_NO_PC_AUTO / ANSX_PC_B_PAIRS (the producer /
 * consumer encoder never / whatever the list length / only on request; pairs per workgroup of its two-round shape),
 * ANSX_ENCODE_MODE2, ANSX_DECODE_PAIR, ANSX_DECODE_SMALL_RING (1 never | 2 always), ANSX_FORGET_HINTS, ANSX_NO_BIG_GEO,
 * ANSX_FIN_ONE_WAVE, ANSX_TEST_SP_BITS (number: words of the rank-space ANSint prelude writer's bit buffer on its first
 * attempt)  (flags: "1" on, "0"/""/NULL off).  Unknown name: ANSX_ERR_ARG. */
int ansx_debug_set(ansx_ctx* ctx, const char* name, const char* value);

/* Bytes of device workspace currently held by the context. */
size_t ansx_workspace_bytes(const ansx_ctx* ctx);

/* One pass of the Zipf generator's rejection loop for a given canonical uniform u01 in [0, 1] (unit tests: the map
 * uniform -> value is compared with include/zipf_dist.hpp:49-59 driven by the same uniforms): candidate value *k and
 * whether it is accepted.  Host only, no device needed. */
int ansx_zipf_from_uniform(double n, double q, double u01, uint32_t* k, int* accepted);

/* Host evaluation of the portable log2 used by the device normaliser (unit tests only). */
double ansx_host_log2(double x);
/* The same function evaluated on the DEVICE for n inputs (host arrays): the normaliser relies on
 * host and device results being bit-identical. */
int ansx_selftest_log2(ansx_ctx* ctx, const double* in, double* out, size_t n);
/* The normaliser's division helper (exact for integer-valued operands below 2^31, see
 * csrc/ansx_dev.h) evaluated on the device: out[i] = a[i] / b[i], to be compared with IEEE division. */
int ansx_selftest_div(ansx_ctx* ctx, const double* a, const double* b, double* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ANSX_H */

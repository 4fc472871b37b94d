/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (clean-room, plain C) of the reference's
 * ANSfold<f> / ANSrfold<f> encode+decode path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (ans_large_alphabet_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement byte-for-byte against
 *   (a) oracle/_ref/libans_ref.so = the unmodified reference headers compiled from /root/reference
 *       (present in the authoring container and, prebuilt, on the GPU box), and
 *   (b) tests/golden/ fixtures generated from (a) by tests/golden/make_golden.py,
 *   (c) the canonical known answers of SURVEY.md section 8c.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef ANS_ORACLE_H
#define ANS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ANS_ORACLE_FOLD = 0, ANS_ORACLE_RFOLD = 1, ANS_ORACLE_MSB_KIND = 2 /* ANSmsb, include/ans_msb.hpp; f ignored */,
    ANS_ORACLE_INT_KIND = 3 /* ANSint, include/ans_int.hpp (methods.hpp:484-497); f ignored; inputs with max value < 16384 or <= n + 1024 */ };

typedef struct {
    uint32_t max_sym;       /* largest folded symbol in the block                         */
    uint32_t log2_frame;    /* log2(M)                                                     */
    uint32_t header_bytes;  /* rfold: 4 or 4+4T; fold: 0                                   */
    uint32_t prelude_bytes; /* vbyte + 1 + interp words (excludes header_bytes)            */
    uint32_t interp_bits;   /* number of valid bits in the interpolative code              */
    uint32_t reorder_flag;  /* rfold: 0/1                                                  */
    uint64_t sigma;         /* rfold: number of distinct input values; fold: #nonzero syms */
    uint64_t final_states[4];
    uint32_t present_syms;  /* symbols with a non-zero frequency in the block's model (every codec) */
    uint32_t reserved;
} ans_oracle_info;

/* include/ans_fold.hpp:38-65 — folded symbol and number of exception bytes */
uint32_t ans_oracle_fold(uint32_t f, uint32_t x, uint32_t* nbytes);
/* include/ans_fold.hpp:150-175 */
uint32_t ans_oracle_unfold(uint32_t f, uint32_t sym, uint32_t* nbytes);

/* include/ans_util.hpp:100-157 + util.hpp:271-298.  freqs[nfreqs]; writes scaled[largest_sym+1];
 * returns frame size M (sum of scaled). */
uint64_t ans_oracle_adjust_freqs(const uint64_t* freqs, size_t nfreqs, uint32_t largest_sym,
    uint32_t* scaled);

/* include/ans_util.hpp:46-63.  Returns bytes written; *valid_bits = bits of interp code. */
size_t ans_oracle_write_prelude(const uint32_t* nfreqs, size_t nsyms, uint64_t frame_size,
    uint8_t* out, uint32_t* valid_bits);
/* include/ans_util.hpp:25-42.  Returns nsyms; *frame_log2 receives log2(M). */
size_t ans_oracle_read_prelude(const uint8_t* in, uint32_t* nfreqs, uint32_t* frame_log2);

/* include/ans_fold.hpp:238-281 (kind 0) / ans_reorder_fold.hpp:312-355 (kind 1),
 * i.e. methods.hpp:535-540 / 555-560.  Returns bytes written (0 on error).
 * If ckpt_interval (multiple of 4) is non-zero, records a decoder restart point at every symbol
 * index i = s*ckpt_interval (s >= 1, i < n - n%4... i.e. wherever a full 4-group starts):
 * ckpt_states[4*(s-1)+j] = state j, ckpt_off[s-1] = stream byte offset at that moment.
 * Returns number of checkpoints through *n_ckpt. */
size_t ans_oracle_encode(int kind, uint32_t f, const uint32_t* in, size_t n, uint8_t* out,
    size_t cap, ans_oracle_info* info, size_t ckpt_interval, uint64_t* ckpt_states,
    uint32_t* ckpt_off, size_t* n_ckpt);

/* include/ans_fold.hpp:283-311 / ans_reorder_fold.hpp:357-385 (methods.hpp:541-546 / 561-566).
 * ref_f3_compat != 0 reproduces the reference's rfold decode defect when flag == 0;
 * 0 decodes correctly.  Returns 0 on success. */
int ans_oracle_decode(int kind, uint32_t f, const uint8_t* in, size_t nbytes, uint32_t* out,
    size_t n, int ref_f3_compat);

/* include/ans_util.hpp:100-157 with require_u16 as given (false = ANSint, ans_int.hpp:50). */
uint64_t ans_oracle_adjust_freqs_ex(const uint64_t* freqs, size_t nfreqs, uint32_t largest_sym,
    uint32_t* scaled, int require_u16);

/* One block of src/pseudo_adaptive.cpp:85-130 (per-block alphabet compaction): alphabet header + codec
 * stream of the block remapped to 1-based ranks; kind 0 (ANSfold<f>), 2 (ANSmsb) or 3 (ANSint). */
typedef struct {
    uint32_t sigma;        /* distinct values in the block                         */
    uint32_t universe;     /* sum of the distinct values + 1 (u32)                 */
    uint32_t header_bytes; /* 8 + interpolative words                              */
    uint32_t interp_bits;  /* valid bits of the header's interpolative code        */
} ans_oracle_pa_info;
size_t ans_oracle_pa_encode(int kind, uint32_t f, const uint32_t* in, size_t n, uint8_t* out, size_t cap,
    ans_oracle_pa_info* pinfo, ans_oracle_info* info, size_t ckpt_interval, uint64_t* ckpt_states,
    uint32_t* ckpt_off, size_t* n_ckpt);
/* Inverse of ans_oracle_pa_encode (the reference harness never decodes: this build's own). */
int ans_oracle_pa_decode(int kind, uint32_t f, const uint8_t* in, size_t nbytes, uint32_t* out, size_t n);

/* Parse hints of a codec prelude for this build's container index (8 words, see ans_oracle.c). */
void ans_oracle_prelude_hints(const uint8_t* prelude, uint32_t* hints);

/* Every block of a list at once, on `threads` host threads (full-size parity: tests compare EVERY block of a 256 Mi-int
 * container with this, not a sample).  Block b = in[b * block_ints ...) is one ans_oracle_encode call with restart points
 * every ckpt_interval ints (0 = none).  Per block: sizes[b] = stream bytes, stream_hash[b] = ans_oracle_hash of the stream,
 * ckpt_digest[b] = sum over restart points s and states j of state * (2654435761 + 2 (4 s + j)) + offset * (40503 + 2 s),
 * modulo 2^64 (what a test recomputes from a parsed container with numpy), max_log2_frame / max_nsyms = the container
 * header's fields.  Returns 0, or -1 if a block failed. */
int ans_oracle_blocks_digest(int kind, uint32_t f, const uint32_t* in, size_t n, size_t block_ints, size_t ckpt_interval,
    int threads, uint32_t* sizes, uint64_t* stream_hash, uint64_t* ckpt_digest, uint32_t* max_log2_frame, uint32_t* max_nsyms);
/* FNV-1a (64 bit) over 8-byte little-endian words of the span, the tail bytes one by one */
uint64_t ans_oracle_hash(const uint8_t* p, size_t n);
/* the same over nspans spans [offs[i], offs[i + 1]) of buf, on `threads` host threads */
void ans_oracle_hash_spans(const uint8_t* buf, const uint64_t* offs, size_t nspans, int threads, uint64_t* out);

/* Worst-case stream size for one encode() call. */
size_t ans_oracle_bound(int kind, uint32_t f, size_t n);
/* ... of ANSint (kind 3) when the list's largest value is known (any value below 2^30) */
size_t ans_oracle_bound_int(size_t n, uint32_t max_value);

#ifdef __cplusplus
}
#endif
#endif

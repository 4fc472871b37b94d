// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
//
// Thin C-ABI shim around the *unmodified* reference headers, compiled where they lie under
// /root/reference/include (nothing is copied into this repository).  The resulting
// oracle/_ref/libans_ref.so is used to (1) validate the clean-room restatement in
// oracle/ans_oracle.c, (2) generate tests/golden/ fixtures, (3) optionally serve as the
// "reference" CPU baseline in bench.py.
//
// Include order matters: ans_fold.hpp uses constants::K / RADIX / RADIX_LOG2, which
// only exist once ans_byte.hpp has been seen (that is what methods.hpp:29 does).
// The wrappers below restate methods.hpp:529-567 (ANSfold<f>/ANSrfold<f>::encode/decode) — those
// three-line forwarding functions cannot be included directly because methods.hpp also pulls in
// un-vendored third-party headers (FastPFor, streamvbyte, FSE).
//
// Build: see oracle/Makefile (clang++ -ftrivial-auto-var-init=zero => canonical zero padding for
// the indeterminate bits of the last interpolative word).

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "ans_byte.hpp"
#include "ans_fold.hpp"
#include "ans_reorder_fold.hpp"
#include "ans_msb.hpp"
#include "ans_int.hpp"
#include "zipf_dist.hpp"
#include "qsufsort.hpp"  // Larsson-Sadakane suffix sort, unmodified (src/generate_bwtmtf.cpp:34)
#include <deque>

namespace {

template <uint32_t f>
size_t fold_enc(const uint32_t* in, size_t n, uint8_t* out, size_t cap)
{
    return ans_fold_compress<f>(out, cap, in, n); // methods.hpp:535-540
}
template <uint32_t f>
void fold_dec(const uint8_t* in, size_t nbytes, uint32_t* out, size_t n)
{
    ans_fold_decompress<f>(out, n, in, nbytes); // methods.hpp:541-546
}
template <uint32_t f>
size_t rfold_enc(const uint32_t* in, size_t n, uint8_t* out, size_t cap)
{
    return ans_reorder_fold_compress<f>(out, cap, in, n); // methods.hpp:555-560
}
template <uint32_t f>
void rfold_dec(const uint8_t* in, size_t nbytes, uint32_t* out, size_t n)
{
    ans_reorder_fold_decompress<f>(out, n, in, nbytes); // methods.hpp:561-566
}

} // namespace

#define DISPATCH_F(fn, ...)                                                    \
    switch (f) {                                                               \
    case 1: return fn<1>(__VA_ARGS__);                                         \
    case 2: return fn<2>(__VA_ARGS__);                                         \
    case 3: return fn<3>(__VA_ARGS__);                                         \
    case 4: return fn<4>(__VA_ARGS__);                                         \
    case 5: return fn<5>(__VA_ARGS__);                                         \
    case 6: return fn<6>(__VA_ARGS__);                                         \
    case 7: return fn<7>(__VA_ARGS__);                                         \
    default: break;                                                            \
    }

extern "C" {

// kind: 0 = ANSfold<f>, 1 = ANSrfold<f>, 2 = ANSmsb (methods.hpp:499-515), 3 = ANSint (methods.hpp:484-497)
size_t ref_encode(int kind, int f, const uint32_t* in, size_t n, uint8_t* out, size_t cap)
{
    if (kind == 3) return ans_int_compress(out, cap, in, n);
    if (kind == 2) return ans_msb_compress(out, cap, in, n);
    if (kind == 0) {
        DISPATCH_F(fold_enc, in, n, out, cap)
    } else {
        DISPATCH_F(rfold_enc, in, n, out, cap)
    }
    return 0;
}

void ref_decode(int kind, int f, const uint8_t* in, size_t nbytes, uint32_t* out, size_t n)
{
    if (kind == 3) {
        ans_int_decompress(out, n, in, nbytes);
        return;
    }
    if (kind == 2) {
        ans_msb_decompress(out, n, in, nbytes);
        return;
    }
    if (kind == 0) {
        DISPATCH_F(fold_dec, in, nbytes, out, n)
    } else {
        DISPATCH_F(rfold_dec, in, nbytes, out, n)
    }
}

// adjust_freqs(freqs, largest_sym, require_u16=true) — ans_util.hpp:100-157.
// freqs has nfreqs entries; writes largest_sym+1 normalised freqs; returns the frame size M.
uint64_t ref_adjust_freqs(const uint64_t* freqs, size_t nfreqs, uint32_t largest_sym, uint32_t* scaled_out)
{
    std::vector<uint64_t> F(freqs, freqs + nfreqs);
    auto S = adjust_freqs(F, largest_sym, true);
    uint64_t M = 0;
    for (size_t i = 0; i < S.size(); i++) {
        scaled_out[i] = S[i];
        M += S[i];
    }
    return M;
}

// ans_serialize_interp — ans_util.hpp:46-63.  Returns total prelude bytes (vbyte + 1 + interp words).
size_t ref_serialize_prelude(const uint32_t* nfreqs, size_t nsyms, uint64_t frame_size, uint8_t* out)
{
    std::vector<uint32_t> v(nfreqs, nfreqs + nsyms);
    uint8_t* p = out;
    ans_serialize_interp(v, frame_size, p);
    return size_t(p - out);
}

// ans_load_interp — ans_util.hpp:25-42.  Returns number of symbols (max_sym+1).
size_t ref_load_prelude(const uint8_t* in, uint32_t* nfreqs_out)
{
    auto v = ans_load_interp(in);
    for (size_t i = 0; i < v.size(); i++) nfreqs_out[i] = v[i];
    return v.size();
}

// One block of src/pseudo_adaptive.cpp's run<t_compressor>() (:85-130): the bytes that harness writes for the
// block at enc_ptr -- u32 alphabet size, u32 universe, interpolative code of the running sums of the
// block's distinct values (:106-113), then t_compressor::encode of the block remapped to 1-based ranks
// (:91-103,115-123; skipped when the block has one distinct value).  The statements are the harness's,
// re-typed around the UNMODIFIED interpolative_internal::encode and codec functions (the harness itself
// needs Boost and cannot be built here).  Returns total bytes; *hdr_bytes = 8 + interpolative bytes.
size_t ref_pa_encode(int kind, int f, const uint32_t* in_ptr, size_t block_size, uint8_t* enc_ptr, size_t enc_size,
    size_t* hdr_bytes)
{
    uint32_t max_sym = *std::max_element(in_ptr, in_ptr + block_size);
    std::vector<uint32_t> remapped_block_data(block_size);
    std::vector<uint32_t> remapped_alphabet(size_t(max_sym) + 1, 0);
    std::vector<uint32_t> block_alphabet;
    for (size_t k = 0; k < block_size; k++) remapped_alphabet[in_ptr[k]] = 1;
    if (remapped_alphabet[0] == 1) block_alphabet.push_back(0);
    for (size_t k = 1; k <= max_sym; k++) {
        if (remapped_alphabet[k] == 1) block_alphabet.push_back(k);
        remapped_alphabet[k] += remapped_alphabet[k - 1];
    }
    for (size_t k = 0; k < block_size; k++) remapped_block_data[k] = remapped_alphabet[in_ptr[k]];
    for (size_t k = 1; k < block_alphabet.size(); k++) block_alphabet[k] += block_alphabet[k - 1];
    uint32_t* enc_ptr_u32 = (uint32_t*)enc_ptr;
    *enc_ptr_u32++ = block_alphabet.size();
    *enc_ptr_u32++ = block_alphabet.back() + 1;
    auto bytes_written = interpolative_internal::encode(
        enc_ptr_u32, block_alphabet.data(), block_alphabet.size(), block_alphabet.back() + 1);
    enc_ptr += (bytes_written + 8);
    enc_size -= (bytes_written + 8);
    size_t total = bytes_written + 8;
    if (hdr_bytes) *hdr_bytes = total;
    if (block_alphabet.size() != 1) total += ref_encode(kind, f, remapped_block_data.data(), block_size, enc_ptr, enc_size);
    return total;
}

uint32_t ref_fold_mapping(int f, uint32_t x)
{
    DISPATCH_F(ans_fold_mapping, x)
    return 0;
}

uint32_t ref_fold_undo_mapping(int f, uint32_t s)
{
    DISPATCH_F(ans_fold_undo_mapping, s)
    return 0;
}

uint32_t ref_fold_exception_bytes(int f, uint32_t s)
{
    DISPATCH_F(ans_fold_exception_bytes, s)
    return 0;
}


// include/zipf_dist.hpp:49-59 driven by std::mt19937(seed) exactly as src/generate_inputs.cpp:63-79 does, with the
// uniforms it consumed recorded: libstdc++'s uniform_real_distribution draws generate_canonical<double, 53> and
// maps it to [H(x1), H(n)); a clone of the generator is replayed until it catches up, so u01[] holds, per output
// value, every canonical uniform of its rejection loop (the last one is the accepted draw).
// Returns the number of uniforms written (stops early when cap would be exceeded; *n_values = values completed).
size_t ref_zipf_trace(uint32_t n, double q, uint32_t seed, size_t count, uint32_t* values, uint32_t* ndraws, double* u01,
    size_t cap, size_t* n_values)
{
    std::mt19937 g(seed), g2(seed);
    zipf_distribution<uint32_t> z(n, q);
    size_t nu = 0, i = 0;
    for (; i < count; i++) {
        const uint32_t v = z(g);
        uint32_t k = 0;
        const size_t start = nu;
        do {
            if (nu >= cap) {
                *n_values = i;
                return start;
            }
            u01[nu++] = std::generate_canonical<double, 53>(g2);
            k++;
        } while (!(g2 == g));
        values[i] = v;
        ndraws[i] = k;
    }
    *n_values = i;
    return nu;
}


// All-cores CPU row for bench.py (SURVEY 8d "optional all-cores row"): blocks are independent encode() calls, so
// `threads` std::threads take blocks of block_ints ints from a shared counter -- first every block is encoded
// (timed), then every block decoded (timed) and compared.  Returns 0 when every block round-trips.
int ref_blocks_mt(int kind, int f, const uint32_t* in, size_t n, size_t block_ints, int threads, double* enc_seconds,
    double* dec_seconds, size_t* total_bytes)
{
    const size_t nb = (n + block_ints - 1) / block_ints;
    std::vector<std::vector<uint8_t>> streams(nb);
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto run = [&](bool enc) {
        next = 0;
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, enc] {
                std::vector<uint32_t> back;
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= nb) break;
                    const size_t lo = b * block_ints, cnt = std::min(block_ints, n - lo);
                    if (enc) {
                        streams[b].resize(cnt * 8 + 65536 + 4 * 16400);
                        streams[b].resize(ref_encode(kind, f, in + lo, cnt, streams[b].data(), streams[b].size()));
                    } else {
                        back.resize(cnt);
                        ref_decode(kind, f, streams[b].data(), streams[b].size(), back.data(), cnt);
                        if (memcmp(back.data(), in + lo, cnt * 4) != 0) bad = 1;
                    }
                }
            });
        for (auto& t : th) t.join();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    *enc_seconds = run(true);
    *dec_seconds = run(false);
    size_t tot = 0;
    for (const auto& s : streams) tot += s.size();
    *total_bytes = tot;
    return bad.load();
}

// src/generate_bwtmtf.cpp:142-173 around the UNMODIFIED include/qsufsort.hpp: suffix array of the parsed text T (n1 ints,
// the last one the 0 terminator word_parse / byte_parse append), BWT, move-to-front ranks.  That TU needs Boost (its
// option parser and boost::split) and cannot be built here, so its statements are re-typed, as ref_pa_encode does for
// pseudo_adaptive.cpp; the suffix sort is the reference's own.  Writes min(n1 - 1, n) ranks, returns how many.
size_t ref_bwtmtf(const int32_t* T_in, size_t n1, size_t n, uint32_t* mtf_out)
{
    std::vector<int> T(T_in, T_in + n1);
    std::vector<int> text = T;
    std::vector<int> SA(T.size());
    const auto [min, max] = std::minmax_element(T.begin(), T.end() - 1);
    const int max_sym = *max;                                              // (suffixsort overwrites T)
    suffixsort(T.data(), SA.data(), T.size() - 1, *max + 1, *min);         // :148
    std::vector<int> BWT(T.size());
    for (size_t i = 0; i < text.size(); i++)                               // :152-156
        BWT[i] = SA[i] != 0 ? text[SA[i] - 1] : text.back();
    size_t seq_len = text.size() - 1;                                      // :158-160
    if (seq_len > n) seq_len = n;
    std::deque<int> alphabet;                                              // :163-171, get_mtf_rank :111-118
    for (size_t i = 0; i <= (size_t)max_sym; i++) alphabet.push_back(i);
    for (size_t i = 0; i < seq_len; i++) {
        auto sym = BWT[i];
        auto itr = std::find(std::begin(alphabet), std::end(alphabet), sym);
        mtf_out[i] = (uint32_t)std::distance(std::begin(alphabet), itr);
        alphabet.erase(itr);
        alphabet.push_front(sym);
    }
    return seq_len;
}

} // extern "C"

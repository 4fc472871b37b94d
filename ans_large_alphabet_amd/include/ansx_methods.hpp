// C++17 host mirror of the reference's codec concept for the ANSfold / ANSrfold path.
//
// The reference registers codecs as structs with three static members
// (/root/reference/include/methods.hpp:529-567):
//
//     static std::string name();
//     static size_t encode(const uint32_t* in_ptr, size_t in_size_u32,
//                          uint8_t* out_ptr, size_t out_size_u8, uint8_t* buf = NULL);
//     static void   decode(const uint8_t* in_ptr, size_t in_size_u8,
//                          uint32_t* out_ptr, size_t out_size_u32, uint8_t* buf = NULL);
//
// and consumes them purely as template arguments: run<t_compressor>(inputs)
// (src/table_efficiency.cpp:64-65,176-179; benchmark.cpp:179-192; fold_effectiveness.cpp:132-148).
// ANSfoldGPU<f> / ANSrfoldGPU<f> below have exactly those signatures, names and argument
// meanings, so they drop into any such harness:  run<ANSfoldGPU<1>>(inputs);
//
// Differences, all forced by the boundary:
//  * encode() writes the ansx container (include/ansx.h): one unmodified reference stream per
//    block behind a small index.  ANSfoldGPUStream<f> / ANSrfoldGPUStream<f> emit exactly one
//    plain reference stream instead (bit-identical to ANSfold<f>::encode on the whole list).
//  * the reference never checks out_size_u8 and has no error path (malformed input is UB); here
//    a failure throws std::runtime_error with the ansx status text, mirroring quit()'s
//    "print and stop" (include/util.hpp:101-115) without killing the process.
//  * `buf` is accepted and ignored, as in the reference (ans_fold.hpp never touches it).
//
// Link with -lansx (ans_large_alphabet_amd/libansx.so).  Not thread-safe per context: the
// process-wide context below is guarded by a mutex.
#pragma once

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../include/ansx.h"

namespace ansx {

struct Runtime {
    ansx_ctx* ctx = nullptr;
    std::mutex mu;
    static Runtime& get()
    {
        static Runtime r;
        return r;
    }
    ansx_ctx* context()
    {
        if (!ctx) {
            int st = ansx_init(-1, &ctx);
            if (st != ANSX_OK) throw std::runtime_error(std::string("ansx_init: ") + ansx_strerror(st));
        }
        return ctx;
    }
    ~Runtime()
    {
        if (ctx) ansx_destroy(ctx);
    }
};

template <int KIND, uint32_t fidelity, uint32_t BLOCK_INTS, uint32_t FLAGS = 0u> struct Codec {
    static std::string name()  // methods.hpp:530-533 / 550-553 / 500 / 485
    {
        if (KIND == ANSX_MSB) return "ANSmsb";
        if (KIND == ANSX_INT) return "ANS";
        return std::string(KIND == ANSX_RFOLD ? "ANSrfold-" : "ANSfold-") + std::to_string(fidelity);
    }
    static ansx_opts opts()
    {
        ansx_opts o;
        o.block_ints = BLOCK_INTS;
        o.ckpt_interval = 0;
        o.flags = FLAGS;
        o.reserved = 0;
        return o;
    }
    static size_t encode(const uint32_t* in_ptr, size_t in_size_u32, uint8_t* out_ptr, size_t out_size_u8,
        uint8_t* buf = NULL)
    {
        (void)buf;
        Runtime& R = Runtime::get();
        std::lock_guard<std::mutex> lock(R.mu);
        ansx_opts o = opts();
        size_t written = 0;
        int st = ansx_encode(R.context(), KIND, (int)fidelity, in_ptr, in_size_u32, out_ptr, out_size_u8,
            &written, &o);
        if (st != ANSX_OK) throw std::runtime_error(name() + " encode: " + ansx_strerror(st));
        return written;
    }
    static void decode(const uint8_t* in_ptr, size_t in_size_u8, uint32_t* out_ptr, size_t out_size_u32,
        uint8_t* buf = NULL)
    {
        (void)buf;
        Runtime& R = Runtime::get();
        std::lock_guard<std::mutex> lock(R.mu);
        ansx_opts o = opts();
        int st = ansx_decode(R.context(), KIND, (int)fidelity, in_ptr, in_size_u8, out_ptr, out_size_u32, &o);
        if (st != ANSX_OK) throw std::runtime_error(name() + " decode: " + ansx_strerror(st));
    }
};

}  // namespace ansx

// drop-in names (block container, library default block size)
template <uint32_t fidelity> using ANSfoldGPU = ansx::Codec<ANSX_FOLD, fidelity, 0u>;
template <uint32_t fidelity> using ANSrfoldGPU = ansx::Codec<ANSX_RFOLD, fidelity, 0u>;
// ANSmsb (methods.hpp:499-515): the fixed-threshold MSB fold, same kernels, no fidelity
using ANSmsbGPU = ansx::Codec<ANSX_MSB, 0u, 0u>;
using ANSmsbGPUStream = ansx::Codec<ANSX_MSB, 0u, ANSX_SINGLE_STREAM>;
// ANSint (methods.hpp:484-497): 32-bit frequencies over the values themselves -- on the GPU through the
// per-block alphabet compaction of src/pseudo_adaptive.cpp:85-130 (every block = alphabet header + ANSint
// stream of the block's 1-based ranks, as that harness writes it), which is also available for the others
using ANSintGPU = ansx::Codec<ANSX_INT, 0u, 0u, ANSX_FLAG_COMPACT_ALPHABET>;
// ANSint on the values themselves (any below 2^30; from 16384 on: blocks / single-stream lists with at most 16384 distinct values): block container, or exactly the bytes of ANSint::encode
using ANSintGPUPlain = ansx::Codec<ANSX_INT, 0u, 0u>;
using ANSintGPUStream = ansx::Codec<ANSX_INT, 0u, ANSX_SINGLE_STREAM>;
using ANSmsbGPUCompact = ansx::Codec<ANSX_MSB, 0u, 0u, ANSX_FLAG_COMPACT_ALPHABET>;
template <uint32_t fidelity> using ANSfoldGPUCompact = ansx::Codec<ANSX_FOLD, fidelity, 0u, ANSX_FLAG_COMPACT_ALPHABET>;
// exactly one reference stream, byte-identical to ANSfold<f>::encode / ANSrfold<f>::encode
template <uint32_t fidelity> using ANSfoldGPUStream = ansx::Codec<ANSX_FOLD, fidelity, ANSX_SINGLE_STREAM>;
template <uint32_t fidelity> using ANSrfoldGPUStream = ansx::Codec<ANSX_RFOLD, fidelity, ANSX_SINGLE_STREAM>;

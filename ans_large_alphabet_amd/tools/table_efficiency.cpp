// table_efficiency (ansx edition): the reference's Table-10 harness for the ANSfold / ANSrfold
// rows, driven through the drop-in codec structs of include/ansx_methods.hpp.
//
// Behaviour follows /root/reference/src/table_efficiency.cpp for this path:
//   -h/--help, -t/--text, -i/--input <dir|file>           (:34-62; no Boost here)
//   files matching .*\.u32 (or .*\.txt with -t), or a single file, sorted by name (:128-157)
//   text = one decimal per line, binary = raw little-endian uint32 (util.hpp:160-192)
//   per codec: output buffers of n*8 bytes (:73-74), NUM_RUNS = 5 timed encodes and decodes,
//   minimum kept (:32,78-101), round trip verified like REQUIRE_EQUAL (cutil.hpp:30-50),
//   rows printed as "\method{name}  &" then "%15.4f  &  %15.4f" ints/s per file (:67,112-120)
//   codecs: ANSfold-1, ANSfold-5, ANSrfold-1, ANSrfold-5 (:176-179)
//
// Extra: --stream times the single-reference-stream variants instead of the block container;
//        --bits also prints bits/int (the by-product table_effectiveness.cpp reports).
//
// Build: make -C ans_large_alphabet_amd/tools   (g++ -std=c++17, links libansx.so)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <limits>
#include <string>
#include <vector>

#include "../include/ansx_methods.hpp"
#include "harness_common.hpp"

static const int NUM_RUNS = 5;
static bool g_bits = false;

template <class t_compressor> void run(std::vector<std::vector<uint32_t>>& inputs)
{
    printf("\\method{%s}  &\n", t_compressor::name().c_str());
    std::vector<double> enc_speed, dec_speed, bpi;
    for (const auto& input : inputs) {
        std::vector<uint8_t> encoded_data(input.size() * 8 + 4096);
        std::vector<uint8_t> tmp_buf(16);
        size_t encoded_bytes = 0;
        size_t enc_min = std::numeric_limits<size_t>::max();
        for (int i = 0; i < NUM_RUNS; i++) {
            auto t0 = std::chrono::high_resolution_clock::now();
            encoded_bytes = t_compressor::encode(input.data(), input.size(), encoded_data.data(),
                encoded_data.size(), tmp_buf.data());
            auto t1 = std::chrono::high_resolution_clock::now();
            enc_min = std::min((size_t)(t1 - t0).count(), enc_min);
        }
        encoded_data.resize(encoded_bytes);
        std::vector<uint32_t> recover(input.size());
        size_t dec_min = std::numeric_limits<size_t>::max();
        for (int i = 0; i < NUM_RUNS; i++) {
            auto t0 = std::chrono::high_resolution_clock::now();
            t_compressor::decode(encoded_data.data(), encoded_data.size(), recover.data(), recover.size(),
                tmp_buf.data());
            auto t1 = std::chrono::high_resolution_clock::now();
            dec_min = std::min((size_t)(t1 - t0).count(), dec_min);
        }
        require_equal(input.data(), recover.data(), input.size(), t_compressor::name());
        enc_speed.push_back(double(input.size()) / (double(enc_min) / 1e9));  // util.hpp:307-311
        dec_speed.push_back(double(input.size()) / (double(dec_min) / 1e9));
        bpi.push_back(8.0 * double(encoded_bytes) / double(input.size()));
    }
    for (size_t i = 0; i < enc_speed.size(); i++) {
        for (size_t j = 0; j < i * 4; j++) printf(" ");
        printf("%15.4f  &  %15.4f  ", enc_speed[i], dec_speed[i]);
        if (g_bits) printf("%% %.4f bits/int  ", bpi[i]);
        if (i + 1 == enc_speed.size()) printf("\\\\ \n\n");
        else printf("&\n");
    }
}

int main(int argc, char const* argv[])
{
    std::string input;
    bool text = false, stream = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            printf("Allowed options:\n  -h [ --help ]   produce help message\n  -t [ --text ]   text input "
                   "(default is uint32_t binary)\n  -i [ --input ] arg  the input dir\n  --stream        one "
                   "reference stream per file instead of the block container\n  --bits          also print "
                   "bits/int\n");
            return EXIT_SUCCESS;
        } else if (a == "-t" || a == "--text") {
            text = true;
        } else if ((a == "-i" || a == "--input") && i + 1 < argc) {
            input = argv[++i];
        } else if (a == "--stream") {
            stream = true;
        } else if (a == "--bits") {
            g_bits = true;
        } else {
            fprintf(stderr, "Error parsing cmdargs: unknown option %s\n", a.c_str());
            return EXIT_FAILURE;
        }
    }
    if (input.empty()) {
        fprintf(stderr, "Missing required option: --input\n");
        return EXIT_FAILURE;
    }
    std::vector<std::vector<uint32_t>> inputs = load_inputs(input, text);
    try {
        if (!stream) {
            run<ANSfoldGPU<1>>(inputs);
            run<ANSfoldGPU<5>>(inputs);
            run<ANSrfoldGPU<1>>(inputs);
            run<ANSrfoldGPU<5>>(inputs);
            run<ANSmsbGPU>(inputs);  // the reference runs ANSmsb in benchmark.cpp:174 / table_effectiveness.cpp:146
        } else {
            run<ANSfoldGPUStream<1>>(inputs);
            run<ANSfoldGPUStream<5>>(inputs);
        }
    } catch (const std::exception& e) {
        quit(e.what());
    }
    return EXIT_SUCCESS;
}

// TEST INFRASTRUCTURE ONLY (lives under oracle/, never part of the package).
//
// The CPU rows of the reference's Table-10 harness (src/table_efficiency.cpp:64-121,176-179) for the codecs this
// build covers, produced by the REFERENCE ITSELF: the codec functions come from oracle/_ref/libans_ref.so, i.e. the
// unmodified reference headers compiled by oracle/Makefile.  The reference's own table_efficiency.cpp cannot be
// built here (Boost, un-vendored submodules); this file restates its run<>() loop -- output buffers of n * 8 bytes,
// NUM_RUNS = 5 timed encodes and decodes, minimum kept, round trip verified, rows printed as
// "\method{name}  &" then "%15.4f  &  %15.4f" ints/s per file -- and its input handling (a directory of .u32 files,
// or .txt with -t, or a single file; sorted by name).
//
//   oracle/_ref/table10_cpu.x [-t] -i <dir|file>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <limits>
#include <string>
#include <vector>

extern "C" {
size_t ref_encode(int kind, int f, const uint32_t* in, size_t n, uint8_t* out, size_t cap);
void ref_decode(int kind, int f, const uint8_t* in, size_t nbytes, uint32_t* out, size_t n);
}

namespace fs = std::filesystem;
static const int NUM_RUNS = 5;  // table_efficiency.cpp:32

static std::vector<uint32_t> read_u32(const std::string& name)
{
    FILE* f = fopen(name.c_str(), "rb");
    if (!f) { fprintf(stderr, "opening file %s failed\n", name.c_str()); exit(EXIT_FAILURE); }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> v((size_t)sz / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) { fprintf(stderr, "reading file content failed\n"); exit(EXIT_FAILURE); }
    fclose(f);
    return v;
}
static std::vector<uint32_t> read_text(const std::string& name)
{
    std::vector<uint32_t> v;
    FILE* f = fopen(name.c_str(), "r");
    if (!f) { fprintf(stderr, "opening file %s failed\n", name.c_str()); exit(EXIT_FAILURE); }
    uint32_t num;
    while (fscanf(f, "%u\n", &num) == 1) v.push_back(num);
    fclose(f);
    return v;
}

static void run(const char* name, int kind, int f, const std::vector<std::vector<uint32_t>>& inputs)
{
    printf("\\method{%s}  &\n", name);
    std::vector<double> enc_speed, dec_speed;
    for (const auto& input : inputs) {
        std::vector<uint8_t> encoded(input.size() * 8 + 65536);  // (:73; + the rfold header of tiny inputs)
        size_t bytes = 0, enc_min = std::numeric_limits<size_t>::max(), dec_min = enc_min;
        for (int i = 0; i < NUM_RUNS; i++) {
            const auto t0 = std::chrono::high_resolution_clock::now();
            bytes = ref_encode(kind, f, input.data(), input.size(), encoded.data(), encoded.size());
            const auto t1 = std::chrono::high_resolution_clock::now();
            enc_min = std::min((size_t)(t1 - t0).count(), enc_min);
        }
        std::vector<uint32_t> recover(input.size());
        for (int i = 0; i < NUM_RUNS; i++) {
            const auto t0 = std::chrono::high_resolution_clock::now();
            ref_decode(kind, f, encoded.data(), bytes, recover.data(), recover.size());
            const auto t1 = std::chrono::high_resolution_clock::now();
            dec_min = std::min((size_t)(t1 - t0).count(), dec_min);
        }
        if (recover != input) { fprintf(stderr, "%s NOT EQUAL!\n", name); exit(EXIT_FAILURE); }
        enc_speed.push_back(double(input.size()) / (double(enc_min) / 1e9));  // util.hpp:307-311
        dec_speed.push_back(double(input.size()) / (double(dec_min) / 1e9));
    }
    for (size_t i = 0; i < enc_speed.size(); i++) {  // :112-120
        for (size_t j = 0; j < i * 4; j++) printf(" ");
        printf("%15.4f  &  %15.4f  ", enc_speed[i], dec_speed[i]);
        if (i + 1 == enc_speed.size()) printf("\\\\ \n\n");
        else printf("&\n");
    }
}

int main(int argc, char** argv)
{
    std::string input;
    bool text = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-t" || a == "--text") text = true;
        else if ((a == "-i" || a == "--input") && i + 1 < argc) input = argv[++i];
        else { fprintf(stderr, "usage: table10_cpu.x [-t] -i <dir|file>\n"); return EXIT_FAILURE; }
    }
    if (input.empty()) { fprintf(stderr, "Missing required option: --input\n"); return EXIT_FAILURE; }
    const std::string ext = text ? ".txt" : ".u32";
    std::vector<std::string> files;
    if (fs::is_regular_file(fs::path(input))) files.push_back(input);
    else
        for (const auto& e : fs::directory_iterator(fs::path(input))) {
            const std::string fn = e.path().filename().string();
            if (e.is_regular_file() && fn.size() >= ext.size() && fn.compare(fn.size() - ext.size(), ext.size(), ext) == 0)
                files.push_back(e.path().string());
        }
    std::sort(files.begin(), files.end());
    std::vector<std::vector<uint32_t>> inputs;
    for (const auto& fn : files) inputs.push_back(text ? read_text(fn) : read_u32(fn));
    if (inputs.empty()) { fprintf(stderr, "no input files\n"); return EXIT_FAILURE; }
    run("ANS", 3, 0, inputs);         // ANSint        table_efficiency.cpp:175
    run("ANSfold-1", 0, 1, inputs);   //               :176
    run("ANSfold-5", 0, 5, inputs);   //               :177
    run("ANSrfold-1", 1, 1, inputs);  //               :178
    run("ANSrfold-5", 1, 5, inputs);  //               :179
    run("ANSmsb", 2, 0, inputs);      // benchmark.cpp:174 / table_effectiveness.cpp:146
    return EXIT_SUCCESS;
}

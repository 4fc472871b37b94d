// generate_inputs -- the reference's synthetic benchmark files (/root/reference/src/generate_inputs.cpp:
// 94-122: uniform08/12/16/20, geom0.01 ... geom0.99, zipf12, zipf20; -n numbers each, .u32 or .txt) produced
// by the library's counter-based generators (include/ansx.h ansx_generate_host -- the same values the HIP
// kernel k_generate writes on the device).  The distributions are the reference's; the random stream is
// not (std::mt19937(0) + libstdc++ + libm), see include/ansx.h.
//
//   generate_inputs -n <numbers per file> -o <output directory> [-t] [-s <seed>]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ansx.h"

static void emit(int dist, double a, double b, uint64_t seed, size_t n, const std::string& path, bool text)
{
    printf("generating file %s\n", path.c_str());
    std::vector<uint32_t> v(n);
    if (ansx_generate_host(dist, a, b, seed, 0, v.data(), n) != ANSX_OK) {
        fprintf(stderr, "error: generator rejected its parameters\n");
        exit(EXIT_FAILURE);
    }
    FILE* f = fopen((path + (text ? ".txt" : ".u32")).c_str(), text ? "w" : "wb");
    if (!f) {
        fprintf(stderr, "error: opening output file %s failed\n", path.c_str());
        exit(EXIT_FAILURE);
    }
    if (text) for (uint32_t x : v) fprintf(f, "%u\n", x);
    else fwrite(v.data(), 4, v.size(), f);
    fclose(f);
}

int main(int argc, char** argv)
{
    std::string out;
    size_t n = 0;
    bool text = false;
    uint64_t seed = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if ((a == "-o" || a == "--output") && i + 1 < argc) out = argv[++i];
        else if ((a == "-n" || a == "--num") && i + 1 < argc) n = strtoull(argv[++i], nullptr, 10);
        else if ((a == "-s" || a == "--seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        else if (a == "-t" || a == "--text") text = true;
        else {
            fprintf(stderr, "usage: %s -n <num> -o <output path> [-t] [-s <seed>]\n", argv[0]);
            return EXIT_FAILURE;
        }
    }
    if (out.empty() || n == 0) {
        fprintf(stderr, "error: missing required option (-n, -o)\n");
        return EXIT_FAILURE;
    }
    for (int bits : { 8, 12, 16, 20 }) {  // generate_inputs.cpp:94-101
        char nm[32];
        snprintf(nm, sizeof(nm), "/uniform%02d", bits);
        emit(ANSX_GEN_UNIFORM, 0, (double)((1u << bits) - 1), seed, n, out + nm, text);
    }
    for (const char* p : { "0.01", "0.1", "0.2", "0.4", "0.6", "0.8", "0.9", "0.99" })  // :103-118
        emit(ANSX_GEN_GEOMETRIC, atof(p), 0, seed, n, out + "/geom" + p, text);
    emit(ANSX_GEN_ZIPF, (double)(1 << 12), 1.0, seed, n, out + "/zipf12", text);  // :119-122
    emit(ANSX_GEN_ZIPF, (double)(1 << 20), 1.0, seed, n, out + "/zipf20", text);
    return 0;
}

// table_effectiveness (ansx edition): the reference's Table-9 harness (bits per integer) for the
// ANSfold / ANSrfold / ANSmsb rows, driven through the drop-in codec structs of
// include/ansx_methods.hpp.
//
// Behaviour follows /root/reference/src/table_effectiveness.cpp for this path:
//   -h/--help, -t/--text, -i/--input <dir|file>                        (:34-62; no Boost here)
//   per codec: one encode per file into an n*8-byte buffer (:71-74), BPI = 8 * bytes / n (:75),
//   rows printed as "name  &" then "%2.4f  " per file, "&" between files, "\\\\" at the end (:67,78-88)
//   codecs: ANSmsb (:146) plus the fold rows of fold_effectiveness.cpp:132-148 (ANSfold-1..5, ANSrfold-1..5)
//
// Extra: --stream measures the single-reference-stream variants: those bytes are identical to the
//        reference's own encode() output, so the numbers are the reference's Table-9 numbers exactly;
//        the default rows are the block container (index + restart points included).
//        Every encode is decoded again and compared (the reference's harness does not).
//
// Build: make -C ans_large_alphabet_amd/tools   (g++ -std=c++17, links libansx.so)
#include "../include/ansx_methods.hpp"
#include "harness_common.hpp"

template <class t_compressor> void run(std::vector<std::vector<uint32_t>>& inputs)
{
    printf("%s  &\n", t_compressor::name().c_str());
    std::vector<double> BPIs;
    for (const auto& input : inputs) {
        std::vector<uint8_t> encoded_data(input.size() * 8 + 4096);
        std::vector<uint8_t> tmp_buf(16);
        const size_t encoded_bytes = t_compressor::encode(input.data(), input.size(), encoded_data.data(),
            encoded_data.size(), tmp_buf.data());
        std::vector<uint32_t> recover(input.size());
        t_compressor::decode(encoded_data.data(), encoded_bytes, recover.data(), recover.size(), tmp_buf.data());
        require_equal(input.data(), recover.data(), input.size(), t_compressor::name());
        BPIs.push_back(double(encoded_bytes * 8) / double(input.size()));
    }
    for (size_t i = 0; i < BPIs.size(); i++) {
        for (size_t j = 0; j < i * 4; j++) printf(" ");
        printf("%2.4f  ", BPIs[i]);
        if (i + 1 == BPIs.size()) printf("\\\\ \n");
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
                   "reference stream per file instead of the block container\n");
            return EXIT_SUCCESS;
        } else if (a == "-t" || a == "--text") {
            text = true;
        } else if ((a == "-i" || a == "--input") && i + 1 < argc) {
            input = argv[++i];
        } else if (a == "--stream") {
            stream = true;
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
            run<ANSintGPU>(inputs);         // "ANS" (table_effectiveness.cpp:145), per-block compacted alphabet
            run<ANSmsbGPU>(inputs);
            run<ANSmsbGPUCompact>(inputs);  // the pseudo_adaptive.cpp pairing (:253-254)
            run<ANSfoldGPU<1>>(inputs);
            run<ANSfoldGPU<2>>(inputs);
            run<ANSfoldGPU<3>>(inputs);
            run<ANSfoldGPU<4>>(inputs);
            run<ANSfoldGPU<5>>(inputs);
            run<ANSfoldGPU<6>>(inputs);  // fold_effectiveness.cpp:132-139 sweeps 1..8; 8 is unsound upstream
            run<ANSfoldGPU<7>>(inputs);
            run<ANSrfoldGPU<1>>(inputs);
            run<ANSrfoldGPU<2>>(inputs);
            run<ANSrfoldGPU<3>>(inputs);
            run<ANSrfoldGPU<4>>(inputs);
            run<ANSrfoldGPU<5>>(inputs);
            run<ANSrfoldGPU<6>>(inputs);
            run<ANSrfoldGPU<7>>(inputs);
        } else {
            run<ANSmsbGPUStream>(inputs);
            run<ANSfoldGPUStream<1>>(inputs);
            run<ANSfoldGPUStream<3>>(inputs);
            run<ANSfoldGPUStream<5>>(inputs);
            run<ANSrfoldGPUStream<1>>(inputs);
            run<ANSrfoldGPUStream<3>>(inputs);
            run<ANSrfoldGPUStream<5>>(inputs);
        }
    } catch (const std::exception& e) {
        quit(e.what());
    }
    return EXIT_SUCCESS;
}

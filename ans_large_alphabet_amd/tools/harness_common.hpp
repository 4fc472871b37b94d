// Shared input handling of the ansx harnesses (table_efficiency, table_effectiveness): behaviour of
// /root/reference/src/table_efficiency.cpp:128-157 and include/util.hpp:101-115,160-192 without Boost:
// files matching .*\.u32 (or .*\.txt with -t) of a directory, or one file, sorted by name; text = one
// decimal per line, binary = raw little-endian uint32; errors print and stop (quit()).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <string>
#include <vector>

namespace fs = std::filesystem;

[[noreturn]] static void quit(const std::string& msg)
{
    fprintf(stderr, "error: %s\n", msg.c_str());
    exit(EXIT_FAILURE);
}

static std::vector<uint32_t> read_file_text(const std::string& name)
{
    std::vector<uint32_t> v;
    FILE* f = fopen(name.c_str(), "r");
    if (!f) quit("opening file " + name + " failed");
    uint32_t num;
    while (fscanf(f, "%u\n", &num) == 1) v.push_back(num);
    fclose(f);
    return v;
}

static std::vector<uint32_t> read_file_u32(const std::string& name)
{
    FILE* f = fopen(name.c_str(), "rb");
    if (!f) quit("opening file " + name + " failed");
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz % 4 != 0) quit("reading file content failed: file size % 32bit != 0");
    std::vector<uint32_t> v((size_t)sz / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) quit("reading file content failed");
    fclose(f);
    return v;
}

static void require_equal(const uint32_t* a, const uint32_t* b, size_t n, const std::string& name)
{
    int errors = 0;
    for (size_t i = 0; i < n; i++) {
        if (a[i] != b[i]) {
            errors++;
            fprintf(stderr, "%s not equal at position %zu/%zu -> expected=%u is=%u\n", name.c_str(), i, n - 1,
                a[i], b[i]);
            if (errors == 5) quit(name + " not equal");
        }
    }
    if (errors != 0) quit("NOT EQUAL!");
}


static std::vector<std::vector<uint32_t>> load_inputs(const std::string& input, bool text)
{
    const std::string ext = text ? ".txt" : ".u32";
    std::vector<std::string> files;
    fs::path p(input);
    if (fs::is_regular_file(p)) {
        files.push_back(p.string());
    } else {
        for (const auto& e : fs::directory_iterator(p)) {
            if (!e.is_regular_file()) continue;
            const std::string fn = e.path().filename().string();
            if (fn.size() >= ext.size() && fn.compare(fn.size() - ext.size(), ext.size(), ext) == 0)
                files.push_back(e.path().string());
        }
    }
    std::sort(files.begin(), files.end());
    std::vector<std::vector<uint32_t>> inputs;
    for (const auto& f : files) inputs.push_back(text ? read_file_text(f) : read_file_u32(f));
    if (inputs.empty()) quit("no input files");
    return inputs;
}

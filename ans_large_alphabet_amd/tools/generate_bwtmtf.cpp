// generate_bwtmtf -- word- or byte-parsed text -> BWT -> move-to-front ranks, the pipeline behind the
// reference's "*-WORD-BWTMTF" datasets (/root/reference/src/generate_bwtmtf.cpp:67-99 word_parse,
// :101-115 byte_parse, :142-173 SA -> BWT -> MTF, scripts/download_data.sh:24-30).  Same options and
// output files; no Boost, own suffix sorter (prefix doubling instead of the vendored qsufsort: the
// terminal 0 is the unique smallest symbol in word mode, so the suffix order -- and with it every
// output byte -- does not depend on the algorithm) and a Fenwick tree instead of the O(alphabet)
// deque scan per symbol for the move-to-front ranks (same ranks).
//
//   generate_bwtmtf -i <text file> -n <max symbols> -o <output prefix> [-w] [-t]
//     -w  word parse (lower-cased, split at any run of ";, \n.?'()-\"", ids in order of first appearance)
//     -t  decimal text output, one number per line (default: raw little-endian uint32)
//   writes <prefix>-WORD.u32 / -CHAR.u32 (the parsed symbols) and <prefix>-...-BWTMTF.u32 (the ranks)
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

[[noreturn]] static void quit(const std::string& msg)
{
    fprintf(stderr, "error: %s\n", msg.c_str());
    exit(EXIT_FAILURE);
}

static std::vector<uint8_t> read_file_u8(const std::string& name)
{
    FILE* f = fopen(name.c_str(), "rb");
    if (!f) quit("opening file " + name + " failed");
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)sz);
    if (sz && fread(v.data(), 1, v.size(), f) != v.size()) quit("reading file content failed");
    fclose(f);
    return v;
}

static void write_file(const std::vector<uint32_t>& v, const std::string& name, bool text)
{
    FILE* f = fopen(name.c_str(), text ? "w" : "wb");
    if (!f) quit("opening output file " + name + " failed");
    if (text) {
        for (uint32_t x : v) fprintf(f, "%u\n", x);
    } else if (!v.empty() && fwrite(v.data(), 4, v.size(), f) != v.size()) {
        quit("writing " + name + " failed");
    }
    fclose(f);
}

// generate_bwtmtf.cpp:67-99.  boost::split(..., is_any_of(...), token_compress_on): tokens are the
// stretches between RUNS of delimiters; a leading / trailing run yields an empty first / last token.
static std::vector<int> word_parse(const std::string& file, size_t n)
{
    std::vector<uint8_t> c = read_file_u8(file);
    for (auto& ch : c) ch = (uint8_t)std::tolower(ch);
    bool delim[256] = {};
    for (const char* d = ";, \n.?'()-\""; *d; d++) delim[(uint8_t)*d] = true;
    std::vector<int> T;
    std::unordered_map<std::string, uint32_t> str2id;
    const char* base = (const char*)c.data();
    size_t i = 0;
    const size_t N = c.size();
    for (;;) {
        size_t j = i;
        while (j < N && !delim[c[j]]) j++;
        const std::string word(base + i, j - i);
        auto it = str2id.find(word);
        if (it != str2id.end()) T.push_back((int)it->second);
        else {
            const uint32_t id = (uint32_t)str2id.size() + 1;
            T.push_back((int)id);
            str2id.emplace(word, id);
        }
        if (T.size() >= n || j >= N) break;
        while (j < N && delim[c[j]]) j++;  // token_compress_on
        i = j;                             // (j == N here: one more, empty, token -- as boost::split yields)
    }
    T.push_back(0);
    return T;
}

// generate_bwtmtf.cpp:101-115 (n is clamped to the file size; the reference reads past a shorter file)
static std::vector<int> byte_parse(const std::string& file, size_t n)
{
    const std::vector<uint8_t> c = read_file_u8(file);
    if (n > c.size()) n = c.size();
    std::vector<int> T(n + 1);
    for (size_t i = 0; i < n; i++) T[i] = c[i];
    T[n] = 0;
    return T;
}

// Suffix array of T[0..N) by prefix doubling: rank pairs (rank[i], rank[i + k]) sorted with a 64-bit key.
// Suffixes that run off the end compare smaller (rank -1 -> key part 0), which is what a unique smallest
// terminal gives anyway.
static std::vector<int> suffix_array(const std::vector<int>& T)
{
    const size_t N = T.size();
    std::vector<int> sa(N), rnk(N), tmp(N);
    std::iota(sa.begin(), sa.end(), 0);
    std::sort(sa.begin(), sa.end(), [&](int a, int b) { return T[a] < T[b]; });
    rnk[sa[0]] = 0;
    for (size_t i = 1; i < N; i++) rnk[sa[i]] = rnk[sa[i - 1]] + (T[sa[i]] != T[sa[i - 1]] ? 1 : 0);
    std::vector<uint64_t> key(N);
    for (size_t k = 1; k < N && (size_t)rnk[sa[N - 1]] + 1 < N; k <<= 1) {
        for (size_t i = 0; i < N; i++) {
            const uint64_t hi = (uint64_t)rnk[i] + 1, lo = i + k < N ? (uint64_t)rnk[i + k] + 1 : 0;
            key[i] = (hi << 32) | lo;
        }
        // only groups that are still tied need sorting: sort within runs of equal first rank
        size_t a = 0;
        while (a < N) {
            size_t b = a + 1;
            const int ra = rnk[sa[a]];
            while (b < N && rnk[sa[b]] == ra) b++;
            if (b - a > 1) std::sort(sa.begin() + a, sa.begin() + b, [&](int x, int y) { return key[x] < key[y]; });
            a = b;
        }
        tmp[sa[0]] = 0;
        for (size_t i = 1; i < N; i++) tmp[sa[i]] = tmp[sa[i - 1]] + (key[sa[i]] != key[sa[i - 1]] ? 1 : 0);
        rnk.swap(tmp);
    }
    return sa;
}

// move-to-front ranks (generate_bwtmtf.cpp:117-124,159-166): the alphabet starts as 0, 1, ..., max; the rank
// of a symbol is the number of distinct symbols in front of it.  Every symbol carries the time of its last
// move to the front (initially -sym: smaller symbols are further ahead); rank = symbols with a later time.
static std::vector<uint32_t> mtf_ranks(const std::vector<int>& bwt, size_t len, int max_sym)
{
    const size_t A = (size_t)max_sym + 1, slots = A + len + 1;
    std::vector<uint32_t> fen(slots + 1, 0);
    auto add = [&](size_t i, int d) {
        for (i++; i <= slots; i += i & (~i + 1)) fen[i] += (uint32_t)d;
    };
    auto prefix = [&](size_t i) {  // sum of [0, i)
        uint32_t s = 0;
        for (; i > 0; i -= i & (~i + 1)) s += fen[i];
        return s;
    };
    // time slot of symbol s initially: A - 1 - s (symbol 0 has the latest time = front)
    std::vector<size_t> when(A);
    for (size_t s = 0; s < A; s++) {
        when[s] = A - 1 - s;
        add(when[s], 1);
    }
    std::vector<uint32_t> out(len);
    size_t now = A;
    for (size_t i = 0; i < len; i++) {
        const size_t s = (size_t)bwt[i];
        out[i] = (uint32_t)(A - prefix(when[s] + 1));  // symbols with a later time
        add(when[s], -1);
        when[s] = now++;
        add(when[s], 1);
    }
    return out;
}

int main(int argc, char** argv)
{
    std::string input, prefix;
    size_t n = 0;
    bool text = false, words = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if ((a == "-i" || a == "--input") && i + 1 < argc) input = argv[++i];
        else if ((a == "-o" || a == "--output") && i + 1 < argc) prefix = argv[++i];
        else if ((a == "-n" || a == "--num") && i + 1 < argc) n = strtoull(argv[++i], nullptr, 10);
        else if (a == "-t" || a == "--text") text = true;
        else if (a == "-w" || a == "--word") words = true;
        else {
            fprintf(stderr, "usage: %s -i <input file> -n <num> -o <output prefix> [-w] [-t]\n", argv[0]);
            return a == "-h" || a == "--help" ? EXIT_SUCCESS : EXIT_FAILURE;
        }
    }
    if (input.empty() || prefix.empty() || n == 0) quit("missing required option (-i, -n, -o)");
    std::string file_name = prefix + (words ? "-WORD" : "-CHAR");
    const std::vector<int> T = words ? word_parse(input, n) : byte_parse(input, n);
    const int max_sym = *std::max_element(T.begin(), T.end() - 1);
    printf("text size = %zu min_sym = %d max_sym = %d\n", T.size(), *std::min_element(T.begin(), T.end() - 1), max_sym);
    const std::vector<int> SA = suffix_array(T);
    std::vector<int> BWT(T.size());
    for (size_t i = 0; i < T.size(); i++) BWT[i] = SA[i] != 0 ? T[(size_t)SA[i] - 1] : T.back();  // :152-157
    size_t seq_len = T.size() - 1;
    if (seq_len > n) seq_len = n;
    const std::vector<uint32_t> MTF = mtf_ranks(BWT, seq_len, max_sym);
    std::vector<uint32_t> text_u32(seq_len);
    for (size_t i = 0; i < seq_len; i++) text_u32[i] = (uint32_t)T[i];
    const char* ext = text ? ".txt" : ".u32";
    write_file(text_u32, file_name + ext, text);
    write_file(MTF, file_name + "-BWTMTF" + ext, text);
    return 0;
}

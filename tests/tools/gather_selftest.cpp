// ansx_gather_containers over RCCL, driven from C++ with one thread per GPU: every rank draws its slice of
// ONE list (ansx_generate_dev, indices [r n, (r+1) n)), encodes it, the containers are gathered on the root and
// merged; the root decodes the merged container and compares it with the whole list drawn on the host.
//   gather_selftest [max_ranks] [ints_per_rank]      (ranks = min(max_ranks, visible GPUs); exit code 0 = OK)
// Build: g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I<repo>/include gather_selftest.cpp \
//            -L<repo>/ans_large_alphabet_amd -lansx -L/opt/rocm/lib -lamdhip64 -lrccl -lpthread
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ansx.h"

#define CHECK(x)                                                                 \
    do {                                                                         \
        if (!(x)) {                                                              \
            fprintf(stderr, "gather_selftest: %s failed (line %d)\n", #x, __LINE__); \
            exit(2);                                                             \
        }                                                                        \
    } while (0)

int main(int argc, char** argv)
{
    int want = argc > 1 ? atoi(argv[1]) : 8;
    const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)(5 * 16384);  // whole blocks per rank
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0);
    const int N = want < ndev ? want : ndev;
    std::vector<int> devs(N);
    for (int i = 0; i < N; i++) devs[i] = i;
    std::vector<ncclComm_t> comms(N);
    CHECK(ncclCommInitAll(comms.data(), N, devs.data()) == ncclSuccess);
    const int root = N - 1;  // (not rank 0 on purpose)
    const uint64_t seed = 4242;
    std::vector<int> status(N, -1);
    std::vector<size_t> merged_bytes(N, 0);
    std::vector<uint8_t*> d_merged(N, nullptr);
    std::vector<ansx_ctx*> ctxs(N, nullptr);
    const ansx_opts opts = { 16384, 1024, 0, 0 };
    const size_t slot = (ansx_bound(ANSX_FOLD, 1, n, &opts) + 15) / 16 * 16;
    auto worker = [&](int r) {
        CHECK(hipSetDevice(r) == hipSuccess);
        ansx_ctx* c = nullptr;
        CHECK(ansx_init(r, &c) == ANSX_OK);
        ctxs[r] = c;
        uint32_t* d_in = nullptr;
        uint8_t *d_out = nullptr, *d_recv = nullptr;
        CHECK(hipMalloc((void**)&d_in, n * 4) == hipSuccess && hipMalloc((void**)&d_out, slot) == hipSuccess);
        if (r == root) CHECK(hipMalloc((void**)&d_recv, slot * N) == hipSuccess && hipMalloc((void**)&d_merged[r], slot * N + 4096) == hipSuccess);
        CHECK(ansx_generate_dev(c, ANSX_GEN_ZIPF, 1048576.0, 1.2, seed, (uint64_t)r * n, d_in, n, nullptr) == ANSX_OK);
        size_t bytes = 0;
        CHECK(ansx_encode_dev(c, ANSX_FOLD, 1, d_in, n, d_out, slot, &bytes, &opts, nullptr) == ANSX_OK);
        status[r] = ansx_gather_containers(c, comms[r], r, N, root, d_out, bytes, d_recv, slot, d_merged[r], slot * N + 4096,
            &merged_bytes[r], nullptr);
        CHECK(hipDeviceSynchronize() == hipSuccess);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < N; r++) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (int r = 0; r < N; r++) CHECK(status[r] == ANSX_OK);
    for (int r = 0; r < N; r++) CHECK((merged_bytes[r] != 0) == (r == root));
    // the root decodes the merged container: it must be the whole list
    CHECK(hipSetDevice(root) == hipSuccess);
    uint32_t* d_back = nullptr;
    CHECK(hipMalloc((void**)&d_back, n * N * 4) == hipSuccess);
    CHECK(ansx_decode_dev(ctxs[root], ANSX_FOLD, 1, d_merged[root], merged_bytes[root], d_back, n * N, &opts, nullptr) == ANSX_OK);
    std::vector<uint32_t> got(n * N), want_v(n * N);
    CHECK(hipMemcpy(got.data(), d_back, n * N * 4, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(ansx_generate_host(ANSX_GEN_ZIPF, 1048576.0, 1.2, seed, 0, want_v.data(), n * N) == ANSX_OK);
    CHECK(memcmp(got.data(), want_v.data(), n * N * 4) == 0);
    printf("gather_selftest OK: %d rank(s), %zu ints each, merged container %zu bytes on rank %d\n", N, n, merged_bytes[root], root);
    return 0;
}

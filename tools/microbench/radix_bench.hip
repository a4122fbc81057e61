// Micro-benchmark + correctness check of plx::radix::sort_pairs (simplex_gp_amd/csrc/plx_radix.h) against std::stable_sort
// and rocPRIM's radix_sort_pairs at the sizes of the lattice build.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../include -I../../simplex_gp_amd/csrc radix_bench.hip -o radix_bench
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <numeric>
#include <random>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "plx_radix.h"
namespace plx { void set_error(const char *, ...) {} }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class K>
static int run(size_t n, int end_bit, bool check, hipStream_t s)
{
    std::vector<K> h(n);
    std::vector<uint32_t> hv(n);
    std::mt19937_64 rng(n + end_bit);
    const K mask = end_bit >= (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << end_bit) - 1);
    for (auto &x : h) x = (K)rng() & mask & ((rng() & 7) ? ~(K)0 : (K)0xFFFF);       // many duplicate keys: stability matters
    std::iota(hv.begin(), hv.end(), 0u);
    K *ka, *kb; uint32_t *va, *vb; void *tmp, *rtmp = nullptr; size_t rtb = 0;
    CK(hipMalloc(&ka, n * sizeof(K))); CK(hipMalloc(&kb, n * sizeof(K)));
    CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4));
    CK(hipMalloc(&tmp, plx::radix::temp_bytes(n)));
    int second = 0;
    if (check) {
        CK(hipMemcpy(ka, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
        CK(hipMemcpy(va, hv.data(), n * 4, hipMemcpyHostToDevice));
        if (plx::radix::sort_pairs<K>(tmp, ka, kb, va, vb, (int64_t)n, end_bit, &second, s) != 0) return 1;
        CK(hipStreamSynchronize(s));
        std::vector<K> gk(n); std::vector<uint32_t> gv(n);
        CK(hipMemcpy(gk.data(), second ? kb : ka, n * sizeof(K), hipMemcpyDeviceToHost));
        CK(hipMemcpy(gv.data(), second ? vb : va, n * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h[a] < h[b]; });
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += (gv[i] != order[i]) || (gk[i] != h[order[i]]);
        printf("check n=%8zu key=%zu B bits=%2d: %s (%zu mismatches)\n", n, sizeof(K), end_bit, bad ? "FAILED" : "ok", bad);
        if (bad) return 1;
    }
    CK(hipMemcpy(ka, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) plx::radix::sort_pairs<K>(tmp, ka, kb, va, vb, (int64_t)n, end_bit, &second, s);
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) plx::radix::sort_pairs<K>(tmp, ka, kb, va, vb, (int64_t)n, end_bit, &second, s);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const float mine = ms / reps * 1e3f;
    CK((rocprim::radix_sort_pairs(nullptr, rtb, ka, kb, va, vb, n, 0, (unsigned)end_bit, s)));
    CK(hipMalloc(&rtmp, rtb + 16));
    for (int i = 0; i < 3; ++i) CK((rocprim::radix_sort_pairs(rtmp, rtb, ka, kb, va, vb, n, 0, (unsigned)end_bit, s)));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) CK((rocprim::radix_sort_pairs(rtmp, rtb, ka, kb, va, vb, n, 0, (unsigned)end_bit, s)));
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("time  n=%8zu key=%zu B bits=%2d: plx::radix %7.1f us   rocPRIM %7.1f us\n", n, sizeof(K), end_bit, mine, ms / reps * 1e3f);
    CK(hipFree(ka)); CK(hipFree(kb)); CK(hipFree(va)); CK(hipFree(vb)); CK(hipFree(tmp)); CK(hipFree(rtmp));
    return 0;
}

int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    for (size_t n : {1ul, 2ul, 63ul, 4096ul, 4097ul, 100003ul})
        if (run<uint64_t>(n, 37, true, s) || run<uint32_t>(n, 19, true, s)) return 1;
    if (run<uint64_t>(1000000, 36, true, s)) return 1;
    if (run<uint64_t>(1000000, 40, false, s)) return 1;
    if (run<uint64_t>(1000000, 63, true, s)) return 1;
    if (run<uint64_t>(4000000, 36, false, s)) return 1;
    if (run<uint64_t>(400000, 40, true, s)) return 1;
    if (run<uint64_t>(1730000, 40, false, s)) return 1;
    if (run<uint32_t>(2770000, 19, true, s)) return 1;
    if (run<uint32_t>(9000000, 19, false, s)) return 1;
    if (run<uint32_t>(9000000, 22, false, s)) return 1;
    printf("RADIX_BENCH_OK\n");
    return 0;
}

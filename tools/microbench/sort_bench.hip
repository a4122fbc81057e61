// Micro-benchmark: rocPRIM radix_sort_pairs by size, key width, key bits and algorithm (default = merge sort up to 2^20
// items, else Onesweep; forced Onesweep).  The lattice build runs three such sorts.
// Build: hipcc -O3 --offload-arch=gfx950 sort_bench.hip -o sort_bench ; run on the GPU box.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

using Force = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

template <class K, class Cfg>
static int run(const char *name, size_t n, int end_bit, hipStream_t s)
{
    std::vector<K> h(n);
    std::mt19937_64 rng(1);
    const K mask = end_bit >= (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << end_bit) - 1);
    for (auto &x : h) x = (K)rng() & mask;
    K *ki, *ko; uint32_t *vi, *vo; void *tmp = nullptr; size_t tb = 0;
    CK(hipMalloc(&ki, n * sizeof(K))); CK(hipMalloc(&ko, n * sizeof(K)));
    CK(hipMalloc(&vi, n * 4)); CK(hipMalloc(&vo, n * 4));
    CK(hipMemcpy(ki, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
    CK(hipMemset(vi, 0, n * 4));
    CK((rocprim::radix_sort_pairs<Cfg>(nullptr, tb, ki, ko, vi, vo, n, 0, (unsigned)end_bit, s)));
    CK(hipMalloc(&tmp, tb + 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK((rocprim::radix_sort_pairs<Cfg>(tmp, tb, ki, ko, vi, vo, n, 0, (unsigned)end_bit, s)));
    CK(hipEventRecord(a, s));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) CK((rocprim::radix_sort_pairs<Cfg>(tmp, tb, ki, ko, vi, vo, n, 0, (unsigned)end_bit, s)));
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-10s n=%8zu key=%2zu B bits=%2d  %8.1f us\n", name, n, sizeof(K), end_bit, ms / reps * 1e3);
    CK(hipFree(ki)); CK(hipFree(ko)); CK(hipFree(vi)); CK(hipFree(vo)); CK(hipFree(tmp));
    return 0;
}

template <unsigned B, unsigned I, unsigned R>
using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<B, I>, rocprim::kernel_config<B, I>, R,
                                                                          rocprim::block_radix_rank_algorithm::match>, 0>;

int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    // wider digits: fewer passes
    for (size_t n : {2770000ul, 5200000ul}) {
        if (run<uint32_t, OS<256, 12, 10>>("os256x12r10", n, 19, s)) return 1;
        if (run<uint32_t, OS<512, 8, 10>>("os512x8r10", n, 19, s)) return 1;
        if (run<uint32_t, OS<1024, 4, 10>>("os1024x4r10", n, 19, s)) return 1;
        if (run<uint32_t, OS<256, 16, 7>>("os256x16r7", n, 19, s)) return 1;
        if (run<uint32_t, OS<512, 12, 10>>("os512x12r10", n, 20, s)) return 1;
    }
    for (size_t n : {400000ul, 1000000ul}) {
        if (run<uint64_t, OS<256, 12, 10>>("os256x12r10", n, 60, s)) return 1;
        if (run<uint64_t, OS<512, 8, 10>>("os512x8r10", n, 40, s)) return 1;
        if (run<uint64_t, OS<256, 8, 9>>("os256x8r9", n, 63, s)) return 1;
    }
    if (0) for (size_t n : {400000ul, 1000000ul, 2770000ul}) {
        for (int bits : {64, 48, 40, 32}) {
            if (run<uint64_t, rocprim::default_config>("default", n, bits, s)) return 1;
            if (run<uint64_t, Force>("onesweep", n, bits, s)) return 1;
        }
        for (int bits : {32, 24, 19, 16}) {
            if (run<uint32_t, rocprim::default_config>("default", n, bits, s)) return 1;
            if (run<uint32_t, Force>("onesweep", n, bits, s)) return 1;
        }
    }
    return 0;
}

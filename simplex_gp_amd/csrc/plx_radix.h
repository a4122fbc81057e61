// plx_radix.h -- a small stable LSD radix sort of (key, 32-bit value) pairs for the lattice build.
//
// Why not the library sort: the build sorts 4e5 .. 3e6 items three times (points along their Z-curve, vertices along
// their Morton curve, block rows by vertex) and rocPRIM is launch / latency bound at these sizes -- it merge-sorts
// below 2^20 items (1e6 64-bit keys: 201 us in ~22 launches) and its Onesweep passes take 25-40 us each
// (tools/microbench/sort_bench.hip).  The keys here carry few significant bits (the callers compact them first), so
// a plain three-kernel pass -- per-tile digit counts, per-bin row scans, stable scatter -- over exactly
// the significant digits (8 to 10 bits each) is 3 short launches per pass and nothing else.
//
//   count    tile t (2048 or 4096 keys, 256 threads): LDS histogram of the digit        -> counts[bin][tile]
//   scan     workgroup b: exclusive scan of row b over the tiles, row total     -> counts[bin][tile], totals[bin]
//   scatter  tile t: bin starts (scan of totals over the bins) + its row offsets; every wave owns a contiguous quarter
//            of the tile and walks it 64 keys at a time: the lanes holding equal digits find each other with one
//            ballot per digit bit, the first of them advances the wave's running offset of that digit in LDS, and every key goes
//            to offset + (equal-digit lanes below it).  Order of equal digits = (wave, step, lane) = input order: stable.
//
// Included by plx_sort.hip only.  No MFMA, no atomics on memory; LDS atomics only for the (order-free) histograms.
#pragma once

#include "plx_internal.h"

namespace plx {
namespace radix {

constexpr int kThreads = 256;
constexpr int kMaxDigitBits = 10;                     // widest digit: 1024 bins (11- and 12-bit digits were measured slower per
                                                      // key bit at every size: the scatter's write locality goes first)

template <class K, int DB, int KPT>
__global__ __launch_bounds__(kThreads) void count_kernel(const K *__restrict__ keys, int n, int shift, int ntiles,
                                                         int *__restrict__ counts)
{
    constexpr int BINS = 1 << DB, kKeysPerThread = KPT, kTile = kThreads * KPT;
    __shared__ int hist[BINS];
    const int tid = threadIdx.x, tile = blockIdx.x;
    for (int b = tid; b < BINS; b += kThreads) hist[b] = 0;
    __syncthreads();
    const int base = tile * kTile;
    K k[kKeysPerThread];
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const int i = base + j * kThreads + tid;
        k[j] = i < n ? keys[i] : (K)0;
    }
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j)
        if (base + j * kThreads + tid < n) atomicAdd(&hist[(int)((k[j] >> shift) & (K)(BINS - 1))], 1);
    __syncthreads();
    for (int b = tid; b < BINS; b += kThreads) counts[(size_t)b * ntiles + tile] = hist[b];
}

// exclusive scan of one int per thread over the workgroup (kThreads threads); *total = the sum
__device__ __forceinline__ int radix_wg_scan(int v, int *wsum, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
        const int s = wsum[w];
        if (w < wave) before += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return before + incl - v;
}

// row b of counts (one digit value, all tiles): exclusive scan in place, total to totals[b]
__global__ __launch_bounds__(kThreads) void scan_rows_kernel(int *__restrict__ counts, int ntiles, int *__restrict__ totals)
{
    __shared__ int wsum[kThreads / 64];
    int *row = counts + (size_t)blockIdx.x * ntiles;
    const int tid = threadIdx.x;
    int carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += kThreads) {       // same trip count in every thread
        const int t = t0 + tid;
        const int v = t < ntiles ? row[t] : 0;
        int total;
        const int ex = radix_wg_scan(v, wsum, &total);
        if (t < ntiles) row[t] = carry + ex;
        carry += total;
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

template <class K, int DB, int KPT, bool HAS_VALS>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                           K *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int n,
                                                           int shift, int ntiles, const int *__restrict__ counts,
                                                           const int *__restrict__ totals)
{
    constexpr int kKeysPerThread = KPT, kTile = kThreads * KPT;
    constexpr int BINS = 1 << DB, W = kThreads / 64, PER_WAVE = kTile / W, BPT = BINS / kThreads;   // bins per thread
    extern __shared__ int run_lds[];                    // [W][BINS] per wave: count of each digit, then the running output offset
    __shared__ int wsum[W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
    int *run = run_lds + wave * BINS;
    for (int b = tid; b < W * BINS; b += kThreads) run_lds[b] = 0;
    // this wave's keys: positions base + j * 64 + lane
    const int base = tile * kTile + wave * PER_WAVE;
    K k[kKeysPerThread];
    uint32_t v[kKeysPerThread];
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const int i = base + j * 64 + lane;
        k[j] = i < n ? keys_in[i] : (K)0;
        if (HAS_VALS) v[j] = i < n ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;      // vals_in == nullptr: the values are 0, 1, 2, ... (first pass)
    }
    // start of every bin in the output (exclusive scan of the bin totals; thread t owns bins t * BPT ..) + this tile's
    // offset inside the bin
    int tot[BPT], mine = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) { tot[q] = totals[tid * BPT + q]; mine += tot[q]; }
    int all;
    int at = radix_wg_scan(mine, wsum, &all);           // (its barriers also publish the zeroed run_lds)
    int start[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        start[q] = at + counts[(size_t)(tid * BPT + q) * ntiles + tile];
        at += tot[q];
    }
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j)
        if (base + j * 64 + lane < n) atomicAdd(&run[(int)((k[j] >> shift) & (K)(BINS - 1))], 1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BPT; ++q) {                     // the waves' counts of a digit -> their first output offsets
        int o = start[q];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int c = run_lds[w * BINS + tid * BPT + q];
            run_lds[w * BINS + tid * BPT + q] = o;
            o += c;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kKeysPerThread; ++j) {
        const bool live = base + j * 64 + lane < n;
        const int d = (int)((k[j] >> shift) & (K)(BINS - 1));
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const bool bit = (d >> b) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int rank = __popcll(peers & below);
        // the running offsets pass between the lanes of this wave (and between its unrolled steps) through LDS: volatile
        // accesses and a wave barrier on both sides, so that neither the compiler nor a future non-lockstep wave may
        // serve a peer's read from a stale register or let the leader's store overtake it
        volatile int *vrun = run;
        const int old = vrun[d];                        // every peer reads the offset before the first of them advances it
        __builtin_amdgcn_wave_barrier();
        if (live && rank == 0) vrun[d] = old + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (live) {
            keys_out[old + rank] = k[j];
            if (HAS_VALS) vals_out[old + rank] = v[j];
        }
    }
}

// keys per thread: 2048-key tiles below 7e5 keys (4e5 64-bit keys: 71 vs 87 us -- twice the workgroups per CU), 4096-key
// tiles above (2.8e6 keys: 80 vs 112 us)
inline int keys_per_thread(int64_t n) { return n < 700000 ? 8 : 16; }
inline int num_tiles(int64_t n) { const int tile = kThreads * keys_per_thread(n); return (int)((n + tile - 1) / tile); }
inline size_t temp_bytes(int64_t n) { return (((size_t)1 << kMaxDigitBits) * num_tiles(n) + ((size_t)1 << kMaxDigitBits)) * sizeof(int) + 64; }

template <class K, int DB, int KPT>
static int one_pass_kpt(int *counts, int *totals, const K *src, K *dst, const uint32_t *vsrc, uint32_t *vdst, int n, int shift,
                        int ntiles, hipStream_t stream)
{
    constexpr int BINS = 1 << DB;
    const size_t lds = (size_t)(kThreads / 64) * BINS * sizeof(int);
    count_kernel<K, DB, KPT><<<ntiles, kThreads, 0, stream>>>(src, n, shift, ntiles, counts);
    scan_rows_kernel<<<BINS, kThreads, 0, stream>>>(counts, ntiles, totals);
    if (vdst)      // (vsrc may be null: the values of the first pass are the positions themselves)
        scatter_kernel<K, DB, KPT, true><<<ntiles, kThreads, lds, stream>>>(src, vsrc, dst, vdst, n, shift, ntiles, counts, totals);
    else
        scatter_kernel<K, DB, KPT, false><<<ntiles, kThreads, lds, stream>>>(src, nullptr, dst, nullptr, n, shift, ntiles, counts, totals);
    return PLX_OK;
}

template <class K, int DB>
static int one_pass(int *counts, int *totals, const K *src, K *dst, const uint32_t *vsrc, uint32_t *vdst, int n, int shift,
                    int ntiles, hipStream_t stream)
{
    return keys_per_thread(n) == 8 ? one_pass_kpt<K, DB, 8>(counts, totals, src, dst, vsrc, vdst, n, shift, ntiles, stream)
                                   : one_pass_kpt<K, DB, 16>(counts, totals, src, dst, vsrc, vdst, n, shift, ntiles, stream);
}

// Digit width for end_bit key bits of n keys.  Measured (MI355X, us per pass, 8- / 10-bit digits): n = 1e6 64-bit keys
// 21 / 33, n = 2.8e6 32-bit keys 34 / 40, n = 9e6 85 / 89: a 10-bit pass costs 1.5x an 8-bit one on small inputs (more
// bins than keys per wave: scattered writes) and 1.05 - 1.2x on large ones, so wide digits pay only when they save a
// third of the passes (19 bits: 2 instead of 3) or the input is large.
inline int digit_bits(int64_t n, int end_bit)
{
    int best = 8;
    double best_cost = 1e30;
    for (int db = 8; db <= kMaxDigitBits; ++db) {
        const int passes = (end_bit + db - 1) / db;
        const double cost = passes * (1.0 + (db - 8) * (n < 2000000 ? 0.25 : 0.08));
        if (cost < best_cost - 1e-9) { best = db; best_cost = cost; }
    }
    return best;
}

// Sorts bits [0, end_bit) of the keys in passes of digit_bits(n, end_bit) bits, ping-ponging between
// the two buffer pairs; *in_second tells where the result is (0: keys_a / vals_a, 1: keys_b / vals_b).  vals may be null
// (keys only).  n <= 2^31 - 4096.  temp: temp_bytes(n).
// first_keys (optional): the first pass reads its keys from there (left untouched) and takes the values to be 0, 1, 2, ...;
// the later passes ping-pong between the two buffer pairs as usual (n > 1 and end_bit > 0 required: there must BE a pass).
template <class K>
int sort_pairs(void *temp, K *keys_a, K *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n, int end_bit, int *in_second,
               hipStream_t stream, const K *first_keys = nullptr)
{
    *in_second = 0;
    if (n <= 1 || end_bit <= 0) {
        if (first_keys) { set_error("radix::sort_pairs: a sort from constant keys needs at least one pass"); return PLX_ERR_INVALID; }
        return PLX_OK;
    }
    const int ntiles = num_tiles(n);
    const int db = digit_bits(n, end_bit);
    int *counts = reinterpret_cast<int *>(temp);
    int *totals = counts + ((size_t)1 << db) * ntiles;
    const K *src = first_keys ? first_keys : keys_a;
    K *dst = keys_b;
    const uint32_t *vsrc = first_keys ? nullptr : vals_a;
    uint32_t *vdst = vals_b;
    for (int shift = 0; shift < end_bit; shift += db) {
        switch (db) {
        case 8: one_pass<K, 8>(counts, totals, src, dst, vsrc, vdst, (int)n, shift, ntiles, stream); break;
        case 9: one_pass<K, 9>(counts, totals, src, dst, vsrc, vdst, (int)n, shift, ntiles, stream); break;
        default: one_pass<K, 10>(counts, totals, src, dst, vsrc, vdst, (int)n, shift, ntiles, stream); break;
        }
        src = dst;
        dst = dst == keys_b ? keys_a : keys_b;
        vsrc = vdst;
        vdst = vdst == vals_b ? vals_a : vals_b;
        *in_second ^= 1;
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace radix
}  // namespace plx

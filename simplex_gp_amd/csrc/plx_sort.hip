// plx_sort.hip -- the sorts of the build: the stable LSD radix sort of (key, value) pairs (plx_radix.h; rocPRIM's device
// sort takes over for 64-bit keys above 3e6 items) and the per-block LDS sort that builds the block tables (rocPRIM's
// block radix sort).  Its own translation unit because the rocPRIM headers dominate compile time.
#include <cstring>
#include <algorithm>

#include "plx_internal.h"

#include <rocprim/rocprim.hpp>

#include "plx_radix.h"

namespace plx {

// 64-bit keys above kLibrarySortFrom items go to rocPRIM's Onesweep, which is bandwidth bound there and faster (4e6 keys,
// 36 bits: 252 vs 337 us); below, the three-launch passes of plx::radix win (1e6: 105 vs 172 us, 4e5: 71 vs 132 us).
constexpr int64_t kLibrarySortFrom = 3000000;

size_t radix_temp_bytes(int64_t n)
{
    size_t own = radix::temp_bytes(n), lib = 0;
    if (n > kLibrarySortFrom)
        (void)rocprim::radix_sort_pairs(nullptr, lib, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                        (uint32_t *)nullptr, (size_t)n, 0, 64u, (hipStream_t)0);
    return std::max(own, lib + 64);
}

int radix_sort_pairs64(void *temp, uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n, int end_bit,
                       int *in_second, hipStream_t stream)
{
    if (n > kLibrarySortFrom && vals_a) {
        size_t lib = 0;
        PLX_HIP_TRY(rocprim::radix_sort_pairs(nullptr, lib, keys_a, keys_b, vals_a, vals_b, (size_t)n, 0, (unsigned)end_bit, stream));
        PLX_HIP_TRY(rocprim::radix_sort_pairs(temp, lib, keys_a, keys_b, vals_a, vals_b, (size_t)n, 0, (unsigned)end_bit, stream));
        *in_second = 1;
        return PLX_OK;
    }
    return radix::sort_pairs<uint64_t>(temp, keys_a, keys_b, vals_a, vals_b, n, end_bit, in_second, stream);
}

int radix_sort_pairs32(void *temp, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n, int end_bit,
                       int *in_second, hipStream_t stream, const uint32_t *first_keys)
{
    return radix::sort_pairs<uint32_t>(temp, keys_a, keys_b, vals_a, vals_b, n, end_bit, in_second, stream, first_keys);
}

// ---- self test of the radix sort (plx_selftest_sort): keys with many duplicates, values = positions; sorted + stable?
template <class K>
__global__ void selftest_fill_kernel(K *keys, uint32_t *vals, int64_t n, K mask, uint64_t seed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const K wide = (K)h & mask;
    keys[i] = (h >> 61) ? wide : (wide & (K)0x3FF);          // one key in eight from a 1024-value pool: long runs of equal keys
    vals[i] = (uint32_t)i;
}

template <class K>
__global__ void selftest_check_kernel(const K *keys, const uint32_t *vals, int64_t n, K mask, uint64_t seed,
                                      unsigned long long *bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = (uint64_t)vals[i] * 0x9E3779B97F4A7C15ull + seed;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const K wide = (K)h & mask;
    const K want = (h >> 61) ? wide : (wide & (K)0x3FF);
    bool wrong = keys[i] != want;                              // the value still belongs to its key
    if (i > 0) wrong = wrong || keys[i - 1] > keys[i] || (keys[i - 1] == keys[i] && vals[i - 1] >= vals[i]);   // sorted, stable
    if (wrong) atomicAdd(bad, 1ull);
}

template <class K>
static int selftest_typed(int64_t n, int end_bit, uint64_t seed, hipStream_t stream, int64_t *mismatches)
{
    K *ka = nullptr, *kb = nullptr;
    uint32_t *va = nullptr, *vb = nullptr;
    void *tmp = nullptr;
    unsigned long long *bad = nullptr, h_bad = 0;
    const K mask = end_bit >= (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << end_bit) - 1);
    // every exit path frees whatever was allocated (an early PLX_HIP_TRY return used to leak the earlier buffers)
    auto body = [&]() -> int {
        PLX_HIP_TRY(hipMalloc(&ka, n * sizeof(K))); PLX_HIP_TRY(hipMalloc(&kb, n * sizeof(K)));
        PLX_HIP_TRY(hipMalloc(&va, n * 4)); PLX_HIP_TRY(hipMalloc(&vb, n * 4));
        PLX_HIP_TRY(hipMalloc(&tmp, radix_temp_bytes(n))); PLX_HIP_TRY(hipMalloc(&bad, 8));
        PLX_HIP_TRY(hipMemsetAsync(bad, 0, 8, stream));
        const int grid = (int)((n + 255) / 256);
        selftest_fill_kernel<K><<<grid, 256, 0, stream>>>(ka, va, n, mask, seed);
        int second = 0;
        PLX_TRY(sizeof(K) == 8 ? radix_sort_pairs64(tmp, (uint64_t *)ka, (uint64_t *)kb, va, vb, n, end_bit, &second, stream)
                               : radix_sort_pairs32(tmp, (uint32_t *)ka, (uint32_t *)kb, va, vb, n, end_bit, &second, stream));
        selftest_check_kernel<K><<<grid, 256, 0, stream>>>(second ? kb : ka, second ? vb : va, n, mask, seed, bad);
        PLX_HIP_TRY(hipMemcpyAsync(&h_bad, bad, 8, hipMemcpyDeviceToHost, stream));
        PLX_HIP_TRY(hipStreamSynchronize(stream));
        *mismatches = (int64_t)h_bad;
        return PLX_OK;
    };
    const int rc = body();
    if (rc != PLX_OK) (void)hipStreamSynchronize(stream);
    (void)hipFree(ka); (void)hipFree(kb); (void)hipFree(va); (void)hipFree(vb); (void)hipFree(tmp); (void)hipFree(bad);
    return rc;
}

int selftest_sort(int64_t n, int key_bytes, int end_bit, uint64_t seed, hipStream_t stream, int64_t *mismatches)
{
    return key_bytes == 8 ? selftest_typed<uint64_t>(n, end_bit, seed, stream, mismatches)
                          : selftest_typed<uint32_t>(n, end_bit, seed, stream, mismatches);
}

// ----------------------------------------------------------------------------
// Block tables (plx_block.hip), built by ONE workgroup per point block, in LDS: the block's <= 4096 (16 per thread) or 6144 (24 per thread) corners are
// loaded in (corner, point) order and sorted by vertex id with rocPRIM's block radix sort (stable, so equal vertices
// keep that order -- the order a global stable sort of (block, vertex) keys gives, which is what the other block
// sizes use); row heads / ends, block-local row numbers and every per-corner record follow from the sorted registers.
// Replaces a global radix sort of all nnz keys + a count pass + a fill pass (0.37 + 0.02 + 0.11 ms at N = 1e6, d = 8).
// The block's vertex list goes to a block-strided scratch (rows_tmp[b * cpb + row]); blk_compact_kernel moves it
// behind the row offsets once those are scanned.
// PACKED (round 5): the block-local corner index rides in the low CB bits of the key (vertex id above it) and the sort moves
// keys only, RBITS bits per pass -- half the LDS exchange traffic per pass of the (key, value) form and fewer passes than
// rocPRIM's default 4 bits (20 key bits: 5 -> 4 passes).  The sort covers the vertex bits alone and is stable, so equal
// vertices keep (corner, point) order exactly as before: same tables, bit for bit.  Needs vbits + CB <= 32 (with 2^vbits > m).
template <int E, bool PACKED, int RBITS, int T>
__global__ __launch_bounds__(T) void blk_sort_fill_kernel(const int *__restrict__ evid, const float *__restrict__ ew, int n,
                                                            int own_begin, int n_own, int P, int d1, int cpb, int vbits,
                                                            uint16_t *__restrict__ bc_pt, float *__restrict__ bc_w,
                                                            uint16_t *__restrict__ srow, int64_t sstride,
                                                            int *__restrict__ rows_tmp, int *__restrict__ rows)
{
    // a block holds 256 * E corners (E = 16 or 24), sorted by T threads with IPT = 256 E / T corners each (T = 512: half the
    // serial depth per thread of every sort pass; N = 1e6: 133 -> 108 us; T = 1024: 139)
    constexpr int IPT = 256 * E / T;
    static_assert(IPT % 4 == 0, "the blocked stores move 4 corners at a time");
    using Sort = rocprim::block_radix_sort<uint32_t, T, IPT, uint32_t, 1, 1, RBITS>;
    using SortKeys = rocprim::block_radix_sort<uint32_t, T, IPT, rocprim::empty_type, 1, 1, RBITS>;
    constexpr int CB = E == 16 ? 12 : 13;              // bits of a block-local corner index (256 * E corners)
    __shared__ union { typename Sort::storage_type pairs; typename SortKeys::storage_type keys; } storage;
    __shared__ uint32_t edge_key[T + 1];               // first key of every thread (+ a sentinel)
    __shared__ uint32_t last_key[T];                   // last key of every thread
    __shared__ int wave_sum[T / 64];
    extern __shared__ uint16_t srow_tile[];            // [d1][np] block-local row of every corner, point major per corner
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int p0 = b * P, np = min(P, n_own - p0), nc = np * d1;
    uint32_t keys[IPT], vals[IPT];                      // vals: local corner c = r * np + i
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int c = tid * IPT + j;
        if (c < nc) {
            const int r = c / np, i = c - r * np;
            keys[j] = (uint32_t)evid[(size_t)r * n + own_begin + p0 + i];
            vals[j] = (uint32_t)c;
            if (PACKED) keys[j] = (keys[j] << CB) | (uint32_t)c;
        } else {
            keys[j] = 0xFFFFFFFFu;                      // padding sorts behind every vertex (vbits <= 30)
            vals[j] = 0u;
        }
    }
    if (PACKED) {
        SortKeys().sort(keys, storage.keys, (unsigned)CB, (unsigned)(CB + vbits));   // (vbits: 2^vbits > m, so the padding's all-ones field sorts last)
#pragma unroll
        for (int j = 0; j < IPT; ++j) { vals[j] = keys[j] & ((1u << CB) - 1u); keys[j] = keys[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : keys[j] >> CB; }
    } else {
        Sort().sort(keys, vals, storage.pairs, 0, (unsigned)vbits + 1);   // + 1: the padding key's top bit must take part
    }
    // neighbours across threads: the last key of the thread before, the first key of the thread after
    edge_key[tid] = keys[0];
    last_key[tid] = keys[IPT - 1];
    if (tid == 0) edge_key[T] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t next_first = edge_key[tid + 1];
    const uint32_t prev_last = tid == 0 ? 0u : last_key[tid - 1];
    // heads: first corner of a vertex row; the block's first corner is one by construction
    int heads = 0;
    bool head[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int pos = tid * IPT + j;
        const uint32_t prev = j == 0 ? prev_last : keys[j - 1];
        head[j] = pos < nc && (pos == 0 || keys[j] != prev);
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) heads += head[j] ? 1 : 0;
    // exclusive scan of the head counts over the workgroup
    int incl = heads;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = incl - heads, total = 0;
#pragma unroll
    for (int wv = 0; wv < T / 64; ++wv) {
        if (wv < wave) before += wave_sum[wv];
        total += wave_sum[wv];
    }
    if (tid == 0) rows[b] = total;
    // per-corner records
    const size_t k0 = (size_t)b * cpb;
    int lrow = before - 1;
    uint32_t ptw[IPT / 2];
    float wq[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int pos = tid * IPT + j;
        lrow += head[j] ? 1 : 0;
        const uint32_t nxt = j + 1 < IPT ? keys[j + 1] : next_first;
        const bool live = pos < nc;
        const bool end = live && (pos + 1 == nc || nxt != keys[j]);
        const uint32_t c = vals[j];
        const uint32_t r = live ? c / (uint32_t)np : 0u, i = live ? c - r * (uint32_t)np : 0u;
        const uint32_t rec = live ? (i | (end ? 0x8000u : 0u)) : 0u;
        if (j & 1) ptw[j / 2] |= rec << 16; else ptw[j / 2] = rec;
        wq[j] = live ? ew[(size_t)r * n + own_begin + p0 + i] : 0.f;
        if (live) srow_tile[c] = (uint16_t)lrow;
        if (head[j]) rows_tmp[k0 + lrow] = (int)keys[j];
    }
    // blocked stores: IPT consecutive corners per thread = 2 IPT bytes of bc_pt, 4 IPT bytes of bc_w (the arrays have slack)
    if (tid * IPT < nc) {
        if constexpr (IPT % 8 == 0) {
#pragma unroll
            for (int q = 0; q < IPT / 8; ++q)
                *reinterpret_cast<uint4 *>(bc_pt + k0 + tid * IPT + 8 * q) = make_uint4(ptw[4 * q], ptw[4 * q + 1], ptw[4 * q + 2], ptw[4 * q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < IPT / 4; ++q)
                *reinterpret_cast<uint2 *>(bc_pt + k0 + tid * IPT + 4 * q) = make_uint2(ptw[2 * q], ptw[2 * q + 1]);
        }
#pragma unroll
        for (int q = 0; q < IPT / 4; ++q)
            *reinterpret_cast<float4 *>(bc_w + k0 + tid * IPT + 4 * q) = make_float4(wq[4 * q], wq[4 * q + 1], wq[4 * q + 2], wq[4 * q + 3]);
    }
    __syncthreads();
    for (int c = tid; c < nc; c += T) {
        const int r = c / np, i = c - r * np;
        srow[(size_t)r * sstride + p0 + i] = srow_tile[c];
    }
}

__global__ __launch_bounds__(256) void blk_compact_kernel(const int *__restrict__ rows_tmp, const int *__restrict__ brow_ptr,
                                                          int cpb, int *__restrict__ brow_vid)
{
    const int b = blockIdx.x;
    const int base = brow_ptr[b], rows = brow_ptr[b + 1] - base;
    for (int j = threadIdx.x; j < rows; j += 256) brow_vid[base + j] = rows_tmp[(size_t)b * cpb + j];
}

int sort_fill_blocks_lds(const int *evid, const float *ew, int n, int own_begin, int n_own, int P, int d1, int cpb, int vbits,
                         int64_t m_vertices, int ipt, int64_t nblocks, uint16_t *bc_pt, float *bc_w, uint16_t *srow, int64_t sstride, int *rows_tmp,
                         int *rows, hipStream_t stream)
{
    if ((ipt != 16 && ipt != 24) || cpb > 256 * ipt || vbits > 30) { set_error("sort_fill_blocks_lds: %d corners per thread, %d per block", ipt, cpb); return PLX_ERR_INVALID; }
    const int cb = ipt == 16 ? 12 : 13;
    // packed keys: the vertex field must leave the all-ones value to the padding (ids < m < 2^vb) and fit above the corner bits
    const int vb = vbits + (((int64_t)1 << vbits) == (int64_t)m_vertices ? 1 : 0);
    const int mode = (g_blk_sort != 0 && vb + cb <= 32) ? g_blk_sort : 0;      // 0: (key, value) pairs, 4 bits per pass
#define PLX_BLK_SORT(E, PACKED, RBITS, T)                                                                                       \
    blk_sort_fill_kernel<E, PACKED, RBITS, T><<<(unsigned)nblocks, T, (size_t)cpb * 2, stream>>>(                               \
        evid, ew, n, own_begin, n_own, P, d1, cpb, mode ? vb : vbits, bc_pt, bc_w, srow, sstride, rows_tmp, rows)
    // mode: 0 = (key, value) pairs, 4 bits per pass; 4 / 5 / 6 = packed keys at that many bits per pass, 256 threads;
    // 15 = packed keys, 5 bits, 512 threads
    if (ipt == 16) {
        if (mode == 0) PLX_BLK_SORT(16, false, 4, 256); else if (mode == 4) PLX_BLK_SORT(16, true, 4, 256);
        else if (mode == 6) PLX_BLK_SORT(16, true, 6, 256); else if (mode == 5) PLX_BLK_SORT(16, true, 5, 256);
        else PLX_BLK_SORT(16, true, 5, 512);
    } else {
        if (mode == 0) PLX_BLK_SORT(24, false, 4, 256); else if (mode == 4) PLX_BLK_SORT(24, true, 4, 256);
        else if (mode == 6) PLX_BLK_SORT(24, true, 6, 256); else if (mode == 5) PLX_BLK_SORT(24, true, 5, 256);
        else PLX_BLK_SORT(24, true, 5, 512);
    }
#undef PLX_BLK_SORT
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

int compact_block_rows(const int *rows_tmp, const int *brow_ptr, int cpb, int64_t nblocks, int *brow_vid, hipStream_t stream)
{
    blk_compact_kernel<<<(unsigned)nblocks, 256, 0, stream>>>(rows_tmp, brow_ptr, cpb, brow_vid);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

// plx_block.hip -- block-structured splat / slice for single-column right-hand sides (vd = 1) on lattices
// whose vertices are shared by many simplex corners (the coarse / medium regimes: m <= nnz / 2).
//
// Reference: splat value accumulation h:478-479, slice h:497-510.
//
// Why: with one column, splat and slice are 4-byte gathers -- 9e6 of them per side at N = 1e6, d = 8 -- and a
// gather that misses the 32 KB vector L1 costs one L2 request however few bytes it wants; both stages ran at the
// L2 request rate (round 1: splat 0.16, slice 0.32 of the HBM roofline with every array cache resident).  The
// points are in lattice order, so a run of consecutive points shares most of its vertices: at N = 1e6, d = 8,
// lengthscale 1 a block of 1820 points (16,380 corners) touches ~3,600 distinct vertices.  So:
//
//   build   the owned points are cut into blocks of P points; the corners of a block are sorted by vertex
//           ("block rows": one per distinct vertex of the block, R_b in total, ~0.22 nnz above), every corner
//           keeps a 15-bit block-local point index + a row-end flag (bc_pt), its weight (bc_w), and -- point
//           major, for slice -- the 16-bit block-local row it belongs to (srow);
//   splat   (1) one workgroup per block stages its P source values in LDS (this is also where the caller's
//           row order is undone: no separate gather-in pass), streams its corners (6 bytes each, 16 per
//           thread), forms w * src from LDS and reduces them by block row with a segmented scan whose tree is
//           fixed (bitwise reproducible) -> partial[R_b]; rows never cross a block, so there is no fix-up pass;
//           (2) the block rows, sorted by vertex, are reduced by vertex with the same kind of scan, one wave per
//           256 rows -> values[m]: R_b gathers instead of nnz;
//   slice   one workgroup per block gathers the values of its block rows into LDS (R_b gathers in total), then
//           every point reads its d+1 (row, weight) pairs and the values from LDS.
//
// No MFMA: this is a gather / segmented-reduce path, bound by the corner stream (6 B per corner per side).

#include "plx_kernels.h"

namespace plx {

constexpr int kBlkT = 256;       // threads per block workgroup
constexpr int kCombineRun = 256; // block rows per wave of splat_combine_kernel
constexpr int kBlkMaxP = 1024;    // most points per block (d <= 2: fewer corners than a block could hold)

// exclusive scan of one int per thread over a kBlock-thread workgroup
__device__ __forceinline__ int wg_exclusive_scan(int val, int *total)
{
    __shared__ int wsum[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = val;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        const int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - val;
}

// ----------------------------------------------------------------------------
// build helpers (the per-block LDS sort lives in plx_sort.hip)

// rows[] -> exclusive offsets (rows[nblocks] = total); counters[0] = total, counters[1] = largest block
__global__ __launch_bounds__(kBlock) void blk_scan_kernel(int *__restrict__ rows, int nblocks, int *__restrict__ counters)
{
    __shared__ int wmax[kBlock / 64];
    int carry = 0, biggest = 0;
    for (int base = 0; base < nblocks; base += kBlock) {
        const int i = base + threadIdx.x;
        const int v = (i < nblocks) ? rows[i] : 0;
        biggest = max(biggest, v);
        int total;
        const int ex = wg_exclusive_scan(v, &total);
        if (i < nblocks) rows[i] = carry + ex;
        carry += total;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) biggest = max(biggest, __shfl_xor(biggest, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = biggest;
    __syncthreads();
    if (threadIdx.x == 0) {
        int mx = 0;
        for (int w = 0; w < kBlock / 64; ++w) mx = max(mx, wmax[w]);
        rows[nblocks] = carry;
        counters[0] = carry;
        counters[1] = mx;
    }
}

// ----------------------------------------------------------------------------
// splat, stage 1: one workgroup per block.

template <int E>
__global__ __launch_bounds__(kBlkT) void splat_block_kernel(const uint16_t *__restrict__ bc_pt, const float *__restrict__ bc_w,
                                                           const int *__restrict__ brow_ptr, const float *__restrict__ src,
                                                           const uint32_t *__restrict__ perm, int own_begin, int n_own,
                                                           int nnz, int P, int cpb, float *__restrict__ partial)
{
    constexpr int T = kBlkT;
    extern __shared__ float lds_src[];                  // [P] source values of the block's points, then [rows] row sums
    __shared__ int w_cnt[T / 64];
    __shared__ float w_tail[T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int p0 = b * P, np = min(P, n_own - p0);
    const int k0 = b * cpb, nc = min(cpb, nnz - k0);
    const int kb = tid * E;
    const int base = brow_ptr[b], rows = brow_ptr[b + 1] - base;   // needed last: loaded first, off the critical path

    // corner records first: their latency overlaps the window gather below
    uint32_t ptw[E / 2];
    float w[E];
    if (kb + E <= nc) {
#pragma unroll
        for (int q = 0; q < E / 8; ++q) {
            const uint4 a = *reinterpret_cast<const uint4 *>(bc_pt + k0 + kb + 8 * q);
            ptw[4 * q] = a.x; ptw[4 * q + 1] = a.y; ptw[4 * q + 2] = a.z; ptw[4 * q + 3] = a.w;
        }
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const float4 f = *reinterpret_cast<const float4 *>(bc_w + k0 + kb + 4 * q);
            w[4 * q] = f.x; w[4 * q + 1] = f.y; w[4 * q + 2] = f.z; w[4 * q + 3] = f.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; j += 2) {
            const uint32_t a = (kb + j < nc) ? bc_pt[k0 + kb + j] : 0u;
            const uint32_t c = (kb + j + 1 < nc) ? bc_pt[k0 + kb + j + 1] : 0u;
            ptw[j / 2] = a | (c << 16);
            w[j] = (kb + j < nc) ? bc_w[k0 + kb + j] : 0.f;
            w[j + 1] = (kb + j + 1 < nc) ? bc_w[k0 + kb + j + 1] : 0.f;
        }
    }
    // source window; perm == nullptr: rows are already in lattice order
    for (int i = tid; i < np; i += T) {
        const int row = perm ? (int)perm[own_begin + p0 + i] - own_begin : p0 + i;
        lds_src[i] = src[row];
    }
    __syncthreads();

    // products, and this thread's scan element: (row ends seen, sum since the last row end)
    float prod[E];
    int cnt = 0;
    float tail = 0.f;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const uint32_t pt = (ptw[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
        prod[j] = w[j] * lds_src[pt & 0x7FFFu];
        const bool end = (pt & 0x8000u) != 0;
        tail = end ? 0.f : tail + prod[j];
        cnt += end ? 1 : 0;
    }
    // inclusive scan over the wave: combine(l, r) = (l.cnt + r.cnt, r.cnt ? r.tail : l.tail + r.tail)
    int icnt = cnt;
    float itail = tail;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ocnt = __shfl_up(icnt, off);
        const float otail = __shfl_up(itail, off);
        if (lane >= off) {
            itail = icnt > 0 ? itail : otail + itail;
            icnt += ocnt;
        }
    }
    if (lane == 63) { w_cnt[wave] = icnt; w_tail[wave] = itail; }
    int xcnt = __shfl_up(icnt, 1);
    float xtail = __shfl_up(itail, 1);
    if (lane == 0) { xcnt = 0; xtail = 0.f; }
    __syncthreads();
    int pcnt = 0;
    float ptail = 0.f;
    for (int wv = 0; wv < wave; ++wv) {                // waves before this one, left to right
        const int c = w_cnt[wv];
        const float t = w_tail[wv];
        ptail = c > 0 ? t : ptail + t;
        pcnt += c;
    }
    float run = xcnt > 0 ? xtail : ptail + xtail;       // what the open row holds when it reaches this thread
    // Row sums go through LDS (behind the source window) and leave as coalesced stores: written straight from
    // here they are one 4-byte request per row end (2.8e6 of them at N = 1e6: 6 us of the kernel).
    float *rowsum = lds_src + P;
    int r = pcnt + xcnt;                                // first row that ends in this thread
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool end = ((ptw[j / 2] >> (16 * (j & 1))) & 0x8000u) != 0;
        run += prod[j];
        if (end) { rowsum[r++] = run; run = 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < rows; j += T) partial[base + j] = rowsum[j];
}

// splat, stage 2: values[v] = sum of v's block-row partials, in block order.  The block rows are sorted by vertex
// (s2_idx, bit 31 = last row of its vertex; s2_vid = the vertex, read at row ends only).  One wave per ~kCombineRun
// entries -- a balanced share whatever the row lengths (1 .. number of blocks): the wave's range starts at the first
// vertex row that begins at or after entry w * kCombineRun (s2_wave: built with the tables), the lanes take
// consecutive entries, gather their partials and reduce them by vertex with the same fixed-tree segmented scan as
// stage 1; a row that crosses a 64-entry step is carried in a register.  (A first version with one thread per
// vertex and a serial loop over its rows ran 35 us at N = 1e6: every wave waited for its longest row.)

// DENSE: every vertex has at least one block row (the lattice's own points touch every vertex), so the block rows of
// a wave's range belong to consecutive vertices and the vertex of a row end is the wave's first vertex (s2_wave_v) plus
// the row ends counted so far -- the 4-byte-per-row s2_vid stream is not read at all and the stores of a step fall on
// consecutive floats.  !DENSE (a rank's share of a sharded lattice): the vertex comes from s2_vid at row ends.
template <bool DENSE>
__global__ __launch_bounds__(kBlock) void splat_combine_kernel(const int *__restrict__ s2_wave, const int *__restrict__ s2_wave_v,
                                                               const int *__restrict__ s2_idx, const int *__restrict__ s2_vid,
                                                               const float *__restrict__ partial, int nwaves,
                                                               float *__restrict__ values, int ntiles, int remap,
                                                               int ablate)
{
    // XCD-aware tile order: every XCD sweeps one contiguous eighth of the vertex-sorted rows, whose partials lie in
    // about one eighth of the partial array (blocks and vertices both follow the lattice order): L2-resident gathers
    ablate = PLX_DIAG_VALUE(ablate);
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63;
    const int w = tile * (kBlock / 64) + (threadIdx.x >> 6);
    if (w >= nwaves) return;                            // whole waves leave; no workgroup barrier below
    const int e0 = s2_wave[w], e1 = s2_wave[w + 1];
    int vnext = DENSE ? s2_wave_v[w] : 0;               // vertex of the next row end of this wave's range
    float carry = 0.f;                                  // what the row open at the start of a step already holds
    // EPL consecutive entries per lane (16-byte loads, aligned: the step starts at e0 rounded down and the entries
    // before e0 -- the tail of the previous wave's last row -- are masked), summed in the thread, then ONE wave scan
    // per 64 * EPL entries (a scan per 64 entries made the kernel issue-bound: 12 us with every load switched off)
    constexpr int EPL = 4;
    for (int base = e0 & ~3; base < e1; base += EPL * 64) {
        const int k0 = base + EPL * lane;
        int idx[EPL], vid[EPL];
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            int4 ix = make_int4(0, 0, 0, 0), vx = make_int4(0, 0, 0, 0);
            if (k0 + 4 * q < e1) {
                const int kq = k0 + 4 * q;
                ix = (ablate & 2) ? make_int4(kq, kq + 1, kq + 2, (int)(0x80000000u | (uint32_t)(kq + 3)))
                                  : *reinterpret_cast<const int4 *>(s2_idx + kq);
                if (!DENSE) vx = *reinterpret_cast<const int4 *>(s2_vid + kq);
            }
            idx[4 * q] = ix.x; idx[4 * q + 1] = ix.y; idx[4 * q + 2] = ix.z; idx[4 * q + 3] = ix.w;
            vid[4 * q] = vx.x; vid[4 * q + 1] = vx.y; vid[4 * q + 2] = vx.z; vid[4 * q + 3] = vx.w;
        }
        float val[EPL];
        bool end[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool live = k0 + j >= e0 && k0 + j < e1;
            end[j] = live && idx[j] < 0;
            val[j] = live ? ((ablate & 1) ? 1.0f : partial[idx[j] & 0x7FFFFFFF]) : 0.f;
        }
        // this thread's scan element (row ends seen, sum since the last row end): as in splat_block_kernel
        int icnt = 0;
        float itail = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            itail = end[j] ? 0.f : itail + val[j];
            icnt += end[j] ? 1 : 0;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int ocnt = __shfl_up(icnt, off);
            const float otail = __shfl_up(itail, off);
            if (lane >= off) {
                itail = icnt > 0 ? itail : otail + itail;
                icnt += ocnt;
            }
        }
        int xcnt = __shfl_up(icnt, 1);
        float xtail = __shfl_up(itail, 1);
        if (lane == 0) { xcnt = 0; xtail = 0.f; }
        float run = xcnt > 0 ? xtail : carry + xtail;   // what the open row holds when it reaches this thread
        int v = vnext + xcnt;                           // DENSE: vertex of this thread's first row end
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            run += val[j];
            if (end[j]) {
                if (!(ablate & 4)) values[DENSE ? v : vid[j]] = run;
                ++v;
                run = 0.f;
            }
        }
        const int tcnt = __shfl(icnt, 63);
        const float ttail = __shfl(itail, 63);
        carry = tcnt > 0 ? ttail : carry + ttail;
        vnext += tcnt;
    }
}

// The three small tables over the vertex-sorted block rows in ONE launch (round 6: they were blk_rowptr_kernel and
// blk_s2_finish_kernel, 9 + 7 us behind a launch boundary each):
//   s2_ptr[u] = first k with skeys[k] >= u: the block rows of vertex u are s2_idx[s2_ptr[u] .. s2_ptr[u+1])   (u <= m)
//   bit 31 of s2_idx[k] = last block row of its vertex                                                      (k < nrows)
//   s2_wave[w] = first entry >= w * kCombineRun that starts a vertex row, s2_wave_v[w] = the vertex of that row (w <= nwaves)
__device__ __forceinline__ int s2_lower_bound(const uint32_t *__restrict__ skeys, int nrows, uint32_t u)
{
    int lo = 0, hi = nrows;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skeys[mid] < u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void blk_s2_tables_kernel(int *__restrict__ s2_idx, const uint32_t *__restrict__ s2_vid,
                                                               int *__restrict__ s2_ptr, int nrows, int nwaves, int m,
                                                               int *__restrict__ s2_wave, int *__restrict__ s2_wave_v)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k <= m) s2_ptr[k] = s2_lower_bound(s2_vid, nrows, (uint32_t)k);
    if (k < nrows) {
        const bool end = (k + 1 == nrows) || s2_vid[k + 1] != s2_vid[k];
        if (end) s2_idx[k] |= (int)0x80000000u;
    }
    if (k <= nwaves) {
        int e = k * kCombineRun;
        if (e >= nrows) e = nrows;
        else if (e > 0 && s2_vid[e - 1] == s2_vid[e]) e = s2_lower_bound(s2_vid, nrows, s2_vid[e] + 1u);   // inside a row: it belongs to the wave before
        s2_wave[k] = e;
        s2_wave_v[k] = e < nrows ? (int)s2_vid[e] : m;
    }
}

int splat_block_impl(plx_lattice *L, const float *d_src, float *d_values, hipStream_t stream)
{
    PLX_TRY(ensure_s2(L, stream));
    const int n_own = (int)(L->own_end - L->own_begin);
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const size_t lds = ((size_t)L->blk_P + (size_t)L->blk_max_rows) * 4;
#define PLX_LAUNCH(E)                                                                                                  \
    splat_block_kernel<E><<<(unsigned)L->nblocks, kBlkT, lds, stream>>>(                                               \
        L->bc_pt.as<uint16_t>(), L->bc_w.as<float>(), L->brow_ptr.as<int>(), d_src, perm, (int)L->own_begin, n_own,    \
        (int)L->nnz, L->blk_P, L->blk_cpb, L->partial.as<float>())
    if (L->blk_E == 16) PLX_LAUNCH(16);
    else PLX_LAUNCH(24);
#undef PLX_LAUNCH
    const bool dense = L->n_shards == 1 && !L->partial_cover && g_block_dense_combine != 0;
    // vertices no owned point touches (a rank's share of a sharded lattice) have no block rows: zero them first
    if (L->n_shards != 1 || L->partial_cover) PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)L->m * 4, stream));
    const int nt = ceil_div(L->n_s2waves, kBlock / 64);
    if (dense)
        splat_combine_kernel<true><<<tile_grid(nt, g_xcd_remap), kBlock, 0, stream>>>(
            L->s2_wave.as<int>(), L->s2_wave_v.as<int>(), L->s2_idx.as<int>(), L->s2_vid.as<int>(), L->partial.as<float>(),
            (int)L->n_s2waves, d_values, nt, g_xcd_remap, PLX_DIAG_VALUE(g_block_ablate));
    else
        splat_combine_kernel<false><<<tile_grid(nt, g_xcd_remap), kBlock, 0, stream>>>(
            L->s2_wave.as<int>(), L->s2_wave_v.as<int>(), L->s2_idx.as<int>(), L->s2_vid.as<int>(), L->partial.as<float>(),
            (int)L->n_s2waves, d_values, nt, g_xcd_remap, PLX_DIAG_VALUE(g_block_ablate));
    L->kn_splat = "splat_block_kernel+splat_combine_kernel";
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ----------------------------------------------------------------------------
// slice: one workgroup per block; out = sum_r w_r * values[v_r] / (1 + 2^-d) with the block's vertex values staged
// in LDS.  Same arithmetic and order as slice_v1_kernel (plx_slice.hip).

template <int D1>
__global__ __launch_bounds__(kBlkT) void slice_block_kernel(const uint16_t *__restrict__ srow, int64_t sstride,
                                                           const float *__restrict__ ew, int n,
                                                           const int *__restrict__ brow_ptr, const int *__restrict__ brow_vid,
                                                           const float *__restrict__ values, const uint32_t *__restrict__ perm,
                                                           int own_begin, int n_own, int P, float rden,
                                                           float *__restrict__ out, const float *__restrict__ affine,
                                                           const float *__restrict__ src, int store_mode)
{
    extern __shared__ float lds_val[];                  // values of the block's rows
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int p0 = b * P, np = min(P, n_own - p0);
    const int base = brow_ptr[b], rows = brow_ptr[b + 1] - base;
    // the first point's replay records: in flight while the rows are gathered (two points per round and thread was
    // measured 10 % slower)
    uint32_t v[D1];
    float w[D1];
    const int T = blockDim.x;
    int i = tid;
    if (i < np) {
#pragma unroll
        for (int r = 0; r < D1; ++r) {
            v[r] = srow[(size_t)r * sstride + p0 + i];
            w[r] = ew[(size_t)r * n + own_begin + p0 + i];
        }
    }
    for (int j = tid; j < rows; j += T) lds_val[j] = values[brow_vid[base + j]];
    __syncthreads();
    while (i < np) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < D1; ++r) acc += w[r] * lds_val[v[r]] * rden;
        const int row = perm ? (int)perm[own_begin + p0 + i] - own_begin : p0 + i;
        if (affine) acc = affine[0] * acc + affine[1] * src[row];      // out = a K src + b src (plx_apply_affine)
        if (store_mode == 1) __builtin_nontemporal_store(acc, out + row);
        else if (store_mode == 2) __hip_atomic_store(out + row, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[row] = acc;
        i += T;
        if (i < np) {
#pragma unroll
            for (int r = 0; r < D1; ++r) {
                v[r] = srow[(size_t)r * sstride + p0 + i];
                w[r] = ew[(size_t)r * n + own_begin + p0 + i];
            }
        }
    }
}

// Caller row order on the way out, as a GATHER: out[j] = tmp[inv_perm[j]] (+ the affine epilogue), 4 rows per thread.
// Scattering the results from the slice kernel costs one partial-line write per point that the XCD L2s cannot merge
// (36 MB of write traffic for a 4 MB output, +10 us at N = 1e6); reading 4 bytes from a random place is cheaper than
// writing 4 bytes to one, and the stores here are whole 16-byte vectors.
__global__ __launch_bounds__(kBlock) void unpermute_kernel(const float *__restrict__ tmp, const uint32_t *__restrict__ inv,
                                                           int n_own, float *__restrict__ out,
                                                           const float *__restrict__ affine, const float *__restrict__ src)
{
    const int j0 = (blockIdx.x * kBlock + threadIdx.x) * 4;
    if (j0 >= n_own) return;
    const bool full = j0 + 4 <= n_own;
    uint32_t q[4] = {0, 0, 0, 0};
    if (full) {
        const uint4 v = *reinterpret_cast<const uint4 *>(inv + j0);
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    } else {
        for (int k = 0; k < 4; ++k) q[k] = j0 + k < n_own ? inv[j0 + k] : 0u;
    }
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = tmp[q[k]];
    if (affine) {
        const float a = affine[0], b = affine[1];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (j0 + k < n_own) r[k] = a * r[k] + b * src[j0 + k];
    }
    if (full && (reinterpret_cast<uintptr_t>(out + j0) & 15) == 0) {
        *reinterpret_cast<float4 *>(out + j0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        for (int k = 0; k < 4; ++k)
            if (j0 + k < n_own) out[j0 + k] = r[k];
    }
}

// The same for rows of vd columns: tmp is [n_own][vdp] in lattice order (whole 16-byte chunks), out is the caller's
// [n_own][vd].  One thread per (caller row, chunk): a 16-byte gather, then up to four floats of the output row -- lanes
// run along the output, so the stores of a wave are one contiguous stretch.  (slice_vec_kernel writing the caller's rows
// itself scatters 4 vd-byte rows: 677 MB of memory-side writes for a 176 MB result at N = 4e6, vd = 11.)
__global__ __launch_bounds__(kBlock) void unpermute_rows_kernel(const float4 *__restrict__ tmp, const uint32_t *__restrict__ inv,
                                                                int n_own, int nch, int vd, float *__restrict__ out,
                                                                const float *__restrict__ affine, const float *__restrict__ src)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= (int64_t)n_own * nch) return;
    const int j = (int)(item / nch), ch = (int)(item - (int64_t)j * nch);
    float4 r = tmp[(size_t)inv[j] * nch + ch];
    float *o = out + (size_t)j * vd + 4 * ch;
    const int left = vd - 4 * ch;
    if (affine) {
        const float a = affine[0], b = affine[1];
        const float *sp = src + (size_t)j * vd + 4 * ch;
        r.x = a * r.x + b * sp[0];
        if (left > 1) r.y = a * r.y + b * sp[1];
        if (left > 2) r.z = a * r.z + b * sp[2];
        if (left > 3) r.w = a * r.w + b * sp[3];
    }
    if (left >= 4 && (vd & 3) == 0) {
        *reinterpret_cast<float4 *>(o) = r;
    } else {
        o[0] = r.x;
        if (left > 1) o[1] = r.y;
        if (left > 2) o[2] = r.z;
        if (left > 3) o[3] = r.w;
    }
}

// The same through LDS: a workgroup owns kBlock consecutive caller rows, i.e. ONE contiguous stretch of the output
// (kBlock * vd floats, 16-byte aligned because kBlock is a multiple of 4).  Phase 1 gathers the rows' 16-byte chunks from
// the lattice-ordered scratch into an LDS image of that stretch; phase 2 streams the image out as whole 16-byte vectors
// (affine epilogue applied there, against the equally contiguous rows of src): every line of the output is written
// whole, whatever vd -- the per-chunk form above writes a 44-byte row (vd = 11) as 11 four-byte stores.
__global__ __launch_bounds__(kBlock) void unpermute_rows_lds_kernel(const float4 *__restrict__ tmp, const uint32_t *__restrict__ inv,
                                                                    int n_own, int nch, int vd, float *__restrict__ out,
                                                                    const float *__restrict__ affine, const float *__restrict__ src)
{
    extern __shared__ float img[];          // kBlock * vd floats
    const int j0 = blockIdx.x * kBlock;
    const int rows = min(kBlock, n_own - j0);
    for (int item = threadIdx.x; item < rows * nch; item += kBlock) {
        const int r = item / nch, ch = item - r * nch;
        const float4 g = tmp[(size_t)inv[j0 + r] * nch + ch];
        float *o = img + r * vd + 4 * ch;
        const int left = vd - 4 * ch;
        o[0] = g.x;
        if (left > 1) o[1] = g.y;
        if (left > 2) o[2] = g.z;
        if (left > 3) o[3] = g.w;
    }
    __syncthreads();
    const int total = rows * vd;
    float *o = out + (size_t)j0 * vd;
    const float *sp = src ? src + (size_t)j0 * vd : nullptr;
    const float a = affine ? affine[0] : 1.f, b = affine ? affine[1] : 0.f;
    const int quads = total >> 2;
    for (int q = threadIdx.x; q < quads; q += kBlock) {
        float4 v = *reinterpret_cast<const float4 *>(img + 4 * q);
        if (affine) {
            const float4 sv = *reinterpret_cast<const float4 *>(sp + 4 * q);
            v.x = a * v.x + b * sv.x; v.y = a * v.y + b * sv.y; v.z = a * v.z + b * sv.z; v.w = a * v.w + b * sv.w;
        }
        *reinterpret_cast<float4 *>(o + 4 * q) = v;
    }
    for (int k = 4 * quads + threadIdx.x; k < total; k += kBlock) o[k] = affine ? a * img[k] + b * sp[k] : img[k];
}

int unpermute_rows(plx_lattice *L, const float *d_tmp, int vd, float *d_out, const float *d_affine, const float *d_src,
                   hipStream_t stream)
{
    const int n_own = (int)(L->own_end - L->own_begin), nch = values_stride(vd) / 4;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_out) | (d_affine ? reinterpret_cast<uintptr_t>(d_src) : 0)) & 15) == 0;
    if (g_perm_rows && vd <= 48 && aligned) {
        unpermute_rows_lds_kernel<<<ceil_div(n_own, kBlock), kBlock, (size_t)kBlock * vd * 4, stream>>>(
            reinterpret_cast<const float4 *>(d_tmp), L->inv_perm.as<uint32_t>(), n_own, nch, vd, d_out, d_affine, d_src);
        L->kn_slice = "slice_vec_kernel+unpermute_rows_lds_kernel";
        PLX_HIP_TRY(hipGetLastError());
        return PLX_OK;
    }
    unpermute_rows_kernel<<<ceil_div((int64_t)n_own * nch, kBlock), kBlock, 0, stream>>>(
        reinterpret_cast<const float4 *>(d_tmp), L->inv_perm.as<uint32_t>(), n_own, nch, vd, d_out, d_affine, d_src);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// inv_perm[caller row within the shard] = position of that row in lattice order (within the shard)
__global__ __launch_bounds__(kBlock) void inv_perm_kernel(const uint32_t *__restrict__ perm, int own_begin, int n_own,
                                                          uint32_t *__restrict__ inv)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_own) inv[perm[own_begin + i] - (uint32_t)own_begin] = (uint32_t)i;
}

int slice_block_impl(plx_lattice *L, const float *d_values, float *d_out, hipStream_t stream, const float *d_affine,
                     const float *d_src)
{
    const int n_own = (int)(L->own_end - L->own_begin);
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const size_t lds = (size_t)L->blk_max_rows * 4;
    const float rden = 1.0f / L->slice_denom;
    // caller row order: slice into a lattice-ordered scratch (coalesced), then gather the rows out
    const bool two_step = perm != nullptr && g_unpermute_gather != 0;
    float *final_out = d_out;
    const float *final_affine = d_affine;
    if (two_step) {
        PLX_TRY(ensure(L->ssrc, (size_t)n_own * 4 + 16));
        d_out = L->ssrc.as<float>();
        perm = nullptr;
        d_affine = nullptr;
    }
    switch (L->d + 1) {
#define PLX_CASE(D1)                                                                                                    \
    case D1:                                                                                                            \
        slice_block_kernel<D1><<<(unsigned)L->nblocks, kBlkT, lds, stream>>>(                                           \
            L->srow.as<uint16_t>(), L->srow_stride, L->ew.as<float>(), (int)L->n, L->brow_ptr.as<int>(),                \
            L->brow_vid.as<int>(), d_values, perm, (int)L->own_begin, n_own, L->blk_P, rden, d_out, d_affine, d_src,   \
            perm ? g_scatter_store : 0);                                                                                \
        break;
        PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9)
        PLX_CASE(10) PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17)
        PLX_CASE(18) PLX_CASE(19) PLX_CASE(20) PLX_CASE(21) PLX_CASE(22) PLX_CASE(23) PLX_CASE(24) PLX_CASE(25)
        PLX_CASE(26) PLX_CASE(27) PLX_CASE(28) PLX_CASE(29) PLX_CASE(30) PLX_CASE(31) PLX_CASE(32) PLX_CASE(33)
#undef PLX_CASE
    }
    L->kn_slice = "slice_block_kernel";
    if (two_step) {
        unpermute_kernel<<<ceil_div(ceil_div(n_own, 4), kBlock), kBlock, 0, stream>>>(d_out, L->inv_perm.as<uint32_t>(), n_own,
                                                                                    final_out, final_affine, d_src);
        L->kn_slice = "slice_block_kernel+unpermute_kernel";
    }
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ----------------------------------------------------------------------------
// Block tables for the owned points of a built lattice (evid / ew / perm final), built by their first user.

// Corners per thread of the block kernels, i.e. the block size (256 * e corners): the kernels are latency bound and a
// CU holds 8 block workgroups (32 waves), so what counts is how many rounds of 2048 workgroups a launch takes and how
// long a workgroup lives (~e): at N = 1e6, d = 8, 448-point blocks (e = 16) are 2,232 workgroups = 2 rounds, 672-point
// blocks (e = 24) 1,489 = 1 round.
// Points per block: a full block's corners (P * d1) are a whole number of threads (e corners each) and of 16-byte
// vectors of the per-corner arrays, so no thread of one block ever stores into the next block's records; at most
// kBlkMaxP (the block-local point index has 15 bits; d <= 2 has fewer corners than a block could hold).
static int block_points(int e, int d1)
{
    const int mult = e == 24 ? 48 : 16;
    const int P = (kBlkT * e / d1) / mult * mult;
    return std::min(P, kBlkMaxP / mult * mult);
}

static int choose_block_e(int n_own, int d1)
{
    if (g_block_e == 16 || g_block_e == 24) return g_block_e;
    const int64_t slots = 2048;
    int best = 16;
    int64_t best_cost = -1;
    for (int e : {16, 24}) {
        const int P = block_points(e, d1);
        if (P < 16) continue;
        const int64_t nb = ((int64_t)n_own + P - 1) / P;
        const int64_t cost = ((nb + slots - 1) / slots) * e;
        if (best_cost < 0 || cost < best_cost) { best = e; best_cost = cost; }
    }
    return best;
}

// lattice-order position of every caller row of the shard: the gather-out passes of slice (built by their first user)
int ensure_inv_perm(plx_lattice *L, hipStream_t stream)
{
    if (L->inv_perm_ready) return PLX_OK;
    const int n_own = (int)(L->own_end - L->own_begin);
    PLX_TRY(ensure(L->inv_perm, (size_t)n_own * 4 + 16));
    if (n_own > 0)
        inv_perm_kernel<<<ceil_div(n_own, kBlock), kBlock, 0, stream>>>(L->perm.as<uint32_t>(), (int)L->own_begin, n_own,
                                                                        L->inv_perm.as<uint32_t>());
    PLX_HIP_TRY(hipGetLastError());
    L->inv_perm_ready = true;
    return PLX_OK;
}

int ensure_blocks(plx_lattice *L, hipStream_t stream)
{
    if (L->blocks_ready) return PLX_OK;
    PLX_TRY(refuse_under_capture(stream, "the block tables of this lattice"));
    return build_blocks(L, stream);
}

int build_blocks(plx_lattice *L, hipStream_t stream)
{
    L->use_blocks = false;
    L->blocks_ready = true;
    L->s2_ready = false;
    const int d1 = L->d + 1;
    const int n = (int)L->n, n_own = (int)(L->own_end - L->own_begin);
    const int64_t nnz = L->nnz, m = L->m;
    if (g_block_path == 0 || nnz == 0 || m == 0) return PLX_OK;
    // the path pays when corners share vertices: with m > nnz / 2 most block rows hold a single corner and the
    // two-stage splat only adds a pass (the sparse regime keeps the vertex-sorted CSR path)
    if (g_block_path == 1 && 2 * m > nnz) return PLX_OK;
    const int E = choose_block_e(n_own, d1);
    const int P = block_points(E, d1);
    if (P < 16) return PLX_OK;
    const int cpb = P * d1;                      // corners per (full) block
    const int64_t nblocks = (n_own + P - 1) / P;
    int vbits = 1;
    while ((1ll << vbits) < m) ++vbits;
    if (vbits > 30) return PLX_OK;
    L->blk_P = P; L->blk_E = E; L->blk_cpb = cpb; L->nblocks = nblocks;
    L->srow_stride = ((int64_t)n_own + 7) & ~7ll;

    PLX_TRY(ensure(L->sort_keys_in, (size_t)nblocks * cpb * 4 + 64));      // the blocks' vertex lists, block strided
    PLX_TRY(ensure(L->bc_pt, (size_t)nnz * 2 + 128));     // slack: the last thread of the last block stores whole vectors
    PLX_TRY(ensure(L->bc_w, (size_t)nnz * 4 + 128));
    PLX_TRY(ensure(L->srow, (size_t)d1 * L->srow_stride * 2 + 64));
    PLX_TRY(ensure(L->brow_ptr, (size_t)(nblocks + 1) * 4));
    // one workgroup per block: sort in LDS, per-corner records, row counts; the vertex lists go to sort_keys_in
    PLX_TRY(sort_fill_blocks_lds(L->evid.as<int>(), splat_weights(L), n, (int)L->own_begin, n_own, P, d1, cpb, vbits, m, E, nblocks,
                                 L->bc_pt.as<uint16_t>(), L->bc_w.as<float>(), L->srow.as<uint16_t>(), L->srow_stride,
                                 L->sort_keys_in.as<int>(), L->brow_ptr.as<int>(), stream));
    blk_scan_kernel<<<1, kBlock, 0, stream>>>(L->brow_ptr.as<int>(), (int)nblocks, L->counters.as<int>() + 40);
    int h_rows[2];
    PLX_TRY(read_back(L, L->counters.as<int>() + 40, 2, h_rows, stream));   // R_b sizes the row tables
    const int64_t nrows = h_rows[0];
    L->n_brows = nrows;
    L->blk_max_rows = h_rows[1];
    if (g_block_path == 1 && 10 * nrows > 7 * nnz) return PLX_OK;   // too little sharing inside blocks: CSR path
    PLX_TRY(ensure(L->brow_vid, (size_t)nrows * 4 + 16));
    PLX_TRY(compact_block_rows(L->sort_keys_in.as<int>(), L->brow_ptr.as<int>(), cpb, nblocks, L->brow_vid.as<int>(), stream));
    PLX_TRY(ensure_inv_perm(L, stream));
    PLX_HIP_TRY(hipGetLastError());
    L->use_blocks = true;
    return PLX_OK;
}

// The second half of the tables, needed by the block splat only: the block rows sorted by vertex.
int ensure_s2(plx_lattice *L, hipStream_t stream)
{
    if (L->s2_ready) return PLX_OK;
    const int64_t nrows = L->n_brows, m = L->m;
    int vbits = 1;
    while ((1ll << vbits) < m) ++vbits;
    PLX_TRY(ensure(L->s2_idx, (size_t)nrows * 4 + 64));
    PLX_TRY(ensure(L->s2_ptr, (size_t)(m + 2) * 4));
    PLX_TRY(ensure(L->s2_vid, (size_t)nrows * 4 + 64));
    L->n_s2waves = (nrows + kCombineRun - 1) / kCombineRun;
    PLX_TRY(ensure(L->s2_wave, (size_t)(L->n_s2waves + 2) * 4));
    PLX_TRY(ensure(L->s2_wave_v, (size_t)(L->n_s2waves + 2) * 4));
    PLX_TRY(ensure(L->partial, (size_t)nrows * 4 + 16));
    PLX_TRY(ensure(L->sort_temp, radix_temp_bytes(nrows)));
    PLX_TRY(ensure(L->sort_keys_in, (size_t)nrows * 4 + 64));
    PLX_TRY(ensure(L->sort_vals_in, (size_t)nrows * 4 + 64));
    // the block rows sorted by vertex: the first pass reads brow_vid itself (which stays as it is: the block kernels
    // gather through it) with the row numbers implied -- no copy pass in front of the sort
    int second = 0;
    PLX_TRY(radix_sort_pairs32(L->sort_temp.p, L->sort_keys_in.as<uint32_t>(), L->s2_vid.as<uint32_t>(),
                               L->sort_vals_in.as<uint32_t>(), L->s2_idx.as<uint32_t>(), nrows, vbits, &second, stream,
                               nrows > 1 ? L->brow_vid.as<uint32_t>() : nullptr));
    if (nrows <= 1) {      // (nothing to sort: one row, or none)
        PLX_HIP_TRY(hipMemcpyAsync(L->s2_vid.p, L->brow_vid.p, (size_t)nrows * 4, hipMemcpyDeviceToDevice, stream));
        PLX_HIP_TRY(hipMemsetAsync(L->s2_idx.p, 0, (size_t)nrows * 4 + 4, stream));
        second = 1;
    }
    if (!second) { std::swap(L->sort_keys_in, L->s2_vid); std::swap(L->sort_vals_in, L->s2_idx); }
    blk_s2_tables_kernel<<<ceil_div(std::max<int64_t>(std::max<int64_t>(nrows, m + 1), L->n_s2waves + 1), kBlock), kBlock, 0, stream>>>(
        L->s2_idx.as<int>(), L->s2_vid.as<uint32_t>(), L->s2_ptr.as<int>(), (int)nrows, (int)L->n_s2waves, (int)m,
        L->s2_wave.as<int>(), L->s2_wave_v.as<int>());
    PLX_HIP_TRY(hipGetLastError());
    L->s2_ready = true;
    return PLX_OK;
}

// Which kernels serve a vd-column MVM on this lattice.  Builds the block tables when the answer depends on them.
// Multi-column right-hand sides keep to the vertex-sorted CSR kernels: block kernels for rows of 2..4 sixteen-byte chunks
// were built and measured in round 3 (LDS-staged vertex rows in slice, per-block segmented scan + per-vertex combine in
// splat) and lost on every lattice tried -- N = 4e6, d = 8, vd = 11: slice 381 vs 431 us in caller row order but 318 vs
// 252 us in lattice row order, splat 764 - 1020 vs 660 us ("multi-column block tables").
int choose_paths(plx_lattice *L, int vd, hipStream_t stream, bool *splat_blocks, bool *slice_blocks)
{
    *splat_blocks = *slice_blocks = false;
    if (vd != 1 || L->nnz == 0) return PLX_OK;
    PLX_TRY(ensure_blocks(L, stream));
    *splat_blocks = *slice_blocks = L->use_blocks;
    return PLX_OK;
}

// plx_prepare: every table a vd-column MVM will read, now instead of inside the first MVM
int prepare_tables(plx_lattice *L, int vd, hipStream_t stream)
{
    bool sp = false, sl = false;
    PLX_TRY(choose_paths(L, vd, stream, &sp, &sl));
    if (L->nnz == 0) return PLX_OK;
    if (sp) { PLX_TRY(ensure_s2(L, stream)); return PLX_OK; }
    if (vd == 1) {
        PLX_TRY(ensure_first(L, stream));
        if (L->use_first) return PLX_OK;
    }
    PLX_TRY(ensure_csr(L, stream));
    return PLX_OK;
}

}  // namespace plx

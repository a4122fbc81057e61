// plx_apply.hip -- the per-MVM kernels: splat, blur, slice.
//
// Reference: cpp/permutohedral.h ("h") splat value accumulation h:478-479,
// blur h:513-572, slice h:497-510.  The reference's CUDA path does the splat
// with one float atomicAdd per (point, corner, channel) and re-hashes every
// neighbour in every blur pass; here
//   splat = segmented reduction over simplex corners sorted by vertex (no
//           atomics, bitwise reproducible),
//   blur  = d+1 gather-accumulate passes over a precomputed neighbour table,
//   slice = per-point gather through SoA (vertex id, weight) planes.
// All three are HBM/cache-bandwidth bound gather stencils; no MFMA.
//
// Column tiling for vd > 1: a work item is (row, column) with the column
// fastest, VT = 2^logvt columns per tile and blockIdx.y selecting the tile, so
// the vd values of one vertex/point row are read by adjacent lanes.

#include "plx_internal.h"

#include <type_traits>

namespace plx {

static int g_splat_impl = 1;   // 0 row loop (first version), 1 segmented scan
static int g_blur_vpt = 4;     // vertices per thread in the vd = 1 blur
static int g_slice_impl = 1;   // 0 runtime loop, 1 unrolled per dimension
static int g_splat_ablate = 0; // diagnostics only: 1 no value gather, 2 no stores, 4 no row-id loads

Tunable *tunables()
{
    static Tunable t[] = {{"splat_impl", &g_splat_impl}, {"blur_vpt", &g_blur_vpt},
                          {"slice_impl", &g_slice_impl}, {"splat_ablate", &g_splat_ablate}, {nullptr, nullptr}};
    return t;
}

// ----------------------------------------------------------------------------
// splat, version 1: segmented scan.
//
// The corners of the owned points are sorted by vertex (csr_pt / csr_w; the sign
// bit of csr_pt marks the first corner of each vertex row).  A workgroup takes
// kSplatChunk consecutive corners, 4 per thread, forms w * src[point] in
// registers and runs one segmented inclusive scan over the chunk (in-thread,
// then wave shuffles, then four wave totals through LDS).  A thread whose
// corner closes a row stores the row sum; the row ids (csr_vid) are read only
// at row ends.  Rows that cross a chunk edge leave head / tail partial sums for
// splat_fixup_kernel.  The scan tree is fixed, so results are reproducible.

template <int VT>
__global__ __launch_bounds__(kBlock) void splat_scan_kernel(const int *__restrict__ csr_pt,
                                                            const float *__restrict__ csr_w,
                                                            const int *__restrict__ csr_vid,
                                                            const float *__restrict__ src, int vd, int nnz,
                                                            float *__restrict__ values,
                                                            float *__restrict__ head_partial,
                                                            float *__restrict__ tail_partial, int ablate)
{
    constexpr int EPT = kSplatChunk / kBlock;   // corners per thread
    static_assert(EPT == 4, "vector loads below assume 4 corners per thread");
    __shared__ int wave_cnt[kBlock / 64];
    __shared__ float wave_sum[kBlock / 64][VT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x;
    const int col0 = blockIdx.y * VT;
    const int k0 = c * kSplatChunk;
    const int kb = k0 + tid * EPT;

    // first round of loads: 4 corners + the corner after them (csr_pt has slack past nnz)
    int pt[EPT + 1];
    float w[EPT];
    if (kb + EPT <= nnz) {
        const int4 a = *reinterpret_cast<const int4 *>(csr_pt + kb);
        const float4 b = *reinterpret_cast<const float4 *>(csr_w + kb);
        pt[0] = a.x; pt[1] = a.y; pt[2] = a.z; pt[3] = a.w;
        w[0] = b.x; w[1] = b.y; w[2] = b.z; w[3] = b.w;
        pt[EPT] = (kb + EPT < nnz) ? csr_pt[kb + EPT] : -1;
    } else {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const bool ok = kb + j < nnz;
            pt[j] = ok ? csr_pt[kb + j] : 0;
            w[j] = ok ? csr_w[kb + j] : 0.f;
        }
        pt[EPT] = -1;
    }
    // a corner closes its row when the next corner is a head (sign bit) or the data ends
    bool head[EPT], row_ends[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        head[j] = pt[j] < 0;
        row_ends[j] = (kb + j + 1 >= nnz) ? true : (pt[j + 1] < 0);
    }
    // second round, all independent: value gathers and the row ids needed at row ends
    float p[EPT][VT];
    int vrow[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int q = pt[j] & 0x7FFFFFFF;
#pragma unroll
        for (int cc = 0; cc < VT; ++cc)
            p[j][cc] = (col0 + cc < vd) ? w[j] * ((ablate & 1) ? 1.0f : src[(size_t)q * vd + col0 + cc]) : 0.f;
        vrow[j] = (row_ends[j] && kb + j < nnz) ? ((ablate & 4) ? (kb + j) & 1023 : csr_vid[kb + j]) : 0;
    }

    // in-thread: heads, and the sum since the last head (or of all four)
    int cnt = 0;
    float run[VT];
#pragma unroll
    for (int cc = 0; cc < VT; ++cc) run[cc] = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        cnt += head[j] ? 1 : 0;
#pragma unroll
        for (int cc = 0; cc < VT; ++cc) run[cc] = head[j] ? p[j][cc] : run[cc] + p[j][cc];
    }

    // wave inclusive scan of (cnt, run): combine(left, right) = (l.cnt + r.cnt, r.cnt ? r.sum : l.sum + r.sum)
    int icnt = cnt;
    float isum[VT];
#pragma unroll
    for (int cc = 0; cc < VT; ++cc) isum[cc] = run[cc];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ocnt = __shfl_up(icnt, off);
        float osum[VT];
#pragma unroll
        for (int cc = 0; cc < VT; ++cc) osum[cc] = __shfl_up(isum[cc], off);
        if (lane >= off) {
#pragma unroll
            for (int cc = 0; cc < VT; ++cc) isum[cc] = (icnt > 0) ? isum[cc] : osum[cc] + isum[cc];
            icnt += ocnt;
        }
    }
    if (lane == 63) {
        wave_cnt[wave] = icnt;
#pragma unroll
        for (int cc = 0; cc < VT; ++cc) wave_sum[wave][cc] = isum[cc];
    }
    // exclusive value inside the wave
    int xcnt = __shfl_up(icnt, 1);
    float xsum[VT];
#pragma unroll
    for (int cc = 0; cc < VT; ++cc) xsum[cc] = __shfl_up(isum[cc], 1);
    if (lane == 0) {
        xcnt = 0;
#pragma unroll
        for (int cc = 0; cc < VT; ++cc) xsum[cc] = 0.f;
    }
    __syncthreads();
    // fold the totals of the waves before this one, left to right
    int pcnt = 0;
    float psum[VT];
#pragma unroll
    for (int cc = 0; cc < VT; ++cc) psum[cc] = 0.f;
    for (int wv = 0; wv < wave; ++wv) {
        const int wc = wave_cnt[wv];
#pragma unroll
        for (int cc = 0; cc < VT; ++cc) psum[cc] = (wc > 0) ? wave_sum[wv][cc] : psum[cc] + wave_sum[wv][cc];
        pcnt += wc;
    }
    int hc = pcnt + xcnt;                 // heads in the chunk before this thread's corners
#pragma unroll
    for (int cc = 0; cc < VT; ++cc) run[cc] = (xcnt > 0) ? xsum[cc] : psum[cc] + xsum[cc];

#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int k = kb + j;
        if (k >= nnz) break;
        if (head[j]) ++hc;
#pragma unroll
        for (int cc = 0; cc < VT; ++cc) run[cc] = head[j] ? p[j][cc] : run[cc] + p[j][cc];
        const bool chunk_ends = (j == EPT - 1 && tid == kBlock - 1);
        if ((ablate & 2) && run[0] != 12345.678f) continue;
        if (row_ends[j]) {
            if (hc == 0) {
#pragma unroll
                for (int cc = 0; cc < VT; ++cc)
                    if (col0 + cc < vd) head_partial[(size_t)c * vd + col0 + cc] = run[cc];
            } else {
#pragma unroll
                for (int cc = 0; cc < VT; ++cc)
                    if (col0 + cc < vd) values[(size_t)vrow[j] * vd + col0 + cc] = run[cc];
            }
        } else if (chunk_ends) {
            float *dst = (hc == 0) ? head_partial : tail_partial;
#pragma unroll
            for (int cc = 0; cc < VT; ++cc)
                if (col0 + cc < vd) dst[(size_t)c * vd + col0 + cc] = run[cc];
        }
    }
}

__device__ __forceinline__ bool chunk_has_head(const int *__restrict__ csr_pt, const int *__restrict__ csr_vid,
                                               int k0, int k1)
{
    return csr_pt[k0] < 0 || csr_vid[k0] != csr_vid[k1 - 1];
}

// A vertex row that starts inside chunk c and runs past its end: add the head
// partials of the chunks it covers, in chunk order.
__global__ __launch_bounds__(kBlock) void splat_scan_fixup_kernel(const int *__restrict__ csr_pt,
                                                                  const int *__restrict__ csr_vid,
                                                                  int nchunks, int nnz, int vd,
                                                                  const float *__restrict__ head_partial,
                                                                  const float *__restrict__ tail_partial,
                                                                  float *__restrict__ values)
{
    const int it = blockIdx.x * kBlock + threadIdx.x;
    if (it >= nchunks * vd) return;
    const int c = it / vd, col = it - c * vd;
    const int k0 = c * kSplatChunk, k1 = min(k0 + kSplatChunk, nnz);
    if (k1 >= nnz || csr_pt[k1] < 0 || !chunk_has_head(csr_pt, csr_vid, k0, k1)) return;
    float total = tail_partial[(size_t)c * vd + col];
    for (int c2 = c + 1; c2 < nchunks; ++c2) {
        total += head_partial[(size_t)c2 * vd + col];
        const int a = c2 * kSplatChunk, b = min(a + kSplatChunk, nnz);
        if (b >= nnz || csr_pt[b] < 0 || chunk_has_head(csr_pt, csr_vid, a, b)) break;
    }
    values[(size_t)csr_vid[k1 - 1] * vd + col] = total;
}

// ----------------------------------------------------------------------------
// splat, version 0 (kept for A/B): stage products in LDS, one work item per row.

template <int VT>
__device__ __forceinline__ void splat_emit(int v, int col, float s, int ra, int rb, int k0, int k1, int c,
                                           int vd, float *__restrict__ values,
                                           float *__restrict__ head_partial,
                                           float *__restrict__ tail_partial)
{
    if (col >= vd) return;
    const bool started_before = ra < k0, ends_after = rb > k1;
    if (!started_before && !ends_after) values[(size_t)v * vd + col] = s;
    else if (started_before) head_partial[(size_t)c * vd + col] = s;
    else tail_partial[(size_t)c * vd + col] = s;
}

template <int VT>
__global__ __launch_bounds__(kBlock) void splat_rows_kernel(const int *__restrict__ csr_pt,
                                                            const float *__restrict__ csr_w,
                                                            const int *__restrict__ row_ptr,
                                                            const int *__restrict__ csr_vid,
                                                            const float *__restrict__ src, int vd, int nnz,
                                                            float *__restrict__ values,
                                                            float *__restrict__ head_partial,
                                                            float *__restrict__ tail_partial)
{
    __shared__ float prod[kSplatChunk * VT];
    __shared__ int long_rows[kSplatChunk / 32 + 2];
    __shared__ int n_long;
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const int col0 = blockIdx.y * VT;
    const int k0 = c * kSplatChunk;
    const int k1 = min(k0 + kSplatChunk, nnz);
    const int len = k1 - k0;
    if (tid == 0) n_long = 0;

    for (int it = tid; it < len * VT; it += kBlock) {
        const int i = it / VT, cc = it % VT, col = col0 + cc;
        const int pt = csr_pt[k0 + i] & 0x7FFFFFFF;
        const float w = csr_w[k0 + i];
        prod[it] = (col < vd) ? w * src[(size_t)pt * vd + col] : 0.f;
    }
    __syncthreads();

    const int vf = csr_vid[k0], vl = csr_vid[k1 - 1];
    const int nrows = vl - vf + 1;
    for (int it = tid; it < nrows * VT; it += kBlock) {
        const int v = vf + it / VT, cc = it % VT;
        const int ra = row_ptr[v], rb = row_ptr[v + 1];
        const int a = max(ra, k0), b = min(rb, k1);
        if (b - a > 32) {
            if (cc == 0) long_rows[atomicAdd(&n_long, 1)] = v;
            continue;
        }
        float s = 0.f;
        for (int k = a; k < b; ++k) s += prod[(k - k0) * VT + cc];
        splat_emit<VT>(v, col0 + cc, s, ra, rb, k0, k1, c, vd, values, head_partial, tail_partial);
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    constexpr int EP = 64 / VT;
    for (int q = wave; q < n_long; q += kBlock / 64) {
        const int v = long_rows[q];
        const int ra = row_ptr[v], rb = row_ptr[v + 1];
        const int a = max(ra, k0), b = min(rb, k1);
        const int cc = lane % VT, eo = lane / VT;
        float s = 0.f;
        for (int k = a + eo; k < b; k += EP) s += prod[(k - k0) * VT + cc];
#pragma unroll
        for (int off = 32; off >= VT; off >>= 1) s += __shfl_xor(s, off);
        if (lane < VT)
            splat_emit<VT>(v, col0 + lane, s, ra, rb, k0, k1, c, vd, values, head_partial, tail_partial);
    }
}

__global__ __launch_bounds__(kBlock) void splat_rows_fixup_kernel(const int *__restrict__ row_ptr,
                                                                  const int *__restrict__ csr_vid,
                                                                  int nchunks, int nnz, int vd,
                                                                  const float *__restrict__ head_partial,
                                                                  const float *__restrict__ tail_partial,
                                                                  float *__restrict__ values)
{
    const int it = blockIdx.x * kBlock + threadIdx.x;
    if (it >= nchunks * vd) return;
    const int c = it / vd, col = it - c * vd;
    const int k0 = c * kSplatChunk, k1 = min(k0 + kSplatChunk, nnz);
    const int v = csr_vid[k1 - 1];
    const int ra = row_ptr[v], rb = row_ptr[v + 1];
    if (ra < k0 || rb <= k1) return;
    float total = tail_partial[(size_t)c * vd + col];
    for (int c2 = c + 1; c2 < nchunks; ++c2) {
        total += head_partial[(size_t)c2 * vd + col];
        if (rb <= min((c2 + 1) * kSplatChunk, nnz)) break;
    }
    values[(size_t)v * vd + col] = total;
}

int splat_impl(plx_lattice *L, const float *d_src, int vd, float *d_values, hipStream_t stream)
{
    const int64_t m = L->m;
    if (L->nnz == 0) {
        PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)m * vd * 4, stream));
        return PLX_OK;
    }
    const bool all_rows_touched = (L->own_begin == 0 && L->own_end == L->n);
    if (!all_rows_touched) PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)m * vd * 4, stream));
    PLX_TRY(ensure(L->head_partial, (size_t)L->nchunks * vd * 4));
    PLX_TRY(ensure(L->tail_partial, (size_t)L->nchunks * vd * 4));
    const int *pt = L->csr_pt.as<int>();
    const float *w = L->csr_w.as<float>();
    const int *vid = L->sort_keys_out.as<int>();   // sorted vertex id of every corner
    float *hp = L->head_partial.as<float>(), *tp = L->tail_partial.as<float>();
    const int nnz = (int)L->nnz, nch = (int)L->nchunks;
    if (g_splat_impl == 1) {
        const int vt = vd == 1 ? 1 : vd == 2 ? 2 : 4;
        dim3 grid((unsigned)nch, (unsigned)ceil_div(vd, vt));
        switch (vt) {
        case 1: splat_scan_kernel<1><<<grid, kBlock, 0, stream>>>(pt, w, vid, d_src, vd, nnz, d_values, hp, tp, g_splat_ablate); break;
        case 2: splat_scan_kernel<2><<<grid, kBlock, 0, stream>>>(pt, w, vid, d_src, vd, nnz, d_values, hp, tp, g_splat_ablate); break;
        default: splat_scan_kernel<4><<<grid, kBlock, 0, stream>>>(pt, w, vid, d_src, vd, nnz, d_values, hp, tp, g_splat_ablate); break;
        }
        splat_scan_fixup_kernel<<<ceil_div((int64_t)nch * vd, kBlock), kBlock, 0, stream>>>(pt, vid, nch, nnz, vd, hp,
                                                                                              tp, d_values);
    } else {
        const int vt = vd == 1 ? 1 : vd == 2 ? 2 : vd <= 4 ? 4 : 8;
        dim3 grid((unsigned)nch, (unsigned)ceil_div(vd, vt));
        const int *rp = L->row_ptr.as<int>();
        switch (vt) {
        case 1: splat_rows_kernel<1><<<grid, kBlock, 0, stream>>>(pt, w, rp, vid, d_src, vd, nnz, d_values, hp, tp); break;
        case 2: splat_rows_kernel<2><<<grid, kBlock, 0, stream>>>(pt, w, rp, vid, d_src, vd, nnz, d_values, hp, tp); break;
        case 4: splat_rows_kernel<4><<<grid, kBlock, 0, stream>>>(pt, w, rp, vid, d_src, vd, nnz, d_values, hp, tp); break;
        default: splat_rows_kernel<8><<<grid, kBlock, 0, stream>>>(pt, w, rp, vid, d_src, vd, nnz, d_values, hp, tp); break;
        }
        splat_rows_fixup_kernel<<<ceil_div((int64_t)nch * vd, kBlock), kBlock, 0, stream>>>(rp, vid, nch, nnz, vd, hp,
                                                                                              tp, d_values);
    }
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ----------------------------------------------------------------------------
// blur: one Jacobi pass along one lattice axis,
//   out[i] = sum_{nid=-r..r} c[nid+r] * old[nbr(i, nid)]      (h:539-549)
// accumulated from zero in tap order like the reference.

template <int ORDER>   // 0 = runtime order
__global__ __launch_bounds__(kBlock) void blur_axis_kernel(const float *__restrict__ old,
                                                           float *__restrict__ out,
                                                           const int *__restrict__ nbr, int m,
                                                           int64_t mstride, int vd, int logvt,
                                                           int order_rt, TapArgs taps)
{
    const int order = ORDER > 0 ? ORDER : order_rt;
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int i = (int)(item >> logvt);
    const int col = (blockIdx.y << logvt) + (int)(item & ((1 << logvt) - 1));
    if (i >= m || col >= vd) return;
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < order; ++s) {
        const int nb = nbr[s * mstride + i];
        const float val = nb >= 0 ? old[(size_t)nb * vd + col] : 0.f;
        acc += taps.c[s] * val;
    }
    acc += taps.c[order] * old[(size_t)i * vd + col];
#pragma unroll
    for (int s = 0; s < order; ++s) {
        const int nb = nbr[(order + s) * mstride + i];
        const float val = nb >= 0 ? old[(size_t)nb * vd + col] : 0.f;
        acc += taps.c[order + 1 + s] * val;
    }
    out[(size_t)i * vd + col] = acc;
}

// vd == 1: VPT consecutive vertices per thread, 4*VPT-byte loads from every plane
template <int ORDER, int VPT>
__global__ __launch_bounds__(kBlock) void blur_axis_v1_kernel(const float *__restrict__ old,
                                                              float *__restrict__ out,
                                                              const int *__restrict__ nbr, int m,
                                                              int64_t mstride, TapArgs taps)
{
    using ivec = typename std::conditional<VPT == 4, int4, int2>::type;
    using fvec = typename std::conditional<VPT == 4, float4, float2>::type;
    const int i0 = (blockIdx.x * kBlock + threadIdx.x) * VPT;
    if (i0 >= m) return;
    if (i0 + VPT <= m) {
        int nb[2 * ORDER][VPT];
#pragma unroll
        for (int s = 0; s < 2 * ORDER; ++s) {
            const ivec v = *reinterpret_cast<const ivec *>(nbr + s * mstride + i0);
            const int *pv = reinterpret_cast<const int *>(&v);
#pragma unroll
            for (int j = 0; j < VPT; ++j) nb[s][j] = pv[j];
        }
        const fvec cv = *reinterpret_cast<const fvec *>(old + i0);
        const float *pc = reinterpret_cast<const float *>(&cv);
        float g[2 * ORDER][VPT];
#pragma unroll
        for (int s = 0; s < 2 * ORDER; ++s)
#pragma unroll
            for (int j = 0; j < VPT; ++j) g[s][j] = nb[s][j] >= 0 ? old[nb[s][j]] : 0.f;
        fvec res;
        float *pr = reinterpret_cast<float *>(&res);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < ORDER; ++s) acc += taps.c[s] * g[s][j];
            acc += taps.c[ORDER] * pc[j];
#pragma unroll
            for (int s = 0; s < ORDER; ++s) acc += taps.c[ORDER + 1 + s] * g[ORDER + s][j];
            pr[j] = acc;
        }
        *reinterpret_cast<fvec *>(out + i0) = res;
    } else {
        for (int i = i0; i < m; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < ORDER; ++s) {
                const int nbi = nbr[s * mstride + i];
                acc += taps.c[s] * (nbi >= 0 ? old[nbi] : 0.f);
            }
            acc += taps.c[ORDER] * old[i];
#pragma unroll
            for (int s = 0; s < ORDER; ++s) {
                const int nbi = nbr[(ORDER + s) * mstride + i];
                acc += taps.c[ORDER + 1 + s] * (nbi >= 0 ? old[nbi] : 0.f);
            }
            out[i] = acc;
        }
    }
}

static inline int pick_logvt(int vd) { return vd == 1 ? 0 : vd == 2 ? 1 : vd <= 4 ? 2 : vd <= 8 ? 3 : 4; }

template <int ORDER>
static void launch_blur_v1(const float *cur, float *nxt, const int *nb, int m, int64_t mstride, const TapArgs &taps,
                           hipStream_t stream)
{
    if (g_blur_vpt == 4)
        blur_axis_v1_kernel<ORDER, 4><<<ceil_div(ceil_div(m, 4), kBlock), kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, taps);
    else
        blur_axis_v1_kernel<ORDER, 2><<<ceil_div(ceil_div(m, 2), kBlock), kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, taps);
}

int blur_impl(plx_lattice *L, float *d_values, float *d_scratch, int vd, int *result_in_scratch,
              hipStream_t stream)
{
    const int m = (int)L->m, d1 = L->d + 1, order = L->order;
    const int logvt = pick_logvt(vd);
    const int vt = 1 << logvt;
    dim3 grid((unsigned)ceil_div((int64_t)m * vt, kBlock), (unsigned)ceil_div(vd, vt));
    const bool vec = (vd == 1 && order >= 1 && order <= 3 && g_blur_vpt > 1);
    float *cur = d_values, *nxt = d_scratch;
    for (int axis = 0; axis < d1; ++axis) {
        const int *nb = L->nbr.as<int>() + (size_t)axis * 2 * order * L->mstride;
        if (vec) {
            switch (order) {
            case 1: launch_blur_v1<1>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            case 2: launch_blur_v1<2>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            default: launch_blur_v1<3>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            }
        } else {
            switch (order) {
            case 1: blur_axis_kernel<1><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
            case 2: blur_axis_kernel<2><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
            case 3: blur_axis_kernel<3><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
            default: blur_axis_kernel<0><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
            }
        }
        float *t = cur; cur = nxt; nxt = t;
    }
    tmark(L, stream);
    *result_in_scratch = (cur == d_scratch) ? 1 : 0;
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ----------------------------------------------------------------------------
// slice: out[p][c] = sum_r w_r * values[v_r][c] / (1 + 2^-d)     (h:502-509)

__global__ __launch_bounds__(kBlock) void slice_kernel(const int *__restrict__ evid,
                                                       const float *__restrict__ ew, int n, int own_begin,
                                                       int n_own, int d1, const float *__restrict__ values,
                                                       int vd, int logvt, float denom,
                                                       float *__restrict__ out)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int pl = (int)(item >> logvt);
    const int col = (blockIdx.y << logvt) + (int)(item & ((1 << logvt) - 1));
    if (pl >= n_own || col >= vd) return;
    const int p = own_begin + pl;
    float acc = 0.f;
    for (int r = 0; r < d1; ++r) {
        const int v = evid[(size_t)r * n + p];
        const float w = ew[(size_t)r * n + p];
        acc += w * values[(size_t)v * vd + col] / denom;
    }
    out[(size_t)pl * vd + col] = acc;
}

// all d+1 (id, weight) loads first, then all gathers, then the ordered sum
template <int D1>
__global__ __launch_bounds__(kBlock) void slice_unrolled_kernel(const int *__restrict__ evid,
                                                                const float *__restrict__ ew, int n,
                                                                int own_begin, int n_own,
                                                                const float *__restrict__ values, int vd,
                                                                int logvt, float denom, float *__restrict__ out)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int pl = (int)(item >> logvt);
    const int col = (blockIdx.y << logvt) + (int)(item & ((1 << logvt) - 1));
    if (pl >= n_own || col >= vd) return;
    const int p = own_begin + pl;
    int v[D1];
    float w[D1], g[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) {
        v[r] = evid[(size_t)r * n + p];
        w[r] = ew[(size_t)r * n + p];
    }
#pragma unroll
    for (int r = 0; r < D1; ++r) g[r] = values[(size_t)v[r] * vd + col];
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < D1; ++r) acc += w[r] * g[r] / denom;
    out[(size_t)pl * vd + col] = acc;
}

int slice_impl(plx_lattice *L, const float *d_values, int vd, float *d_out, hipStream_t stream)
{
    const int n_own = (int)(L->own_end - L->own_begin);
    if (n_own == 0) return PLX_OK;
    const int logvt = pick_logvt(vd);
    const int vt = 1 << logvt;
    dim3 grid((unsigned)ceil_div((int64_t)n_own * vt, kBlock), (unsigned)ceil_div(vd, vt));
    const int *evid = L->evid.as<int>();
    const float *ew = L->ew.as<float>();
    const int n = (int)L->n, ob = (int)L->own_begin;
    if (g_slice_impl == 1) {
        switch (L->d + 1) {
#define PLX_CASE(D1) \
    case D1: slice_unrolled_kernel<D1><<<grid, kBlock, 0, stream>>>(evid, ew, n, ob, n_own, d_values, vd, logvt, L->slice_denom, d_out); break;
            PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9)
            PLX_CASE(10) PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17)
            PLX_CASE(18) PLX_CASE(19) PLX_CASE(20) PLX_CASE(21) PLX_CASE(22) PLX_CASE(23) PLX_CASE(24) PLX_CASE(25)
            PLX_CASE(26) PLX_CASE(27) PLX_CASE(28) PLX_CASE(29) PLX_CASE(30) PLX_CASE(31) PLX_CASE(32) PLX_CASE(33)
#undef PLX_CASE
        }
    } else {
        slice_kernel<<<grid, kBlock, 0, stream>>>(evid, ew, n, ob, n_own, L->d + 1, d_values, vd, logvt,
                                                  L->slice_denom, d_out);
    }
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

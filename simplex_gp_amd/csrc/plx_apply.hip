// plx_apply.hip -- the per-MVM kernels: splat, blur, slice.
//
// Reference: cpp/permutohedral.h ("h") splat value accumulation h:478-479,
// blur h:513-572, slice h:497-510.  The reference's CUDA path does the splat
// with one float atomicAdd per (point, corner, channel) and re-hashes every
// neighbour in every blur pass; here
//   splat = CSR segmented reduction over corners sorted by vertex (no atomics,
//           bitwise reproducible),
//   blur  = d+1 gather-accumulate passes over a precomputed neighbour table,
//   slice = per-point gather through SoA (vertex id, weight) planes.
// All three are HBM/cache-bandwidth bound gather stencils; no MFMA.
//
// Column tiling: a work item is (row, column) with the column fastest, VT =
// 2^logvt columns per tile and blockIdx.y selecting the tile, so the vd values
// of one vertex/point row are read by adjacent lanes.

#include "plx_internal.h"

namespace plx {

// ----------------------------------------------------------------------------
// splat

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

// One workgroup stages kSplatChunk consecutive CSR entries (w * src[point]) in
// LDS, then sums each vertex row that intersects the chunk.  Rows wholly inside
// the chunk are stored; the two rows that may cross the chunk edges leave
// partial sums for splat_fixup_kernel.  Short rows: one work item per (row,
// column).  Rows longer than 32 entries: one wave per row.
template <int VT>
__global__ __launch_bounds__(kBlock) void splat_kernel(const int *__restrict__ csr_pt,
                                                       const float *__restrict__ csr_w,
                                                       const int *__restrict__ row_ptr,
                                                       const int *__restrict__ chunk_first,
                                                       const int *__restrict__ chunk_last,
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
        const int pt = csr_pt[k0 + i];
        const float w = csr_w[k0 + i];
        prod[it] = (col < vd) ? w * src[(size_t)pt * vd + col] : 0.f;
    }
    __syncthreads();

    const int vf = chunk_first[c], vl = chunk_last[c];
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
    constexpr int EP = 64 / VT;   // entries per wave step
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

// A vertex row that starts in chunk c and runs past its end: add the partial
// sums of the chunks it covers, in chunk order (deterministic).
__global__ __launch_bounds__(kBlock) void splat_fixup_kernel(const int *__restrict__ row_ptr,
                                                             const int *__restrict__ chunk_last,
                                                             int nchunks, int nnz, int vd,
                                                             const float *__restrict__ head_partial,
                                                             const float *__restrict__ tail_partial,
                                                             float *__restrict__ values)
{
    const int it = blockIdx.x * kBlock + threadIdx.x;
    if (it >= nchunks * vd) return;
    const int c = it / vd, col = it - c * vd;
    const int v = chunk_last[c];
    const int k0 = c * kSplatChunk, k1 = min(k0 + kSplatChunk, nnz);
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
    const int vt = vd == 1 ? 1 : vd == 2 ? 2 : vd <= 4 ? 4 : 8;
    dim3 grid((unsigned)L->nchunks, (unsigned)ceil_div(vd, vt));
#define PLX_SPLAT(VT)                                                                                   \
    splat_kernel<VT><<<grid, kBlock, 0, stream>>>(L->csr_pt.as<int>(), L->csr_w.as<float>(),            \
                                                  L->row_ptr.as<int>(), L->chunk_first.as<int>(),       \
                                                  L->chunk_last.as<int>(), d_src, vd, (int)L->nnz,      \
                                                  d_values, L->head_partial.as<float>(),                \
                                                  L->tail_partial.as<float>())
    switch (vt) {
    case 1: PLX_SPLAT(1); break;
    case 2: PLX_SPLAT(2); break;
    case 4: PLX_SPLAT(4); break;
    default: PLX_SPLAT(8); break;
    }
#undef PLX_SPLAT
    splat_fixup_kernel<<<ceil_div(L->nchunks * vd, kBlock), kBlock, 0, stream>>>(
        L->row_ptr.as<int>(), L->chunk_last.as<int>(), (int)L->nchunks, (int)L->nnz, vd,
        L->head_partial.as<float>(), L->tail_partial.as<float>(), d_values);
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

static inline int pick_logvt(int vd) { return vd == 1 ? 0 : vd == 2 ? 1 : vd <= 4 ? 2 : vd <= 8 ? 3 : 4; }

int blur_impl(plx_lattice *L, float *d_values, float *d_scratch, int vd, int *result_in_scratch,
              hipStream_t stream)
{
    const int m = (int)L->m, d1 = L->d + 1, order = L->order;
    const int logvt = pick_logvt(vd);
    const int vt = 1 << logvt;
    dim3 grid((unsigned)ceil_div((int64_t)m * vt, kBlock), (unsigned)ceil_div(vd, vt));
    float *cur = d_values, *nxt = d_scratch;
    for (int axis = 0; axis < d1; ++axis) {
        const int *nb = L->nbr.as<int>() + (size_t)axis * 2 * order * L->mstride;
        switch (order) {
        case 1: blur_axis_kernel<1><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
        case 2: blur_axis_kernel<2><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
        case 3: blur_axis_kernel<3><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
        default: blur_axis_kernel<0><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, L->mstride, vd, logvt, order, L->taps); break;
        }
        float *t = cur; cur = nxt; nxt = t;
    }
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

int slice_impl(plx_lattice *L, const float *d_values, int vd, float *d_out, hipStream_t stream)
{
    const int n_own = (int)(L->own_end - L->own_begin);
    if (n_own == 0) return PLX_OK;
    const int logvt = pick_logvt(vd);
    const int vt = 1 << logvt;
    dim3 grid((unsigned)ceil_div((int64_t)n_own * vt, kBlock), (unsigned)ceil_div(vd, vt));
    slice_kernel<<<grid, kBlock, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), (int)L->n,
                                              (int)L->own_begin, n_own, L->d + 1, d_values, vd, logvt,
                                              L->slice_denom, d_out);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

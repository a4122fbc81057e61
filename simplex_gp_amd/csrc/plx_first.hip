// plx_first.hip -- single-column splat for lattices on which almost every corner owns its vertex (fine regimes:
// m >= 0.9 nnz; at N = 1e6, d = 8, lengthscale 0.25 99.2 % of the vertices are touched by exactly one corner).
//
// h:478-479 accumulates w * v per vertex.  The general path sorts the corners by vertex and runs a segmented scan
// (three kernels, 0.27 of the HBM roofline there).  Here the FIRST-TOUCH corner of every vertex -- the build knows it
// (flag_kernel: the bit mask per point that numbers the vertices, h:73-79) -- simply STORES its product, streamed in
// point order: with first-touch numbering the ids of a point's new vertices are consecutive, so the stores of
// neighbouring lanes fall on the same lines.  Every vertex has exactly one first-touch corner, so every row is written
// exactly once and nothing is zeroed.  The remaining corners (nnz - m: under 1 % in that regime) are kept as a short
// list sorted by vertex (stable radix sort over entries taken in (point, corner) order); one thread per RUN of the list
// adds its products in list order onto the stored value.  No atomics; the sum per vertex is first-touch product, then the
// extras in (point, corner) order -- fixed, reproducible.  A vertex with more than kMaxRun extras (a heavy tail: many
// points on one spot) sends the lattice back to the sorted-corner path.
#include "plx_kernels.h"

#include <algorithm>

namespace plx {

constexpr int kMaxRun = 64;

__device__ __forceinline__ int popc2(uint32_t lo, uint32_t hi) { return __popc(lo) + __popc(hi); }

// corners per 256 points that are NOT the first touch of their vertex
__global__ __launch_bounds__(kBlock) void extra_count_kernel(const uint32_t *__restrict__ flagmask, int n, int d1,
                                                             int *__restrict__ blockcnt)
{
    __shared__ int wsum[kBlock / 64];
    const int p = blockIdx.x * kBlock + threadIdx.x;
    int c = 0;
    if (p < n) c = d1 - popc2(flagmask[2 * (size_t)p], flagmask[2 * (size_t)p + 1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += wsum[w];
        blockcnt[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kBlock) void extra_scan_kernel(int *__restrict__ blockcnt, int nblocks, int *__restrict__ total)
{
    __shared__ int wsum[kBlock / 64];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblocks; base += kBlock) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? blockcnt[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < nblocks) blockcnt[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// the extras in (point, corner) order: key = vertex, value = entry index r * n + p
__global__ __launch_bounds__(kBlock) void extra_fill_kernel(const uint32_t *__restrict__ flagmask, const int *__restrict__ blockoff,
                                                            const int *__restrict__ evid, int n, int d1,
                                                            uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    __shared__ int wsum[kBlock / 64];
    const int p = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
    int c = 0;
    if (p < n) { lo = flagmask[2 * (size_t)p]; hi = flagmask[2 * (size_t)p + 1]; c = d1 - popc2(lo, hi); }
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pos = blockoff[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
    if (p >= n || c == 0) return;
    for (int r = 0; r < d1; ++r) {
        const bool first = r < 32 ? ((lo >> r) & 1u) : ((hi >> (r - 32)) & 1u);
        if (!first) {
            const size_t e = (size_t)r * n + p;
            keys[pos] = (uint32_t)evid[e];
            vals[pos] = (uint32_t)e;
            ++pos;
        }
    }
}

// sorted extras -> (vertex, point, weight) records; flag[0] |= 1 when a vertex has more than kMaxRun of them
__global__ __launch_bounds__(kBlock) void extra_finalize_kernel(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ svals,
                                                                const float *__restrict__ ew, int n, int count,
                                                                int *__restrict__ ex_vid, int *__restrict__ ex_pt,
                                                                float *__restrict__ ex_w, int *__restrict__ flag)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    const uint32_t v = skeys[j], e = svals[j];
    ex_vid[j] = (int)v;
    ex_pt[j] = (int)(e % (uint32_t)n);
    ex_w[j] = ew[e];
    if (j + kMaxRun < count && skeys[j + kMaxRun] == v) atomicOr(flag, 1);
}

// pass 1: one point per thread, every first-touch corner stores w * v at its vertex
template <int D1>
__global__ __launch_bounds__(kBlock) void splat_first_kernel(const uint32_t *__restrict__ flagmask, const int *__restrict__ evid,
                                                             const float *__restrict__ ew, const uint32_t *__restrict__ perm,
                                                             const float *__restrict__ src, int n, float *__restrict__ values,
                                                             int ntiles, int remap)
{
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int p = tile * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t lo = flagmask[2 * (size_t)p], hi = flagmask[2 * (size_t)p + 1];
    const float v = src[perm ? perm[p] : (uint32_t)p];
    int id[D1];
    float w[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) { id[r] = evid[(size_t)r * n + p]; w[r] = ew[(size_t)r * n + p]; }
#pragma unroll
    for (int r = 0; r < D1; ++r) {
        const bool first = r < 32 ? ((lo >> r) & 1u) : ((hi >> (r - 32)) & 1u);
        if (first) values[id[r]] = w[r] * v;
    }
}

// pass 1 under FIRST-TOUCH NUMBERING (h:73-79 applied to the lattice-ordered points: what the build uses on these
// lattices): ids are handed out in (point, corner) order, so the first-touch corners of a workgroup's 256 points own
// one CONTIGUOUS id range.  The products are packed into LDS at their rank inside the workgroup (one scan of the
// per-point counts) and stored as one coalesced run; no id is loaded except the range's first.
template <int D1>
__global__ __launch_bounds__(kBlock) void splat_first_seq_kernel(const uint32_t *__restrict__ flagmask, const int *__restrict__ evid,
                                                                 const float *__restrict__ ew, const uint32_t *__restrict__ perm,
                                                                 const float *__restrict__ src, int n, float *__restrict__ values,
                                                                 int ntiles, int remap)
{
    __shared__ float prod[kBlock * D1];
    __shared__ int wsum[kBlock / 64];
    __shared__ int s_base;
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int p = tile * kBlock + threadIdx.x;
    const bool live = p < n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t lo = 0, hi = 0;
    float v = 0.f, w[D1];
    if (live) {
        lo = flagmask[2 * (size_t)p]; hi = flagmask[2 * (size_t)p + 1];
        v = src[perm ? perm[p] : (uint32_t)p];
#pragma unroll
        for (int r = 0; r < D1; ++r) w[r] = ew[(size_t)r * n + p];
    }
    const int c = popc2(lo, hi);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = incl - c, total = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) {
        if (q < wave) off += wsum[q];
        total += wsum[q];
    }
    if (off == 0 && c > 0) {               // exactly one thread: the owner of the range's first id
        const int r0 = lo ? __ffs(lo) - 1 : 32 + __ffs(hi) - 1;
        s_base = evid[(size_t)r0 * n + p];
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < D1; ++r) {
            const bool first = r < 32 ? ((lo >> r) & 1u) : ((hi >> (r - 32)) & 1u);
            if (first) prod[off++] = w[r] * v;
        }
    }
    __syncthreads();
    if (total == 0) return;
    const int base = s_base;
    for (int k = threadIdx.x; k < total; k += kBlock) values[base + k] = prod[k];
}

// pass 2: one thread per run of the sorted extras
__global__ __launch_bounds__(kBlock) void splat_extras_kernel(const int *__restrict__ ex_vid, const int *__restrict__ ex_pt,
                                                              const float *__restrict__ ex_w, const uint32_t *__restrict__ perm,
                                                              const float *__restrict__ src, int count, float *__restrict__ values)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    const int v = ex_vid[j];
    if (j > 0 && ex_vid[j - 1] == v) return;
    float acc = values[v];
    for (int q = j; q < count && ex_vid[q] == v; ++q) {
        const int p = ex_pt[q];
        acc += ex_w[q] * src[perm ? perm[p] : (uint32_t)p];
    }
    values[v] = acc;
}

static int ensure_first_body(plx_lattice *L, hipStream_t stream);

// first_ready is set only once the list is built (or known not to apply): a transient failure -- an allocation, the
// read-back -- is reported and the next call tries again, instead of pinning the lattice to the sorted-corner path for the
// rest of its build with n_extra left stale
int ensure_first(plx_lattice *L, hipStream_t stream)
{
    if (L->first_ready) return PLX_OK;
    PLX_TRY(refuse_under_capture(stream, "the first-touch splat list of this lattice"));
    L->use_first = false;
    L->n_extra = 0;
    const int rc = ensure_first_body(L, stream);
    if (rc != PLX_OK) { L->use_first = false; L->n_extra = 0; return rc; }
    L->first_ready = true;
    return PLX_OK;
}

static int ensure_first_body(plx_lattice *L, hipStream_t stream)
{
    const int64_t nnz = L->nnz, m = L->m;
    const int n = (int)L->n, d1 = L->d + 1;
    if (g_splat_first == 0 || L->n_shards != 1 || L->partial_cover || !L->flags_valid || nnz == 0 || nnz != (int64_t)n * d1) return PLX_OK;
    const int64_t extra = nnz - m;
    if (extra < 0 || (g_splat_first == 1 && 10 * extra > nnz)) return PLX_OK;
    L->n_extra = extra;
    if (extra == 0) { L->use_first = true; return PLX_OK; }
    const int nblk = ceil_div(n, kBlock);
    PLX_TRY(ensure(L->blockcnt, (size_t)(nblk + 1) * 4));
    PLX_TRY(ensure(L->sort_keys_in, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->sort_vals_in, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->sort_vals_out, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->ex_vid, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->ex_pt, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->ex_w, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->ex_keys, (size_t)extra * 4 + 16));
    PLX_TRY(ensure(L->sort_temp, radix_temp_bytes(extra)));
    int *cnt = L->counters.as<int>() + 56;       // {total, long-run flag}
    PLX_HIP_TRY(hipMemsetAsync(cnt, 0, 8, stream));
    extra_count_kernel<<<nblk, kBlock, 0, stream>>>(L->flagmask.as<uint32_t>(), n, d1, L->blockcnt.as<int>());
    extra_scan_kernel<<<1, kBlock, 0, stream>>>(L->blockcnt.as<int>(), nblk, cnt);
    extra_fill_kernel<<<nblk, kBlock, 0, stream>>>(L->flagmask.as<uint32_t>(), L->blockcnt.as<int>(), L->evid.as<int>(), n, d1,
                                                   L->sort_keys_in.as<uint32_t>(), L->sort_vals_in.as<uint32_t>());
    int end_bit = 1;
    while ((1ll << end_bit) < m) ++end_bit;
    int second = 0;
    PLX_TRY(radix_sort_pairs32(L->sort_temp.p, L->sort_keys_in.as<uint32_t>(), L->ex_keys.as<uint32_t>(), L->sort_vals_in.as<uint32_t>(),
                               L->sort_vals_out.as<uint32_t>(), extra, end_bit, &second, stream));
    const uint32_t *sk = second ? L->ex_keys.as<uint32_t>() : L->sort_keys_in.as<uint32_t>();
    const uint32_t *sv = second ? L->sort_vals_out.as<uint32_t>() : L->sort_vals_in.as<uint32_t>();
    extra_finalize_kernel<<<ceil_div(extra, kBlock), kBlock, 0, stream>>>(sk, sv, splat_weights(L), n, (int)extra, L->ex_vid.as<int>(),
                                                                         L->ex_pt.as<int>(), L->ex_w.as<float>(), cnt + 1);
    PLX_HIP_TRY(hipGetLastError());
    int h[2];
    PLX_TRY(read_back(L, cnt, 2, h, stream));
    if (h[0] != (int)extra) { set_error("ensure_first: %d extras counted, %lld expected", h[0], (long long)extra); return PLX_ERR_STATE; }
    L->use_first = h[1] == 0;
    return PLX_OK;
}

int splat_first_impl(plx_lattice *L, const float *d_src, float *d_values, hipStream_t stream)
{
    const int n = (int)L->n;
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const int nt = ceil_div(n, kBlock);
    const int grid = tile_grid(nt, g_xcd_remap);
    const bool seq = L->vertex_order == 0 && g_splat_first != 3;      // first-touch numbering: contiguous id ranges (3: A/B switch)
    switch (L->d + 1) {
#define PLX_CASE(D1) \
    case D1: \
        if (seq) splat_first_seq_kernel<D1><<<grid, kBlock, 0, stream>>>(L->flagmask.as<uint32_t>(), L->evid.as<int>(), splat_weights(L), perm, d_src, n, d_values, nt, g_xcd_remap); \
        else splat_first_kernel<D1><<<grid, kBlock, 0, stream>>>(L->flagmask.as<uint32_t>(), L->evid.as<int>(), splat_weights(L), perm, d_src, n, d_values, nt, g_xcd_remap); \
        break;
        PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9)
        PLX_CASE(10) PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17)
        PLX_CASE(18) PLX_CASE(19) PLX_CASE(20) PLX_CASE(21) PLX_CASE(22) PLX_CASE(23) PLX_CASE(24) PLX_CASE(25)
        PLX_CASE(26) PLX_CASE(27) PLX_CASE(28) PLX_CASE(29) PLX_CASE(30) PLX_CASE(31) PLX_CASE(32) PLX_CASE(33)
#undef PLX_CASE
    }
    if (L->n_extra > 0)
        splat_extras_kernel<<<ceil_div(L->n_extra, kBlock), kBlock, 0, stream>>>(L->ex_vid.as<int>(), L->ex_pt.as<int>(), L->ex_w.as<float>(),
                                                                                perm, d_src, (int)L->n_extra, d_values);
    L->kn_splat = seq ? (L->n_extra > 0 ? "splat_first_seq_kernel+splat_extras_kernel" : "splat_first_seq_kernel")
                      : (L->n_extra > 0 ? "splat_first_kernel+splat_extras_kernel" : "splat_first_kernel");
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

// plx_lanczos.hip -- what a Lanczos step does next to its MVM (the variance cache of the reference's evaluation:
// gpytorch.settings.fast_pred_var + max_root_decomposition_size(lanc_iter), experiments/train_simplexgp.py:63-72):
// full re-orthogonalisation of w = A q_i against the basis q_0 .. q_i, the two recurrence coefficients, the next
// basis vector.  In torch this is ~19 launches per step (a gemv pair, a dot, a norm, a dozen element-wise and indexing
// kernels); at the sizes of the reference's UCI sets a launch is ~3.6 us of GPU time whatever it does (N = 10,623,
// d = 18: the MVM's own 21 launches take 75 us, the 19 around it 68 us).  Here it is four launches and two streams of
// the basis (what the gemv pair reads):
//
//   project           p0[g][j]  = sum over the rows of group g of Q[j][r] w[r]                       j = i - 1, i
//   subtract+project  c = sum_g p0[g];  w -= c_{i-1} Q[i-1] + c_i Q[i];  p1[g][j] = sum Q[j][r] w[r]     j <= i
//   subtract+norm     c2 = sum_g p1[g]; w -= sum_{j <= i} c2_j Q[j]; s[g] = sum w[r]^2;  alpha_i = c_i + c2_i
//   scale             beta_i = sqrt(sum_g s[g]);  Q[i+1] = w / beta_i
//
// i.e. first the two directions in which w is large (the alpha q_i and beta q_{i-1} terms of the three-term recurrence),
// then one classical Gram-Schmidt pass against the whole basis.  The order matters: a Gram-Schmidt pass leaves
// -E c in w (E = Q^T Q - I, c the coefficients it removed), so with the large coefficients alpha, beta still in w the
// departure of q_i from orthogonality to an early q_j is multiplied by alpha / beta_i per step -- measured in fp32 on a
// diagonal-plus-low-rank operator: |Q^T Q - I| = 0.77 after 40 steps with the full pass first, 4e-7 with the two large
// terms removed first (the torch form's order), 5e-7 with two full passes (a third stream of the basis).  Rows are split
// into at most kLzMaxGroups groups of whole workgroup spans; every sum over groups is taken redundantly by each
// workgroup of the next launch in a fixed order (no atomics, no "last block" tickets): the step is deterministic.
#include "plx_internal.h"

#include <algorithm>

namespace plx {

constexpr int kLzMaxRows = 256;       // basis vectors a step can project on (the reference's lanc_iter default is 100)
constexpr int kLzMaxGroups = 256;

struct LzShape {
    int threads, span, groups;
};

// a workgroup of 16 waves owns `span` consecutive rows.  Small problems take many short spans (a launch is latency there,
// parallelism is what hides it: at span 256 the 16 waves share the basis rows of the projection and the four quarters of
// the workgroup share the j's of the subtraction); large ones at most kLzMaxGroups spans, so that the sums over groups
// stay a few KB per workgroup.  More than 8192 x kLzMaxGroups = 2,097,152 rows: not served (plx_lanczos_work_floats < 0).
static LzShape lanczos_shape(int64_t n)
{
    LzShape s;
    s.threads = 1024;
    if (n <= 256 * (int64_t)kLzMaxGroups) s.span = 256;
    else if (n <= 1024 * (int64_t)kLzMaxGroups) s.span = 1024;
    else if (n <= 4096 * (int64_t)kLzMaxGroups) s.span = 4096;
    else s.span = 8192;
    s.groups = (int)std::min<int64_t>(1 << 30, std::max<int64_t>(1, ceil_div(n, (int64_t)s.span)));
    return s;
}

__device__ __forceinline__ float lz_wave_sum(float a)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) a += __shfl_xor(a, off);
    return a;
}

constexpr int kLzThreads = 1024;

// c[j] = sum over g < groups of partial[g][j], first <= j < rows (0 elsewhere), in LDS; fixed order: the groups of a thread slice strided, then
// the four slices in sequence.  red: kLzThreads floats of LDS.  Valid after the trailing barrier.
__device__ __forceinline__ void lz_sum_groups(const float *__restrict__ partial, int groups, int first, int rows, float *red,
                                              float *c)
{
    constexpr int JW = kLzMaxRows;
    constexpr int SL = kLzThreads / JW;                     // 4 slices of the groups
    const int jj = threadIdx.x % JW, sl = threadIdx.x / JW;
    float a0 = 0.f, a1 = 0.f;
    if (jj >= first && jj < rows) {
        int g = sl;
        for (; g + SL < groups; g += 2 * SL) {
            const float p0 = partial[(size_t)g * kLzMaxRows + jj], p1 = partial[(size_t)(g + SL) * kLzMaxRows + jj];
            a0 += p0;
            a1 += p1;
        }
        if (g < groups) a0 += partial[(size_t)g * kLzMaxRows + jj];
    }
    red[threadIdx.x] = a0 + a1;
    __syncthreads();
    if ((int)threadIdx.x < JW) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SL; ++k) s += red[k * JW + threadIdx.x];
        c[threadIdx.x] = ((int)threadIdx.x >= first && (int)threadIdx.x < rows) ? s : 0.f;
    }
    __syncthreads();
}

// partial_out[j] = sum over this group's rows of Q[j][r] wv[r - r0]: a wave per basis row (strided), lanes across the
// group's rows (coalesced 256-byte segments of the basis row), two basis rows in flight per wave
template <int SPAN>
__device__ __forceinline__ void lz_project(const float *__restrict__ Q, int64_t ld, int first, int rows, int64_t r0, int64_t n,
                                           const float *wv, float *__restrict__ partial_out)
{
    constexpr int W = kLzThreads / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lim = (int)min((int64_t)SPAN, n - r0);
    for (int j = first + wave; j < rows; j += 2 * W) {
        const float *qa = Q + (size_t)j * ld + r0;
        const bool two = j + W < rows;
        const float *qb = two ? qa + (size_t)W * ld : qa;
        float aa = 0.f, ab = 0.f;
#pragma unroll 4
        for (int r = lane; r < lim; r += 64) {
            const float x = wv[r];
            aa += qa[r] * x;
            ab += qb[r] * x;
        }
        aa = lz_wave_sum(aa);
        ab = lz_wave_sum(ab);
        if (lane == 0) {
            partial_out[j] = aa;
            if (two) partial_out[j + W] = ab;
        }
    }
}

// wv[r] -= sum over first <= j < rows of c[j] Q[j][r0 + r] for this group's rows; returns the sum of squares of the
// entries this thread wrote.  The loads of a batch of basis rows are issued together, branch-free (rows past n read a
// clamped, valid address and are dropped afterwards): with one predicated load per row in flight the pass ran at
// 2.2 TB/s at N = 1e6, against 6 TB/s for the projection.
//   SPAN >= 4096: thread t owns 4 consecutive rows per 4096 (one 16-byte load per basis row; ld % 4 == 0);
//   SPAN == 1024: thread t owns row t;
//   SPAN == 256:  the workgroup's four quarters share the j's of a row (j = first + quarter, + 4, ...) and meet in LDS
//                 (red: kLzThreads floats), summed in quarter order by the row's first thread.
template <int SPAN>
__device__ __forceinline__ float lz_subtract(const float *__restrict__ Q, int64_t ld, int first, int rows, int64_t r0, int64_t n,
                                             const float *c, float *wv, float *red, float *__restrict__ w)
{
    float ss = 0.f;
    if constexpr (SPAN >= 4096) {
        constexpr int V = SPAN / (4 * kLzThreads);          // 16-byte pieces per thread and basis row: 1 or 2
        constexpr int U = V == 1 ? 8 : 4;                   // basis rows per batch
        float4 acc[V];
        int64_t off[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t col = r0 + 4 * ((int64_t)threadIdx.x + v * kLzThreads);
            off[v] = (col + 3 < ld ? col : 0) / 4;           // (a clamped piece is never used: its rows are >= n)
        }
        const float4 *q4 = reinterpret_cast<const float4 *>(Q);
        const int64_t ld4 = ld / 4;
        int j = first;
        for (; j + U <= rows; j += U) {
            float4 x[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) x[u][v] = q4[(int64_t)(j + u) * ld4 + off[v]];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float cj = c[j + u];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    acc[v].x += cj * x[u][v].x; acc[v].y += cj * x[u][v].y; acc[v].z += cj * x[u][v].z; acc[v].w += cj * x[u][v].w;
                }
            }
        }
        for (; j < rows; ++j) {
            const float cj = c[j];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float4 x = q4[(int64_t)j * ld4 + off[v]];
                acc[v].x += cj * x.x; acc[v].y += cj * x.y; acc[v].z += cj * x.z; acc[v].w += cj * x.w;
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int r = 4 * (threadIdx.x + v * kLzThreads);
            const float a[4] = {acc[v].x, acc[v].y, acc[v].z, acc[v].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (r0 + r + e < n) {
                    const float val = wv[r + e] - a[e];
                    wv[r + e] = val;
                    w[r0 + r + e] = val;
                    ss += val * val;
                }
        }
    } else {
        constexpr int JS = kLzThreads / SPAN;               // 1 (SPAN 1024) or 4 (SPAN 256)
        constexpr int U = 8;
        const int r = threadIdx.x % SPAN, js = threadIdx.x / SPAN;
        const bool ok = r0 + r < n;
        const float *q = Q + (ok ? r0 + r : 0);
        float acc = 0.f;
        int j = first + js;
        for (; j + (U - 1) * JS < rows; j += U * JS) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = q[(int64_t)(j + u * JS) * ld];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += c[j + u * JS] * x[u];
        }
        for (; j < rows; j += JS) acc += c[j] * q[(int64_t)j * ld];
        if constexpr (JS > 1) {
            red[threadIdx.x] = acc;
            __syncthreads();
            acc = 0.f;
            if (js == 0) {
#pragma unroll
                for (int k = 0; k < JS; ++k) acc += red[k * SPAN + r];
            }
        }
        if (js == 0 && ok) {
            const float val = wv[r] - acc;
            wv[r] = val;
            w[r0 + r] = val;
            ss = val * val;
        }
    }
    return ss;
}

template <int SPAN>
__device__ __forceinline__ void lz_stage(const float *__restrict__ w, int64_t r0, int64_t n, float *wv)
{
    for (int r = threadIdx.x; r < SPAN; r += kLzThreads) wv[r] = r0 + r < n ? w[r0 + r] : 0.f;
    __syncthreads();
}

template <int SPAN>
__global__ __launch_bounds__(kLzThreads) void lanczos_project_kernel(const float *__restrict__ Q, int64_t ld,
                                                                     const float *__restrict__ w, int64_t n, int rows,
                                                                     float *__restrict__ partial)
{
    __shared__ float wv[SPAN];
    const int64_t r0 = (int64_t)blockIdx.x * SPAN;
    lz_stage<SPAN>(w, r0, n, wv);
    lz_project<SPAN>(Q, ld, max(0, rows - 2), rows, r0, n, wv, partial + (size_t)blockIdx.x * kLzMaxRows);
}

template <int SPAN>
__global__ __launch_bounds__(kLzThreads) void lanczos_subtract_project_kernel(const float *__restrict__ Q, int64_t ld,
                                                                              float *__restrict__ w, int64_t n, int rows,
                                                                              const float *__restrict__ partial_in, int groups,
                                                                              float *__restrict__ partial_out,
                                                                              float *__restrict__ c_out)
{
    __shared__ float wv[SPAN];
    __shared__ float red[kLzThreads];
    __shared__ float c[kLzMaxRows];
    const int64_t r0 = (int64_t)blockIdx.x * SPAN;
    const int first = max(0, rows - 2);
    lz_sum_groups(partial_in, groups, first, rows, red, c);
    if (blockIdx.x == 0 && threadIdx.x == 0) c_out[0] = c[rows - 1];          // the first part of alpha_i
    lz_stage<SPAN>(w, r0, n, wv);
    lz_subtract<SPAN>(Q, ld, first, rows, r0, n, c, wv, red, w);
    __syncthreads();
    lz_project<SPAN>(Q, ld, 0, rows, r0, n, wv, partial_out + (size_t)blockIdx.x * kLzMaxRows);
}

template <int SPAN>
__global__ __launch_bounds__(kLzThreads) void lanczos_subtract_norm_kernel(const float *__restrict__ Q, int64_t ld,
                                                                           float *__restrict__ w, int64_t n, int rows,
                                                                           const float *__restrict__ partial_in, int groups,
                                                                           const float *__restrict__ c_first,
                                                                           float *__restrict__ alphas,
                                                                           float *__restrict__ sumsq)
{
    __shared__ float wv[SPAN];
    __shared__ float red[kLzThreads];
    __shared__ float c[kLzMaxRows];
    const int64_t r0 = (int64_t)blockIdx.x * SPAN;
    lz_sum_groups(partial_in, groups, 0, rows, red, c);
    if (blockIdx.x == 0 && threadIdx.x == 0) alphas[rows - 1] = c_first[0] + c[rows - 1];
    lz_stage<SPAN>(w, r0, n, wv);
    float ss = lz_subtract<SPAN>(Q, ld, 0, rows, r0, n, c, wv, red, w);
    // sum of squares of the group: within the waves, then the waves in sequence
    const float ws = lz_wave_sum(ss);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < kLzThreads / 64; ++k) s += red[k];
        sumsq[blockIdx.x] = s;
    }
}

template <int SPAN>
__global__ __launch_bounds__(kLzThreads) void lanczos_scale_kernel(const float *__restrict__ w, int64_t n,
                                                                   const float *__restrict__ sumsq, int groups,
                                                                   float *__restrict__ qnext, float *__restrict__ betas, int i)
{
    __shared__ float red[kLzMaxGroups];
    __shared__ float inv_s;
    if ((int)threadIdx.x < kLzMaxGroups) red[threadIdx.x] = (int)threadIdx.x < groups ? sumsq[threadIdx.x] : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += red[g];
        const float beta = sqrtf(s);
        inv_s = 1.f / fmaxf(beta, 1e-30f);
        if (blockIdx.x == 0) betas[i] = beta;
    }
    __syncthreads();
    const float inv = inv_s;
    const int64_t r0 = (int64_t)blockIdx.x * SPAN;
    for (int k = threadIdx.x; k < SPAN; k += kLzThreads)
        if (r0 + k < n) qnext[r0 + k] = w[r0 + k] * inv;
}

template <int SPAN>
static void lanczos_launch(float *Q, int64_t ld, float *w, int64_t n, int i, float *alphas, float *betas, float *work,
                           int groups, hipStream_t s)
{
    float *p0 = work, *p1 = work + (size_t)kLzMaxGroups * kLzMaxRows, *c0 = p1 + (size_t)kLzMaxGroups * kLzMaxRows,
          *sumsq = c0 + kLzMaxRows;
    const int rows = i + 1;
    lanczos_project_kernel<SPAN><<<groups, kLzThreads, 0, s>>>(Q, ld, w, n, rows, p0);
    lanczos_subtract_project_kernel<SPAN><<<groups, kLzThreads, 0, s>>>(Q, ld, w, n, rows, p0, groups, p1, c0);
    lanczos_subtract_norm_kernel<SPAN><<<groups, kLzThreads, 0, s>>>(Q, ld, w, n, rows, p1, groups, c0, alphas, sumsq);
    lanczos_scale_kernel<SPAN><<<groups, kLzThreads, 0, s>>>(w, n, sumsq, groups, Q + (size_t)(i + 1) * ld, betas, i);
}

} // namespace plx

using namespace plx;

extern "C" int plx_lanczos_max_rows(void) { return kLzMaxRows; }

extern "C" int64_t plx_lanczos_work_floats(int64_t n)
{
    if (n <= 0 || lanczos_shape(n).groups > kLzMaxGroups) return -1;
    return 2 * (int64_t)kLzMaxGroups * kLzMaxRows + kLzMaxRows + kLzMaxGroups;
}

extern "C" int plx_lanczos_step(float *d_q, int64_t ld, float *d_w, int64_t n, int i, float *d_alphas, float *d_betas,
                                float *d_work, void *stream)
{
    if (!d_q || !d_w || !d_alphas || !d_betas || !d_work) {
        set_error("plx_lanczos_step: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (n <= 0 || ld < n || i < 0 || i + 1 > kLzMaxRows) {
        set_error("plx_lanczos_step: n = %lld, ld = %lld, step %d (at most %d basis vectors, ld >= n)", (long long)n, (long long)ld, i,
                  kLzMaxRows);
        return PLX_ERR_INVALID;
    }
    if (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(d_q) & 15) != 0) {
        set_error("plx_lanczos_step: the basis must be 16-byte aligned with ld a multiple of 4 (ld = %lld)", (long long)ld);
        return PLX_ERR_INVALID;
    }
    const LzShape sh = lanczos_shape(n);
    if (sh.groups > kLzMaxGroups) {
        set_error("plx_lanczos_step: n = %lld is more than %lld rows", (long long)n, (long long)kLzMaxGroups * 8192);
        return PLX_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (sh.span) {
    case 256: lanczos_launch<256>(d_q, ld, d_w, n, i, d_alphas, d_betas, d_work, sh.groups, s); break;
    case 1024: lanczos_launch<1024>(d_q, ld, d_w, n, i, d_alphas, d_betas, d_work, sh.groups, s); break;
    case 4096: lanczos_launch<4096>(d_q, ld, d_w, n, i, d_alphas, d_betas, d_work, sh.groups, s); break;
    default: lanczos_launch<8192>(d_q, ld, d_w, n, i, d_alphas, d_betas, d_work, sh.groups, s); break;
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// plx_linalg.hip -- the one dense reduction the CG caller needs next to the MVM:
// column-wise dot products of two row-major [n][vd] matrices (vd = 1 + probes is
// small, n is large), deterministic.  torch's (a*b).sum(0) takes 1.3 ms on
// [1e6][11]; this is two streaming reads (~20 us).
#include "plx_internal.h"

#include <algorithm>

namespace plx {

constexpr int kDotBlocks = 1024;

// threads = (row lane, column lane), column fastest, vd lanes per row: a workgroup step covers kBlock / vd whole rows,
// i.e. consecutive threads read consecutive floats (with a power-of-two lane count per row, 5 of 16 lanes idled at
// vd = 11 and every row was its own 44-byte segment: cg_step_update 51-55 -> 49 us, 46 us being its streaming floor)
__global__ __launch_bounds__(kBlock) void coldot_partial_kernel(const float *__restrict__ a,
                                                                const float *__restrict__ b, int64_t n, int vd,
                                                                int cw, float *__restrict__ partial)
{
    __shared__ float red[kBlock];
    const int c = threadIdx.x % cw;
    const int rl = threadIdx.x / cw;
    const int rows_per_step = kBlock / cw;
    const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, n);
    float acc = 0.f;
    if (rl < rows_per_step)
        for (int64_t r = r0 + rl; r < r1; r += rows_per_step) acc += a[r * vd + c] * b[r * vd + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    // fixed-order sum over the row lanes of each column
    if (rl == 0 && c < vd) {
        float s = 0.f;
        for (int k = 0; k < rows_per_step; ++k) s += red[k * cw + c];
        partial[(size_t)blockIdx.x * vd + c] = s;
    }
}

// One workgroup per column; 1024 threads with four independent loads in flight each (the partials of the fused
// slice + dot are one row per 256-thread tile: 11,719 rows at N = 1e6, vd = 12 -- 256 threads with one load in flight
// took 11 us, twice per CG iteration).  Fixed summation order: thread-strided partial sums, then a tree.
constexpr int kFinalBlock = 1024;
__global__ __launch_bounds__(kFinalBlock) void coldot_final_kernel(const float *__restrict__ partial, int nblocks, int vd,
                                                                   float *__restrict__ out)
{
    __shared__ float red[kFinalBlock];
    const int c = blockIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = threadIdx.x;
    for (; k + 3 * kFinalBlock < nblocks; k += 4 * kFinalBlock) {
        const float p0 = partial[(size_t)k * vd + c], p1 = partial[(size_t)(k + kFinalBlock) * vd + c];
        const float p2 = partial[(size_t)(k + 2 * kFinalBlock) * vd + c], p3 = partial[(size_t)(k + 3 * kFinalBlock) * vd + c];
        a0 += p0; a1 += p1; a2 += p2; a3 += p3;
    }
    for (; k < nblocks; k += kFinalBlock) a0 += partial[(size_t)k * vd + c];
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int s = kFinalBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// CG vector updates fused into one pass each (the torch formulation is 2 addcmul_ + a column dot = three
// passes over [n][vd] plus a mul_/add_ pair for the direction).
//   update:    X += P * alpha;  R -= AP * alpha;  partial[c] += R[.][c]^2      (alpha per column)
//   direction: P  = R + P * beta                                                 (beta per column)
__global__ __launch_bounds__(kBlock) void cg_update_kernel(float *__restrict__ X, float *__restrict__ R,
                                                           const float *__restrict__ P, const float *__restrict__ AP,
                                                           const float *__restrict__ alpha, int64_t n, int vd,
                                                           int cw, float *__restrict__ partial)
{
    __shared__ float red[kBlock];
    const int c = threadIdx.x % cw;
    const int rl = threadIdx.x / cw;
    const int rows_per_step = kBlock / cw;
    const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, n);
    float acc = 0.f;
    if (rl < rows_per_step) {                            // (the last kBlock % vd threads have no row lane)
        const float a = alpha[c];
        for (int64_t r = r0 + rl; r < r1; r += rows_per_step) {
            const int64_t i = r * vd + c;
            X[i] += P[i] * a;
            const float res = R[i] - AP[i] * a;
            R[i] = res;
            acc += res * res;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c < vd) {
        float s = 0.f;
        for (int k = 0; k < rows_per_step; ++k) s += red[k * cw + c];
        partial[(size_t)blockIdx.x * vd + c] = s;
    }
}

// The same two updates with the CG coefficients formed on the device from the iteration's
// scalars (no host round trip, none of the ten tiny elementwise launches a tensor-library
// formulation of "alpha = active ? rs / pAp : 0" costs per iteration):
//   step_update:     alpha = active ? rs / max(pAp, tiny) : 0;  X += alpha P;  R -= alpha AP;  partial |R|^2
//   step_direction:  beta = active ? rs_new / max(rs, tiny) : 0;  P = R + beta P;
//                    active' = active and sqrt(rs_new) / b_norm > tol
__global__ __launch_bounds__(kBlock) void cg_step_update_kernel(float *__restrict__ X, float *__restrict__ R,
                                                                const float *__restrict__ P, const float *__restrict__ AP,
                                                                const float *__restrict__ rs, const float *__restrict__ pAp,
                                                                const float *__restrict__ active, int64_t n, int vd,
                                                                int cw, float *__restrict__ partial,
                                                                float *__restrict__ alpha_out)
{
    __shared__ float red[kBlock];
    const int c = threadIdx.x % cw;
    const int rl = threadIdx.x / cw;
    const int rows_per_step = kBlock / cw;
    const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, n);
    float acc = 0.f;
    if (rl < rows_per_step) {                            // (the last kBlock % vd threads have no row lane)
        const float a = active[c] > 0.f ? rs[c] / fmaxf(pAp[c], 1e-30f) : 0.f;
        if (blockIdx.x == 0 && rl == 0) alpha_out[c] = a;
        for (int64_t r = r0 + rl; r < r1; r += rows_per_step) {
            const int64_t i = r * vd + c;
            X[i] += P[i] * a;
            const float res = R[i] - AP[i] * a;
            R[i] = res;
            acc += res * res;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c < vd) {
        float s = 0.f;
        for (int k = 0; k < rows_per_step; ++k) s += red[k * cw + c];
        partial[(size_t)blockIdx.x * vd + c] = s;
    }
}

__global__ __launch_bounds__(kBlock) void cg_step_direction_kernel(float *__restrict__ P, const float *__restrict__ R,
                                                                   const float *__restrict__ rs_new,
                                                                   const float *__restrict__ rs,
                                                                   const float *__restrict__ active,
                                                                   const float *__restrict__ b_norm, float tol,
                                                                   int64_t total, int vd, float *__restrict__ beta_out,
                                                                   float *__restrict__ active_out)
{
    // beta per column once per workgroup; the column of element i = blockIdx * kBlock + tid from 32-bit
    // residues (a 64-bit i % vd per element costs more than the update itself)
    __shared__ float sbeta[kBlock];
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        sbeta[c] = active[c] > 0.f ? rs_new[c] / fmaxf(rs[c], 1e-30f) : 0.f;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < total) {
        const uint32_t uvd = (uint32_t)vd;
        const uint32_t bm = ((blockIdx.x % uvd) * ((uint32_t)kBlock % uvd)) % uvd;      // wave-uniform
        const uint32_t c = (bm + threadIdx.x) % uvd;
        P[i] = R[i] + P[i] * sbeta[c];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const bool on = active[c] > 0.f;
        beta_out[c] = on ? rs_new[c] / fmaxf(rs[c], 1e-30f) : 0.f;
        active_out[c] = (on && sqrtf(rs_new[c]) / b_norm[c] > tol) ? 1.f : 0.f;
    }
}

// the same four elements per thread (16-byte loads / stores; total a multiple of 4, P and R 16-byte aligned)
__global__ __launch_bounds__(kBlock) void cg_step_direction4_kernel(float4 *__restrict__ P, const float4 *__restrict__ R,
                                                                    const float *__restrict__ rs_new,
                                                                    const float *__restrict__ rs,
                                                                    const float *__restrict__ active,
                                                                    const float *__restrict__ b_norm, float tol,
                                                                    int64_t quads, int vd, float *__restrict__ beta_out,
                                                                    float *__restrict__ active_out)
{
    __shared__ float sbeta[kBlock];
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        sbeta[c] = active[c] > 0.f ? rs_new[c] / fmaxf(rs[c], 1e-30f) : 0.f;
    }
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q < quads) {
        const uint32_t uvd = (uint32_t)vd;
        // column of element 4 q = 4 (blockIdx kBlock + tid) mod vd, from 32-bit residues
        const uint32_t bm = ((blockIdx.x % uvd) * ((4u * (uint32_t)kBlock) % uvd)) % uvd;      // wave-uniform
        uint32_t c = (bm + 4u * threadIdx.x) % uvd;
        const float4 r = R[q];
        float4 p = P[q];
        p.x = r.x + p.x * sbeta[c]; c = c + 1 == uvd ? 0 : c + 1;
        p.y = r.y + p.y * sbeta[c]; c = c + 1 == uvd ? 0 : c + 1;
        p.z = r.z + p.z * sbeta[c]; c = c + 1 == uvd ? 0 : c + 1;
        p.w = r.w + p.w * sbeta[c];
        P[q] = p;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const bool on = active[c] > 0.f;
        beta_out[c] = on ? rs_new[c] / fmaxf(rs[c], 1e-30f) : 0.f;
        active_out[c] = (on && sqrtf(rs_new[c]) / b_norm[c] > tol) ? 1.f : 0.f;
    }
}

// ----------------------------------------------------------------------------
// The same two steps WITHOUT the two stand-alone reductions between them (round 6: a CG iteration was MVM + coldot_final
// + update + coldot_final + direction; each coldot_final is a 7 us kernel behind a dependent-launch boundary of about
// 5 us).  Rows of whole 16-byte chunks (vd = 4 NCH, NCH <= 4: the padded widths solvers.khat_solve runs at).
//   update_fused:    every workgroup first sums the slice kernel's per-tile partial sums of <P, AP> itself (ntiles rows
//                    of NCH chunks: 0.56 MB at N = 1e6, 12 columns -- re-read by each of the kFusedBlocks workgroups from
//                    its XCD's L2), forms alpha, then streams its share of X, R, P, AP; |R|^2 leaves as ONE partial
//                    row per workgroup ([kFusedBlocks][vd]).
//   direction_fused: every workgroup sums those kFusedBlocks rows (12 KB), forms beta and the next activity mask,
//                    then streams its share of P, R; workgroup 0 also stores rs_new, beta, active_out.
// All sums in a fixed order (thread-strided partial sums, a fixed xor butterfly over the lanes, the waves in order):
// every workgroup computes bit-identical coefficients, runs are reproducible.
constexpr int kFusedBlocks = 256;        // update_fused: one 1024-thread workgroup per CU
constexpr int kFusedThreads = 1024;
constexpr int kFusedDirBlocks = 1024;    // direction_fused: persistent 256-thread workgroups

__device__ __forceinline__ float4 f4_xor_sum(float4 a)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
    }
    return a;
}

// column sums of `rows` rows of NCH float4 chunks, by all T threads of the workgroup; result in out[4 NCH] (LDS), valid
// after the trailing barrier.  wsum: [T / 64][NCH] float4 of LDS scratch.
template <int NCH, int T, int U>
__device__ __forceinline__ void block_column_sums(const float4 *__restrict__ part, int rows, float4 *wsum, float *out)
{
    // U loads of every thread are in flight together (predicated: no tail loop whose trips wait for each other -- with 8
    // in flight and a scalar tail the prologue cost the update 8 us and the direction 3.7 us); chunk f belongs to column
    // group f % NCH, tracked without a division
    const int total = rows * NCH;                          // chunks
    float4 acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = (int)threadIdx.x % NCH;                        // chunk-in-row of this thread's current chunk
    for (int f = threadIdx.x; f < total; f += U * T) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (f + u * T < total) ? part[f + u * T] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                if (k == j) { acc[j].x += v[u].x; acc[j].y += v[u].y; acc[j].z += v[u].z; acc[j].w += v[u].w; }
            k = (k + T % NCH) % NCH;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const float4 r = f4_xor_sum(acc[j]);
        if (lane == 0) wsum[wave * NCH + j] = r;
    }
    __syncthreads();
    if ((int)threadIdx.x < 4 * NCH) {
        const int j = threadIdx.x >> 2, e = threadIdx.x & 3;
        float sacc = 0.f;
        for (int w = 0; w < T / 64; ++w) sacc += reinterpret_cast<const float *>(&wsum[w * NCH + j])[e];
        out[threadIdx.x] = sacc;
    }
    __syncthreads();
}

template <int NCH>
__global__ __launch_bounds__(kFusedThreads) void cg_step_update_fused_kernel(float *__restrict__ X, float *__restrict__ R,
                                                                             const float *__restrict__ P,
                                                                             const float *__restrict__ AP,
                                                                             const float *__restrict__ rs,
                                                                             const float4 *__restrict__ pap_partial, int ntiles,
                                                                             const float *__restrict__ active, int64_t n,
                                                                             float *__restrict__ rs_partial,
                                                                             float *__restrict__ alpha_out)
{
    constexpr int vd = 4 * NCH, T = kFusedThreads;
    __shared__ float4 wsum[(T / 64) * NCH];
    __shared__ float pap[4 * NCH], salpha[4 * NCH];
    __shared__ float red[T];
    block_column_sums<NCH, T, 12>(pap_partial, ntiles, wsum, pap);
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const float a = active[c] > 0.f ? rs[c] / fmaxf(pap[c], 1e-30f) : 0.f;
        salpha[c] = a;
        if (blockIdx.x == 0) alpha_out[c] = a;
    }
    __syncthreads();
    const int c = threadIdx.x % vd, rl = threadIdx.x / vd;
    constexpr int rows_per_step = T / vd;
    const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, n);
    float acc = 0.f;
    if (rl < rows_per_step) {
        const float a = salpha[c];
        for (int64_t r = r0 + rl; r < r1; r += rows_per_step) {
            const int64_t i = r * vd + c;
            X[i] += P[i] * a;
            const float res = R[i] - AP[i] * a;
            R[i] = res;
            acc += res * res;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        float s = 0.f;
        for (int k = 0; k < rows_per_step; ++k) s += red[k * vd + c];
        rs_partial[(size_t)blockIdx.x * vd + c] = s;
    }
}

template <int NCH>
__global__ __launch_bounds__(kBlock) void cg_step_direction_fused_kernel(float4 *__restrict__ P, const float4 *__restrict__ R,
                                                                         const float4 *__restrict__ rs_partial, int nparts,
                                                                         const float *__restrict__ rs,
                                                                         const float *__restrict__ active,
                                                                         const float *__restrict__ b_norm, float tol,
                                                                         int64_t quads, float *__restrict__ rs_new_out,
                                                                         float *__restrict__ beta_out,
                                                                         float *__restrict__ active_out)
{
    constexpr int vd = 4 * NCH;
    __shared__ float4 wsum[(kBlock / 64) * NCH];
    __shared__ float rsn[4 * NCH];
    __shared__ float4 sbeta[NCH];
    block_column_sums<NCH, kBlock, NCH>(rs_partial, nparts, wsum, rsn);
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const bool on = active[c] > 0.f;
        const float b = on ? rsn[c] / fmaxf(rs[c], 1e-30f) : 0.f;
        reinterpret_cast<float *>(sbeta)[c] = b;
        if (blockIdx.x == 0) {
            rs_new_out[c] = rsn[c];
            beta_out[c] = b;
            active_out[c] = (on && sqrtf(rsn[c]) / b_norm[c] > tol) ? 1.f : 0.f;
        }
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < quads; q += stride) {
        const float4 b = sbeta[(int)(q % NCH)];
        const float4 r = R[q];
        float4 p = P[q];
        p.x = r.x + p.x * b.x; p.y = r.y + p.y * b.y; p.z = r.z + p.z * b.z; p.w = r.w + p.w * b.w;
        P[q] = p;
    }
}

// the preconditioned iteration's direction step fed the same way: <R, Z> as the partial sums plx_pcg_apply left behind
// (nrz rows), |R|^2 as the partial sums of update_fused (kFusedBlocks rows); P = Z + beta P, activity from the TRUE residual
template <int NCH>
__global__ __launch_bounds__(kBlock) void pcg_step_direction_fused_kernel(float4 *__restrict__ P, const float4 *__restrict__ Z,
                                                                          const float4 *__restrict__ rz_partial, int nrz,
                                                                          const float4 *__restrict__ rr_partial, int nrr,
                                                                          const float *__restrict__ rz,
                                                                          const float *__restrict__ active,
                                                                          const float *__restrict__ b_norm, float tol,
                                                                          int64_t quads, float *__restrict__ rz_new_out,
                                                                          float *__restrict__ rr_out, float *__restrict__ beta_out,
                                                                          float *__restrict__ active_out)
{
    constexpr int vd = 4 * NCH;
    __shared__ float4 wsum[(kBlock / 64) * NCH];
    __shared__ float rzn[4 * NCH], rr[4 * NCH];
    __shared__ float4 sbeta[NCH];
    block_column_sums<NCH, kBlock, 8>(rz_partial, nrz, wsum, rzn);
    block_column_sums<NCH, kBlock, NCH>(rr_partial, nrr, wsum, rr);
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const bool on = active[c] > 0.f;
        const float b = on ? rzn[c] / fmaxf(rz[c], 1e-30f) : 0.f;
        reinterpret_cast<float *>(sbeta)[c] = b;
        if (blockIdx.x == 0) {
            rz_new_out[c] = rzn[c];
            rr_out[c] = rr[c];
            beta_out[c] = b;
            active_out[c] = (on && sqrtf(rr[c]) / b_norm[c] > tol) ? 1.f : 0.f;
        }
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < quads; q += stride) {
        const float4 b = sbeta[(int)(q % NCH)];
        const float4 z = Z[q];
        float4 p = P[q];
        p.x = z.x + p.x * b.x; p.y = z.y + p.y * b.y; p.z = z.z + p.z * b.z; p.w = z.w + p.w * b.w;
        P[q] = p;
    }
}

__global__ __launch_bounds__(kBlock) void cg_direction_kernel(float *__restrict__ P, const float *__restrict__ R,
                                                              const float *__restrict__ beta, int64_t total, int vd)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < total) P[i] = R[i] + P[i] * beta[i % vd];
}

// Backward pass of the lattice filter with respect to the positions (bilateral_kernel.py:113-122), the two
// elementwise ends of it in one pass each:
//   stack:    out[p] = [ g (L) | g (x) x (L*d) | src (L) | src (x) x (L*d) ],  (a (x) x)[l*d + k] = a[p][l] * x[p][k]
//   contract: grad_x[p][k] = -2 sum_l ( src_l x_k wg_l - src_l wgx_{l,k} + g_l x_k ws_l - g_l wsx_{l,k} )
//             with [wg | wgx | ws | wsx] = the filtered stack
__global__ __launch_bounds__(kBlock) void backward_stack_kernel(const float *__restrict__ g, const float *__restrict__ src,
                                                                const float *__restrict__ x, int64_t n, int L, int d,
                                                                float *__restrict__ out)
{
    const int W = 2 * L * (1 + d);
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= n * W) return;
    const int64_t p = item / W;
    int c = (int)(item - p * W);
    const float *a = g;
    if (c >= L * (1 + d)) { a = src; c -= L * (1 + d); }
    float v;
    if (c < L) v = a[p * L + c];
    else { const int q = c - L, l = q / d, k = q - l * d; v = a[p * L + l] * x[p * d + k]; }
    out[item] = v;
}

__global__ __launch_bounds__(kBlock) void backward_contract_kernel(const float *__restrict__ g,
                                                                   const float *__restrict__ src,
                                                                   const float *__restrict__ x,
                                                                   const float *__restrict__ f, int64_t n, int L, int d,
                                                                   float *__restrict__ grad_x)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= n * d) return;
    const int64_t p = item / d;
    const int k = (int)(item - p * d);
    const int W = 2 * L * (1 + d);
    const float *fp = f + p * W;
    const float *wg = fp, *wgx = fp + L, *ws = fp + L + L * d, *wsx = fp + 2 * L + L * d;
    const float xk = x[item];
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
        const float s = src[p * L + l], gg = g[p * L + l];
        acc += s * xk * wg[l] - s * wgx[l * d + k] + gg * xk * ws[l] - gg * wsx[l * d + k];
    }
    grad_x[item] = -2.0f * acc;
}

}  // namespace plx

namespace plx {
int coldot_final(const float *d_partial, int nblocks, int vd, float *d_out, hipStream_t stream)
{
    coldot_final_kernel<<<vd, kFinalBlock, 0, stream>>>(d_partial, nblocks, vd, d_out);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}
}  // namespace plx

using namespace plx;

extern "C" int plx_backward_stack(const float *d_g, const float *d_src, const float *d_x, int64_t n, int L, int d,
                                  float *d_out, void *stream)
{
    if (!d_g || !d_src || !d_x || !d_out) { set_error("plx_backward_stack: NULL argument"); return PLX_ERR_INVALID; }
    if (n < 0 || L < 1 || d < 1) { set_error("plx_backward_stack: bad shape"); return PLX_ERR_INVALID; }
    const int64_t total = n * 2 * L * (1 + d);
    if (total > 0)
        backward_stack_kernel<<<ceil_div(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(d_g, d_src, d_x, n, L, d, d_out);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_backward_contract(const float *d_g, const float *d_src, const float *d_x, const float *d_filtered,
                                     int64_t n, int L, int d, float *d_grad_x, void *stream)
{
    if (!d_g || !d_src || !d_x || !d_filtered || !d_grad_x) { set_error("plx_backward_contract: NULL argument"); return PLX_ERR_INVALID; }
    if (n < 0 || L < 1 || d < 1) { set_error("plx_backward_contract: bad shape"); return PLX_ERR_INVALID; }
    if (n * d > 0)
        backward_contract_kernel<<<ceil_div(n * d, kBlock), kBlock, 0, (hipStream_t)stream>>>(d_g, d_src, d_x, d_filtered, n,
                                                                                            L, d, d_grad_x);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_cg_update(float *d_x, float *d_r, const float *d_p, const float *d_ap, const float *d_alpha,
                             int64_t n, int vd, float *d_rs_new, float *d_work, void *stream)
{
    if (!d_x || !d_r || !d_p || !d_ap || !d_alpha || !d_rs_new || !d_work) { set_error("plx_cg_update: NULL argument"); return PLX_ERR_INVALID; }
    if (n < 0 || vd < 1 || vd > kBlock) { set_error("plx_cg_update: vd = %d outside 1..%d", vd, kBlock); return PLX_ERR_INVALID; }
    const int cw = vd;      // lanes per row
    hipStream_t s = (hipStream_t)stream;
    cg_update_kernel<<<kDotBlocks, kBlock, 0, s>>>(d_x, d_r, d_p, d_ap, d_alpha, n, vd, cw, d_work);
    coldot_final_kernel<<<vd, kFinalBlock, 0, s>>>(d_work, kDotBlocks, vd, d_rs_new);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_cg_step_update(float *d_x, float *d_r, const float *d_p, const float *d_ap, const float *d_rs,
                                  const float *d_pap, const float *d_active, int64_t n, int vd, float *d_rs_new,
                                  float *d_alpha, float *d_work, void *stream)
{
    if (!d_x || !d_r || !d_p || !d_ap || !d_rs || !d_pap || !d_active || !d_rs_new || !d_alpha || !d_work) {
        set_error("plx_cg_step_update: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (n < 0 || vd < 1 || vd > kBlock) { set_error("plx_cg_step_update: vd = %d outside 1..%d", vd, kBlock); return PLX_ERR_INVALID; }
    const int cw = vd;      // lanes per row
    hipStream_t s = (hipStream_t)stream;
    cg_step_update_kernel<<<kDotBlocks, kBlock, 0, s>>>(d_x, d_r, d_p, d_ap, d_rs, d_pap, d_active, n, vd, cw, d_work, d_alpha);
    coldot_final_kernel<<<vd, kFinalBlock, 0, s>>>(d_work, kDotBlocks, vd, d_rs_new);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_cg_step_direction(float *d_p, const float *d_r, const float *d_rs_new, const float *d_rs,
                                     const float *d_active, const float *d_b_norm, float tol, int64_t n, int vd,
                                     float *d_beta, float *d_active_out, void *stream)
{
    if (!d_p || !d_r || !d_rs_new || !d_rs || !d_active || !d_b_norm || !d_beta || !d_active_out) {
        set_error("plx_cg_step_direction: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (d_active == d_active_out) { set_error("plx_cg_step_direction: active and active_out must be different buffers"); return PLX_ERR_INVALID; }
    if (n < 0 || vd < 1 || vd > kBlock) { set_error("plx_cg_step_direction: vd = %d outside 1..%d", vd, kBlock); return PLX_ERR_INVALID; }
    const int64_t total = n * vd;
    if (total > 0 && (total & 3) == 0 && ((reinterpret_cast<uintptr_t>(d_p) | reinterpret_cast<uintptr_t>(d_r)) & 15) == 0) {
        const int64_t quads = total / 4;
        cg_step_direction4_kernel<<<ceil_div(quads, kBlock), kBlock, 0, (hipStream_t)stream>>>(
            reinterpret_cast<float4 *>(d_p), reinterpret_cast<const float4 *>(d_r), d_rs_new, d_rs, d_active, d_b_norm, tol,
            quads, vd, d_beta, d_active_out);
        PLX_HIP_TRY(hipGetLastError());
        return PLX_OK;
    }
    const int grid = total > 0 ? ceil_div(total, kBlock) : 1;
    cg_step_direction_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(d_p, d_r, d_rs_new, d_rs, d_active, d_b_norm, tol, total,
                                                                     vd, d_beta, d_active_out);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int64_t plx_cg_fused_work_floats(int vd) { return (vd >= 4 && vd <= 16 && vd % 4 == 0) ? (int64_t)kFusedBlocks * vd : -1; }

extern "C" int plx_cg_step_update_fused(float *d_x, float *d_r, const float *d_p, const float *d_ap, const float *d_rs,
                                        const float *d_pap_partial, int ntiles, const float *d_active, int64_t n, int vd,
                                        float *d_alpha, float *d_work, void *stream)
{
    if (!d_x || !d_r || !d_p || !d_ap || !d_rs || !d_pap_partial || !d_active || !d_alpha || !d_work) {
        set_error("plx_cg_step_update_fused: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (n < 0 || ntiles < 0 || plx_cg_fused_work_floats(vd) < 0) {
        set_error("plx_cg_step_update_fused: vd = %d is not 4, 8, 12 or 16 (use plx_cg_step_update)", vd);
        return PLX_ERR_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(d_pap_partial) & 15) != 0) { set_error("plx_cg_step_update_fused: d_pap_partial must be 16-byte aligned"); return PLX_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    const float4 *pp = reinterpret_cast<const float4 *>(d_pap_partial);
    switch (vd / 4) {
    case 1: cg_step_update_fused_kernel<1><<<kFusedBlocks, kFusedThreads, 0, s>>>(d_x, d_r, d_p, d_ap, d_rs, pp, ntiles, d_active, n, d_work, d_alpha); break;
    case 2: cg_step_update_fused_kernel<2><<<kFusedBlocks, kFusedThreads, 0, s>>>(d_x, d_r, d_p, d_ap, d_rs, pp, ntiles, d_active, n, d_work, d_alpha); break;
    case 3: cg_step_update_fused_kernel<3><<<kFusedBlocks, kFusedThreads, 0, s>>>(d_x, d_r, d_p, d_ap, d_rs, pp, ntiles, d_active, n, d_work, d_alpha); break;
    default: cg_step_update_fused_kernel<4><<<kFusedBlocks, kFusedThreads, 0, s>>>(d_x, d_r, d_p, d_ap, d_rs, pp, ntiles, d_active, n, d_work, d_alpha); break;
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_cg_step_direction_fused(float *d_p, const float *d_r, const float *d_work, const float *d_rs,
                                           const float *d_active, const float *d_b_norm, float tol, int64_t n, int vd,
                                           float *d_rs_new, float *d_beta, float *d_active_out, void *stream)
{
    if (!d_p || !d_r || !d_work || !d_rs || !d_active || !d_b_norm || !d_rs_new || !d_beta || !d_active_out) {
        set_error("plx_cg_step_direction_fused: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (d_active == d_active_out || d_rs == d_rs_new) { set_error("plx_cg_step_direction_fused: active / rs and their outputs must be different buffers"); return PLX_ERR_INVALID; }
    if (n < 0 || plx_cg_fused_work_floats(vd) < 0) {
        set_error("plx_cg_step_direction_fused: vd = %d is not 4, 8, 12 or 16 (use plx_cg_step_direction)", vd);
        return PLX_ERR_INVALID;
    }
    if (((reinterpret_cast<uintptr_t>(d_p) | reinterpret_cast<uintptr_t>(d_r) | reinterpret_cast<uintptr_t>(d_work)) & 15) != 0) {
        set_error("plx_cg_step_direction_fused: d_p, d_r and d_work must be 16-byte aligned");
        return PLX_ERR_INVALID;
    }
    const int64_t quads = n * vd / 4;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(kFusedDirBlocks, ceil_div(quads, kBlock)));
    hipStream_t s = (hipStream_t)stream;
    float4 *p4 = reinterpret_cast<float4 *>(d_p);
    const float4 *r4 = reinterpret_cast<const float4 *>(d_r), *w4 = reinterpret_cast<const float4 *>(d_work);
    switch (vd / 4) {
    case 1: cg_step_direction_fused_kernel<1><<<grid, kBlock, 0, s>>>(p4, r4, w4, kFusedBlocks, d_rs, d_active, d_b_norm, tol, quads, d_rs_new, d_beta, d_active_out); break;
    case 2: cg_step_direction_fused_kernel<2><<<grid, kBlock, 0, s>>>(p4, r4, w4, kFusedBlocks, d_rs, d_active, d_b_norm, tol, quads, d_rs_new, d_beta, d_active_out); break;
    case 3: cg_step_direction_fused_kernel<3><<<grid, kBlock, 0, s>>>(p4, r4, w4, kFusedBlocks, d_rs, d_active, d_b_norm, tol, quads, d_rs_new, d_beta, d_active_out); break;
    default: cg_step_direction_fused_kernel<4><<<grid, kBlock, 0, s>>>(p4, r4, w4, kFusedBlocks, d_rs, d_active, d_b_norm, tol, quads, d_rs_new, d_beta, d_active_out); break;
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_pcg_step_direction_fused(float *d_p, const float *d_z, const float *d_rz_partial, int nrz,
                                            const float *d_rr_partial, const float *d_rz, const float *d_active,
                                            const float *d_b_norm, float tol, int64_t n, int vd, float *d_rz_new, float *d_rr,
                                            float *d_beta, float *d_active_out, void *stream)
{
    if (!d_p || !d_z || !d_rz_partial || !d_rr_partial || !d_rz || !d_active || !d_b_norm || !d_rz_new || !d_rr || !d_beta || !d_active_out) {
        set_error("plx_pcg_step_direction_fused: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (d_active == d_active_out || d_rz == d_rz_new) { set_error("plx_pcg_step_direction_fused: active / rz and their outputs must be different buffers"); return PLX_ERR_INVALID; }
    if (n < 0 || nrz < 0 || plx_cg_fused_work_floats(vd) < 0) {
        set_error("plx_pcg_step_direction_fused: vd = %d is not 4, 8, 12 or 16 (use plx_pcg_step_direction)", vd);
        return PLX_ERR_INVALID;
    }
    if (((reinterpret_cast<uintptr_t>(d_p) | reinterpret_cast<uintptr_t>(d_z) | reinterpret_cast<uintptr_t>(d_rz_partial) |
          reinterpret_cast<uintptr_t>(d_rr_partial)) & 15) != 0) {
        set_error("plx_pcg_step_direction_fused: d_p, d_z and the partial sums must be 16-byte aligned");
        return PLX_ERR_INVALID;
    }
    const int64_t quads = n * vd / 4;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(kFusedDirBlocks, ceil_div(quads, kBlock)));
    hipStream_t s = (hipStream_t)stream;
    float4 *p4 = reinterpret_cast<float4 *>(d_p);
    const float4 *z4 = reinterpret_cast<const float4 *>(d_z), *a4 = reinterpret_cast<const float4 *>(d_rz_partial),
                 *b4 = reinterpret_cast<const float4 *>(d_rr_partial);
#define PLX_CASE(NCH) pcg_step_direction_fused_kernel<NCH><<<grid, kBlock, 0, s>>>(p4, z4, a4, nrz, b4, kFusedBlocks, d_rz, d_active, d_b_norm, tol, quads, d_rz_new, d_rr, d_beta, d_active_out)
    switch (vd / 4) {
    case 1: PLX_CASE(1); break;
    case 2: PLX_CASE(2); break;
    case 3: PLX_CASE(3); break;
    default: PLX_CASE(4); break;
    }
#undef PLX_CASE
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_cg_direction(float *d_p, const float *d_r, const float *d_beta, int64_t n, int vd, void *stream)
{
    if (!d_p || !d_r || !d_beta) { set_error("plx_cg_direction: NULL argument"); return PLX_ERR_INVALID; }
    if (n < 0 || vd < 1) { set_error("plx_cg_direction: bad shape"); return PLX_ERR_INVALID; }
    const int64_t total = n * vd;
    if (total > 0)
        cg_direction_kernel<<<ceil_div(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(d_p, d_r, d_beta, total, vd);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_coldot(const float *d_a, const float *d_b, int64_t n, int vd, float *d_out, float *d_work,
                          void *stream)
{
    if (!d_a || !d_b || !d_out || !d_work) { set_error("plx_coldot: NULL argument"); return PLX_ERR_INVALID; }
    if (n < 0 || vd < 1 || vd > kBlock) { set_error("plx_coldot: vd = %d outside 1..%d", vd, kBlock); return PLX_ERR_INVALID; }
    const int cw = vd;      // lanes per row
    hipStream_t s = (hipStream_t)stream;
    coldot_partial_kernel<<<kDotBlocks, kBlock, 0, s>>>(d_a, d_b, n, vd, cw, d_work);
    coldot_final_kernel<<<vd, kFinalBlock, 0, s>>>(d_work, kDotBlocks, vd, d_out);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int64_t plx_coldot_work_floats(int vd) { return (int64_t)kDotBlocks * (vd > 0 ? vd : 1); }

// plx_pcg.hip -- the vector work of PRECONDITIONED batched CG next to the MVM, and the construction of the
// preconditioner itself.
//
// The reference trains with gpytorch.settings.max_preconditioner_size(100) (experiments/train_simplexgp.py:36,
// configs/simplexgp.yml): every solve of (s K + sigma^2 I) is preconditioned by P = L L^T + sigma^2 I, L [n][k] the
// rank-k pivoted Cholesky factor of s K.  GPyTorch does that with torch ops; here
//
//   L^T is stored row-major [kp][ld] (kp = k rounded up to 16, zero rows; ld = n rounded up to 64, zero tail), rows of
//   the n dimension in the SAME order as the CG vectors (lattice row order in solvers.py), so no per-iteration
//   permutation is left, and one preconditioner application is two streaming passes over L^T:
//     pcg_gram_kernel     G = L^T R        MFMA f32 16x16x4 (exact fp32 products, fp32 accumulate), per-workgroup partials
//     pcg_project_kernel  T = C^-1 G       C = sigma^2 I + L^T L, its inverse precomputed in fp64 [kp][kp]
//     pcg_apply_kernel    Z = (R - L T) / sigma^2, with the per-workgroup partials of <R, Z> from the same registers
//   pcg_step_direction   beta = rz' / rz, P = Z + beta P, active' from the TRUE residual norm
//
//   the factor is built in BATCHES of up to 16 speculated pivots (plx_pchol_*): the next nb pivots are the nb largest
//   entries of the residual diagonal (ties: lower caller row first, as torch.argmax breaks them); their nb kernel
//   rows come out of ONE nb-column MVM instead of nb single-column ones, the update against the finished part of the
//   factor is one panel pass, and the nb in-batch steps then run in pivot order, each one checking on the device that
//   its speculated pivot still is the argmax of the updated diagonal -- the first one that is not ends the batch
//   (nothing after it is used), so the factor is exactly the sequential algorithm's.
#include "plx_internal.h"

#include <hip/hip_fp16.h>

#include <stdlib.h>

#include <algorithm>

namespace plx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGramBlocksMax = 1024;  // the work buffer is sized for this many workgroups of the gram kernel
// workgroups of the gram kernel = partial sums per (j, c); PLX_GRAM_BLOCKS overrides it for A/B runs
static int gram_blocks()
{
    static const int v = [] {
        const char *e = getenv("PLX_GRAM_BLOCKS");
        const int x = e ? atoi(e) : 512;
        return x >= 64 && x <= kGramBlocksMax ? x : 512;
    }();
    return v;
}
constexpr int kPcgCols = 16;          // column tile: the N of the 16x16x4 MFMA; T is [kp][16]

// ---- G = L^T R ---------------------------------------------------------------------------------------------------
// One wave per 64-row tile.  MFMA 16x16x4 f32: A[j = lane & 15][k = lane >> 4], B[k = lane >> 4][c = lane & 15],
// D[j = 4 (lane >> 4) + reg][c = lane & 15].  The k slot of a lane is a ROW of the tile: lane l loads the 16 bytes
// lt[j][i0 + 16 s + 4 (l >> 4) .. + 3] (s = 0..3), component q of that load is the A operand of MFMA (s, q), whose B
// operand is R[i0 + 16 s + 4 (l >> 4) + q][l & 15]: the same row on both sides, whatever order the rows come in.
template <int JT, bool HALF>
__global__ __launch_bounds__(kBlock) void pcg_gram_kernel(const void *__restrict__ lt_, int64_t ld, int kp, int j0,
                                                          const float *__restrict__ R, int64_t n, int t, int ntiles,
                                                          float *__restrict__ partial)
{
    // fp32 factor: 4 loads of 4 rows per lane and 16-row tile quarter (S = 4 steps); fp16: 2 loads of 8 rows (S = 2)
    constexpr int S = HALF ? 2 : 4, Q = HALF ? 8 : 4;
    __shared__ float red[JT * 16 * kPcgCols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    f32x4 acc[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wstride = gridDim.x * (kBlock / 64);
    for (int tile = blockIdx.x * (kBlock / 64) + wave; tile < ntiles; tile += wstride) {
        const int64_t i0 = (int64_t)tile * 64;
        // every load of the tile is issued before the first MFMA: 28 KB (fp32) / 14 KB (fp16) in flight per wave -- with the
        // loads issued row block by row block two waves per SIMD kept 32 KB per CU in flight and the pass ran at 3.1 TB/s
        uint4 araw[JT][S];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const char *row = (const char *)lt_ + ((int64_t)(j0 + 16 * jt + lr) * ld + i0 + Q * lk) * (HALF ? 2 : 4);
#pragma unroll
            for (int s = 0; s < S; ++s) araw[jt][s] = *reinterpret_cast<const uint4 *>(row + (size_t)s * (64 / S) * (HALF ? 2 : 4));
        }
        float b[S * Q];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int64_t i = i0 + (64 / S) * s + Q * lk + q;
                b[s * Q + q] = (lr < t && i < n) ? R[i * t + lr] : 0.f;
            }
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float a[Q];
                const uint4 v = araw[jt][s];
                if constexpr (HALF) {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        a[2 * h] = __half2float(__ushort_as_half((unsigned short)(w[h] & 0xFFFFu)));
                        a[2 * h + 1] = __half2float(__ushort_as_half((unsigned short)(w[h] >> 16)));
                    }
                } else {
                    a[0] = __uint_as_float(v.x); a[1] = __uint_as_float(v.y); a[2] = __uint_as_float(v.z); a[3] = __uint_as_float(v.w);
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[s * Q + q], acc[jt], 0, 0, 0);
            }
    }
    // the four waves add into one LDS image in wave order (fixed summation order)
    for (int w = 0; w < kBlock / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = (jt * 16 + 4 * lk + e) * kPcgCols + lr;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + acc[jt][e];
                }
        }
        __syncthreads();
    }
    // partial[workgroup][c][j]: consecutive threads write consecutive j
    for (int x = threadIdx.x; x < JT * 16 * kPcgCols; x += kBlock) {
        const int jj = x % (JT * 16), c = x / (JT * 16);
        partial[((size_t)blockIdx.x * kPcgCols + c) * kp + j0 + jj] = red[jj * kPcgCols + c];
    }
}

// ---- T = C^-1 G, one workgroup per column: the partials are summed in fp64 in workgroup order, C^-1 is fp64 [kp][kp]
// (symmetric: column j is read as row j, consecutive threads consecutive addresses) --------------------------------
__global__ __launch_bounds__(1024) void pcg_project_kernel(const float *__restrict__ partial, int nparts, int kp,
                                                           const double *__restrict__ cinv, float *__restrict__ Tm)
{
    extern __shared__ double sm[];          // g[kp] | grp[groups][kp]
    double *g = sm, *grp = sm + kp;
    const int c = blockIdx.x;
    // thread (j, q): the terms q, q + groups, ... of entry j, eight loads in flight (one thread per entry walking its 512
    // partial sums took 52 us; two loads in flight per thread 24 us)
    const int groups = max(1, (int)blockDim.x / kp);
    const int j = threadIdx.x % kp, q = threadIdx.x / kp;
    if (q < groups) {
        double s = 0.0;
        int w = q;
        for (; w + 7 * groups < nparts; w += 8 * groups) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((size_t)(w + u * groups) * kPcgCols + c) * kp + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; w < nparts; w += groups) s += (double)partial[((size_t)w * kPcgCols + c) * kp + j];
        grp[q * kp + j] = s;
    }
    __syncthreads();
    for (int jj = threadIdx.x; jj < kp; jj += blockDim.x) {
        double s = 0.0;
        for (int qq = 0; qq < groups; ++qq) s += grp[qq * kp + jj];
        g[jj] = s;
    }
    __syncthreads();
    // T = C^-1 g the same way: thread (j, q) takes the terms q, q + groups, ... of row j (C^-1 is symmetric: its column j
    // is read as row j, consecutive threads consecutive addresses)
    if (q < groups) {
        double s = 0.0;
        int qq = q;
        for (; qq + 3 * groups < kp; qq += 4 * groups) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = cinv[(size_t)(qq + u * groups) * kp + j];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u] * g[qq + u * groups];
        }
        for (; qq < kp; qq += groups) s += cinv[(size_t)qq * kp + j] * g[qq];
        grp[q * kp + j] = s;
    }
    __syncthreads();
    for (int jj = threadIdx.x; jj < kp; jj += blockDim.x) {
        double s = 0.0;
        for (int qq = 0; qq < groups; ++qq) s += grp[qq * kp + jj];
        Tm[jj * kPcgCols + c] = (float)s;
    }
}

// ---- Z = (in_scale R - L T) out_scale, one row per lane: the k rows of L^T stream through coalesced, the k x T
// coefficients are wave-uniform (scalar loads).  TRANSPOSED: Z is written as T rows of ldz floats (the panel of the
// batched pivoted Cholesky); otherwise row-major [n][T] like R, with the workgroup's partial <R, Z> per column. ------
template <int T, bool TRANSPOSED, bool HALF>
__global__ __launch_bounds__(kBlock) void pcg_apply_kernel(const void *__restrict__ lt_, int64_t ld, int k,
                                                           const float *__restrict__ R, int64_t n,
                                                           const float *__restrict__ Tm, const float *__restrict__ scal,
                                                           float *__restrict__ Z, int64_t ldz, float *__restrict__ partial)
{
    constexpr int ROWS = HALF ? 2 : 1;      // an fp16 factor: two adjacent rows per lane, so that a lane still loads 4 bytes
    __shared__ float red[TRANSPOSED ? 1 : (kBlock / 64) * T];
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * ROWS;
    float acc[ROWS][T];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < T; ++c) acc[r][c] = 0.f;
    // (rows beyond n read the zero tail of the factor: ld is a multiple of 64)
    const int64_t ic = i < ld ? i : 0;
#pragma unroll 4
    for (int j = 0; j < k; ++j) {
        float l[ROWS];
        if constexpr (HALF) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>((const unsigned short *)lt_ + (int64_t)j * ld + ic);
            l[0] = __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
            l[1] = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
        } else {
            l[0] = ((const float *)lt_)[(int64_t)j * ld + ic];
        }
        const float *tr = Tm + j * kPcgCols;
#pragma unroll
        for (int c = 0; c < T; ++c) {
            const float tc = tr[c];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r][c] = fmaf(l[r], tc, acc[r][c]);
        }
    }
    const float in_scale = scal[0], out_scale = scal[1];
    float dot[T];
#pragma unroll
    for (int c = 0; c < T; ++c) dot[c] = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int64_t ir = i + r;
        const bool live = ir < n;
        const int64_t ii = live ? ir : n - 1;
        float rv[T], z[T];
        if constexpr (T % 4 == 0) {
#pragma unroll
            for (int q = 0; q < T / 4; ++q) {
                const float4 v = reinterpret_cast<const float4 *>(R + ii * T)[q];
                rv[4 * q] = v.x; rv[4 * q + 1] = v.y; rv[4 * q + 2] = v.z; rv[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < T; ++c) rv[c] = R[ii * T + c];
        }
#pragma unroll
        for (int c = 0; c < T; ++c) z[c] = (rv[c] * in_scale - acc[r][c]) * out_scale;
        if (live) {
            if constexpr (TRANSPOSED) {
#pragma unroll
                for (int c = 0; c < T; ++c) Z[(int64_t)c * ldz + ir] = z[c];
            } else if constexpr (T % 4 == 0) {
#pragma unroll
                for (int q = 0; q < T / 4; ++q)
                    reinterpret_cast<float4 *>(Z + ir * T)[q] = make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
            } else {
#pragma unroll
                for (int c = 0; c < T; ++c) Z[ir * T + c] = z[c];
            }
#pragma unroll
            for (int c = 0; c < T; ++c) dot[c] += rv[c] * z[c];
        }
    }
    if constexpr (!TRANSPOSED) {
        // <R, Z> per column: lanes of a wave by a fixed xor tree, the four waves in order
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < T; ++c) {
            float p = dot[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            if (lane == 0) red[wave * T + c] = p;
        }
        __syncthreads();
        if ((int)threadIdx.x < T) {
            float s = 0.f;
            for (int w = 0; w < kBlock / 64; ++w) s += red[w * T + threadIdx.x];
            partial[(size_t)blockIdx.x * T + threadIdx.x] = s;
        }
    }
}

// fp32 factor -> fp16 copy (round to nearest even), whole [kp][ld] image
__global__ __launch_bounds__(kBlock) void pcg_to_half_kernel(const float4 *__restrict__ src, int64_t quads, uint2 *__restrict__ dst)
{
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= quads) return;
    const float4 v = src[q];
    const uint32_t lo = (uint32_t)__half_as_ushort(__float2half_rn(v.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(v.y)) << 16);
    const uint32_t hi = (uint32_t)__half_as_ushort(__float2half_rn(v.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(v.w)) << 16);
    dst[q] = make_uint2(lo, hi);
}

// ---- direction of a preconditioned iteration: beta = active ? rz' / rz : 0; P = Z + beta P; a column stays active
// while its TRUE residual, sqrt(rr) / |b|, is above tol (rr = |R|^2 from plx_cg_step_update) -------------------------
__global__ __launch_bounds__(kBlock) void pcg_step_direction_kernel(float *__restrict__ P, const float *__restrict__ Z,
                                                                    const float *__restrict__ rz_new,
                                                                    const float *__restrict__ rz,
                                                                    const float *__restrict__ rr,
                                                                    const float *__restrict__ active,
                                                                    const float *__restrict__ b_norm, float tol,
                                                                    int64_t total, int vd, float *__restrict__ beta_out,
                                                                    float *__restrict__ active_out)
{
    __shared__ float sbeta[kBlock];
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        sbeta[c] = active[c] > 0.f ? rz_new[c] / fmaxf(rz[c], 1e-30f) : 0.f;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < total) {
        const uint32_t uvd = (uint32_t)vd;
        const uint32_t bm = ((blockIdx.x % uvd) * ((uint32_t)kBlock % uvd)) % uvd;      // wave-uniform
        const uint32_t c = (bm + threadIdx.x) % uvd;
        P[i] = Z[i] + P[i] * sbeta[c];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const bool on = active[c] > 0.f;
        beta_out[c] = on ? rz_new[c] / fmaxf(rz[c], 1e-30f) : 0.f;
        active_out[c] = (on && sqrtf(rr[c]) / b_norm[c] > tol) ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(kBlock) void pcg_step_direction4_kernel(float4 *__restrict__ P, const float4 *__restrict__ Z,
                                                                     const float *__restrict__ rz_new,
                                                                     const float *__restrict__ rz,
                                                                     const float *__restrict__ rr,
                                                                     const float *__restrict__ active,
                                                                     const float *__restrict__ b_norm, float tol,
                                                                     int64_t quads, int vd, float *__restrict__ beta_out,
                                                                     float *__restrict__ active_out)
{
    __shared__ float sbeta[kBlock];
    if ((int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        sbeta[c] = active[c] > 0.f ? rz_new[c] / fmaxf(rz[c], 1e-30f) : 0.f;
    }
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q < quads) {
        const uint32_t uvd = (uint32_t)vd;
        const uint32_t bm = ((blockIdx.x % uvd) * ((4u * (uint32_t)kBlock) % uvd)) % uvd;      // wave-uniform
        uint32_t c = (bm + 4u * threadIdx.x) % uvd;
        const float4 z = Z[q];
        float4 p = P[q];
        p.x = z.x + p.x * sbeta[c]; c = c + 1 == uvd ? 0 : c + 1;
        p.y = z.y + p.y * sbeta[c]; c = c + 1 == uvd ? 0 : c + 1;
        p.z = z.z + p.z * sbeta[c]; c = c + 1 == uvd ? 0 : c + 1;
        p.w = z.w + p.w * sbeta[c];
        P[q] = p;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < vd) {
        const int c = threadIdx.x;
        const bool on = active[c] > 0.f;
        beta_out[c] = on ? rz_new[c] / fmaxf(rz[c], 1e-30f) : 0.f;
        active_out[c] = (on && sqrtf(rr[c]) / b_norm[c] > tol) ? 1.f : 0.f;
    }
}

// =====================================================================================================================
// batched pivoted Cholesky
// =====================================================================================================================

constexpr int kPcholParts = 1024;     // workgroups of the diagonal passes = argmax partials
constexpr int kPcholMaxBatch = 16;

// order of the residual diagonal: larger value first, ties by LOWER rank (rank = the caller's row of the entry when
// the vectors are in lattice order; the index itself otherwise) -- what torch.argmax does on the caller-order vector.
// Diagonal entries are >= 0, so their bit patterns order like the values.  0 = "nothing".
__device__ inline uint64_t pchol_key(float v, uint32_t rank) { return ((uint64_t)__float_as_uint(v) << 32) | (uint32_t)(~rank); }

__device__ inline void wave_argmax(uint64_t &key, int &idx)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t hi = __shfl_xor((uint32_t)(key >> 32), o, 64), lo = __shfl_xor((uint32_t)key, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        const uint64_t ok = ((uint64_t)hi << 32) | lo;
        if (ok > key) { key = ok; idx = oi; }
    }
}

// workgroup maximum of (key, idx); result valid in thread 0.  scratch: kBlock / 64 entries each.
__device__ inline void block_argmax(uint64_t &key, int &idx, uint64_t *skey, int *sidx)
{
    wave_argmax(key, idx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { skey[wave] = key; sidx[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (skey[w] > key) { key = skey[w]; idx = sidx[w]; }
    }
}

// the nb largest entries of every workgroup's chunk, in order: round r takes the largest key below round r - 1's
__global__ __launch_bounds__(kBlock) void pchol_top_partial_kernel(const float *__restrict__ diag,
                                                                   const uint32_t *__restrict__ rank, int64_t n, int nb,
                                                                   uint64_t *__restrict__ pkey, int *__restrict__ pidx)
{
    __shared__ uint64_t skey[kBlock / 64];
    __shared__ int sidx[kBlock / 64];
    __shared__ uint64_t s_last;
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(lo + chunk, n);
    uint64_t last = ~0ull;
    for (int r = 0; r < nb; ++r) {
        uint64_t best = 0;
        int bi = -1;
        for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
            const uint64_t key = pchol_key(diag[i], rank ? rank[i] : (uint32_t)i);
            if (key < last && key > best) { best = key; bi = (int)i; }
        }
        block_argmax(best, bi, skey, sidx);
        if (threadIdx.x == 0) {
            pkey[(size_t)blockIdx.x * nb + r] = best;
            pidx[(size_t)blockIdx.x * nb + r] = bi;
            s_last = best;
        }
        __syncthreads();
        last = s_last;
        __syncthreads();
    }
}

// Batch state (device ints, two slots of 8: slot[0] = stop (the argmax is not among the batch's unused candidates),
// slot[1] = bit mask of the candidates used so far, slot[2] = the candidate (batch column) the step uses, slot[3] = the
// residual diagonal at its pivot (float bits).  Step b runs under state S_b; S_b (b >= 1) follows from S_(b-1) and the
// argmax partials step b - 1 left.
// The candidates are the nb largest diagonal entries when the batch starts, but the ORDER in which the sequential
// algorithm takes them is only known step by step (an entry touched by an earlier pivot's column falls behind untouched
// ones): every step looks its true argmax up among the unused candidates, so a batch only ends when the argmax is an
// entry whose kernel row was not computed.

// cand[0..nsel-1) = the nsel - 1 largest keys' entries, *knext = the nsel-th largest key (0: none) -- a bound on every
// entry that is NOT a candidate.  Every part's list is sorted (largest first): a `parts`-way merge by ONE wave, every lane
// in charge of 16 lists whose first kDepth keys sit in LDS (a deeper list is read from memory: the largest entries of a
// diagonal are spread over the parts).  (One list head per thread of a 1024-thread workgroup and a workgroup argmax per
// output: 27 us for 12 outputs; scanning all parts * nsel keys per output: 47 us.)
constexpr int kTopDepth = 4;
__global__ __launch_bounds__(1024) void pchol_top_final_kernel(const uint64_t *__restrict__ pkey, const int *__restrict__ pidx,
                                                               int parts, int nsel, int *__restrict__ cand,
                                                               uint64_t *__restrict__ knext)
{
    __shared__ uint64_t keys[1024 * kTopDepth];
    __shared__ unsigned char head[1024];
    __shared__ int win_part[kPcholMaxBatch + 1], win_depth[kPcholMaxBatch + 1];
    for (int x = threadIdx.x; x < 1024 * kTopDepth; x += 1024) {
        const int p = x / kTopDepth, h = x % kTopDepth;
        keys[x] = (p < parts && h < nsel) ? pkey[(size_t)p * nsel + h] : 0;
    }
    head[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    for (int r = 0; r < nsel; ++r) {
        uint64_t best = 0;
        int who = -1;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int p = lane * 16 + u, h = head[p];
            const uint64_t key = h < kTopDepth ? keys[p * kTopDepth + h] : ((p < parts && h < nsel) ? pkey[(size_t)p * nsel + h] : 0);
            if (key > best) { best = key; who = p; }
        }
        // the maximum key by a butterfly over the KEY alone (two shuffles a stage instead of three), its owner from a
        // ballot: keys are distinct (the rank is part of the key), so at most one lane holds it
        uint64_t top = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t other = ((uint64_t)__shfl_xor((uint32_t)(top >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((uint32_t)top, o, 64);
            top = other > top ? other : top;
        }
        const uint64_t holders = __ballot(best == top && top != 0);
        who = holders ? __shfl(who, __ffsll((long long)holders) - 1, 64) : -1;
        best = top;
        if (lane == 0) {
            win_part[r] = best ? who : -1;
            win_depth[r] = best ? head[who] : 0;
            if (r == nsel - 1) *knext = best;
        }
        if (best && (who >> 4) == lane) head[who] += 1;          // keys are distinct (the rank is part of the key)
    }
    if (lane < nsel - 1) cand[lane] = win_part[lane] >= 0 ? pidx[(size_t)win_part[lane] * nsel + win_depth[lane]] : -1;
}

__global__ void pchol_onehot_kernel(const int *__restrict__ cand, int nb, int t, float *__restrict__ rhs)
{
    const int b = threadIdx.x;
    if (b < nb && cand[b] >= 0) rhs[(int64_t)cand[b] * t + b] = 1.f;
}

// W[j][b] = L[cand_b][j] for the finished columns j < m: the coefficients of the panel update
__global__ __launch_bounds__(kBlock) void pchol_gather_kernel(const float *__restrict__ lt, int64_t ld, int m,
                                                              const int *__restrict__ cand, int nb, float *__restrict__ W)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= m * kPcgCols) return;
    const int j = x / kPcgCols, b = x % kPcgCols;
    W[x] = (b < nb && cand[b] >= 0) ? lt[(int64_t)j * ld + cand[b]] : 0.f;
}

// ---- a batch in one pass ---------------------------------------------------------------------------------------------
// Which candidates the sequential algorithm takes, in which order, can be decided WITHOUT touching the n-vectors as long
// as one bound holds.  The candidates are the nb largest entries of the residual diagonal when the batch starts and
// K_next is the largest key outside them; a step only ever LOWERS diagonal entries, so K_next bounds every non-candidate
// for the whole batch.  The candidates' own entries after each step follow from the nb x nb block of the panel at the
// candidate rows alone (a small pivoted Cholesky, pchol_plan_kernel).  While the largest unused candidate's key stays
// above K_next it IS the global argmax -- the sequential algorithm's next pivot; the first step where it does not ends
// the plan (the true argmax may still be a candidate: the exact step kernels below take over from there when asked to).
// On a lattice whose kernel rows are sparse (the regime where a factor column touches few of the other candidates) whole
// batches are planned, and pchol_multi_step_kernel writes all their columns in ONE pass over the n-vectors: nb panel
// rows in, nb columns out, the diagonal once -- against nb passes that each re-read the columns before them.
// Arithmetic: the column formula is written with explicit fmaf in the three kernels that evaluate it (plan, multi-step,
// step), so the plan sees bit for bit the values the n-vector passes store.
static_assert(kBlock == kPcholMaxBatch * kPcholMaxBatch, "pchol_plan_kernel gathers the candidates' panel block with one thread per entry");

struct PcholPlan {
    int a;                                     // planned steps (>= 1: the first candidate is the argmax by construction)
    int used;                                  // bit mask of the candidates they use
    int order[kPcholMaxBatch];                 // candidate (batch column) of step b
    float dmax[kPcholMaxBatch];                // residual diagonal at its pivot when the step runs
    float w[kPcholMaxBatch][kPcholMaxBatch];   // w[b][q] = L[pivot_b][m + q], q < b
};

__device__ inline float pchol_col(float v, bool ok, float root) { return ok ? v / root : 0.f; }
__device__ inline float pchol_diag(float d, float col) { return fmaxf(fmaf(-col, col, d), 0.f); }

__global__ __launch_bounds__(kBlock) void pchol_plan_kernel(const float *__restrict__ rowsT, int64_t ld,
                                                            const int *__restrict__ cand, int nb,
                                                            const float *__restrict__ diag, const uint32_t *__restrict__ rank,
                                                            const uint64_t *__restrict__ knext_p, float tol_abs,
                                                            PcholPlan *__restrict__ plan)
{
    __shared__ float P[kPcholMaxBatch][kPcholMaxBatch + 1];
    {
        const int c = threadIdx.x / kPcholMaxBatch, c2 = threadIdx.x % kPcholMaxBatch;      // kBlock = 16 * 16
        float v = 0.f;
        if (c < nb && c2 < nb && cand[c] >= 0 && cand[c2] >= 0) v = rowsT[(int64_t)c * ld + cand[c2]];
        P[c][c2] = v;                                                                       // panel row of candidate c at candidate c2's entry
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int c = threadIdx.x;
    const bool valid = c < nb && cand[c] >= 0;
    float dv = valid ? diag[cand[c]] : 0.f;
    const uint32_t rk = valid ? (rank ? rank[cand[c]] : (uint32_t)cand[c]) : 0u;
    const uint64_t knext = *knext_p;
    float L[kPcholMaxBatch];
#pragma unroll
    for (int q = 0; q < kPcholMaxBatch; ++q) L[q] = 0.f;
    uint32_t used = 0;
    int a = 0;
    bool open = true;                              // (uniform) the plan has not ended
#pragma unroll
    for (int b = 0; b < kPcholMaxBatch; ++b) {
        uint64_t key = (open && b < nb && valid && !((used >> c) & 1)) ? pchol_key(dv, rk) : 0;
        int cs = c;
        wave_argmax(key, cs);                      // every lane: the largest unused candidate
        open = open && b < nb && key != 0 && (b == 0 || key > knext);
        if (open) {
            const float dmax = __uint_as_float((uint32_t)(key >> 32));
            const bool ok = dmax > tol_abs;
            const float root = sqrtf(fmaxf(dmax, 1e-30f));
            float v = P[cs][c < kPcholMaxBatch ? c : 0];
#pragma unroll
            for (int q = 0; q < b; ++q) {
                const float wq = __shfl(L[q], cs, 64);       // L[pivot][m + q]
                v = fmaf(-L[q], wq, v);
                if (c == 0) plan->w[b][q] = wq;
            }
            const float col = pchol_col(v, ok, root);
            L[b] = col;
            dv = c == cs ? 0.f : pchol_diag(dv, col);
            used |= 1u << cs;
            if (c == 0) { plan->order[b] = cs; plan->dmax[b] = dmax; }
            a = b + 1;
        }
    }
    if (c == 0) { plan->a = a; plan->used = (int)used; }
}

// steps 0 .. a - 1 of the plan in one pass; leaves the state and the argmax partials step a - 1 would have left, so that
// pchol_step_kernel launches for b = a .. nb - 1 can carry on
__global__ __launch_bounds__(kBlock) void pchol_multi_step_kernel(float *__restrict__ lt, int64_t ld, int m,
                                                                  const float *__restrict__ rowsT, float *__restrict__ diag,
                                                                  const uint32_t *__restrict__ rank, int64_t n,
                                                                  const int *__restrict__ cand, const PcholPlan *__restrict__ plan,
                                                                  float tol_abs, uint64_t *__restrict__ pkey,
                                                                  int *__restrict__ pidx, int *__restrict__ state,
                                                                  int *__restrict__ accepted)
{
    __shared__ uint64_t skey[kBlock / 64];
    __shared__ int sidx[kBlock / 64];
    __shared__ PcholPlan sp;
    __shared__ float s_root[kPcholMaxBatch];
    __shared__ int s_piv[kPcholMaxBatch], s_ok[kPcholMaxBatch];
    for (int x = threadIdx.x; x < (int)(sizeof(PcholPlan) / 4); x += kBlock) reinterpret_cast<int *>(&sp)[x] = reinterpret_cast<const int *>(plan)[x];
    __syncthreads();
    const int a = sp.a;
    if ((int)threadIdx.x < a) {
        s_root[threadIdx.x] = sqrtf(fmaxf(sp.dmax[threadIdx.x], 1e-30f));
        s_ok[threadIdx.x] = sp.dmax[threadIdx.x] > tol_abs;
        s_piv[threadIdx.x] = cand[sp.order[threadIdx.x]];
    }
    __syncthreads();
    uint64_t best = 0;
    int bi = -1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float col[kPcholMaxBatch], row[kPcholMaxBatch];
        float dn = diag[i];
#pragma unroll
        for (int b = 0; b < kPcholMaxBatch; ++b) row[b] = b < a ? rowsT[(int64_t)sp.order[b] * ld + i] : 0.f;    // (all loads in flight before the first column)
#pragma unroll
        for (int b = 0; b < kPcholMaxBatch; ++b) {
            col[b] = 0.f;
            if (b < a) {
                float v = row[b];
#pragma unroll
                for (int q = 0; q < kPcholMaxBatch; ++q)
                    if (q < b) v = fmaf(-col[q], sp.w[b][q], v);
                const float cb = pchol_col(v, s_ok[b] != 0, s_root[b]);
                col[b] = cb;
                lt[(int64_t)(m + b) * ld + i] = cb;
                dn = i == s_piv[b] ? 0.f : pchol_diag(dn, cb);
            }
        }
        diag[i] = dn;
        const uint64_t key = pchol_key(dn, rank ? rank[i] : (uint32_t)i);
        if (key > best) { best = key; bi = (int)i; }
    }
    block_argmax(best, bi, skey, sidx);
    if (threadIdx.x == 0) {
        pkey[(size_t)((a - 1) & 1) * kPcholParts + blockIdx.x] = best;
        pidx[(size_t)((a - 1) & 1) * kPcholParts + blockIdx.x] = bi;
        if (blockIdx.x == 0) {
            accepted[0] = a;
            accepted[1] = a;
            int *slot = state + 8 * ((a - 1) & 1);
            slot[0] = 0; slot[1] = sp.used; slot[2] = sp.order[a - 1]; slot[3] = __float_as_int(sp.dmax[a - 1]);
        }
    }
}

// in-batch step b >= 1 (after the planned steps): column m + b of the factor from the panel row of the candidate the
// step uses, corrected by the in-batch columns before it; the residual diagonal is updated in place and every workgroup
// leaves the argmax of its part of the updated diagonal.  Which candidate step b uses is decided at the START of its
// launch, by every workgroup for itself from the partials step b - 1 left (1024 keys: a few microseconds, against a launch
// of a single-workgroup kernel between every two steps -- 5 us per step, a third of the step); workgroup 0 records the
// decision for step b + 1 in the state slot no workgroup of this launch reads, and the partials alternate between two
// halves of their buffer for the same reason.  Nothing is read that was written earlier in the same launch.
__global__ __launch_bounds__(kBlock) void pchol_step_kernel(float *__restrict__ lt, int64_t ld, int m, int b, int nb,
                                                            const float *__restrict__ rowsT, float *__restrict__ diag,
                                                            const uint32_t *__restrict__ rank, int64_t n,
                                                            const int *__restrict__ cand, int *__restrict__ state,
                                                            float tol_abs, uint64_t *__restrict__ pkey, int *__restrict__ pidx,
                                                            const PcholPlan *__restrict__ plan, int *__restrict__ accepted)
{
    __shared__ uint64_t skey[kBlock / 64];
    __shared__ int sidx[kBlock / 64];
    __shared__ int s_stop, s_col;
    __shared__ float s_dmax, s_w[kPcholMaxBatch];
    if (b < plan->a) return;                       // (uniform) the plan's pass wrote this column
    uint64_t best = 0;
    int bi = -1;
    const int *prev = state + 8 * ((b - 1) & 1);
    int *cur = state + 8 * (b & 1);
    if (prev[0]) {                                 // (uniform) the batch ended at an earlier step
        if (blockIdx.x == 0 && threadIdx.x == 0) cur[0] = 1;
        return;
    }
    const uint64_t *pk = pkey + (size_t)((b - 1) & 1) * kPcholParts;
    const int *pi = pidx + (size_t)((b - 1) & 1) * kPcholParts;
    for (int x = threadIdx.x; x < (int)gridDim.x; x += kBlock) {
        const uint64_t key = pk[x];
        if (key > best) { best = key; bi = pi[x]; }
    }
    block_argmax(best, bi, skey, sidx);
    if (threadIdx.x == 0) {
        // the sequential algorithm's next pivot is the argmax `bi`: is its kernel row in this batch, still unused?
        const int used = prev[1];
        int j = -1;
        for (int q = 0; q < nb; ++q)
            if (!((used >> q) & 1) && cand[q] == bi) { j = q; break; }
        s_stop = j < 0;
        s_col = j;
        s_dmax = __uint_as_float((uint32_t)(best >> 32));
        if (blockIdx.x == 0) { cur[0] = j < 0; cur[1] = used | (j < 0 ? 0 : 1 << j); cur[2] = j; cur[3] = (int)(uint32_t)(best >> 32); }
    }
    __syncthreads();
    if (s_stop) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) accepted[0] = b + 1;     // what the host reads back after the batch
    const float dmax = s_dmax;
    const bool ok = dmax > tol_abs;
    const float root = sqrtf(fmaxf(dmax, 1e-30f));
    const int col_b = s_col;                    // the batch column (candidate) this step uses
    const int piv = cand[col_b];
    if ((int)threadIdx.x < b) s_w[threadIdx.x] = lt[(int64_t)(m + threadIdx.x) * ld + piv];    // L[pivot][m + q] of the in-batch columns so far
    __syncthreads();
    float w[kPcholMaxBatch];
#pragma unroll
    for (int q = 0; q < kPcholMaxBatch; ++q) w[q] = q < b ? s_w[q] : 0.f;
    best = 0;
    bi = -1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float v = rowsT[(int64_t)col_b * ld + i];
#pragma unroll
        for (int q = 0; q < kPcholMaxBatch; ++q)
            if (q < b) v = fmaf(-lt[(int64_t)(m + q) * ld + i], w[q], v);
        const float col = pchol_col(v, ok, root);
        lt[(int64_t)(m + b) * ld + i] = col;
        const float dn = i == piv ? 0.f : pchol_diag(diag[i], col);
        diag[i] = dn;
        const uint64_t key = pchol_key(dn, rank ? rank[i] : (uint32_t)i);
        if (key > best) { best = key; bi = (int)i; }
    }
    block_argmax(best, bi, skey, sidx);
    if (threadIdx.x == 0) {
        pkey[(size_t)(b & 1) * kPcholParts + blockIdx.x] = best;
        pidx[(size_t)(b & 1) * kPcholParts + blockIdx.x] = bi;
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------

static int gram_launch(const void *lt, bool half, int64_t ld, int kp, const float *R, int64_t n, int t, float *partial, hipStream_t s)
{
    const int ntiles = ceil_div(n, 64);
    for (int j0 = 0; j0 < kp; j0 += 128) {
        const int jt = std::min(8, (kp - j0) / 16);
        switch (jt) {
#define PLX_GRAM_CASE(J) case J: \
            if (half) pcg_gram_kernel<J, true><<<gram_blocks(), kBlock, 0, s>>>(lt, ld, kp, j0, R, n, t, ntiles, partial); \
            else pcg_gram_kernel<J, false><<<gram_blocks(), kBlock, 0, s>>>(lt, ld, kp, j0, R, n, t, ntiles, partial); \
            break;
        PLX_GRAM_CASE(1) PLX_GRAM_CASE(2) PLX_GRAM_CASE(3) PLX_GRAM_CASE(4)
        PLX_GRAM_CASE(5) PLX_GRAM_CASE(6) PLX_GRAM_CASE(7) PLX_GRAM_CASE(8)
#undef PLX_GRAM_CASE
        }
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

template <bool TRANSPOSED, bool HALF>
static int apply_launch(const void *lt, int64_t ld, int k, const float *R, int64_t n, int t, const float *Tm, const float *scal,
                        float *Z, int64_t ldz, float *partial, hipStream_t s)
{
    const int grid = ceil_div(n, kBlock * (HALF ? 2 : 1));
    switch (t) {
#define PLX_APPLY_CASE(T) case T: pcg_apply_kernel<T, TRANSPOSED, HALF><<<grid, kBlock, 0, s>>>(lt, ld, k, R, n, Tm, scal, Z, ldz, partial); break;
    PLX_APPLY_CASE(1) PLX_APPLY_CASE(2) PLX_APPLY_CASE(3) PLX_APPLY_CASE(4) PLX_APPLY_CASE(5) PLX_APPLY_CASE(6)
    PLX_APPLY_CASE(7) PLX_APPLY_CASE(8) PLX_APPLY_CASE(9) PLX_APPLY_CASE(10) PLX_APPLY_CASE(11) PLX_APPLY_CASE(12)
    PLX_APPLY_CASE(13) PLX_APPLY_CASE(14) PLX_APPLY_CASE(15) PLX_APPLY_CASE(16)
#undef PLX_APPLY_CASE
    default: set_error("plx_pcg: %d columns (1..16)", t); return PLX_ERR_INVALID;
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

static bool factor_shape_ok(const char *who, const void *lt, int64_t ld, int kp, int64_t n, int t)
{
    if (!lt) { set_error("%s: NULL factor", who); return false; }
    if (n < 1 || ld < n || (ld & 63) || kp < 16 || (kp & 15) || kp > 1024 || t < 1 || t > kPcgCols) {
        set_error("%s: n = %lld, ld = %lld (a multiple of 64, >= n), kp = %d (a multiple of 16, <= 1024), %d columns (1..16)",
                  who, (long long)n, (long long)ld, kp, t);
        return false;
    }
    if ((reinterpret_cast<uintptr_t>(lt) & 15) != 0) { set_error("%s: the factor must be 16-byte aligned", who); return false; }
    return true;
}

// work layout of the pcg calls (floats): gram partials | apply partials
static size_t pcg_gram_floats(int kp) { return (size_t)kGramBlocksMax * kPcgCols * kp; }

// work layout of the pchol calls (bytes)
struct PcholWork {
    float *rowsT; float *W; uint64_t *pkey; int *pidx; int *state; uint64_t *knext; PcholPlan *plan;
    static size_t bytes(int64_t ld, int kp)
    {
        return (size_t)kPcgCols * ld * 4 + (size_t)kp * kPcgCols * 4 + (size_t)kPcholParts * (kPcholMaxBatch + 1) * 12 + 512 + sizeof(PcholPlan);
    }
    PcholWork(void *base, int64_t ld, int kp)
    {
        char *p = (char *)base;
        rowsT = (float *)p; p += (size_t)kPcgCols * ld * 4;
        W = (float *)p; p += (size_t)kp * kPcgCols * 4;
        pkey = (uint64_t *)p; p += (size_t)kPcholParts * (kPcholMaxBatch + 1) * 8;
        pidx = (int *)p; p += (size_t)kPcholParts * (kPcholMaxBatch + 1) * 4;
        state = (int *)p; p += 128;
        knext = (uint64_t *)p; p += 128;
        plan = (PcholPlan *)p;
    }
};

}  // namespace plx

using namespace plx;

extern "C" int64_t plx_pcg_work_floats(int64_t n, int kp, int t)
{
    if (n < 1 || kp < 16 || t < 1) return -1;
    return (int64_t)pcg_gram_floats(kp) + (int64_t)ceil_div(n, kBlock) * kPcgCols;
}

static bool factor_type_ok(const char *who, int factor_type)
{
    if (factor_type == PLX_FACTOR_F32 || factor_type == PLX_FACTOR_F16) return true;
    set_error("%s: factor_type %d (PLX_FACTOR_F32 or PLX_FACTOR_F16)", who, factor_type);
    return false;
}

extern "C" int plx_pcg_project(const void *d_lt, int factor_type, int64_t ld, int kp, const float *d_r, int64_t n, int t,
                               const double *d_cinv, float *d_t, float *d_work, void *stream)
{
    if (!factor_type_ok("plx_pcg_project", factor_type) || !factor_shape_ok("plx_pcg_project", d_lt, ld, kp, n, t)) return PLX_ERR_INVALID;
    if (!d_r || !d_cinv || !d_t || !d_work) { set_error("plx_pcg_project: NULL argument"); return PLX_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    PLX_TRY(gram_launch(d_lt, factor_type == PLX_FACTOR_F16, ld, kp, d_r, n, t, d_work, s));
    const int groups = std::max(1, 1024 / kp);
    pcg_project_kernel<<<t, 1024, (size_t)kp * (1 + groups) * sizeof(double), s>>>(d_work, gram_blocks(), kp, d_cinv, d_t);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_pcg_factor_to_half(const float *d_lt, int64_t ld, int kp, void *d_lt_half, void *stream)
{
    if (!d_lt || !d_lt_half || ld < 64 || (ld & 63) || kp < 16 || (kp & 15)) { set_error("plx_pcg_factor_to_half: bad argument"); return PLX_ERR_INVALID; }
    const int64_t quads = (int64_t)kp * ld / 4;
    pcg_to_half_kernel<<<ceil_div(quads, kBlock), kBlock, 0, (hipStream_t)stream>>>(reinterpret_cast<const float4 *>(d_lt), quads,
                                                                                   reinterpret_cast<uint2 *>(d_lt_half));
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_pcg_apply(const void *d_lt, int factor_type, int64_t ld, int kp, int k, const float *d_r, int64_t n, int t,
                             const float *d_t, const float *d_scale, float *d_z, float *d_rz, float *d_work, void *stream)
{
    if (!factor_type_ok("plx_pcg_apply", factor_type) || !factor_shape_ok("plx_pcg_apply", d_lt, ld, kp, n, t)) return PLX_ERR_INVALID;
    if (!d_r || !d_t || !d_scale || !d_z || !d_work) { set_error("plx_pcg_apply: NULL argument"); return PLX_ERR_INVALID; }
    if (k < 0 || k > kp) { set_error("plx_pcg_apply: k = %d outside 0..kp", k); return PLX_ERR_INVALID; }
    if (t % 4 == 0 && ((reinterpret_cast<uintptr_t>(d_r) | reinterpret_cast<uintptr_t>(d_z)) & 15)) {
        set_error("plx_pcg_apply: rows of a multiple of 4 columns must be 16-byte aligned");
        return PLX_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    float *part = d_work + pcg_gram_floats(kp);
    const bool half = factor_type == PLX_FACTOR_F16;
    if (half) PLX_TRY((apply_launch<false, true>(d_lt, ld, k, d_r, n, t, d_t, d_scale, d_z, 0, part, s)));
    else PLX_TRY((apply_launch<false, false>(d_lt, ld, k, d_r, n, t, d_t, d_scale, d_z, 0, part, s)));
    if (d_rz) PLX_TRY(coldot_final(part, ceil_div(n, kBlock * (half ? 2 : 1)), t, d_rz, s));
    return PLX_OK;
}

// where plx_pcg_apply(d_rz = NULL) leaves the per-workgroup partial sums of <R, Z>: d_work + offset, `rows` rows of t floats
extern "C" int64_t plx_pcg_rz_partial_offset(int kp) { return (kp >= 16 && (kp & 15) == 0) ? (int64_t)pcg_gram_floats(kp) : -1; }
extern "C" int plx_pcg_rz_partial_rows(int64_t n, int factor_type)
{
    if (n < 0 || (factor_type != PLX_FACTOR_F16 && factor_type != PLX_FACTOR_F32)) return -1;
    return ceil_div(n, kBlock * (factor_type == PLX_FACTOR_F16 ? 2 : 1));
}

extern "C" int plx_pcg_step_direction(float *d_p, const float *d_z, const float *d_rz_new, const float *d_rz,
                                      const float *d_rr, const float *d_active, const float *d_b_norm, float tol, int64_t n,
                                      int vd, float *d_beta, float *d_active_out, void *stream)
{
    if (!d_p || !d_z || !d_rz_new || !d_rz || !d_rr || !d_active || !d_b_norm || !d_beta || !d_active_out) {
        set_error("plx_pcg_step_direction: NULL argument");
        return PLX_ERR_INVALID;
    }
    if (d_active == d_active_out) { set_error("plx_pcg_step_direction: active and active_out must be different buffers"); return PLX_ERR_INVALID; }
    if (n < 0 || vd < 1 || vd > kBlock) { set_error("plx_pcg_step_direction: vd = %d outside 1..%d", vd, kBlock); return PLX_ERR_INVALID; }
    const int64_t total = n * vd;
    hipStream_t s = (hipStream_t)stream;
    if (total > 0 && (total & 3) == 0 && ((reinterpret_cast<uintptr_t>(d_p) | reinterpret_cast<uintptr_t>(d_z)) & 15) == 0) {
        const int64_t quads = total / 4;
        pcg_step_direction4_kernel<<<ceil_div(quads, kBlock), kBlock, 0, s>>>(
            reinterpret_cast<float4 *>(d_p), reinterpret_cast<const float4 *>(d_z), d_rz_new, d_rz, d_rr, d_active, d_b_norm, tol,
            quads, vd, d_beta, d_active_out);
    } else {
        const int grid = total > 0 ? ceil_div(total, kBlock) : 1;
        pcg_step_direction_kernel<<<grid, kBlock, 0, s>>>(d_p, d_z, d_rz_new, d_rz, d_rr, d_active, d_b_norm, tol, total, vd,
                                                         d_beta, d_active_out);
    }
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int64_t plx_pchol_work_bytes(int64_t ld, int kp)
{
    if (ld < 64 || (ld & 63) || kp < 16 || (kp & 15)) return -1;
    return (int64_t)PcholWork::bytes(ld, kp);
}

extern "C" int plx_pchol_select(const float *d_diag, const uint32_t *d_rank, int64_t n, int nb, int64_t ld, int kp,
                                int32_t *d_cand, void *d_work, void *stream)
{
    if (!d_diag || !d_cand || !d_work) { set_error("plx_pchol_select: NULL argument"); return PLX_ERR_INVALID; }
    if (n < 1 || nb < 1 || nb > kPcholMaxBatch || nb > n || plx_pchol_work_bytes(ld, kp) < 0 || ld < n) {
        set_error("plx_pchol_select: n = %lld, batch %d (1..%d)", (long long)n, nb, kPcholMaxBatch);
        return PLX_ERR_INVALID;
    }
    PcholWork w(d_work, ld, kp);
    hipStream_t s = (hipStream_t)stream;
    const int parts = std::min<int64_t>(kPcholParts, ceil_div(n, kBlock));
    // one key more than the batch: the largest entry that is NOT a candidate bounds what the batch's plan may assume
    pchol_top_partial_kernel<<<parts, kBlock, 0, s>>>(d_diag, d_rank, n, nb + 1, w.pkey, w.pidx);
    pchol_top_final_kernel<<<1, 1024, 0, s>>>(w.pkey, w.pidx, parts, nb + 1, d_cand, w.knext);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_pchol_onehot(const int32_t *d_cand, int nb, int64_t n, int t, float *d_rhs, void *stream)
{
    if (!d_cand || !d_rhs) { set_error("plx_pchol_onehot: NULL argument"); return PLX_ERR_INVALID; }
    if (nb < 1 || nb > kPcholMaxBatch || t < nb || n < 1) { set_error("plx_pchol_onehot: batch %d, %d columns", nb, t); return PLX_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    PLX_HIP_TRY(hipMemsetAsync(d_rhs, 0, (size_t)n * t * 4, s));
    pchol_onehot_kernel<<<1, 64, 0, s>>>(d_cand, nb, t, d_rhs);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_pchol_factor_batch(float *d_lt, int64_t ld, int kp, int m_done, const float *d_rows, int t,
                                      const float *d_scale, const int32_t *d_cand, int nb, float *d_diag,
                                      const uint32_t *d_rank, int64_t n, float tol_abs, int exact_steps, int32_t *d_accepted,
                                      void *d_work, void *stream)
{
    if (!factor_shape_ok("plx_pchol_factor_batch", d_lt, ld, kp, n, t)) return PLX_ERR_INVALID;
    if (!d_rows || !d_scale || !d_cand || !d_diag || !d_accepted || !d_work) { set_error("plx_pchol_factor_batch: NULL argument"); return PLX_ERR_INVALID; }
    if (nb < 1 || nb > kPcholMaxBatch || nb > t || m_done < 0 || m_done + nb > kp) {
        set_error("plx_pchol_factor_batch: batch %d of %d columns at %d finished columns of %d", nb, t, m_done, kp);
        return PLX_ERR_INVALID;
    }
    PcholWork w(d_work, ld, kp);
    hipStream_t s = (hipStream_t)stream;
    if (m_done > 0) pchol_gather_kernel<<<ceil_div((int64_t)m_done * kPcgCols, kBlock), kBlock, 0, s>>>(d_lt, ld, m_done, d_cand, nb, w.W);
    // panel: rowsT[b][i] = scale * rows[i][b] - sum_{j < m} L[i][j] L[cand_b][j]
    PLX_TRY((apply_launch<true, false>(d_lt, ld, m_done, d_rows, n, t, w.W, d_scale, w.rowsT, ld, nullptr, s)));
    const int parts = std::min<int64_t>(kPcholParts, ceil_div(n, kBlock));
    pchol_plan_kernel<<<1, kBlock, 0, s>>>(w.rowsT, ld, d_cand, nb, d_diag, d_rank, w.knext, tol_abs, w.plan);
    pchol_multi_step_kernel<<<parts, kBlock, 0, s>>>(d_lt, ld, m_done, w.rowsT, d_diag, d_rank, n, d_cand, w.plan, tol_abs, w.pkey,
                                                     w.pidx, w.state, d_accepted);
    if (exact_steps)
        for (int b = 1; b < nb; ++b)
            pchol_step_kernel<<<parts, kBlock, 0, s>>>(d_lt, ld, m_done, b, nb, w.rowsT, d_diag, d_rank, n, d_cand, w.state, tol_abs,
                                                       w.pkey, w.pidx, w.plan, d_accepted);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

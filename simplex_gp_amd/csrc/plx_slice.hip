// plx_slice.hip -- the slice kernels of the per-MVM path (h:497-510) and the fused position gradient (py:113-123).
// Overview of the per-MVM path, value-row layout and shared helpers: plx_kernels.h.

#include "plx_kernels.h"

namespace plx {

// ----------------------------------------------------------------------------
// slice: out[row(p)][c] = sum_r w_r * values[v_r][c] / (1 + 2^-d)     (h:502-509)
// p runs in lattice order; the result is scattered to the caller's row order.  The reference divides
// every term; here every term is multiplied by the rounded reciprocal rden = 1 / (1 + 2^-d): an fp32
// division is ~10 vector instructions, and 4(d+1) of them per thread were over half of the multi-column
// kernel's instruction stream (vd = 11 slice 101 -> 75 us).  The two differ by <= 1 ulp per term.

// vd == 1: all d+1 (id, weight) loads first, then all gathers, then the ordered sum
template <int D1>
__global__ __launch_bounds__(kBlock) void slice_v1_kernel(const int *__restrict__ evid,
                                                          const float *__restrict__ ew,
                                                          const uint32_t *__restrict__ perm, int n, int own_begin,
                                                          int n_own, const float *__restrict__ values, float rden,
                                                          float *__restrict__ out, int ntiles, int remap,
                                                          const float *__restrict__ affine, const float *__restrict__ src)
{
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int pl = tile * kBlock + threadIdx.x;
    if (pl >= n_own) return;
    const int p = own_begin + pl;
    int v[D1];
    float w[D1], g[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) {
        v[r] = evid[(size_t)r * n + p];
        w[r] = ew[(size_t)r * n + p];
    }
    const int row = perm ? (int)perm[p] - own_begin : pl;
#pragma unroll
    for (int r = 0; r < D1; ++r) g[r] = values[v[r]];
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < D1; ++r) acc += w[r] * g[r] * rden;
    if (affine) acc = affine[0] * acc + affine[1] * src[row];      // out = a K src + b src (plx_apply_affine)
    out[row] = acc;
}

// vd > 1: one thread per (point, 16-byte chunk).  D1 > 0: d + 1 compiled in -- all (vertex, weight) pairs are loaded
// first, then all row gathers are issued, then the ordered sum (with a runtime trip count every corner waited for its
// own pair and then for its own gather: vd = 12 slice 100 us at N = 1e6); D1 = 0: runtime d + 1 (d + 1 > kSliceMaxD1).
constexpr int kSliceMaxD1 = 20;
template <int D1>
__global__ __launch_bounds__(kBlock) void slice_vec_kernel(const int *__restrict__ evid,
                                                           const float *__restrict__ ew,
                                                           const uint32_t *__restrict__ perm, int n, int own_begin,
                                                           int n_own, int d1, const float4 *__restrict__ values,
                                                           int nch, int vd, float rden, float *__restrict__ out,
                                                           int ntiles, int remap, const float *__restrict__ affine,
                                                           const float *__restrict__ src, float *__restrict__ dot_partial)
{
    __shared__ float4 red[kBlock];
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;                                  // the whole workgroup leaves together
    const int64_t item = (int64_t)tile * kBlock + threadIdx.x;
    const bool live = item < (int64_t)n_own * nch;
    float4 prod = f4_zero();
    if (live) {
        const int pl = (int)(item / nch), ch = (int)(item - (int64_t)pl * nch);
        const int p = own_begin + pl;
        float4 acc = f4_zero();
        if constexpr (D1 > 0) {
            int v[D1];
            float w[D1];
            float4 g[D1];
#pragma unroll
            for (int r = 0; r < D1; ++r) {
                v[r] = evid[(size_t)r * n + p];
                w[r] = ew[(size_t)r * n + p];
            }
#pragma unroll
            for (int r = 0; r < D1; ++r) g[r] = values[(size_t)v[r] * nch + ch];
#pragma unroll
            for (int r = 0; r < D1; ++r) {
                acc.x += w[r] * g[r].x * rden; acc.y += w[r] * g[r].y * rden;
                acc.z += w[r] * g[r].z * rden; acc.w += w[r] * g[r].w * rden;
            }
        } else {
            for (int r = 0; r < d1; ++r) {
                const int v = evid[(size_t)r * n + p];
                const float w = ew[(size_t)r * n + p];
                const float4 g = values[(size_t)v * nch + ch];
                acc.x += w * g.x * rden; acc.y += w * g.y * rden; acc.z += w * g.z * rden; acc.w += w * g.w * rden;
            }
        }
        const size_t row = perm ? (size_t)((int)perm[p] - own_begin) : (size_t)pl;
        float *o = out + row * vd + 4 * ch;
        const int left = vd - 4 * ch;
        if (affine) {
            const float a = affine[0], b = affine[1];
            const float *sp = src + row * vd + 4 * ch;
            float4 sv = f4_zero();
            sv.x = sp[0];
            if (left > 1) sv.y = sp[1];
            if (left > 2) sv.z = sp[2];
            if (left > 3) sv.w = sp[3];
            acc.x = a * acc.x + b * sv.x;
            if (left > 1) acc.y = a * acc.y + b * sv.y;
            if (left > 2) acc.z = a * acc.z + b * sv.z;
            if (left > 3) acc.w = a * acc.w + b * sv.w;
            // <src, out> per column for the CG caller (plx_apply_affine_dot); padding columns stay 0
            prod = make_float4(sv.x * acc.x, left > 1 ? sv.y * acc.y : 0.f, left > 2 ? sv.z * acc.z : 0.f,
                               left > 3 ? sv.w * acc.w : 0.f);
        }
        if (left >= 4 && (vd & 3) == 0) {
            *reinterpret_cast<float4 *>(o) = acc;
        } else {
            o[0] = acc.x;
            if (left > 1) o[1] = acc.y;
            if (left > 2) o[2] = acc.z;
            if (left > 3) o[3] = acc.w;
        }
    }
    if (dot_partial) {
        // one partial sum per (workgroup, column), in a fixed order: reproducible.  Two steps: K threads per column sum
        // every K-th term of that column (a single thread per column walked all kBlock / nch terms through LDS: 85
        // dependent reads with 12 lanes busy while the workgroup's other waves held their registers), then one thread
        // per column adds the K partial sums in order.
        __shared__ float part[kBlock];
        red[threadIdx.x] = prod;
        __syncthreads();
        const int cols = 4 * nch;                                        // <= 64 for the kernels that use this path
        const int K = kBlock / cols;                                     // >= 4
        const int base = (int)(((int64_t)tile * kBlock) % nch);          // chunk index of thread 0
        if ((int)threadIdx.x < cols * K) {
            const int col = threadIdx.x % cols, k = threadIdx.x / cols;
            const int ch = col >> 2, j = col & 3;
            float s = 0.f;
            for (int t = (ch - base + nch) % nch + k * nch; t < kBlock; t += K * nch) s += reinterpret_cast<const float *>(&red[t])[j];
            part[k * cols + col] = s;
        }
        __syncthreads();
        if ((int)threadIdx.x < cols) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += part[k * cols + threadIdx.x];
            dot_partial[(size_t)tile * cols + threadIdx.x] = s;
        }
    }
}

int slice_impl(plx_lattice *L, const float *d_values, int vd, float *d_out, hipStream_t stream, const float *d_affine,
               const float *d_src, float *d_dot_partial)
{
    const int n_own = (int)(L->own_end - L->own_begin);
    if (n_own == 0) { L->kn_slice = ""; tmark(L, stream); return PLX_OK; }
    bool splat_blocks = false, slice_blocks = false;
    PLX_TRY(choose_paths(L, vd, stream, &splat_blocks, &slice_blocks));
    if (slice_blocks) return slice_block_impl(L, d_values, d_out, stream, d_affine, d_src);
    const int *evid = L->evid.as<int>();
    const float *ew = L->ew.as<float>();
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const int n = (int)L->n, ob = (int)L->own_begin;
    if (vd == 1) {
        const int nt = ceil_div(n_own, kBlock);
        const int grid = tile_grid(nt, g_xcd_remap);
        switch (L->d + 1) {
#define PLX_CASE(D1) \
    case D1: slice_v1_kernel<D1><<<grid, kBlock, 0, stream>>>(evid, ew, perm, n, ob, n_own, d_values, 1.0f / L->slice_denom, d_out, nt, g_xcd_remap, d_affine, d_src); break;
            PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9)
            PLX_CASE(10) PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17)
            PLX_CASE(18) PLX_CASE(19) PLX_CASE(20) PLX_CASE(21) PLX_CASE(22) PLX_CASE(23) PLX_CASE(24) PLX_CASE(25)
            PLX_CASE(26) PLX_CASE(27) PLX_CASE(28) PLX_CASE(29) PLX_CASE(30) PLX_CASE(31) PLX_CASE(32) PLX_CASE(33)
#undef PLX_CASE
        }
        L->kn_slice = "slice_v1_kernel";
    } else {
        L->kn_slice = "slice_vec_kernel";
        const int nch = values_stride(vd) / 4;
        // caller row order on LARGE outputs: slice into a lattice-ordered scratch of whole 16-byte chunks (coalesced), then
        // gather the rows out.  Scattering 4 vd-byte rows from the slice kernel costs 3.8x the output in memory-side writes
        // once the output no longer sits in the caches (N = 4e6, vd = 11: slice 422 -> 371 us with the two steps); while it
        // does, the extra pass costs more than it saves (N = 1e6, vd = 11: 65.7 -> 74.3 us), so the gate is the output size.
        // Not used when the caller wants the column dots (those callers run in lattice row order anyway).
        const bool two_step = perm != nullptr && d_dot_partial == nullptr && g_unpermute_gather != 0 &&
                              (int64_t)n_own * vd * 4 > (96ll << 20);
        float *slice_out = d_out;
        const float *slice_affine = d_affine;
        int slice_vd = vd;
        if (two_step) {
            PLX_TRY(ensure_inv_perm(L, stream));
            PLX_TRY(ensure(L->ssrc, (size_t)n_own * nch * 16 + 16));
            slice_out = L->ssrc.as<float>();
            slice_affine = nullptr;
            slice_vd = 4 * nch;
            perm = nullptr;
        }
        const int nt = ceil_div((int64_t)n_own * nch, kBlock);
        const float4 *v4 = reinterpret_cast<const float4 *>(d_values);
        const float rden = 1.0f / L->slice_denom;
        const int grid = tile_grid(nt, g_xcd_remap);
        switch (L->d + 1 <= kSliceMaxD1 ? L->d + 1 : 0) {
#define PLX_CASE(D1) \
    case D1: slice_vec_kernel<D1><<<grid, kBlock, 0, stream>>>(evid, ew, perm, n, ob, n_own, L->d + 1, v4, nch, slice_vd, rden, slice_out, nt, g_xcd_remap, slice_affine, d_src, d_dot_partial); break;
            PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9) PLX_CASE(10)
            PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17) PLX_CASE(18)
            PLX_CASE(19) PLX_CASE(20)
#undef PLX_CASE
        default: slice_vec_kernel<0><<<grid, kBlock, 0, stream>>>(evid, ew, perm, n, ob, n_own, L->d + 1, v4, nch, slice_vd, rden, slice_out, nt, g_xcd_remap, slice_affine, d_src, d_dot_partial); break;
        }
        if (two_step) {
            L->kn_slice = "slice_vec_kernel+unpermute_rows_kernel";
            PLX_TRY(unpermute_rows(L, slice_out, vd, d_out, d_affine, d_src, stream));      // (names the kernel it launched)
        }
    }
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ----------------------------------------------------------------------------
// fused position gradient (py:113-123): splat of the stacked matrix straight from
// the packed records, the usual blur, then slice and contraction in one kernel --
// neither the stacked matrix nor its filtered image ever reach memory.

// One wave per point: the lanes slice the point's 2L(1+d) filtered columns (same arithmetic as
// slice_vec_kernel), park them in LDS, then lane k < d forms
//   grad_x[k] = -2 sum_l ( s_l x_k wg_l - s_l wgx_{l,k} + g_l x_k ws_l - g_l wsx_{l,k} )      (py:122)
// and lane l < L stores grad_src[l] = wg_l (py:123).
template <int MAXCH>
__global__ __launch_bounds__(kBlock) void slice_contract_kernel(const int *__restrict__ evid, const float *__restrict__ ew,
                                                                const uint32_t *__restrict__ perm, int n, int own_begin,
                                                                int n_own, int d1, const float4 *__restrict__ values,
                                                                int nch, const float *__restrict__ rec, int recw, int L,
                                                                int d, float rden, float *__restrict__ grad_x,
                                                                float *__restrict__ grad_src, int ntiles, int remap)
{
    __shared__ float4 f4s[kBlock / 64][64 * MAXCH];
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = tile * (kBlock / 64) + wave;
    if (pl >= n_own) return;                       // whole waves leave together; no workgroup barrier below
    const int p = own_begin + pl;
    float4 acc[MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) acc[q] = f4_zero();
    int my_v = 0;
    float my_w = 0.f;
    if (lane < d1) { my_v = evid[(size_t)lane * n + p]; my_w = ew[(size_t)lane * n + p]; }   // d1 <= 33 < 64
    int r = 0;
    for (; r + 3 <= d1; r += 3) {
        float4 gq[3][MAXCH];
        float w[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int v = __shfl(my_v, r + u);
            w[u] = __shfl(my_w, r + u);
#pragma unroll
            for (int q = 0; q < MAXCH; ++q)
                gq[u][q] = (lane + 64 * q < nch) ? values[(size_t)v * nch + lane + 64 * q] : f4_zero();
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int q = 0; q < MAXCH; ++q) {
                acc[q].x += w[u] * gq[u][q].x * rden; acc[q].y += w[u] * gq[u][q].y * rden;
                acc[q].z += w[u] * gq[u][q].z * rden; acc[q].w += w[u] * gq[u][q].w * rden;
            }
    }
    for (; r < d1; ++r) {
        const int v = __shfl(my_v, r);
        const float w = __shfl(my_w, r);
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            const int ch = lane + 64 * q;
            if (ch < nch) {
                const float4 gq = values[(size_t)v * nch + ch];
                acc[q].x += w * gq.x * rden; acc[q].y += w * gq.y * rden;
                acc[q].z += w * gq.z * rden; acc[q].w += w * gq.w * rden;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) f4s[wave][lane + 64 * q] = acc[q];
    __builtin_amdgcn_wave_barrier();               // the LDS row is private to this wave
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the ds_writes above have landed
    const float *f = reinterpret_cast<const float *>(f4s[wave]);
    const float *rp = rec + (size_t)pl * recw;
    const size_t row = perm ? (size_t)((int)perm[p] - own_begin) : (size_t)pl;
    const int half = L * (1 + d);
    if (lane < d) {
        const float xk = rp[2 * L + lane];
        float a = 0.f;
        for (int l = 0; l < L; ++l) {
            const float sv = rp[L + l], gv = rp[l];
            a += sv * xk * f[l] - sv * f[L + l * d + lane] + gv * xk * f[half + l] - gv * f[half + L + l * d + lane];
        }
        grad_x[row * d + lane] = -2.0f * a;
    }
    if (grad_src)
        for (int l = lane; l < L; l += 64) grad_src[row * L + l] = f[l];
}

// The same with the corner count compiled in (round 6).  What the run-time form above leaves on the table: its d+1
// row gathers go out three at a time from a loop the compiler cannot pipeline (a wave waits three memory latencies
// per point), and its contraction runs on d lanes for 4 L serial LDS reads.  Here
//   * all d+1 rows (up to kContractLoads 16-byte loads per lane) are requested before the first is used, the row
//     bases in scalar registers (compile-time lane of v_readlane);
//   * the contraction uses all 64 lanes: lane = (l-group, k), each lane sums its l's, one xor-butterfly over the
//     l-groups.  The slice sums are formed in corner order exactly as above (bit-identical f); the contraction's
//     sum over l is associated differently (within rounding of the run-time form).
// One point per wave stays (PPW = 1): walking 2 / 4 / 8 / 16 points per wave with the next point's corner ids, weights
// and record fetched behind the current point's rows was measured SLOWER (N = 1e6, d = 8, L = 11, m = 1.73e6 / 1.1e6,
// us: run-time form 932 / 853; PPW 1 / 2 / 4 / 8 / 16 = 686 / 749 / 758 / 802 / 860 and 617 / 638 / 673 / 703 / 748):
// the launch's parallelism is waves, and a wave that loops holds its registers while it waits.
constexpr int kContractLoads = 12;
constexpr int kContractPoints = 1;

template <int D1, int MAXCH, int PPW = kContractPoints>
__global__ __launch_bounds__(kBlock) void slice_contract_d_kernel(const int *__restrict__ evid, const float *__restrict__ ew,
                                                                  const uint32_t *__restrict__ perm, int n, int own_begin,
                                                                  int n_own, const float4 *__restrict__ values, int nch,
                                                                  const float *__restrict__ rec, int recw, int L, float rden,
                                                                  float *__restrict__ grad_x, float *__restrict__ grad_src,
                                                                  int ntiles, int remap)
{
    constexpr int d = D1 - 1;
    constexpr int DK = d <= 1 ? 1 : d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : 32;   // lanes per l-group
    constexpr int G = 64 / DK;                                                                  // l-groups per wave
    constexpr int RB = (kContractLoads / MAXCH) < D1 ? (kContractLoads / MAXCH) : D1;           // rows in flight
    __shared__ float4 f4s[kBlock / 64][64 * MAXCH];
    __shared__ float recs[kBlock / 64][64];
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl0 = tile * (kBlock / 64) * PPW + wave;
    if (pl0 >= n_own) return;                      // whole waves leave together; no workgroup barrier below
    const int half = L * (1 + d);
    const int k = lane & (DK - 1), lg = lane / DK;
    const int kk = k < d ? k : d - 1;
    const float *f = reinterpret_cast<const float *>(f4s[wave]);
    const float *rc = recs[wave];

    int n_v = 0;
    float n_w = 0.f, n_rec = 0.f;
    auto fetch = [&](int pl) {
        const int p = own_begin + pl;
        n_v = 0; n_w = 0.f; n_rec = 0.f;
        if (pl < n_own) {
            if (lane < D1) { n_v = evid[(size_t)lane * n + p]; n_w = ew[(size_t)lane * n + p]; }
            if (lane < recw) n_rec = rec[(size_t)pl * recw + lane];
        }
    };
    fetch(pl0);
#pragma unroll 1
    for (int it = 0; it < PPW; ++it) {
        const int pl = pl0 + it * (kBlock / 64);
        if (pl >= n_own) break;                    // wave-uniform
        const int my_v = n_v;
        const float my_w = n_w, my_rec = n_rec;
        if constexpr (PPW > 1) fetch(pl + (kBlock / 64));   // the next point's ids / weights / record, behind this point's rows
        float4 acc[MAXCH];
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) acc[q] = f4_zero();
        // a batch of C rows: every load of the batch before its first sum (C compile-time, the first row in a scalar register)
        auto batch = [&](int r0, auto cnt) {
            constexpr int C = decltype(cnt)::value;
            float4 gq[C][MAXCH];
            float w[C];
#pragma unroll
            for (int u = 0; u < C; ++u) {
                const int v = __builtin_amdgcn_readlane(my_v, r0 + u);
                w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), r0 + u));
                const float4 *row = values + (size_t)v * nch;
#pragma unroll
                for (int q = 0; q < MAXCH; ++q) gq[u][q] = (lane + 64 * q < nch) ? row[lane + 64 * q] : f4_zero();
            }
#pragma unroll
            for (int u = 0; u < C; ++u)
#pragma unroll
                for (int q = 0; q < MAXCH; ++q) {
                    acc[q].x += w[u] * gq[u][q].x * rden; acc[q].y += w[u] * gq[u][q].y * rden;
                    acc[q].z += w[u] * gq[u][q].z * rden; acc[q].w += w[u] * gq[u][q].w * rden;
                }
        };
        constexpr int NB = D1 / RB, REM = D1 % RB;
        if constexpr (NB == 1) batch(0, std::integral_constant<int, RB>{});
        else {
#pragma unroll 1
            for (int bq = 0; bq < NB; ++bq) batch(bq * RB, std::integral_constant<int, RB>{});   // (a real loop: the register budget is one batch)
        }
        if constexpr (REM > 0) batch(NB * RB, std::integral_constant<int, REM>{});
        __builtin_amdgcn_wave_barrier();           // the previous point's LDS reads are done (one wave: in order)
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) f4s[wave][lane + 64 * q] = acc[q];
        recs[wave][lane] = my_rec;
        __builtin_amdgcn_wave_barrier();           // the LDS rows are private to this wave
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the ds_writes above have landed
        const size_t row = perm ? (size_t)((int)perm[own_begin + pl] - own_begin) : (size_t)pl;
        const float xk = rc[2 * L + kk];
        float a = 0.f;
        for (int l0 = 0; l0 < L; l0 += G) {        // uniform trip count: every lane takes part in the butterfly below
            const int l = l0 + lg;
            const bool on = l < L;
            const int lc = on ? l : 0;
            const float gv = rc[lc], sv = rc[L + lc];
            const float t = sv * xk * f[lc] - sv * f[L + lc * d + kk] + gv * xk * f[half + lc] - gv * f[half + L + lc * d + kk];
            a += on ? t : 0.f;
        }
#pragma unroll
        for (int off = DK; off < 64; off <<= 1) a += __shfl_xor(a, off);
        if (lane < d) grad_x[row * d + lane] = -2.0f * a;
        if (grad_src)
            for (int l = lane; l < L; l += 64) grad_src[row * L + l] = f[l];
    }
}

template <int MAXCH>
static bool launch_slice_contract_d(plx_lattice *lat, const float4 *res, int nch, const float *rec, int recw, int L,
                                    float *d_grad_x, float *d_grad_src, hipStream_t stream)
{
    const int n_own = (int)(lat->own_end - lat->own_begin);
    const uint32_t *perm = lat->lattice_rows ? nullptr : lat->perm.as<uint32_t>();
    const int nt = ceil_div(n_own, (kBlock / 64) * kContractPoints);
    const int grid = tile_grid(nt, g_xcd_remap);
    switch (lat->d + 1) {
#define PLX_CASE(D1) \
    case D1: slice_contract_d_kernel<D1, MAXCH><<<grid, kBlock, 0, stream>>>(lat->evid.as<int>(), lat->ew.as<float>(), perm, (int)lat->n, (int)lat->own_begin, n_own, res, nch, rec, recw, L, 1.0f / lat->slice_denom, d_grad_x, d_grad_src, nt, g_xcd_remap); return true;
        PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9) PLX_CASE(10)
        PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17) PLX_CASE(18)
        PLX_CASE(19) PLX_CASE(20) PLX_CASE(21)
#undef PLX_CASE
    default: return false;
    }
}

int backward_impl(plx_lattice *lat, const float *d_g, const float *d_src, const float *d_x, int L, float *d_grad_x,
                  float *d_grad_src, hipStream_t stream)
{
    const int d = lat->d, W = 2 * L * (1 + d), vdp = values_stride(W), nch = vdp / 4;
    const int n_own = (int)(lat->own_end - lat->own_begin);
    const int recw = backward_record_width(L, d);
    PLX_TRY(ensure(lat->val_a, (size_t)lat->m * vdp * 4));
    PLX_TRY(ensure(lat->val_b, (size_t)lat->m * vdp * 4));
    float *va = lat->val_a.as<float>(), *vb = lat->val_b.as<float>();
    lat->tev_n = 0;
    tmark(lat, stream);
    PLX_TRY(splat_stack_impl(lat, d_g, d_src, d_x, L, va, stream));      // pack + splat of the never-stored stack
    int in_b = 0;
    PLX_TRY(blur_impl(lat, va, vb, W, &in_b, stream));
    const float4 *res = reinterpret_cast<const float4 *>(in_b ? vb : va);
    const uint32_t *perm = lat->lattice_rows ? nullptr : lat->perm.as<uint32_t>();
    const float *rec = lat->rec.as<float>();
    const int nt = ceil_div(n_own, kBlock / 64);
    const int sgrid = tile_grid(nt, g_xcd_remap);
    const bool compiled = g_contract_v != 0 &&
                          (nch <= 64 ? launch_slice_contract_d<1>(lat, res, nch, rec, recw, L, d_grad_x, d_grad_src, stream)
                                     : launch_slice_contract_d<2>(lat, res, nch, rec, recw, L, d_grad_x, d_grad_src, stream));
    if (compiled) {
    } else if (nch <= 64)
        slice_contract_kernel<1><<<sgrid, kBlock, 0, stream>>>(lat->evid.as<int>(), lat->ew.as<float>(), perm, (int)lat->n,
                                                              (int)lat->own_begin, n_own, d + 1, res, nch, rec, recw, L, d,
                                                              1.0f / lat->slice_denom, d_grad_x, d_grad_src, nt, g_xcd_remap);
    else
        slice_contract_kernel<2><<<sgrid, kBlock, 0, stream>>>(lat->evid.as<int>(), lat->ew.as<float>(), perm, (int)lat->n,
                                                              (int)lat->own_begin, n_own, d + 1, res, nch, rec, recw, L, d,
                                                              1.0f / lat->slice_denom, d_grad_x, d_grad_src, nt, g_xcd_remap);
    tmark(lat, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

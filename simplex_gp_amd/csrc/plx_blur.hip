// plx_blur.hip -- the blur kernels of the per-MVM path: d+1 gather-accumulate passes over the precomputed neighbour table.  Reference: h:513-572.
// Overview of the per-MVM path, value-row layout and shared helpers: plx_kernels.h.

#include "plx_kernels.h"

namespace plx {

// ----------------------------------------------------------------------------
// blur: one Jacobi pass along one lattice axis,
//   out[i] = sum_{nid=-r..r} c[nid+r] * old[nbr(i, nid)]      (h:539-549)
// accumulated from zero in tap order like the reference.

// vd == 1: VPT consecutive vertices per thread, 4*VPT-byte loads from every plane
template <int ORDER, int VPT>
__global__ __launch_bounds__(kBlock) void blur_axis_v1_kernel(const float *__restrict__ old,
                                                              float *__restrict__ out,
                                                              const int *__restrict__ nbr, int m,
                                                              int64_t mstride, TapArgs taps, int ablate, int ntiles,
                                                              int remap)
{
    ablate = PLX_DIAG_VALUE(ablate);                   // diagnostics are compiled into libplx_diag.so only
    using ivec = typename std::conditional<VPT == 4, int4, int2>::type;
    using fvec = typename std::conditional<VPT == 4, float4, float2>::type;
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int i0 = (tile * kBlock + threadIdx.x) * VPT;
    if (i0 >= m) return;
    if (i0 + VPT <= m) {
        int nb[2 * ORDER][VPT];
#pragma unroll
        for (int s = 0; s < 2 * ORDER; ++s) {
            ivec v;
            if (ablate & 2) { int *q = reinterpret_cast<int *>(&v); for (int j = 0; j < VPT; ++j) q[j] = i0 + j; }
            else v = *reinterpret_cast<const ivec *>(nbr + s * mstride + i0);
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
            for (int j = 0; j < VPT; ++j) g[s][j] = nb[s][j] >= 0 ? ((ablate & 1) ? (float)nb[s][j] : old[nb[s][j]]) : 0.f;
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

// vd == 1 on a sparse lattice: the same pass over the COMPACTED neighbour table.  A thread
// owns a quad of 4 vertices; its existing neighbour ids start at
//   cbase[wave] + (sum of popcount(mask) over the lower lanes of the wave)
// so a wave reads one contiguous run of ids instead of 2r full planes that are mostly -1.
// inclusive prefix sum over the 64 lanes of a wave with DPP adds only (row_shr inside rows of 16,
// row_bcast:15 / row_bcast:31 across rows): 6 vector instructions, no LDS crossbar round trips
__device__ __forceinline__ int wave_inclusive_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1, shifted-in lanes read 0
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

template <int ORDER>
__global__ __launch_bounds__(kBlock) void blur_axis_compact_kernel(const float *__restrict__ old,
                                                                   float *__restrict__ out,
                                                                   const uint32_t *__restrict__ cmask,
                                                                   const uint32_t *__restrict__ cbase,
                                                                   const int *__restrict__ cids, int m,
                                                                   uint32_t nquads, TapArgs taps)
{
    constexpr int T2 = 2 * ORDER;
    // 32-bit indices throughout (m < 2^31, at most (d+1) * 2r * m < 2^32 ids is checked by the caller):
    // addresses are scalar base + 32-bit lane offset, no 64-bit vector arithmetic
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    const bool live = q < nquads;
    const uint32_t mask = live ? cmask[q] : 0u;
    const int cnt = __popc(mask);
    const int incl = wave_inclusive_sum(cnt);
    if (!live) return;
    uint32_t pos = cbase[q >> 6] + (uint32_t)(incl - cnt);
    const uint32_t i0 = q * 4u;
    const bool full = i0 + 4u <= (uint32_t)m;
    // Only the existing neighbours are loaded: on the sparse lattices this kernel is for, 80-90 % of
    // the slots are empty, and issuing their loads anyway (branch-free) was measured 30 % slower.
    uint32_t nb[4][T2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < T2; ++s) {
            const bool has = (mask >> (j * T2 + s)) & 1u;
            nb[j][s] = has ? (uint32_t)cids[pos] : 0xFFFFFFFFu;
            pos += has ? 1u : 0u;
        }
    float g[4][T2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < T2; ++s) g[j][s] = nb[j][s] != 0xFFFFFFFFu ? old[nb[j][s]] : 0.f;
    float c[4];
    if (full) {
        const float4 cv = *reinterpret_cast<const float4 *>(old + i0);
        c[0] = cv.x; c[1] = cv.y; c[2] = cv.z; c[3] = cv.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = (i0 + j < (uint32_t)m) ? old[i0 + j] : 0.f;
    }
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < ORDER; ++s) acc += taps.c[s] * g[j][s];
        acc += taps.c[ORDER] * c[j];
#pragma unroll
        for (int s = 0; s < ORDER; ++s) acc += taps.c[ORDER + 1 + s] * g[j][ORDER + s];
        r[j] = acc;
    }
    if (full) {
        *reinterpret_cast<float4 *>(out + i0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < (uint32_t)m) out[i0 + j] = r[j];
    }
}

// ----------------------------------------------------------------------------
// vd == 1, order 1, cache-resident lattices: TWO axes per launch.
//
// On a coarse lattice (N = 1e6, d = 8, lengthscale 1: m = 4e5, 1.6 MB of values, 3.2 MB of neighbour ids per axis) a
// blur pass moves 6.4 MB and takes 5.8 us however it is tiled: it is a dependent-launch boundary plus two memory
// latencies (ids, then gathers) from a cold L2 -- the XCD L2s are written back and invalidated at every kernel
// boundary.  d+1 = 9 of them were 52 us of a 110 us MVM.  Two consecutive passes (axis i, then axis j) are
//     out[v] = sum_b c_b * tmp[nbr_j(v, b)],   tmp[u] = sum_a c_a * old[nbr_i(u, a)]            (h:539-549 twice)
// i.e. a 3x3 stencil over the composite neighbours nbr_i(nbr_j(v, b), a), which depend on the lattice only: they are
// tabulated once per build (pair_nbr, 8 ids per vertex and pair; an absent intermediate vertex nbr_j(v, b) makes its
// three composites absent, exactly as the two-pass form never reads tmp there).  The inner sums are formed from zero
// in tap order like a single pass, then the outer sum likewise: the same fp32 operations in the same order as two
// launches, half the launches.  Costs 8 instead of 2 x 2 id loads and 8 instead of 2 x 2 gathers per vertex, which
// only pays while the pass is latency-bound: lattices of up to kPairMaxVertices vertices (measured at d = 8 under the Morton
// vertex numbering, us per MVM with / without pairs: 94 / 99 at m = 4.0e5, 162 / 168 at m = 5.2e5 (N = 2e6), 286 / 286 at
// m = 6.6e5 (N = 4e6), 124 / 116 at m = 7.9e5, 188 / 172 at m = 1.7e6).
constexpr int kPairMaxVertices = 600000;

// slot = 3 * (b + 1) + (a + 1) without the centre (b = a = 0): 0..3 -> (b,a) = (-1,-1) (-1,0) (-1,+1) (0,-1); 4..7 -> (0,+1) (+1,-1) (+1,0) (+1,+1)
__global__ __launch_bounds__(kBlock) void pair_nbr_kernel(const int *__restrict__ nbr, int m, int64_t mstride,
                                                          int *__restrict__ out_all)
{
    // one launch for all axis pairs (blockIdx.y = pair p: axes 2p, 2p + 1): four launches of a 6 us kernel were 24 us of a build
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= m) return;
    const int axis_i = 2 * blockIdx.y, axis_j = axis_i + 1;
    int *out = out_all + (size_t)blockIdx.y * 8 * mstride;
    const int *ni = nbr + (size_t)axis_i * 2 * mstride, *nj = nbr + (size_t)axis_j * 2 * mstride;
    int slot = 0;
#pragma unroll
    for (int b = -1; b <= 1; ++b) {
        const int u = b == 0 ? v : nj[(size_t)(b < 0 ? 0 : 1) * mstride + v];
#pragma unroll
        for (int a = -1; a <= 1; ++a) {
            if (a == 0 && b == 0) continue;
            int id = -1;
            if (u >= 0) id = a == 0 ? u : ni[(size_t)(a < 0 ? 0 : 1) * mstride + u];
            out[(size_t)slot * mstride + v] = id;
            ++slot;
        }
    }
}

template <int VPT>
__global__ __launch_bounds__(kBlock) void blur_pair_v1_kernel(const float *__restrict__ old, float *__restrict__ out,
                                                              const int *__restrict__ pn, int m, int64_t mstride,
                                                              TapArgs taps, int ntiles, int remap)
{
    using ivec = typename std::conditional<VPT == 4, int4, int2>::type;
    using fvec = typename std::conditional<VPT == 4, float4, float2>::type;
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int i0 = (tile * kBlock + threadIdx.x) * VPT;
    if (i0 >= m) return;
    int id[8][VPT];
    float c[VPT];
    if (i0 + VPT <= m) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const ivec v = *reinterpret_cast<const ivec *>(pn + s * mstride + i0);
            const int *pv = reinterpret_cast<const int *>(&v);
#pragma unroll
            for (int j = 0; j < VPT; ++j) id[s][j] = pv[j];
        }
        const fvec cv = *reinterpret_cast<const fvec *>(old + i0);
        const float *pc = reinterpret_cast<const float *>(&cv);
#pragma unroll
        for (int j = 0; j < VPT; ++j) c[j] = pc[j];
    } else {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const bool in = i0 + j < m;
#pragma unroll
            for (int s = 0; s < 8; ++s) id[s][j] = in ? pn[s * mstride + i0 + j] : -1;
            c[j] = in ? old[i0 + j] : 0.f;
        }
    }
    float g[8][VPT];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < VPT; ++j) g[s][j] = id[s][j] >= 0 ? old[id[s][j]] : 0.f;
    float r[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        // tmp at nbr_j(v, -1), v, nbr_j(v, +1): each from zero in tap order (a = -1, 0, +1), as one pass computes it
        float tm = 0.f, t0 = 0.f, tp = 0.f;
        tm += taps.c[0] * g[0][j]; tm += taps.c[1] * g[1][j]; tm += taps.c[2] * g[2][j];
        t0 += taps.c[0] * g[3][j]; t0 += taps.c[1] * c[j];    t0 += taps.c[2] * g[4][j];
        tp += taps.c[0] * g[5][j]; tp += taps.c[1] * g[6][j]; tp += taps.c[2] * g[7][j];
        // an absent intermediate vertex contributes 0 (its three composites are absent, so its tmp is an exact 0)
        float acc = 0.f;
        acc += taps.c[0] * tm; acc += taps.c[1] * t0; acc += taps.c[2] * tp;
        r[j] = acc;
    }
    if (i0 + VPT <= m) {
        fvec res;
        float *pr = reinterpret_cast<float *>(&res);
#pragma unroll
        for (int j = 0; j < VPT; ++j) pr[j] = r[j];
        *reinterpret_cast<fvec *>(out + i0) = res;
    } else {
#pragma unroll
        for (int j = 0; j < VPT; ++j)
            if (i0 + j < m) out[i0 + j] = r[j];
    }
}

// The same two axes per launch for rows of 2..4 chunks (vd 2..16: every CG iteration).  There the id loads are shared by
// the row: a pair launch moves 32 B of ids + one row read + one row written per vertex where two single passes move
// 2 x (8 B + row + row), 128 against 208 bytes at vd = 12, and the 8 gathered rows come from the L2 / MALL like the
// 2 x 2 of the single passes (they are near in id since the vertices are numbered along the Morton curve of the axis
// coordinates).  Same operations in the same order as two blur_axis_narrow_kernel launches.  (Two items per thread,
// kBlock apart, all loads first: 77 us against 59 us per launch at m = 1.73e6, vd = 12 -- occupancy, not latency.)

template <int ROWLEN>
__global__ __launch_bounds__(kBlock) void blur_pair_narrow_kernel(const float4 *__restrict__ old, float4 *__restrict__ out,
                                                                  const int *__restrict__ pn, uint32_t total,
                                                                  uint32_t mstride, TapArgs taps, int ntiles, int remap,
                                                                  int ablate)
{
    ablate = PLX_DIAG_VALUE(ablate);                   // diagnostics are compiled into libplx_diag.so only
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const uint32_t item = (uint32_t)tile * kBlock + threadIdx.x;
    if (item >= total) return;
    const uint32_t i = item / ROWLEN, ch = item - i * ROWLEN;
    int id[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) id[s] = pn[(uint32_t)s * mstride + i];
    const float4 c = old[item];
    float4 g[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) g[s] = old[(id[s] >= 0 && !(ablate & 1)) ? (uint32_t)id[s] * ROWLEN + ch : item];
    // tmp at nbr_j(v, -1), v, nbr_j(v, +1), each from zero in tap order as blur_axis_narrow_kernel forms it; an absent
    // intermediate vertex (slots 1 / 6) is skipped by the outer sum exactly as the second single pass skips it
    float4 tm = f4_zero(), t0 = f4_zero(), tp = f4_zero();
    tm = f4_sel(id[0] >= 0, f4_add(tm, f4_scale(taps.c[0], g[0])), tm);
    tm = f4_add(tm, f4_scale(taps.c[1], g[1]));
    tm = f4_sel(id[2] >= 0, f4_add(tm, f4_scale(taps.c[2], g[2])), tm);
    t0 = f4_sel(id[3] >= 0, f4_add(t0, f4_scale(taps.c[0], g[3])), t0);
    t0 = f4_add(t0, f4_scale(taps.c[1], c));
    t0 = f4_sel(id[4] >= 0, f4_add(t0, f4_scale(taps.c[2], g[4])), t0);
    tp = f4_sel(id[5] >= 0, f4_add(tp, f4_scale(taps.c[0], g[5])), tp);
    tp = f4_add(tp, f4_scale(taps.c[1], g[6]));
    tp = f4_sel(id[7] >= 0, f4_add(tp, f4_scale(taps.c[2], g[7])), tp);
    float4 acc = f4_zero();
    acc = f4_sel(id[1] >= 0, f4_add(acc, f4_scale(taps.c[0], tm)), acc);
    acc = f4_add(acc, f4_scale(taps.c[1], t0));
    acc = f4_sel(id[6] >= 0, f4_add(acc, f4_scale(taps.c[2], tp)), acc);
    out[item] = acc;
}

static void launch_blur_pair_narrow(const float4 *cur, float4 *nxt, const int *pn, int m, int64_t mstride, int rowlen,
                                    const TapArgs &taps, hipStream_t stream, int remap)
{
    const uint32_t total = (uint32_t)m * (uint32_t)rowlen;
    const int nt = ceil_div((int64_t)total, kBlock);
    const int grid = tile_grid(nt, remap);
    switch (rowlen) {
    case 2: blur_pair_narrow_kernel<2><<<grid, kBlock, 0, stream>>>(cur, nxt, pn, total, (uint32_t)mstride, taps, nt, remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    case 3: blur_pair_narrow_kernel<3><<<grid, kBlock, 0, stream>>>(cur, nxt, pn, total, (uint32_t)mstride, taps, nt, remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    default: blur_pair_narrow_kernel<4><<<grid, kBlock, 0, stream>>>(cur, nxt, pn, total, (uint32_t)mstride, taps, nt, remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    }
}

// Which axes share a launch: (0,1) (2,3) ...; when d + 1 is odd the last axis runs alone.  (Measured with the Morton
// numbering, where the last axis is the one with far neighbours: pairing it -- (0,1) .. (d-4,d-3), d-2 alone, (d-1,d) --
// made the CG blur 6 % slower, the pair's three far gathers cost more than the single pass's two.  Staging a
// workgroup's 768 rows in LDS and reading in-tile neighbours from there: 111 us against 57 us per pair launch.)
static inline int pair_at_axis(int d1, int axis) { return (axis + 1 < d1 && !(axis & 1)) ? axis / 2 : -1; }

// composite neighbour tables for the axis pairs: built with the lattice when its single-column blur
// uses them, otherwise by the first multi-column blur that does
int ensure_blur_pairs(plx_lattice *L, hipStream_t stream)
{
    if (L->pairs_ready) return PLX_OK;
    const int d1 = L->d + 1, m = (int)L->m;
    const int npairs = d1 / 2;
    PLX_TRY(ensure(L->pair_nbr, (size_t)npairs * 8 * L->mstride * 4 + 64));
    if (npairs > 0)
        pair_nbr_kernel<<<dim3(ceil_div(m, kBlock), npairs), kBlock, 0, stream>>>(L->nbr.as<int>(), m, L->mstride,
                                                                                  L->pair_nbr.as<int>());
    PLX_HIP_TRY(hipGetLastError());
    L->pairs_ready = true;
    return PLX_OK;
}

int build_blur_pairs(plx_lattice *L, hipStream_t stream)
{
    L->use_pairs = false;
    L->pairs_ready = false;
    L->active_ready = false;      // (every build passes here: the active-row lists belong to the previous neighbour table)
    const int d1 = L->d + 1, m = (int)L->m;
    if (g_blur_fuse == 0 || L->order != 1 || d1 < 2 || m == 0) return PLX_OK;
    if (g_blur_fuse == 1 && (m > kPairMaxVertices || L->single_use)) return PLX_OK;
    PLX_TRY(ensure_blur_pairs(L, stream));
    L->use_pairs = true;
    return PLX_OK;
}


// vd == 1 on a lattice so small that both ping-pong copies of the vertex values fit in LDS (m <= kSmallM): every
// pass is launch-bound there (a few us of host + device launch cost for < 1 us of work), so ONE workgroup runs all
// d+1 passes with a barrier between them.  Same tap order as the per-axis kernels: bit-identical results.
constexpr int kSmallM = 16384;

template <int ORDER>
__global__ __launch_bounds__(1024) void blur_small_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          const int *__restrict__ nbr, int m, int64_t mstride,
                                                          int d1, TapArgs taps)
{
    __shared__ float buf[2][kSmallM];
    for (int i = threadIdx.x; i < m; i += 1024) buf[0][i] = in[i];
    __syncthreads();
    int cur = 0;
    for (int axis = 0; axis < d1; ++axis) {
        const int *nb = nbr + (size_t)axis * 2 * ORDER * mstride;
        for (int i = threadIdx.x; i < m; i += 1024) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < ORDER; ++s) {
                const int j = nb[s * mstride + i];
                acc += taps.c[s] * (j >= 0 ? buf[cur][j] : 0.f);
            }
            acc += taps.c[ORDER] * buf[cur][i];
#pragma unroll
            for (int s = 0; s < ORDER; ++s) {
                const int j = nb[(ORDER + s) * mstride + i];
                acc += taps.c[ORDER + 1 + s] * (j >= 0 ? buf[cur][j] : 0.f);
            }
            buf[cur ^ 1][i] = acc;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int i = threadIdx.x; i < m; i += 1024) out[i] = buf[cur][i];
}

// general: one thread per (vertex, value element).  V = float handles any order at
// vd = 1; V = float4 handles vd > 1 with rowlen = vdp/4 chunks per vertex (lanes of
// one vertex read the same neighbour id and adjacent 16-byte chunks).
template <class V, int ORDER>   // ORDER 0 = runtime order
__global__ __launch_bounds__(kBlock) void blur_axis_kernel(const V *__restrict__ old, V *__restrict__ out,
                                                           const int *__restrict__ nbr, int m, int64_t mstride,
                                                           int rowlen, int order_rt, TapArgs taps, int ntiles, int remap,
                                                           int ablate)
{
    ablate = PLX_DIAG_VALUE(ablate);                   // diagnostics are compiled into libplx_diag.so only
    using O = VecOps<V>;
    const int order = ORDER > 0 ? ORDER : order_rt;
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    // 32-bit index arithmetic: a 64-bit division costs more VALU time than the whole rest of the thread
    // (blur_impl checks m * rowlen < 2^31)
    const uint32_t item = (uint32_t)tile * kBlock + threadIdx.x;
    if (item >= (uint32_t)m * (uint32_t)rowlen) return;
    const uint32_t iu = item / (uint32_t)rowlen;
    const int i = (int)iu, ch = (int)(item - iu * (uint32_t)rowlen);
    V acc = O::zero();
#pragma unroll
    for (int s = 0; s < order; ++s) {
        const int nb = nbr[s * mstride + i];
        if (nb >= 0) acc = O::add(acc, O::scale(taps.c[s], old[(ablate & 1) ? (size_t)item : (size_t)nb * rowlen + ch]));
    }
    acc = O::add(acc, O::scale(taps.c[order], old[item]));
#pragma unroll
    for (int s = 0; s < order; ++s) {
        const int nb = nbr[(order + s) * mstride + i];
        if (nb >= 0) acc = O::add(acc, O::scale(taps.c[order + 1 + s], old[(ablate & 1) ? (size_t)item : (size_t)nb * rowlen + ch]));
    }
    out[item] = acc;
}

// vd 2..16 (rows of 1..4 chunks, every CG iteration): the general kernel with the row length a
// compile-time constant (the division by 3 alone made a 3-chunk pass slower per byte than a 2-chunk one),
// 32-bit element indices (scalar base + 32-bit lane offset addressing) and no branches: an absent
// neighbour re-reads the centre chunk and is dropped by a select.
template <int ORDER, int ROWLEN>
__global__ __launch_bounds__(kBlock) void blur_axis_narrow_kernel(const float4 *__restrict__ old, float4 *__restrict__ out,
                                                                  const int *__restrict__ nbr, uint32_t total,
                                                                  uint32_t mstride, TapArgs taps, int ntiles, int remap)
{
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const uint32_t item = (uint32_t)tile * kBlock + threadIdx.x;
    if (item >= total) return;
    const uint32_t i = item / ROWLEN, ch = item - i * ROWLEN;
    int nb[2 * ORDER];
#pragma unroll
    for (int s = 0; s < 2 * ORDER; ++s) nb[s] = nbr[(uint32_t)s * mstride + i];
    const float4 c = old[item];
    float4 g[2 * ORDER];
#pragma unroll
    for (int s = 0; s < 2 * ORDER; ++s) g[s] = old[nb[s] >= 0 ? (uint32_t)nb[s] * ROWLEN + ch : item];
    float4 acc = f4_zero();
#pragma unroll
    for (int s = 0; s < ORDER; ++s) acc = f4_sel(nb[s] >= 0, f4_add(acc, f4_scale(taps.c[s], g[s])), acc);
    acc = f4_add(acc, f4_scale(taps.c[ORDER], c));
#pragma unroll
    for (int s = 0; s < ORDER; ++s)
        acc = f4_sel(nb[ORDER + s] >= 0, f4_add(acc, f4_scale(taps.c[ORDER + 1 + s], g[ORDER + s])), acc);
    out[item] = acc;
}

template <int ORDER>
static void launch_blur_narrow(const float4 *cur, float4 *nxt, const int *nb, int m, int64_t mstride, int rowlen,
                               const TapArgs &taps, hipStream_t stream, int remap)
{
    const uint32_t total = (uint32_t)m * (uint32_t)rowlen;
    const int nt = ceil_div((int64_t)total, kBlock);
    const int grid = tile_grid(nt, remap);
    switch (rowlen) {
    case 1: blur_axis_narrow_kernel<ORDER, 1><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, total, (uint32_t)mstride, taps, nt, remap); break;
    case 2: blur_axis_narrow_kernel<ORDER, 2><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, total, (uint32_t)mstride, taps, nt, remap); break;
    case 3: blur_axis_narrow_kernel<ORDER, 3><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, total, (uint32_t)mstride, taps, nt, remap); break;
    default: blur_axis_narrow_kernel<ORDER, 4><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, total, (uint32_t)mstride, taps, nt, remap); break;
    }
}

// vd > 1, the shipped kernel: IPT (vertex, chunk) items per thread, 256 apart, so that a thread has
// all its id loads and centre chunks in flight, then all its gathers, then its stores (one item per
// thread streamed at 4.0-4.6 TB/s with the gathers switched off, a copy kernel reaches 6.3).  The
// vertex / chunk of the next item follow from the previous one without a division.
template <int ORDER, int IPT>
__global__ __launch_bounds__(kBlock) void blur_axis_multi_kernel(const float4 *__restrict__ old, float4 *__restrict__ out,
                                                                 const int *__restrict__ nbr, int m, int64_t mstride,
                                                                 int rowlen, TapArgs taps, int ntiles, int remap)
{
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const uint32_t total = (uint32_t)m * (uint32_t)rowlen;
    const uint32_t item0 = (uint32_t)tile * (kBlock * IPT) + threadIdx.x;
    const uint32_t q256 = (uint32_t)kBlock / (uint32_t)rowlen, r256 = (uint32_t)kBlock - q256 * (uint32_t)rowlen;   // uniform
    uint32_t i = item0 / (uint32_t)rowlen, ch = item0 - i * (uint32_t)rowlen;
    uint32_t item[IPT], src[IPT][2 * ORDER];
    bool live[IPT], have[IPT][2 * ORDER];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        item[k] = item0 + (uint32_t)k * kBlock;
        live[k] = item[k] < total;
        const uint32_t ii = live[k] ? i : 0u;
#pragma unroll
        for (int s = 0; s < 2 * ORDER; ++s) {
            const int nb = nbr[s * mstride + ii];
            have[k][s] = live[k] && nb >= 0;
            src[k][s] = have[k][s] ? (uint32_t)nb * (uint32_t)rowlen + ch : (live[k] ? item[k] : 0u);   // absent: re-read the centre
        }
        ch += r256;
        i += q256;
        if (ch >= (uint32_t)rowlen) { ch -= (uint32_t)rowlen; i += 1; }
    }
    float4 c[IPT], g[IPT][2 * ORDER];
#pragma unroll
    for (int k = 0; k < IPT; ++k) c[k] = old[live[k] ? item[k] : 0u];
    // a gather no lane of the wave needs is not issued at all (wave-uniform branch): on sparse lattices -- the elevators
    // stand-in has every corner on a vertex of its own -- almost every neighbour is absent, and re-reading the centre in its
    // place (what keeps the dense case branch-free) made the pass six L1 reads per element instead of one
#pragma unroll
    for (int k = 0; k < IPT; ++k)
#pragma unroll
        for (int s = 0; s < 2 * ORDER; ++s) {
            g[k][s] = f4_zero();
            if (__ballot(have[k][s]) != 0ull) g[k][s] = old[src[k][s]];
        }
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        float4 acc = f4_zero();
#pragma unroll
        for (int s = 0; s < ORDER; ++s)
            if (have[k][s]) acc = f4_add(acc, f4_scale(taps.c[s], g[k][s]));
        acc = f4_add(acc, f4_scale(taps.c[ORDER], c[k]));
#pragma unroll
        for (int s = 0; s < ORDER; ++s)
            if (have[k][ORDER + s]) acc = f4_add(acc, f4_scale(taps.c[ORDER + 1 + s], g[k][ORDER + s]));
        if (live[k]) out[item[k]] = acc;
    }
}

// ----------------------------------------------------------------------------
// Wide rows on SPARSE lattices (round 6).  The taps of every kernel profile are normalised to a centre tap of exactly 1
// (py:19-28: vals / vals[r]), so a vertex without a neighbour on the pass' axis keeps its row: new = 0 + 1 * old.  On the
// lattices of the high-dimensional data sets every point sits in a simplex of its own (m = n (d + 1)); along one axis
// exactly two of a simplex's d + 1 vertices are neighbours of each other, i.e. 2 / (d + 1) of the rows change (10.5 % at
// d = 18; 24 % at N = 1e6, d = 8, l = 0.25) while blur_axis_multi_kernel streams all of them in and out: the 418-column
// backward filter of the config-5 stand-in spent 19 x 243 us = 4.6 of its 6.0 ms there.  Here a pass works IN PLACE on the
// rows that change: (1) one wave per listed vertex forms its new row -- the dense kernel's operations in the dense kernel's
// order: lower taps, centre, upper taps, from zero -- into the scratch buffer at the vertex' place in the list (every row
// is read before any is written: Jacobi), (2) the rows go back to their vertices.  Rows of vertices that are not listed
// are bit for bit what the dense pass writes (up to the sign of a zero: 0 + 1 * (-0) = +0).  The lists are built by the
// first wide blur that qualifies (ensure_active_lists: ordered compaction, one read-back of the d + 1 counts).
constexpr double kActiveShare = 0.40;      // use the lists while at most this share of the vertices changes per axis (mean over the axes; measured break-even: 0.43-0.45, tools/ab_blur_active_r6.py)

__device__ __forceinline__ bool has_neighbour(const int *__restrict__ nb, int64_t mstride, int taps2, int v)
{
    bool any = false;
    for (int s = 0; s < taps2; ++s) any = any || nb[(size_t)s * mstride + v] >= 0;
    return any;
}

// counts[axis][block] = vertices of the block with a neighbour on the axis
__global__ __launch_bounds__(kBlock) void active_count_kernel(const int *__restrict__ nbr, int m, int64_t mstride, int taps2,
                                                              int nblocks, int *__restrict__ counts)
{
    __shared__ int wsum[kBlock / 64];
    const int v = blockIdx.x * kBlock + threadIdx.x, axis = blockIdx.y;
    const int *nb = nbr + (size_t)axis * taps2 * mstride;
    const bool on = v < m && has_neighbour(nb, mstride, taps2, v);
    const int c = __popcll(__ballot(on));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += wsum[w];
        counts[(size_t)axis * nblocks + blockIdx.x] = t;
    }
}

// one workgroup per axis: counts -> exclusive offsets within the axis; totals[axis] = its sum
__global__ __launch_bounds__(1024) void active_scan_kernel(int *__restrict__ counts, int nblocks, int *__restrict__ totals)
{
    __shared__ int wsum[16];
    int *row = counts + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {           // same trip count in every thread
        const int b = b0 + threadIdx.x;
        const int val = b < nblocks ? row[b] : 0;
        const int incl = wave_inclusive_sum(val);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int sw = wsum[w]; if (w < wave) before += sw; tot += sw; }
        __syncthreads();
        if (b < nblocks) row[b] = carry + before + incl - val;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// list[off[axis] + counts[axis][block] + rank in the block] = v   (ascending within an axis)
struct ActiveOffsets { long long off[PLX_MAX_DIM + 2]; };
__global__ __launch_bounds__(kBlock) void active_fill_kernel(const int *__restrict__ nbr, int m, int64_t mstride, int taps2,
                                                             int nblocks, const int *__restrict__ counts, ActiveOffsets ao,
                                                             int *__restrict__ list)
{
    __shared__ int wsum[kBlock / 64];
    const int v = blockIdx.x * kBlock + threadIdx.x, axis = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int *nb = nbr + (size_t)axis * taps2 * mstride;
    const bool on = v < m && has_neighbour(nb, mstride, taps2, v);
    const unsigned long long mask = __ballot(on);
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (on) {
        const int rank = before + __popcll(mask & ((1ull << lane) - 1ull));
        list[ao.off[axis] + counts[(size_t)axis * nblocks + blockIdx.x] + rank] = v;
    }
}

int ensure_active_lists(plx_lattice *L, hipStream_t stream)
{
    if (L->active_ready) return PLX_OK;
    PLX_TRY(refuse_under_capture(stream, "the active-row lists of this lattice's wide blur"));
    const int d1 = L->d + 1, m = (int)L->m, taps2 = 2 * L->order;
    const int nblocks = ceil_div(m, kBlock);
    PLX_TRY(ensure(L->active_cnt, (size_t)d1 * nblocks * 4 + 64));
    int *totals = L->counters.as<int>() + 2;        // (the build's counters are free between builds: d + 1 <= 33 ints)
    active_count_kernel<<<dim3(nblocks, d1), kBlock, 0, stream>>>(L->nbr.as<int>(), m, L->mstride, taps2, nblocks, L->active_cnt.as<int>());
    active_scan_kernel<<<d1, 1024, 0, stream>>>(L->active_cnt.as<int>(), nblocks, totals);
    int h_tot[PLX_MAX_DIM + 1];
    PLX_TRY(read_back(L, totals, d1, h_tot, stream));
    ActiveOffsets ao;
    int64_t total = 0, longest = 0;
    for (int a = 0; a < d1; ++a) {
        ao.off[a] = total; L->active_off[a] = total;
        total += h_tot[a];
        longest = std::max<int64_t>(longest, h_tot[a]);
    }
    ao.off[d1] = total; L->active_off[d1] = total;
    L->active_max = longest;
    PLX_TRY(ensure(L->active_list, (size_t)total * 4 + 64));
    if (total > 0)
        active_fill_kernel<<<dim3(nblocks, d1), kBlock, 0, stream>>>(L->nbr.as<int>(), m, L->mstride, taps2, nblocks,
                                                                     L->active_cnt.as<int>(), ao, L->active_list.as<int>());
    PLX_HIP_TRY(hipGetLastError());
    L->active_ready = true;
    return PLX_OK;
}

// (1) tmp[k] = the new row of vertex list[k]: one wave per listed vertex, lanes over the 16-byte chunks of the row
template <int ORDER, int MAXCH>
__global__ __launch_bounds__(kBlock) void blur_active_rows_kernel(const float4 *__restrict__ old, float4 *__restrict__ tmp,
                                                                  const int *__restrict__ nbr, const int *__restrict__ list,
                                                                  int count, int64_t mstride, int nch, TapArgs taps)
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (k >= count) return;                         // whole waves leave together
    const int v = __builtin_amdgcn_readfirstlane(list[k]);
    int nb[2 * ORDER];
#pragma unroll
    for (int s = 0; s < 2 * ORDER; ++s) nb[s] = __builtin_amdgcn_readfirstlane(nbr[(size_t)s * mstride + v]);
    float4 c[MAXCH], g[2 * ORDER][MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) c[q] = (lane + 64 * q < nch) ? old[(size_t)v * nch + lane + 64 * q] : f4_zero();
#pragma unroll
    for (int s = 0; s < 2 * ORDER; ++s)
#pragma unroll
        for (int q = 0; q < MAXCH; ++q)
            g[s][q] = (nb[s] >= 0 && lane + 64 * q < nch) ? old[(size_t)nb[s] * nch + lane + 64 * q] : f4_zero();   // (wave-uniform: an absent neighbour issues nothing)
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
        float4 acc = f4_zero();
#pragma unroll
        for (int s = 0; s < ORDER; ++s)
            if (nb[s] >= 0) acc = f4_add(acc, f4_scale(taps.c[s], g[s][q]));
        acc = f4_add(acc, f4_scale(taps.c[ORDER], c[q]));
#pragma unroll
        for (int s = 0; s < ORDER; ++s)
            if (nb[ORDER + s] >= 0) acc = f4_add(acc, f4_scale(taps.c[ORDER + 1 + s], g[ORDER + s][q]));
        if (lane + 64 * q < nch) tmp[(size_t)k * nch + lane + 64 * q] = acc;
    }
}

// (2) values[list[k]] = tmp[k]
__global__ __launch_bounds__(kBlock) void blur_active_store_kernel(const float4 *__restrict__ tmp, float4 *__restrict__ values,
                                                                   const int *__restrict__ list, int count, int nch)
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (k >= count) return;
    const int v = __builtin_amdgcn_readfirstlane(list[k]);
    for (int ch = lane; ch < nch; ch += 64) values[(size_t)v * nch + ch] = tmp[(size_t)k * nch + ch];
}

template <int ORDER>
static void launch_blur_active(const float4 *cur, float4 *tmp, const int *nb, const int *list, int count, int64_t mstride, int nch,
                               const TapArgs &taps, hipStream_t stream)
{
    if (count <= 0) return;
    const int grid = ceil_div(count, kBlock / 64);
    if (nch <= 64) blur_active_rows_kernel<ORDER, 1><<<grid, kBlock, 0, stream>>>(cur, tmp, nb, list, count, mstride, nch, taps);
    else blur_active_rows_kernel<ORDER, 2><<<grid, kBlock, 0, stream>>>(cur, tmp, nb, list, count, mstride, nch, taps);
    blur_active_store_kernel<<<grid, kBlock, 0, stream>>>(tmp, const_cast<float4 *>(cur), list, count, nch);
}

template <int ORDER>
static void launch_blur_v1(const float *cur, float *nxt, const int *nb, int m, int64_t mstride, const TapArgs &taps,
                           hipStream_t stream)
{
    const int nt4 = ceil_div(ceil_div(m, 4), kBlock), nt2 = ceil_div(ceil_div(m, 2), kBlock);
    if (g_blur_vpt == 4)
        blur_axis_v1_kernel<ORDER, 4><<<tile_grid(nt4, g_xcd_remap), kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, taps, PLX_DIAG_VALUE(g_blur_ablate), nt4, g_xcd_remap);
    else
        blur_axis_v1_kernel<ORDER, 2><<<tile_grid(nt2, g_xcd_remap), kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, taps, PLX_DIAG_VALUE(g_blur_ablate), nt2, g_xcd_remap);
}

template <class V>
static void launch_blur_general(const V *cur, V *nxt, const int *nb, int m, int64_t mstride, int rowlen, int order,
                                const TapArgs &taps, hipStream_t stream)
{
    const int nt = ceil_div((int64_t)m * rowlen, kBlock);
    const int grid = tile_grid(nt, g_xcd_remap);
    switch (order) {
    case 1: blur_axis_kernel<V, 1><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    case 2: blur_axis_kernel<V, 2><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    case 3: blur_axis_kernel<V, 3><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    default: blur_axis_kernel<V, 0><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, PLX_DIAG_VALUE(g_blur_ablate)); break;
    }
}

int blur_impl(plx_lattice *L, float *d_values, float *d_scratch, int vd, int *result_in_scratch,
              hipStream_t stream)
{
    const int m = (int)L->m, d1 = L->d + 1, order = L->order;
    const int vdp = values_stride(vd);
    const bool v1 = (vd == 1 && order >= 1 && order <= 3 && (g_blur_vpt == 2 || g_blur_vpt == 4));
    // reference_growth with invisible vertices: one axis per launch, each followed by the centre-tap correction
    const bool nocentre = L->replay.active && L->replay.n_invisible > 0;
    if (v1 && m <= kSmallM && g_blur_small && !nocentre) {
        // result goes where the per-axis path would leave it, so callers see no difference
        float *dst = (d1 & 1) ? d_scratch : d_values;
        const int *nb = L->nbr.as<int>();
        switch (order) {
        case 1: blur_small_kernel<1><<<1, 1024, 0, stream>>>(d_values, dst, nb, m, L->mstride, d1, L->taps); break;
        case 2: blur_small_kernel<2><<<1, 1024, 0, stream>>>(d_values, dst, nb, m, L->mstride, d1, L->taps); break;
        default: blur_small_kernel<3><<<1, 1024, 0, stream>>>(d_values, dst, nb, m, L->mstride, d1, L->taps); break;
        }
        L->kn_blur = "blur_small_kernel";
        tmark(L, stream);
        *result_in_scratch = (d1 & 1) ? 1 : 0;
        PLX_HIP_TRY(hipGetLastError());
        return PLX_OK;
    }
    if ((int64_t)m * (vdp > 1 ? vdp / 4 : 1) >= (1ll << 31)) {
        set_error("blur: %d vertices x %d columns exceed the 31-bit element index of the blur kernels", m, vd);
        return PLX_ERR_TOO_LARGE;
    }
    float *cur = d_values, *nxt = d_scratch;
    bool paired = false;
    const bool pair_vec = !nocentre && vd > 1 && order == 1 && g_blur_fuse_vec != 0 && !(L->single_use && !L->pairs_ready) && g_blur_narrow && vdp / 4 >= 2 && vdp / 4 <= 4 &&
                          d1 >= 2 && m > 0 && (int64_t)8 * L->mstride < (1ll << 32);
    if (pair_vec) PLX_TRY(ensure_blur_pairs(L, stream));
    // wide rows on a sparse lattice whose centre tap is 1: only the rows that change, in place (see blur_active_rows_kernel)
    bool active = false;
    if (!nocentre && g_blur_active != 0 && g_blur_multi && order >= 1 && order <= 3 && vd > 1 && vdp / 4 >= (g_blur_multi >= 2 ? 32 : 17) && vdp / 4 <= 128 &&
        L->taps.c[order] == 1.0f && m > 0) {
        // (cheap gate before the lists exist: the lattices in question have nearly as many vertices as corners)
        if (g_blur_active == 2 || (double)m >= 0.75 * (double)L->n * d1) {
            PLX_TRY(ensure_active_lists(L, stream));
            active = g_blur_active == 2 || (double)L->active_off[d1] <= kActiveShare * (double)m * d1;
        }
    }
    for (int axis = 0; axis < d1; ++axis) {
        const int *nb = L->nbr.as<int>() + (size_t)axis * 2 * order * L->mstride;
        if (active) {
            const int *list = L->active_list.as<int>() + L->active_off[axis];
            const int count = (int)(L->active_off[axis + 1] - L->active_off[axis]);
            const float4 *c4 = reinterpret_cast<const float4 *>(cur);
            float4 *t4 = reinterpret_cast<float4 *>(nxt);
            switch (order) {
            case 1: launch_blur_active<1>(c4, t4, nb, list, count, L->mstride, vdp / 4, L->taps, stream); break;
            case 2: launch_blur_active<2>(c4, t4, nb, list, count, L->mstride, vdp / 4, L->taps, stream); break;
            default: launch_blur_active<3>(c4, t4, nb, list, count, L->mstride, vdp / 4, L->taps, stream); break;
            }
            L->kn_blur = "blur_active_rows_kernel+blur_active_store_kernel";
            continue;                               // in place: no ping-pong swap
        }
        const int pair = pair_at_axis(d1, axis);
        if (pair_vec && pair >= 0) {
            const int *pn = L->pair_nbr.as<int>() + (size_t)pair * 8 * L->mstride;
            launch_blur_pair_narrow(reinterpret_cast<const float4 *>(cur), reinterpret_cast<float4 *>(nxt), pn, m, L->mstride,
                                    vdp / 4, L->taps, stream, g_xcd_remap);
            paired = true;
            ++axis;
            float *t = cur; cur = nxt; nxt = t;
            continue;
        }
        if (!nocentre && v1 && order == 1 && L->use_pairs && !L->use_compact && g_blur_fuse != 0 && pair >= 0) {
            // axes (axis, axis + 1) in one launch
            const int *pn = L->pair_nbr.as<int>() + (size_t)pair * 8 * L->mstride;
            if (g_blur_vpt == 4) {
                const int nt = ceil_div(ceil_div(m, 4), kBlock);
                blur_pair_v1_kernel<4><<<tile_grid(nt, g_xcd_remap), kBlock, 0, stream>>>(cur, nxt, pn, m, L->mstride, L->taps, nt, g_xcd_remap);
            } else {
                const int nt = ceil_div(ceil_div(m, 2), kBlock);
                blur_pair_v1_kernel<2><<<tile_grid(nt, g_xcd_remap), kBlock, 0, stream>>>(cur, nxt, pn, m, L->mstride, L->taps, nt, g_xcd_remap);
            }
            paired = true;
            ++axis;
            float *t = cur; cur = nxt; nxt = t;
            continue;
        }
        if (v1 && L->use_compact) {
            const uint32_t *cm = L->cmask.as<uint32_t>() + (size_t)axis * L->nquads;
            const uint32_t *cb = L->cbase.as<uint32_t>() + (size_t)axis * (L->nqwaves + 1);
            const int *ci = L->cids.as<int>() + L->compact_off[axis];
            const int grid = ceil_div(L->nqwaves * 64, kBlock);
            switch (order) {
            case 1: blur_axis_compact_kernel<1><<<grid, kBlock, 0, stream>>>(cur, nxt, cm, cb, ci, m, (uint32_t)L->nquads, L->taps); break;
            case 2: blur_axis_compact_kernel<2><<<grid, kBlock, 0, stream>>>(cur, nxt, cm, cb, ci, m, (uint32_t)L->nquads, L->taps); break;
            default: blur_axis_compact_kernel<3><<<grid, kBlock, 0, stream>>>(cur, nxt, cm, cb, ci, m, (uint32_t)L->nquads, L->taps); break;
            }
            L->kn_blur = "blur_axis_compact_kernel";
        } else if (v1) {
            switch (order) {
            case 1: launch_blur_v1<1>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            case 2: launch_blur_v1<2>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            default: launch_blur_v1<3>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            }
            L->kn_blur = "blur_axis_v1_kernel";
        } else if (vd == 1) {
            launch_blur_general<float>(cur, nxt, nb, m, L->mstride, 1, order, L->taps, stream);
            L->kn_blur = "blur_axis_kernel";
        } else if (order >= 1 && order <= 3 && vdp / 4 <= 4 && g_blur_narrow && (int64_t)d1 * 2 * order * L->mstride < (1ll << 32)) {
            const float4 *c4 = reinterpret_cast<const float4 *>(cur);
            float4 *n4 = reinterpret_cast<float4 *>(nxt);
            // nb is already offset to this axis: plane offsets inside the kernel stay below 2 * order * mstride
            switch (order) {
            case 1: launch_blur_narrow<1>(c4, n4, nb, m, L->mstride, vdp / 4, L->taps, stream, g_xcd_remap); break;
            case 2: launch_blur_narrow<2>(c4, n4, nb, m, L->mstride, vdp / 4, L->taps, stream, g_xcd_remap); break;
            default: launch_blur_narrow<3>(c4, n4, nb, m, L->mstride, vdp / 4, L->taps, stream, g_xcd_remap); break;
            }
            L->kn_blur = "blur_axis_narrow_kernel";
        } else if (order >= 1 && order <= 3 && g_blur_multi && vdp / 4 >= (g_blur_multi >= 2 ? 32 : 17)) {   // narrower rows: no gain (vd 2..16 measured 0-30 % slower); 17..31 chunks: 5-8 % faster than the general kernel (round 6; blur_multi = 2: the 32-chunk gate of rounds 1-5)
            constexpr int IPT = 4;
            const int rowlen = vdp / 4;
            const int nt = ceil_div((int64_t)m * rowlen, kBlock * IPT);
            // wide rows stream far more than they gather: plain tile order is 8 % faster there
            const int remap = 0;
            const int grid = tile_grid(nt, remap);
            const float4 *c4 = reinterpret_cast<const float4 *>(cur);
            float4 *n4 = reinterpret_cast<float4 *>(nxt);
            switch (order) {
            case 1: blur_axis_multi_kernel<1, IPT><<<grid, kBlock, 0, stream>>>(c4, n4, nb, m, L->mstride, rowlen, L->taps, nt, remap); break;
            case 2: blur_axis_multi_kernel<2, IPT><<<grid, kBlock, 0, stream>>>(c4, n4, nb, m, L->mstride, rowlen, L->taps, nt, remap); break;
            default: blur_axis_multi_kernel<3, IPT><<<grid, kBlock, 0, stream>>>(c4, n4, nb, m, L->mstride, rowlen, L->taps, nt, remap); break;
            }
            L->kn_blur = "blur_axis_multi_kernel";
        } else {
            launch_blur_general<float4>(reinterpret_cast<const float4 *>(cur), reinterpret_cast<float4 *>(nxt), nb, m,
                                        L->mstride, vdp / 4, order, L->taps, stream);
            L->kn_blur = "blur_axis_kernel";
        }
        if (nocentre) PLX_TRY(replay_nocentre_fix(L, cur, nxt, vdp, stream));
        float *t = cur; cur = nxt; nxt = t;
    }
    if (paired) {   // pairs, plus one axis on its own when d + 1 is odd
        if (pair_vec) L->kn_blur = (d1 & 1) ? "blur_pair_narrow_kernel+blur_axis_narrow_kernel" : "blur_pair_narrow_kernel";
        else L->kn_blur = (d1 & 1) ? "blur_pair_v1_kernel+blur_axis_v1_kernel" : "blur_pair_v1_kernel";
    }
    tmark(L, stream);
    *result_in_scratch = (cur == d_scratch) ? 1 : 0;
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

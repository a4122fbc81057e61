// Filter of ONE-HOT right-hand sides on the frontier of their non-zeros (plx_filter_onehot): the kernel rows
// K e_p = slice(blur(splat(e_p))) that a pivoted Cholesky asks for (plx_pcg.hip; experiments/train_simplexgp.py:36 trains
// with max_preconditioner_size(100), GPyTorch takes one such row per pivot).
//
// A one-hot column has d + 1 non-zero vertex rows after the splat and at most (2 r + 1) times as many after every blur
// axis (permutohedral.h:513-572: an axis pass reads a vertex and its 2 r neighbours along that axis), so for most of
// the d + 1 passes almost every row a dense pass streams is zero: at N = 1e6, d = 8, order 1, m = 1.73e6 the union of
// twelve columns' non-zero rows is 108 vertices after the splat and ~ 4.5e5 after the LAST axis (tools/onehot_frontier_study.py).
//
// Representation: the vertices that may hold a non-zero ("the frontier"), in the order they were reached: list[k] = vertex,
// pos[vertex] = k or -1.  The frontier only grows, so positions stay valid from pass to pass, and the values live in
// COMPACT arrays [frontier][stride] that ping-pong through the caller's two [m][stride] buffers (the frontier can never
// outgrow m: no overflow case).  Pass `a` is a gather like the dense kernel's -- for every frontier vertex u:
// out[pos u] = sum_s c_s in[pos nbr_a(u, s)] with the dense kernels' operations in the dense kernels' order (a vertex
// outside the previous frontier reads as zero, which is what the dense array holds there), so the results are the dense
// path's bit for bit -- and the same launch appends the not-yet-listed neighbours along axis a + 1 of every frontier
// vertex for the next pass (the neighbour relation is symmetric: every vertex that pass a + 1 can make non-zero is a
// neighbour along a + 1 of a vertex listed now).  The slice then reads its d + 1 vertex rows through pos.
//
// Not used (plx_filter_onehot runs the dense splat_onehot / blur / slice instead) on lattices whose tables were patched
// by the reference_growth replay (their neighbour relation is deliberately NOT symmetric).
#include "plx_internal.h"
#include "plx_kernels.h"

namespace plx {

// counters (int32, device): n0 = cnt[0] vertices listed by the splat, cnt[1 + a] = vertices appended by the expansion
// along axis a.  Frontier before pass a: n0 + sum_{i < a} cnt[1 + i]; frontier of pass a: that + cnt[1 + a].
constexpr int kOhCounters = PLX_MAX_DIM + 8;
constexpr int kSliceMaxD1Onehot = 17;   // d + 1 compiled into the slice up to here (all loads issued before the sums)

__device__ inline int oh_claim(int *pos, int w)
{
    // -1 -> -2 (claimed, position not yet known) -> k; a plain read first: most neighbours are listed already
    if (__hip_atomic_load(&pos[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1) return 0;
    return atomicCAS(&pos[w], -1, -2) == -1;
}

// splat of the one-hot columns + the expansion along axis 0: nb (d + 1) <= 16 * 33 numbers, one workgroup
__global__ __launch_bounds__(1024) void onehot_seed_kernel(const int *__restrict__ evid, const float *__restrict__ ew,
                                                           const int *__restrict__ cand, int nb, int d1, int n, int stride,
                                                           int *__restrict__ pos, int *__restrict__ list, int *__restrict__ cnt,
                                                           float *__restrict__ val, const int *__restrict__ nbr0, int taps2,
                                                           int64_t mstride)
{
    __shared__ int s_n0, s_n1;
    const int x = threadIdx.x;
    if (x == 0) { s_n0 = 0; s_n1 = 0; }
    if (x < kOhCounters) cnt[x] = 0;
    __syncthreads();
    int v = -1, b = 0;
    float w = 0.f;
    if (x < nb * d1) {
        b = x / d1;
        const int r = x - b * d1, p = cand[b];
        if (p >= 0 && p < n) { v = evid[(size_t)r * n + p]; w = ew[(size_t)r * n + p]; }
    }
    if (v >= 0 && atomicCAS(&pos[v], -1, -2) == -1) {
        const int k = atomicAdd(&s_n0, 1);
        list[k] = v;
        for (int c = 0; c < stride; ++c) val[(size_t)k * stride + c] = 0.f;
        __hip_atomic_store(&pos[v], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // the corners of ONE point are distinct vertices: (vertex, column) pairs are written once
    if (v >= 0) val[(size_t)__hip_atomic_load(&pos[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * stride + b] = w;
    const int n0 = s_n0;
    for (int e = x; e < n0 * taps2; e += 1024) {
        const int u = list[e / taps2], nbv = nbr0[(size_t)(e % taps2) * mstride + u];
        if (nbv >= 0 && oh_claim(pos, nbv)) {
            const int k = n0 + atomicAdd(&s_n1, 1);
            list[k] = nbv;
            __hip_atomic_store(&pos[nbv], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (x == 0) { cnt[0] = n0; cnt[1] = s_n1; }
}

// pass `axis` on the frontier + the expansion along the next axis (nb_next; nullptr after the last pass)
template <class V>
__global__ __launch_bounds__(kBlock) void onehot_axis_kernel(const V *__restrict__ in, V *__restrict__ out,
                                                             const int *__restrict__ nb_a, const int *__restrict__ nb_next,
                                                             int order, int64_t mstride, int rowlen, TapArgs taps,
                                                             int *__restrict__ pos, int *__restrict__ list,
                                                             int *__restrict__ cnt, int axis)
{
    using O = VecOps<V>;
    int cprev = cnt[0];
    for (int i = 0; i < axis; ++i) cprev += cnt[1 + i];
    const int cnew = cprev + cnt[1 + axis];
    int *ext = cnt + 2 + axis;
    const int64_t total = (int64_t)cnew * rowlen;
    for (int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x; item < total; item += (int64_t)gridDim.x * kBlock) {
        const int idx = (int)(item / rowlen), ch = (int)(item - (int64_t)idx * rowlen);
        const int u = list[idx];
        V acc = O::zero();
        for (int s = 0; s < order; ++s) {
            const int w = nb_a[(size_t)s * mstride + u];
            if (w >= 0) {
                const int pw = pos[w];        // (a position written by THIS launch is >= cnew, its transient -2 is negative: both read as "not in the previous frontier")
                if ((unsigned)pw < (unsigned)cprev) acc = O::add(acc, O::scale(taps.c[s], in[(size_t)pw * rowlen + ch]));
            }
        }
        if (idx < cprev) acc = O::add(acc, O::scale(taps.c[order], in[item]));
        for (int s = 0; s < order; ++s) {
            const int w = nb_a[(size_t)(order + s) * mstride + u];
            if (w >= 0) {
                const int pw = pos[w];
                if ((unsigned)pw < (unsigned)cprev) acc = O::add(acc, O::scale(taps.c[order + 1 + s], in[(size_t)pw * rowlen + ch]));
            }
        }
        out[item] = acc;
    }
    if (!nb_next) return;
    // expansion along the next axis: every frontier vertex claims its unlisted neighbours (pos -1 -> -2); a workgroup's
    // claims of one round take their list positions with ONE device atomic (a per-claim atomicAdd on the one counter
    // would serialise ~ 2e5 appends of the late passes at the L2)
    __shared__ int s_count, s_base;
    for (int base = blockIdx.x * kBlock; base < cnew; base += gridDim.x * kBlock) {     // (uniform: barriers inside)
        const int idx = base + threadIdx.x;
        if (threadIdx.x == 0) s_count = 0;
        __syncthreads();
        uint32_t mine = 0;
        int u = 0;
        if (idx < cnew) {
            u = list[idx];
            for (int s = 0; s < 2 * order; ++s) {
                const int w = nb_next[(size_t)s * mstride + u];
                if (w >= 0 && oh_claim(pos, w)) mine |= 1u << s;
            }
        }
        int at = mine ? atomicAdd(&s_count, __popc(mine)) : 0;
        __syncthreads();
        if (threadIdx.x == 0 && s_count) s_base = cnew + atomicAdd(ext, s_count);
        __syncthreads();
        at += s_base;
        while (mine) {
            const int s = __ffs(mine) - 1;
            mine &= mine - 1;
            const int w = nb_next[(size_t)s * mstride + u];
            list[at] = w;
            __hip_atomic_store(&pos[w], at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++at;
        }
    }
}

// slice_vec_kernel / slice_v1_kernel's arithmetic, vertex rows read through pos (absent = zero row)
template <class V, int D1>
__global__ __launch_bounds__(kBlock) void onehot_slice_kernel(const int *__restrict__ evid, const float *__restrict__ ew,
                                                              const uint32_t *__restrict__ perm, int n, int d1,
                                                              const V *__restrict__ val, int rowlen, float rden,
                                                              V *__restrict__ out, const int *__restrict__ pos,
                                                              const int *__restrict__ cnt, int *__restrict__ frontier_out)
{
    using O = VecOps<V>;
    int cfin = cnt[0];
    for (int i = 0; i < d1; ++i) cfin += cnt[1 + i];
    if (frontier_out && blockIdx.x == 0 && threadIdx.x == 0) *frontier_out = cfin;
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= (int64_t)n * rowlen) return;
    const int p = (int)(item / rowlen), ch = (int)(item - (int64_t)p * rowlen);
    V acc = O::zero();
    if constexpr (D1 > 0) {
        int pv[D1];
        float w[D1];
        V g[D1];
#pragma unroll
        for (int r = 0; r < D1; ++r) {
            pv[r] = pos[evid[(size_t)r * n + p]];
            w[r] = ew[(size_t)r * n + p];
        }
#pragma unroll
        for (int r = 0; r < D1; ++r) g[r] = (unsigned)pv[r] < (unsigned)cfin ? val[(size_t)pv[r] * rowlen + ch] : O::zero();
#pragma unroll
        for (int r = 0; r < D1; ++r) acc = O::add(acc, O::scale(rden, O::scale(w[r], g[r])));
    } else {
        for (int r = 0; r < d1; ++r) {
            const int pv = pos[evid[(size_t)r * n + p]];
            const float w = ew[(size_t)r * n + p];
            const V g = (unsigned)pv < (unsigned)cfin ? val[(size_t)pv * rowlen + ch] : O::zero();
            acc = O::add(acc, O::scale(rden, O::scale(w, g)));
        }
    }
    const size_t row = perm ? (size_t)perm[p] : (size_t)p;
    out[row * rowlen + ch] = acc;
}

bool onehot_frontier_ok(const plx_lattice *L, int vd, const float *d_out)
{
    if (L->replay.active || L->n_shards != 1 || L->partial_cover || L->order < 1) return false;
    if (vd != 1 && ((vd & 3) || (reinterpret_cast<uintptr_t>(d_out) & 15))) return false;   // whole 16-byte output rows
    return L->m > 0 && L->m < (1ll << 30);
}

template <class V>
static int onehot_frontier_launch(plx_lattice *L, const int *d_cand, int nb, int vd, float *d_values, float *d_scratch,
                                  float *d_out, int *d_frontier, hipStream_t stream)
{
    const int d1 = L->d + 1, order = L->order, n = (int)L->n, stride = values_stride(vd);
    const int rowlen = stride == 1 ? 1 : stride / 4;
    const int64_t m = L->m;
    PLX_TRY(ensure(L->oh_pos, (size_t)m * 4));
    PLX_TRY(ensure(L->oh_list, (size_t)m * 4));
    PLX_TRY(ensure(L->oh_cnt, (size_t)kOhCounters * 4));
    int *pos = L->oh_pos.as<int>(), *list = L->oh_list.as<int>(), *cnt = L->oh_cnt.as<int>();
    const int *nbr = L->nbr.as<int>();
    const size_t plane = (size_t)2 * order * L->mstride;
    PLX_HIP_TRY(hipMemsetAsync(pos, 0xFF, (size_t)m * 4, stream));
    onehot_seed_kernel<<<1, 1024, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), d_cand, nb, d1, n, stride, pos, list, cnt,
                                               d_values, nbr, 2 * order, L->mstride);
    // the frontier's size is only known on the device: a fixed grid walks it (grid-stride); early passes leave most
    // workgroups without work
    const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div(m * rowlen, kBlock)));
    V *cur = reinterpret_cast<V *>(d_values), *nxt = reinterpret_cast<V *>(d_scratch);
    for (int axis = 0; axis < d1; ++axis) {
        onehot_axis_kernel<V><<<grid, kBlock, 0, stream>>>(cur, nxt, nbr + axis * plane, axis + 1 < d1 ? nbr + (axis + 1) * plane : nullptr,
                                                          order, L->mstride, rowlen, L->taps, pos, list, cnt, axis);
        V *t = cur; cur = nxt; nxt = t;
    }
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const float rden = 1.0f / L->slice_denom;
    const int sgrid = ceil_div((int64_t)n * rowlen, kBlock);
    V *o = reinterpret_cast<V *>(d_out);
    switch (d1 <= kSliceMaxD1Onehot ? d1 : 0) {
#define PLX_CASE(D1) case D1: onehot_slice_kernel<V, D1><<<sgrid, kBlock, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), perm, n, d1, cur, rowlen, rden, o, pos, cnt, d_frontier); break;
        PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9) PLX_CASE(10)
        PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17)
#undef PLX_CASE
    default: onehot_slice_kernel<V, 0><<<sgrid, kBlock, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), perm, n, d1, cur, rowlen, rden, o, pos, cnt, d_frontier); break;
    }
    PLX_HIP_TRY(hipGetLastError());
    L->kn_splat = "onehot_seed_kernel";
    L->kn_blur = "onehot_axis_kernel";
    L->kn_slice = "onehot_slice_kernel";
    return PLX_OK;
}

__global__ void onehot_set_frontier_kernel(int *out, int v) { *out = v; }

int filter_onehot_impl(plx_lattice *L, const int *d_cand, int nb, int vd, float *d_values, float *d_scratch, float *d_out,
                       int sparse, int *d_frontier, hipStream_t stream)
{
    if (sparse && onehot_frontier_ok(L, vd, d_out)) {
        if (vd == 1) return onehot_frontier_launch<float>(L, d_cand, nb, vd, d_values, d_scratch, d_out, d_frontier, stream);
        return onehot_frontier_launch<float4>(L, d_cand, nb, vd, d_values, d_scratch, d_out, d_frontier, stream);
    }
    PLX_TRY(splat_onehot_impl(L, d_cand, nb, vd, d_values, stream));
    int in_scratch = 0;
    PLX_TRY(blur_impl(L, d_values, d_scratch, vd, &in_scratch, stream));
    PLX_TRY(slice_impl(L, in_scratch ? d_scratch : d_values, vd, d_out, stream));
    if (d_frontier) onehot_set_frontier_kernel<<<1, 1, 0, stream>>>(d_frontier, (int)std::min<int64_t>(L->m, 0x7FFFFFFF));
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

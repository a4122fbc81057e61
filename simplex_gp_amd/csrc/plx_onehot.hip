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

// counters (int32, device): n0 = cnt[0] vertices listed by the splat, cnt[1 + p] = vertices appended by the expansion
// for PASS p (a pass = one blur axis, or two where the lattice has the composite tables of plx_blur.hip).  Frontier
// before pass p: n0 + sum_{i < p} cnt[1 + i]; frontier of pass p: that + cnt[1 + p].
constexpr int kOhCounters = PLX_MAX_DIM + 8;
constexpr int kSliceMaxD1Onehot = 17;   // d + 1 compiled into the slice up to here (all loads issued before the sums)

// What the NEXT pass can make non-zero from frontier vertex u (the neighbour relation is symmetric, so "the vertices
// whose gather reads u" are u's own neighbours): mode 1 -- the 2 r neighbours along one axis (planes pa); mode 2 -- a pass
// over the axis pair (i, j): out[v] = sum_b c_b tmp[nbr_j(v, b)], tmp[w] = sum_a c_a old[nbr_i(w, a)] reads u at every
// v = nbr_j(nbr_i(u, a), b) whose intermediate vertex nbr_i(u, a) exists (order 1: pa = planes of axis i, pb = of axis j).
struct OhNext { int mode; const int *pa; const int *pb; int order; int64_t mstride; };

constexpr int kOhTargets = 16;      // 2 * PLX_MAX_ORDER single-axis neighbours, or the 8 composites of a pair

// all loads of a level are issued together: the expansion is a chain of dependent memory round trips (list -> neighbour
// -> (neighbour) -> pos -> claim), and walking the targets one after the other made a pass over a 100-vertex frontier
// take 30 us
__device__ inline void oh_targets(const OhNext &nx, int u, int (&t)[kOhTargets])
{
#pragma unroll
    for (int s = 0; s < kOhTargets; ++s) t[s] = -1;
    if (nx.mode == 1) {
#pragma unroll
        for (int s = 0; s < kOhTargets; ++s)
            if (s < 2 * nx.order) t[s] = nx.pa[(size_t)s * nx.mstride + u];
    } else {
        const int um = nx.pa[u], up = nx.pa[nx.mstride + u];
        t[0] = um;
        t[1] = up;
        t[2] = nx.pb[u];
        t[3] = nx.pb[nx.mstride + u];
        const int um0 = um >= 0 ? nx.pb[um] : -1, um1 = um >= 0 ? nx.pb[nx.mstride + um] : -1;
        const int up0 = up >= 0 ? nx.pb[up] : -1, up1 = up >= 0 ? nx.pb[nx.mstride + up] : -1;
        t[4] = um0; t[5] = um1; t[6] = up0; t[7] = up1;
    }
}

// Expansion: every frontier vertex claims its unlisted targets (pos -1 -> -2); a workgroup's claims of one round take
// their list positions with ONE device atomic (a per-claim atomicAdd on the one counter would serialise ~ 2e5 appends
// of the late passes at the L2).  All threads of the workgroup call it (barriers inside).
__device__ inline void oh_expand(const OhNext &nx, int *__restrict__ pos, int *__restrict__ list, int cnew, int *__restrict__ ext,
                                 int *s_count, int *s_base)
{
    for (int base = blockIdx.x * blockDim.x; base < cnew; base += gridDim.x * blockDim.x) {     // (uniform)
        const int idx = base + threadIdx.x;
        if (threadIdx.x == 0) *s_count = 0;
        __syncthreads();
        uint32_t mine = 0;
        int t[kOhTargets];
#pragma unroll
        for (int s = 0; s < kOhTargets; ++s) t[s] = -1;
        if (idx < cnew) {
            oh_targets(nx, list[idx], t);
            int pv[kOhTargets];
#pragma unroll
            for (int s = 0; s < kOhTargets; ++s)
                pv[s] = t[s] >= 0 ? __hip_atomic_load(&pos[t[s]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int s = 0; s < kOhTargets; ++s)                         // -1 -> -2 (claimed, position not yet known) -> k
                if (t[s] >= 0 && pv[s] == -1 && atomicCAS(&pos[t[s]], -1, -2) == -1) mine |= 1u << s;
        }
        int at = mine ? atomicAdd(s_count, __popc(mine)) : 0;
        __syncthreads();
        if (threadIdx.x == 0 && *s_count) *s_base = cnew + atomicAdd(ext, *s_count);
        __syncthreads();
        at += *s_base;
#pragma unroll
        for (int s = 0; s < kOhTargets; ++s) {
            if ((mine >> s) & 1) {
                list[at] = t[s];
                __hip_atomic_store(&pos[t[s]], at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ++at;
            }
        }
    }
}

// splat of the one-hot columns + the expansion for the first pass: nb (d + 1) <= 16 * 33 numbers, one workgroup
__global__ __launch_bounds__(1024) void onehot_seed_kernel(const int *__restrict__ evid, const float *__restrict__ ew,
                                                           const int *__restrict__ cand, int nb, int d1, int n, int stride,
                                                           int *__restrict__ pos, int *__restrict__ list, int *__restrict__ cnt,
                                                           float *__restrict__ val, OhNext nx)
{
    __shared__ int s_n0, s_count, s_base;
    const int x = threadIdx.x;
    if (x == 0) s_n0 = 0;
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
    if (x == 0) cnt[0] = n0;
    oh_expand(nx, pos, list, n0, cnt + 1, &s_count, &s_base);
}

__device__ inline int oh_before(const int *cnt, int pass)
{
    int c = cnt[0];
    for (int i = 0; i < pass; ++i) c += cnt[1 + i];
    return c;
}

// one blur axis on the frontier (pass index `pass`) + the expansion for the next pass
template <class V>
__global__ __launch_bounds__(kBlock) void onehot_axis_kernel(const V *__restrict__ in, V *__restrict__ out,
                                                             const int *__restrict__ nb_a, int order, int64_t mstride,
                                                             int rowlen, TapArgs taps, int *__restrict__ pos,
                                                             int *__restrict__ list, int *__restrict__ cnt, int pass, OhNext nx)
{
    using O = VecOps<V>;
    __shared__ int s_count, s_base;
    const int cprev = oh_before(cnt, pass), cnew = cprev + cnt[1 + pass];
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
    if (nx.mode) oh_expand(nx, pos, list, cnew, cnt + 2 + pass, &s_count, &s_base);
}

// two blur axes in one pass (order 1, the composite neighbour table pn of the pair: plx_blur.hip pair_nbr_kernel):
// blur_pair_narrow_kernel's operations in its order -- which are two single passes' operations in their order
template <class V>
__global__ __launch_bounds__(kBlock) void onehot_pair_kernel(const V *__restrict__ in, V *__restrict__ out,
                                                             const int *__restrict__ pn, int64_t mstride, int rowlen,
                                                             TapArgs taps, int *__restrict__ pos, int *__restrict__ list,
                                                             int *__restrict__ cnt, int pass, OhNext nx)
{
    using O = VecOps<V>;
    __shared__ int s_count, s_base;
    const int cprev = oh_before(cnt, pass), cnew = cprev + cnt[1 + pass];
    const int64_t total = (int64_t)cnew * rowlen;
    for (int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x; item < total; item += (int64_t)gridDim.x * kBlock) {
        const int idx = (int)(item / rowlen), ch = (int)(item - (int64_t)idx * rowlen);
        const int u = list[idx];
        int id[8];
        V g[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) id[s] = pn[(size_t)s * mstride + u];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int pw = id[s] >= 0 ? pos[id[s]] : -1;
            g[s] = (unsigned)pw < (unsigned)cprev ? in[(size_t)pw * rowlen + ch] : O::zero();
        }
        const V c = idx < cprev ? in[item] : O::zero();
        V tm = O::zero(), t0 = O::zero(), tp = O::zero();
        tm = O::sel(id[0] >= 0, O::add(tm, O::scale(taps.c[0], g[0])), tm);
        tm = O::add(tm, O::scale(taps.c[1], g[1]));
        tm = O::sel(id[2] >= 0, O::add(tm, O::scale(taps.c[2], g[2])), tm);
        t0 = O::sel(id[3] >= 0, O::add(t0, O::scale(taps.c[0], g[3])), t0);
        t0 = O::add(t0, O::scale(taps.c[1], c));
        t0 = O::sel(id[4] >= 0, O::add(t0, O::scale(taps.c[2], g[4])), t0);
        tp = O::sel(id[5] >= 0, O::add(tp, O::scale(taps.c[0], g[5])), tp);
        tp = O::add(tp, O::scale(taps.c[1], g[6]));
        tp = O::sel(id[7] >= 0, O::add(tp, O::scale(taps.c[2], g[7])), tp);
        V acc = O::zero();
        acc = O::sel(id[1] >= 0, O::add(acc, O::scale(taps.c[0], tm)), acc);
        acc = O::add(acc, O::scale(taps.c[1], t0));
        acc = O::sel(id[6] >= 0, O::add(acc, O::scale(taps.c[2], tp)), acc);
        out[item] = acc;
    }
    if (nx.mode) oh_expand(nx, pos, list, cnew, cnt + 2 + pass, &s_count, &s_base);
}

// slice_vec_kernel / slice_v1_kernel's arithmetic, vertex rows read through pos (absent = zero row)
template <class V, int D1>
__global__ __launch_bounds__(kBlock) void onehot_slice_kernel(const int *__restrict__ evid, const float *__restrict__ ew,
                                                              const uint32_t *__restrict__ perm, int n, int d1,
                                                              const V *__restrict__ val, int rowlen, float rden,
                                                              V *__restrict__ out, const int *__restrict__ pos,
                                                              const int *__restrict__ cnt, int npass, int *__restrict__ frontier_out)
{
    using O = VecOps<V>;
    const int cfin = oh_before(cnt, npass);
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
    {   // the lists are sized on first use and the axis-pair tables may have to be built: not while the stream is being captured
        const bool pairs_wanted = order == 1 && d1 >= 2 && g_blur_fuse_vec != 0 && !(L->single_use && !L->pairs_ready);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if ((L->oh_pos.cap < (size_t)m * 4 || L->oh_list.cap < (size_t)m * 4 || L->oh_cnt.cap == 0 || (pairs_wanted && !L->pairs_ready)) &&
            hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
            set_error("plx_filter_onehot: its work lists / axis-pair tables have to be built and the stream is being captured: "
                      "run the call once on this lattice before the capture");
            return PLX_ERR_STATE;
        }
    }
    PLX_TRY(ensure(L->oh_pos, (size_t)m * 4));
    PLX_TRY(ensure(L->oh_list, (size_t)m * 4));
    PLX_TRY(ensure(L->oh_cnt, (size_t)kOhCounters * 4));
    int *pos = L->oh_pos.as<int>(), *list = L->oh_list.as<int>(), *cnt = L->oh_cnt.as<int>();
    const int *nbr = L->nbr.as<int>();
    const size_t plane = (size_t)2 * order * L->mstride;
    // passes: two axes at a time where the lattice has (or may build) the composite tables of the dense two-axis blur --
    // the same numbers either way (plx_blur.hip), half the launches of a chain that is latency-bound until its last links
    const bool pairs = order == 1 && d1 >= 2 && g_blur_fuse_vec != 0 && !(L->single_use && !L->pairs_ready) &&
                       (int64_t)8 * L->mstride < (1ll << 32);
    if (pairs) PLX_TRY(ensure_blur_pairs(L, stream));
    struct Pass { int kind, first_axis; };      // kind 2: axes (first_axis, first_axis + 1); kind 1: first_axis alone
    Pass passes[PLX_MAX_DIM + 2];
    int npass = 0;
    for (int axis = 0; axis < d1;) {
        if (pairs && axis + 1 < d1) { passes[npass++] = {2, axis}; axis += 2; }
        else { passes[npass++] = {1, axis}; axis += 1; }
    }
    auto next_of = [&](int k) {
        OhNext nx{0, nullptr, nullptr, order, L->mstride};
        if (k < npass) {
            nx.mode = passes[k].kind;
            nx.pa = nbr + passes[k].first_axis * plane;
            nx.pb = passes[k].kind == 2 ? nbr + (passes[k].first_axis + 1) * plane : nullptr;
        }
        return nx;
    };
    PLX_HIP_TRY(hipMemsetAsync(pos, 0xFF, (size_t)m * 4, stream));
    onehot_seed_kernel<<<1, 1024, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), d_cand, nb, d1, n, stride, pos, list, cnt,
                                               d_values, next_of(0));
    // the frontier's size is only known on the device: a fixed grid walks it (grid-stride); early passes leave most
    // workgroups without work
    const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div(m * rowlen, kBlock)));
    V *cur = reinterpret_cast<V *>(d_values), *nxt = reinterpret_cast<V *>(d_scratch);
    for (int k = 0; k < npass; ++k) {
        if (passes[k].kind == 2)
            onehot_pair_kernel<V><<<grid, kBlock, 0, stream>>>(cur, nxt, L->pair_nbr.as<int>() + (size_t)(passes[k].first_axis / 2) * 8 * L->mstride,
                                                              L->mstride, rowlen, L->taps, pos, list, cnt, k, next_of(k + 1));
        else
            onehot_axis_kernel<V><<<grid, kBlock, 0, stream>>>(cur, nxt, nbr + passes[k].first_axis * plane, order, L->mstride, rowlen,
                                                              L->taps, pos, list, cnt, k, next_of(k + 1));
        V *t = cur; cur = nxt; nxt = t;
    }
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const float rden = 1.0f / L->slice_denom;
    const int sgrid = ceil_div((int64_t)n * rowlen, kBlock);
    V *o = reinterpret_cast<V *>(d_out);
    switch (d1 <= kSliceMaxD1Onehot ? d1 : 0) {
#define PLX_CASE(D1) case D1: onehot_slice_kernel<V, D1><<<sgrid, kBlock, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), perm, n, d1, cur, rowlen, rden, o, pos, cnt, npass, d_frontier); break;
        PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8) PLX_CASE(9) PLX_CASE(10)
        PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) PLX_CASE(17)
#undef PLX_CASE
    default: onehot_slice_kernel<V, 0><<<sgrid, kBlock, 0, stream>>>(L->evid.as<int>(), L->ew.as<float>(), perm, n, d1, cur, rowlen, rden, o, pos, cnt, npass, d_frontier); break;
    }
    PLX_HIP_TRY(hipGetLastError());
    L->kn_splat = "onehot_seed_kernel";
    L->kn_blur = pairs ? "onehot_pair_kernel" : "onehot_axis_kernel";
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

// plx_apply.hip -- the per-MVM kernels: gather-in, splat, blur, slice.
//
// Reference: cpp/permutohedral.h ("h") splat value accumulation h:478-479,
// blur h:513-572, slice h:497-510.  The reference's CUDA path does the splat
// with one float atomicAdd per (point, corner, channel) and re-hashes every
// neighbour in every blur pass; here
//   gather-in  right-hand side rows into lattice point order (and, for vd > 1,
//              into rows padded to whole 16-byte vectors),
//   splat      segmented scan over simplex corners sorted by vertex (no
//              atomics, bitwise reproducible),
//   blur       d+1 gather-accumulate passes over a precomputed neighbour table,
//   slice      per-point gather through SoA (vertex id, weight) planes, result
//              scattered back to the caller's row order.
// All are HBM/cache-bandwidth bound gather stencils; no MFMA.
//
// Value rows: vd = 1 -> one float per vertex; vd > 1 -> vdp = roundup4(vd)
// floats, i.e. nch = vdp/4 float4 "chunks", and every access of the vector
// kernels is one aligned 16-byte load/store per lane.

#include "plx_internal.h"

#include <type_traits>

namespace plx {

static int g_blur_vpt = 4;     // vertices per thread in the vd = 1 blur (2 or 4)
static int g_blur_small = 1;   // all blur passes in one workgroup when m <= 16384 (vd = 1)
static int g_xcd_remap = 1;    // 1: workgroup b works on tile (b % 8) * ceil(nb/8) + b / 8, so that the 8 XCDs (which
                               // receive workgroups round-robin) each own one contiguous slice of the lattice
static int g_splat_direct = 1;  // vd = 1: gather from d_src through caller-row indices instead of a sorted copy:
                                // 0 never, 1 for launch-bound sizes (<= 2e6 corners: saves a launch; at 9e6 corners
                                // the sorted copy wins, 58 vs 60 us), 2 always
static int g_blur_narrow = 1;   // vd 2..16 blur: row length compiled in, branch-free (0: blur_axis_kernel)
static int g_blur_multi = 1;    // vd > 1 blur: 4 items per thread (0: one item per thread, blur_axis_kernel)
static int g_splat_group = 1;   // vd 2..64: lane-group streaming splat (0: segmented-scan kernel)
static int g_splat_wide = 1;    // row-parallel splat for rows of 32..128 chunks (vd 125..512)
static int g_splat_ablate = 0; // diagnostics only: 1 no value gather, 2 no stores, 4 no row-id loads
static int g_blur_ablate = 0;  // diagnostics only: 1 no neighbour gathers, 2 no neighbour-id loads either
extern int g_sort_points;
extern int g_order_zcurve;
extern int g_compact_nbr;
extern int g_insert_dedupe;
extern int g_nbr_symmetric;

Tunable *tunables()
{
    static Tunable t[] = {{"sort_points", &g_sort_points}, {"order_zcurve", &g_order_zcurve}, {"compact_nbr", &g_compact_nbr}, {"insert_dedupe", &g_insert_dedupe}, {"nbr_symmetric", &g_nbr_symmetric}, {"blur_vpt", &g_blur_vpt}, {"xcd_remap", &g_xcd_remap}, {"blur_small", &g_blur_small}, {"blur_multi", &g_blur_multi}, {"blur_narrow", &g_blur_narrow}, {"splat_group", &g_splat_group},
                          {"splat_direct", &g_splat_direct}, {"splat_wide", &g_splat_wide}, {"splat_ablate", &g_splat_ablate}, {"blur_ablate", &g_blur_ablate}, {nullptr, nullptr}};
    return t;
}

// Tile index for workgroup blockIdx.x.  With remap the launch has 8 * ceil(ntiles / 8) workgroups and
// workgroup b takes tile (b % 8) * per + b / 8: workgroups are dealt to the 8 XCDs round-robin
// (MI355X_MICROARCH.md, Workgroup dispatch), so every XCD sweeps one contiguous eighth of the tiles and its
// gathers -- which follow the lattice order -- stay inside one eighth of the gathered array, i.e. inside
// its own 4 MiB L2.  Placement only affects speed, never results.  Returns -1 for the padding workgroups.
__device__ __forceinline__ int tile_index(int ntiles, int remap)
{
    const int b = blockIdx.x;
    if (!remap) return b < ntiles ? b : -1;
    const int per = (ntiles + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    return ((b >> 3) < per && t < ntiles) ? t : -1;
}
static inline int tile_grid(int ntiles, int remap) { return remap ? 8 * ((ntiles + 7) / 8) : ntiles; }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 f4_sel(bool c, float4 a, float4 b) { return c ? a : b; }
__device__ __forceinline__ float4 f4_shfl_up(float4 a, int off)
{
    return make_float4(__shfl_up(a.x, off), __shfl_up(a.y, off), __shfl_up(a.z, off), __shfl_up(a.w, off));
}

// ----------------------------------------------------------------------------
// gather-in: ssrc[i][0..vdp) = src[perm[own_begin + i] - own_begin][0..vd), zero padded

__global__ __launch_bounds__(kBlock) void gather_in_kernel(const float *__restrict__ src,
                                                           const uint32_t *__restrict__ perm, int own_begin,
                                                           int n_own, int vd, int vdp, float *__restrict__ ssrc)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= (int64_t)n_own * vdp) return;
    const int i = (int)(item / vdp), col = (int)(item - (int64_t)i * vdp);
    const int row = perm ? (int)perm[own_begin + i] - own_begin : i;   // perm == nullptr: rows already in lattice order
    ssrc[item] = (col < vd) ? src[(size_t)row * vd + col] : 0.f;
}

__global__ __launch_bounds__(kBlock) void gather_in_v1_kernel(const float *__restrict__ src,
                                                              const uint32_t *__restrict__ perm, int own_begin,
                                                              int n_own, float *__restrict__ ssrc)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_own) ssrc[i] = src[(int)perm[own_begin + i] - own_begin];
}

// ----------------------------------------------------------------------------
// splat: segmented scan.
//
// The corners of the owned points are sorted by vertex (csr_pt / csr_w; the sign
// bit of csr_pt marks the first corner of each vertex row).  A workgroup takes
// kSplatChunk consecutive corners, 4 per thread, forms w * ssrc[point] in
// registers and runs one segmented inclusive scan over the chunk (in-thread,
// then wave shuffles, then four wave totals through LDS).  A thread whose
// corner closes a row stores the row sum; the row ids (csr_vid) are read only
// at row ends.  Rows that cross a chunk edge leave head / tail partial sums for
// splat_fixup_kernel.  The scan tree is fixed, so results are reproducible.
//
// Scan element: (heads seen, sum since the last head);
//   combine(left, right) = (l.cnt + r.cnt, r.cnt ? r.sum : l.sum + r.sum).
//
// V = float (vd = 1, NCH = 1) or float4 with NCH chunks held per corner.

template <class V> struct VecOps;
template <> struct VecOps<float> {
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ float add(float a, float b) { return a + b; }
    static __device__ __forceinline__ float scale(float s, float a) { return s * a; }
    static __device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }
    static __device__ __forceinline__ float shfl_up(float a, int off) { return __shfl_up(a, off); }
};
template <> struct VecOps<float4> {
    static __device__ __forceinline__ float4 zero() { return f4_zero(); }
    static __device__ __forceinline__ float4 add(float4 a, float4 b) { return f4_add(a, b); }
    static __device__ __forceinline__ float4 scale(float s, float4 a) { return f4_scale(s, a); }
    static __device__ __forceinline__ float4 sel(bool c, float4 a, float4 b) { return f4_sel(c, a, b); }
    static __device__ __forceinline__ float4 shfl_up(float4 a, int off) { return f4_shfl_up(a, off); }
};

// rowlen = V-elements per value row (1 for float, nch for float4); tile0 = first
// chunk of this workgroup's column tile (blockIdx.y * NCH)
template <class V, int NCH>
__global__ __launch_bounds__(kSplatBlock) void splat_scan_kernel(const int *__restrict__ csr_pt,
                                                            const float *__restrict__ csr_w,
                                                            const int *__restrict__ csr_vid,
                                                            const V *__restrict__ ssrc, int rowlen, int nnz,
                                                            V *__restrict__ values, V *__restrict__ head_partial,
                                                            V *__restrict__ tail_partial, int ablate, int nchunks,
                                                            int remap)
{
    using O = VecOps<V>;
    constexpr int EPT = kSplatChunk / kSplatBlock;   // corners per thread
    static_assert(EPT % 4 == 0, "vector loads below take 4 corners at a time");
    __shared__ int wave_cnt[kSplatBlock / 64];
    __shared__ V wave_sum[kSplatBlock / 64][NCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tile_index(nchunks, remap);
    if (c < 0) return;
    const int tile0 = blockIdx.y * NCH;
    const int k0 = c * kSplatChunk;
    const int kb = k0 + tid * EPT;

    // first round of loads: 4 corners + the corner after them (csr_pt has slack past nnz)
    int pt[EPT + 1];
    float w[EPT];
    if (kb + EPT <= nnz) {
#pragma unroll
        for (int q4 = 0; q4 < EPT / 4; ++q4) {
            const int4 a = *reinterpret_cast<const int4 *>(csr_pt + kb + 4 * q4);
            const float4 b = *reinterpret_cast<const float4 *>(csr_w + kb + 4 * q4);
            pt[4 * q4] = a.x; pt[4 * q4 + 1] = a.y; pt[4 * q4 + 2] = a.z; pt[4 * q4 + 3] = a.w;
            w[4 * q4] = b.x; w[4 * q4 + 1] = b.y; w[4 * q4 + 2] = b.z; w[4 * q4 + 3] = b.w;
        }
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
    V p[EPT][NCH];
    int vrow[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int q = pt[j] & 0x7FFFFFFF;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            const bool in = tile0 + cc < rowlen;
            p[j][cc] = in ? O::scale(w[j], (ablate & 1) ? O::zero() : ssrc[(size_t)q * rowlen + tile0 + cc]) : O::zero();
        }
        vrow[j] = (row_ends[j] && kb + j < nnz) ? ((ablate & 4) ? (kb + j) & 1023 : csr_vid[kb + j]) : 0;
    }

    // in-thread: heads, and the sum since the last head (or of all four)
    int cnt = 0;
    V run[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) run[cc] = O::zero();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        cnt += head[j] ? 1 : 0;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) run[cc] = O::sel(head[j], p[j][cc], O::add(run[cc], p[j][cc]));
    }

    // wave inclusive scan
    int icnt = cnt;
    V isum[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) isum[cc] = run[cc];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ocnt = __shfl_up(icnt, off);
        V osum[NCH];
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) osum[cc] = O::shfl_up(isum[cc], off);
        if (lane >= off) {
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) isum[cc] = O::sel(icnt > 0, isum[cc], O::add(osum[cc], isum[cc]));
            icnt += ocnt;
        }
    }
    if (lane == 63) {
        wave_cnt[wave] = icnt;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) wave_sum[wave][cc] = isum[cc];
    }
    // exclusive value inside the wave
    int xcnt = __shfl_up(icnt, 1);
    V xsum[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) xsum[cc] = O::shfl_up(isum[cc], 1);
    if (lane == 0) {
        xcnt = 0;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) xsum[cc] = O::zero();
    }
    __syncthreads();
    // fold the totals of the waves before this one, left to right
    int pcnt = 0;
    V psum[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) psum[cc] = O::zero();
    for (int wv = 0; wv < wave; ++wv) {
        const int wc = wave_cnt[wv];
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) psum[cc] = O::sel(wc > 0, wave_sum[wv][cc], O::add(psum[cc], wave_sum[wv][cc]));
        pcnt += wc;
    }
    int hc = pcnt + xcnt;                 // heads in the chunk before this thread's corners
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) run[cc] = O::sel(xcnt > 0, xsum[cc], O::add(psum[cc], xsum[cc]));

#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int k = kb + j;
        if (k >= nnz) break;
        if (head[j]) ++hc;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) run[cc] = O::sel(head[j], p[j][cc], O::add(run[cc], p[j][cc]));
        const bool chunk_ends = (j == EPT - 1 && tid == kSplatBlock - 1);
        if (ablate & 2) continue;
        if (row_ends[j] || chunk_ends) {
            V *dst;
            size_t base;
            if (row_ends[j] && hc > 0) { dst = values; base = (size_t)vrow[j] * rowlen; }
            else { dst = (hc == 0) ? head_partial : tail_partial; base = (size_t)c * rowlen; }
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc)
                if (tile0 + cc < rowlen) dst[base + tile0 + cc] = run[cc];
        }
    }
}

// ----------------------------------------------------------------------------
// splat for WIDE rows (nch >= 32 chunks, i.e. vd >= 125: the backward pass, py:113-119).  The scan kernel tiles the
// columns, so with 50 chunks it would touch every gathered source row 17-25 times, 32-48 bytes at a time.  Here the
// lanes of a wave own the 16-byte chunks of the value row and the wave walks a range of corners: every gathered
// source row is read once, contiguously (splat_wide_kernel below).

// Where the splatted rows come from.  RowSource: a row-major [n_own][nch] float4 matrix in lattice
// order.  StackSource: the backward pass' stacked matrix [ g | g (x) x | s | s (x) x ] (py:113-118),
// never stored -- every 16-byte chunk is formed from the point's packed record
// rec[p] = [ g (L) | s (L) | x (d) | 0 | 1 | pad ]; a lane keeps, for each of its columns, the record
// slots of its two factors (the 0 and 1 slots serve the padding and the un-multiplied columns).
// stage() prepares a batch of up to 64 corners (lane e holds corner e's point), fetch() only issues
// the loads of corner j (so that several corners are in flight together), value() turns what was
// fetched into the 16-byte chunk.  Lanes beyond the row (chunk >= nch) fetch valid memory (the last
// chunk / the 0 and 1 slots) and are never stored.
struct RowSource {
    const float4 *ssrc;
    int nch;
    struct Lane { uint32_t off; };
    using Raw = float4;
    __device__ __forceinline__ void init(Lane &ln, int chunk) const { ln.off = 16u * (uint32_t)min(chunk, nch - 1); }
    __device__ __forceinline__ int lds_floats_per_corner() const { return 0; }
    __device__ __forceinline__ void stage(float *, int, int, int) const {}
    __device__ __forceinline__ Raw fetch(const Lane &ln, const float *, int, int pt) const
    {
        // wave-uniform row base (scalar registers) + 32-bit lane offset
        const char *row = reinterpret_cast<const char *>(ssrc + (size_t)pt * nch);
        return *reinterpret_cast<const float4 *>(row + ln.off);
    }
    static __device__ __forceinline__ float4 value(const Raw &r) { return r; }
};

// Eight 4-byte loads per corner with lane-dependent addresses cost the texture addresser 16 cycles
// each (measured: 2.0 ms per splat at N = 1e6, d = 8, L = 11, the address unit saturated), so the
// batch's records go through LDS first: coalesced 16-byte loads in, then eight conflict-free
// ds_read_b32 per corner (a record spans distinct banks).
struct StackSource {
    const float *rec;
    int recw, L, d;
    struct Lane { uint32_t a[4], x[4]; };       // float offsets of the two factors of each column
    struct Raw { float a[4], x[4]; };
    __device__ __forceinline__ void init(Lane &ln, int chunk) const
    {
        const int half = L * (1 + d), zero = 2 * L + d, one = zero + 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * chunk + j;
            int a, x;
            if (c >= 2 * half) { a = zero; x = one; }
            else {
                const int base = c < half ? 0 : L, cc = c < half ? c : c - half;
                if (cc < L) { a = base + cc; x = one; }
                else { const int q = cc - L, l = q / d; a = base + l; x = 2 * L + (q - l * d); }
            }
            ln.a[j] = (uint32_t)a;
            ln.x[j] = (uint32_t)x;
        }
    }
    __device__ __forceinline__ int lds_floats_per_corner() const { return recw; }
    __device__ __forceinline__ void stage(float *wl, int my_pt, int len, int lane) const
    {
        const int recw4 = recw / 4;
        const float4 *rec4 = reinterpret_cast<const float4 *>(rec);
        float4 *wl4 = reinterpret_cast<float4 *>(wl);
        for (int t = lane; t < 64 * recw4; t += 64) {          // same trip count in every lane (shuffles inside)
            const int e = t / recw4, c4 = t - e * recw4;
            const int pt = __shfl(my_pt, e);
            if (e < len) wl4[t] = rec4[(size_t)pt * recw4 + c4];
        }
    }
    __device__ __forceinline__ Raw fetch(const Lane &ln, const float *wl, int j, int) const
    {
        const float *r = wl + j * recw;
        Raw o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o.a[k] = r[ln.a[k]]; o.x[k] = r[ln.x[k]]; }
        return o;
    }
    static __device__ __forceinline__ float4 value(const Raw &r)
    {
        return make_float4(r.a[0] * r.x[0], r.a[1] * r.x[1], r.a[2] * r.x[2], r.a[3] * r.x[3]);
    }
};

constexpr int kWideChunk = 256;   // corners per wave of splat_wide_kernel

// One wave per kWideChunk consecutive corners, lanes over the 16-byte chunks of a row.  The corners
// are walked in CSR order: acc += w * row(point); a corner that closes its vertex row stores acc
// and clears it.  The chain corner -> point -> source row would cost two dependent memory
// latencies per corner (and a third per vertex row for its bounds), so (a) the lanes fetch 64
// corners' (point, weight, vertex, closes-row) in four coalesced loads, one batch ahead, and
// (b) the source rows of U corners are requested together before the first is used -- across
// vertex-row boundaries, which only matter to the accumulation.  A row that enters the range
// from the left leaves its sum in head_partial, one that leaves it to the right in
// tail_partial, for splat_fixup_kernel; the order of additions is fixed.
template <int MAXCH, class S>   // chunks a lane group can hold: nch <= 64 * MAXCH
__global__ __launch_bounds__(kBlock) void splat_wide_kernel(const int *__restrict__ csr_pt,
                                                            const float *__restrict__ csr_w,
                                                            const int *__restrict__ csr_vid,
                                                            const S src, int nch, int nnz,
                                                            float4 *__restrict__ values,
                                                            float4 *__restrict__ head_partial,
                                                            float4 *__restrict__ tail_partial, int ntiles, int remap)
{
    constexpr int U = (MAXCH == 1 ? 8 : 4) / (sizeof(typename S::Raw) > 16 ? 2 : 1);   // corners in flight
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = tile * (kBlock / 64) + wave;
    const int k0 = c * kWideChunk, k1 = min(k0 + kWideChunk, nnz);
    if (k0 >= nnz) return;
    typename S::Lane ln[MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) src.init(ln[q], lane + 64 * q);
    extern __shared__ float wide_lds[];                          // StackSource: [waves][64][recw] staged records
    float *wl = wide_lds + (size_t)wave * 64 * src.lds_floats_per_corner();

    float4 acc[MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) acc[q] = f4_zero();
    bool from_left = csr_pt[k0] >= 0;      // the first corner continues a row that began before k0
    bool open = false;                     // acc holds corners of a row that has not closed yet

    auto store = [&](float4 *dst) {
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
            if (lane + 64 * q < nch) dst[lane + 64 * q] = acc[q];
            acc[q] = f4_zero();
        }
    };
    auto fetch = [&](int base, int &pt, float &w, int &vid, int &closes) {
        const int e = base + lane;
        pt = 0; w = 0.f; vid = 0; closes = 0;
        if (e < k1) {
            pt = csr_pt[e] & 0x7FFFFFFF;
            w = csr_w[e];
            vid = csr_vid[e];
            closes = (e + 1 >= nnz) || (csr_pt[e + 1] < 0);
        }
    };

    int n_pt, n_vid, n_closes;
    float n_w;
    fetch(k0, n_pt, n_w, n_vid, n_closes);
    for (int base = k0; base < k1; base += 64) {
        const int my_pt = n_pt, my_vid = n_vid, my_closes = n_closes;
        const float my_w = n_w;
        if (base + 64 < k1) fetch(base + 64, n_pt, n_w, n_vid, n_closes);
        const int len = min(64, k1 - base);
        src.stage(wl, my_pt, len, lane);
        for (int j = 0; j < len; j += U) {
            typename S::Raw r[U][MAXCH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // wave-uniform point index in a scalar register: the row address is scalar base + lane offset
                const int jj = min(j + u, len - 1);                                      // past the end: a harmless re-read
                const int pt = __builtin_amdgcn_readlane(my_pt, jj);
#pragma unroll
                for (int q = 0; q < MAXCH; ++q) r[u][q] = src.fetch(ln[q], wl, jj, pt);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < len) {
                    const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j + u));
#pragma unroll
                    for (int q = 0; q < MAXCH; ++q) acc[q] = f4_add(acc[q], f4_scale(w, S::value(r[u][q])));
                    open = true;
                    if (__builtin_amdgcn_readlane(my_closes, j + u)) {
                        store(from_left ? head_partial + (size_t)c * nch
                                        : values + (size_t)__builtin_amdgcn_readlane(my_vid, j + u) * nch);
                        from_left = false;
                        open = false;
                    }
                }
            }
        }
    }
    if (open) store((from_left ? head_partial : tail_partial) + (size_t)c * nch);
}

// splat for NARROW multi-column rows (2..16 chunks, vd 2..64: every CG iteration runs at vd = 1 + probes).
// The scan kernel keeps every corner's whole row in one thread and scans 12-float elements through six wave
// shuffles (173 us at N = 1e6, d = 8, vd = 11).  Here a wave is cut into groups of NCHP lanes, one lane per
// 16-byte chunk; a group walks kGroupRun consecutive corners the way a splat_wide_kernel wave walks its
// range: acc += w * row(point), a corner that closes its vertex row stores acc.  What a group cannot finish
// alone -- the row that enters its run from the left -- is joined by one segmented scan over the groups of
// the wave (log2(groups) shuffle steps of one float4 per lane), and only rows that cross the edge of the
// wave's corner range go through head / tail partials and splat_fixup_kernel.  The wave's corner indices,
// weights and vertex ids are staged through LDS by coalesced loads (a group region holds its run plus the
// corner after it, stride kGroupRun + 1 words: no bank conflicts between groups).
constexpr int kGroupRun = 32;      // corners per lane group

template <class V, int NCHP, int RUN>
__global__ __launch_bounds__(kBlock) void splat_group_kernel(const int *__restrict__ csr_pt, const float *__restrict__ csr_w,
                                                             const int *__restrict__ csr_vid,
                                                             const V *__restrict__ ssrc, int nch, int nnz,
                                                             V *__restrict__ values,
                                                             V *__restrict__ head_partial,
                                                             V *__restrict__ tail_partial, int ntiles, int remap,
                                                             int ablate)
{
    using O = VecOps<V>;
    constexpr int G = 64 / NCHP;                       // groups per wave
    constexpr int WC = G * RUN;                  // corners per wave = one chunk of the partial protocol
    constexpr int RS = RUN + 1;                  // LDS words per group region
    constexpr int U = 8;                               // source rows in flight per lane
    __shared__ int lds_pt[kBlock / 64][G * RS];
    __shared__ float lds_w[kBlock / 64][G * RS];
    __shared__ int lds_vid[kBlock / 64][G * RS];
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wchunk = tile * (kBlock / 64) + wave;
    const int k0 = wchunk * WC;
    if (k0 >= nnz) return;
    // stage: corner k0 + r -> slot (r / run) * RS + r % run; the first corner of a run also closes the
    // region of the run before it (its sign bit says whether that run's last corner closes a row)
    for (int r = lane; r <= WC; r += 64) {
        const int e = k0 + r;
        const bool in = e < nnz && r < WC;
        const int pt = e < nnz ? csr_pt[e] : (int)0x80000000;          // past the data: reads as a row head
        const float w = in ? csr_w[e] : 0.f;
        const int vid = in ? csr_vid[e] : 0;
        const int rg = r / RUN, rj = r - rg * RUN;
        if (r < WC) { lds_pt[wave][rg * RS + rj] = pt; lds_w[wave][rg * RS + rj] = w; lds_vid[wave][rg * RS + rj] = vid; }
        if (rj == 0 && rg > 0) lds_pt[wave][(rg - 1) * RS + RUN] = pt;
    }
    __builtin_amdgcn_wave_barrier();                   // LDS traffic of one wave is in order; keep the compiler from moving it

    const int g = lane / NCHP, cl = lane - g * NCHP;
    const bool col = cl < nch;
    const int len = min(RUN, nnz - (k0 + g * RUN));   // <= 0: this group has no corners
    const int *gp = lds_pt[wave] + g * RS;
    const float *gw = lds_w[wave] + g * RS;
    const int *gv = lds_vid[wave] + g * RS;
    const uint32_t coff = (uint32_t)min(cl, nch - 1);

    V acc = O::zero(), left_part = O::zero();
    const bool from_left = len > 0 && gp[0] >= 0;      // the first corner continues the row of the run before
    bool left_closed = false, closed_any = false;
    int left_vid = 0;
    // Branch-free per corner except for one predicated store: groups close their rows at different
    // corners, and every divergent branch costs the whole wave its scalar bookkeeping (a first version
    // with nested ifs spent 35 scalar + 28 vector instructions per corner).  Corners past the end of
    // the run have weight 0 and never close.
    for (int j = 0; j < RUN; j += U) {
        int pr[U + 1], vd_[U];
        float w[U];
        V row[U];
#pragma unroll
        for (int u = 0; u <= U; ++u) pr[u] = gp[min(j + u, RUN)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w[u] = gw[j + u];
            vd_[u] = gv[j + u];
            const uint32_t q = (j + u < len && !(ablate & 1)) ? (uint32_t)(pr[u] & 0x7FFFFFFF) : 0u;
            row[u] = ssrc[q * (uint32_t)nch + coff];           // 32-bit index: splat_impl checks n_own * nch < 2^32
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = O::add(acc, O::scale(w[u], row[u]));
            const bool closes = (j + u < len) && pr[u + 1] < 0;   // the next corner starts a row, or the data ends
            const bool first_left = closes && from_left && !closed_any;
            left_part = O::sel(first_left, acc, left_part);
            left_vid = first_left ? vd_[u] : left_vid;
            left_closed = left_closed || first_left;
            if (closes && !first_left && col && !(ablate & 2)) values[(uint32_t)vd_[u] * (uint32_t)nch + (uint32_t)cl] = acc;
            acc = O::sel(closes, O::zero(), acc);
            closed_any = closed_any || closes;
        }
    }
    // segmented scan over the groups: (closes seen, sum since the last close)
    int icnt = closed_any ? 1 : 0;
    V ival = acc;
#pragma unroll
    for (int off = NCHP; off < 64; off <<= 1) {
        const int ocnt = __shfl_up(icnt, off);
        const V oval = O::shfl_up(ival, off);
        if (lane >= off) {
            ival = O::sel(icnt > 0, ival, O::add(oval, ival));
            icnt += ocnt;
        }
    }
    int xcnt = __shfl_up(icnt, NCHP);
    V xval = O::shfl_up(ival, NCHP);
    if (lane < NCHP) { xcnt = 0; xval = O::zero(); }
    const bool wave_from_left = lds_pt[wave][0] >= 0;
    if (left_closed && col) {
        // the row that entered this run from the left: what earlier groups hold of it + this group's part
        V *dst = (xcnt == 0 && wave_from_left) ? head_partial + (size_t)wchunk * nch : values + (size_t)left_vid * nch;
        dst[cl] = O::add(xval, left_part);
    }
    if (g == G - 1 && col && lds_pt[wave][(G - 1) * RS + RUN] >= 0) {
        // the row still open at the end of the wave's range
        V *dst = (icnt == 0 && wave_from_left) ? head_partial : tail_partial;
        dst[(size_t)wchunk * nch + cl] = ival;
    }
}

__device__ __forceinline__ bool chunk_has_head(const int *__restrict__ csr_pt, const int *__restrict__ csr_vid,
                                               int k0, int k1)
{
    return csr_pt[k0] < 0 || csr_vid[k0] != csr_vid[k1 - 1];
}

// A vertex row that starts inside chunk c and runs past its end: add the head
// partials of the chunks it covers, in chunk order.  One thread per (chunk, float).
__global__ __launch_bounds__(kBlock) void splat_fixup_kernel(const int *__restrict__ csr_pt,
                                                             const int *__restrict__ csr_vid, int nchunks,
                                                             int chunk, int nnz, int vdp,
                                                             const float *__restrict__ head_partial,
                                                             const float *__restrict__ tail_partial,
                                                             float *__restrict__ values)
{
    const int it = blockIdx.x * kBlock + threadIdx.x;
    if (it >= nchunks * vdp) return;
    const int c = it / vdp, col = it - c * vdp;
    const int k0 = c * chunk, k1 = min(k0 + chunk, nnz);
    if (k1 >= nnz || csr_pt[k1] < 0 || !chunk_has_head(csr_pt, csr_vid, k0, k1)) return;
    float total = tail_partial[(size_t)c * vdp + col];
    for (int c2 = c + 1; c2 < nchunks; ++c2) {
        total += head_partial[(size_t)c2 * vdp + col];
        const int a = c2 * chunk, b = min(a + chunk, nnz);
        if (b >= nnz || csr_pt[b] < 0 || chunk_has_head(csr_pt, csr_vid, a, b)) break;
    }
    values[(size_t)csr_vid[k1 - 1] * vdp + col] = total;
}

int splat_impl(plx_lattice *L, const float *d_src, int vd, float *d_values, hipStream_t stream)
{
    const int64_t m = L->m;
    const int vdp = values_stride(vd);
    const int n_own = (int)(L->own_end - L->own_begin);
    if (L->nnz == 0) {
        PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)m * vdp * 4, stream));
        return PLX_OK;
    }
    const bool all_rows_touched = (L->n_shards == 1 && !L->partial_cover);
    if (!all_rows_touched) PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)m * vdp * 4, stream));
    PLX_TRY(ensure(L->head_partial, (size_t)L->nchunks * vdp * 4));
    PLX_TRY(ensure(L->tail_partial, (size_t)L->nchunks * vdp * 4));
    PLX_TRY(ensure(L->ssrc, (size_t)n_own * vdp * 4));
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const float *ss = L->ssrc.as<float>();
    const int *pt = L->csr_pt.as<int>();
    const bool direct = g_splat_direct == 2 || (g_splat_direct == 1 && L->nnz <= 2000000);
    if (vd == 1 && (L->lattice_rows || direct)) {
        // single column: no padding needed, so gather straight from the caller's buffer -- through the
        // lattice-order indices when the rows are in lattice order, else through the caller-row indices
        ss = d_src;
        if (!L->lattice_rows) pt = L->csr_row.as<int>();
    } else if (vd == 1) {
        gather_in_v1_kernel<<<ceil_div(n_own, kBlock), kBlock, 0, stream>>>(d_src, perm, (int)L->own_begin, n_own,
                                                                            L->ssrc.as<float>());
    } else {
        gather_in_kernel<<<ceil_div((int64_t)n_own * vdp, kBlock), kBlock, 0, stream>>>(
            d_src, perm, (int)L->own_begin, n_own, vd, vdp, L->ssrc.as<float>());
    }
    const float *w = L->csr_w.as<float>();
    const int *vid = L->sort_keys_out.as<int>();   // sorted vertex id of every corner
    float *hp = L->head_partial.as<float>(), *tp = L->tail_partial.as<float>();
    const int nnz = (int)L->nnz, nch_total = vdp / 4, nchunks = (int)L->nchunks;
    if (vd == 1) {
        // (one lane per run of corners -- splat_group_kernel<float, 1, 16> -- was measured 18-75 % slower here:
        // the scan kernel's 16-byte index loads and coalesced stores win on single-column rows)
        splat_scan_kernel<float, 1><<<tile_grid(nchunks, g_xcd_remap), kSplatBlock, 0, stream>>>(pt, w, vid, ss, 1, nnz, d_values, hp, tp, g_splat_ablate, nchunks, g_xcd_remap);
    } else {
        const float4 *s4 = reinterpret_cast<const float4 *>(ss);
        float4 *v4 = reinterpret_cast<float4 *>(d_values), *h4 = reinterpret_cast<float4 *>(hp), *t4 = reinterpret_cast<float4 *>(tp);
        if (nch_total >= 32 && nch_total <= 128 && g_splat_wide) {   // measured: 16 chunks 8 % slower, 50 chunks 1.8x faster
            const int nwide = ceil_div(nnz, kWideChunk), nt = ceil_div(nwide, kBlock / 64);
            PLX_TRY(ensure(L->head_partial, (size_t)nwide * vdp * 4));
            PLX_TRY(ensure(L->tail_partial, (size_t)nwide * vdp * 4));
            h4 = reinterpret_cast<float4 *>(L->head_partial.as<float>());
            t4 = reinterpret_cast<float4 *>(L->tail_partial.as<float>());
            const int grid = tile_grid(nt, g_xcd_remap);
            const RowSource rows{s4, nch_total};
            if (nch_total <= 64)
                splat_wide_kernel<1, RowSource><<<grid, kBlock, 0, stream>>>(pt, w, vid, rows, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap);
            else
                splat_wide_kernel<2, RowSource><<<grid, kBlock, 0, stream>>>(pt, w, vid, rows, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap);
            splat_fixup_kernel<<<ceil_div((int64_t)nwide * vdp, kBlock), kBlock, 0, stream>>>(
                pt, vid, nwide, kWideChunk, nnz, vdp, L->head_partial.as<float>(), L->tail_partial.as<float>(), d_values);
            tmark(L, stream);
            PLX_HIP_TRY(hipGetLastError());
            return PLX_OK;
        }
        // one chunk (vd 2..4): the scan kernel is 25 % faster; two and more: the group kernel by 5 % .. 4x
        if (nch_total >= 2 && nch_total <= 16 && g_splat_group && (int64_t)n_own * nch_total < (1ll << 32) && (int64_t)m * nch_total < (1ll << 32)) {
            const int nchp = nch_total <= 2 ? 2 : (nch_total <= 4 ? 4 : (nch_total <= 8 ? 8 : 16));
            const int wc = (64 / nchp) * kGroupRun;
            const int nwchunks = ceil_div(nnz, wc), nt = ceil_div(nwchunks, kBlock / 64);
            PLX_TRY(ensure(L->head_partial, (size_t)nwchunks * vdp * 4));
            PLX_TRY(ensure(L->tail_partial, (size_t)nwchunks * vdp * 4));
            h4 = reinterpret_cast<float4 *>(L->head_partial.as<float>());
            t4 = reinterpret_cast<float4 *>(L->tail_partial.as<float>());
            const int grid = tile_grid(nt, g_xcd_remap);
            switch (nchp) {
            case 2: splat_group_kernel<float4, 2, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, g_splat_ablate); break;
            case 4: splat_group_kernel<float4, 4, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, g_splat_ablate); break;
            case 8: splat_group_kernel<float4, 8, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, g_splat_ablate); break;
            default: splat_group_kernel<float4, 16, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, g_splat_ablate); break;
            }
            splat_fixup_kernel<<<ceil_div((int64_t)nwchunks * vdp, kBlock), kBlock, 0, stream>>>(
                pt, vid, nwchunks, wc, nnz, vdp, L->head_partial.as<float>(), L->tail_partial.as<float>(), d_values);
            tmark(L, stream);
            PLX_HIP_TRY(hipGetLastError());
            return PLX_OK;
        }
        // up to 3 chunks (12 columns) per workgroup in registers; wider rows take more column tiles
        const int nch = nch_total <= 3 ? nch_total : (nch_total % 3 == 0 ? 3 : (nch_total % 2 == 0 ? 2 : 3));
        dim3 grid((unsigned)tile_grid(nchunks, g_xcd_remap), (unsigned)ceil_div(nch_total, nch));
        switch (nch) {
        case 1: splat_scan_kernel<float4, 1><<<grid, kSplatBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, g_splat_ablate, nchunks, g_xcd_remap); break;
        case 2: splat_scan_kernel<float4, 2><<<grid, kSplatBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, g_splat_ablate, nchunks, g_xcd_remap); break;
        default: splat_scan_kernel<float4, 3><<<grid, kSplatBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, g_splat_ablate, nchunks, g_xcd_remap); break;
        }
    }
    splat_fixup_kernel<<<ceil_div((int64_t)nchunks * vdp, kBlock), kBlock, 0, stream>>>(pt, vid, nchunks, kSplatChunk, nnz, vdp,
                                                                                         hp, tp, d_values);
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

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
#pragma unroll
    for (int k = 0; k < IPT; ++k)
#pragma unroll
        for (int s = 0; s < 2 * ORDER; ++s) g[k][s] = old[src[k][s]];
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

template <int ORDER>
static void launch_blur_v1(const float *cur, float *nxt, const int *nb, int m, int64_t mstride, const TapArgs &taps,
                           hipStream_t stream)
{
    const int nt4 = ceil_div(ceil_div(m, 4), kBlock), nt2 = ceil_div(ceil_div(m, 2), kBlock);
    if (g_blur_vpt == 4)
        blur_axis_v1_kernel<ORDER, 4><<<tile_grid(nt4, g_xcd_remap), kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, taps, g_blur_ablate, nt4, g_xcd_remap);
    else
        blur_axis_v1_kernel<ORDER, 2><<<tile_grid(nt2, g_xcd_remap), kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, taps, g_blur_ablate, nt2, g_xcd_remap);
}

template <class V>
static void launch_blur_general(const V *cur, V *nxt, const int *nb, int m, int64_t mstride, int rowlen, int order,
                                const TapArgs &taps, hipStream_t stream)
{
    const int nt = ceil_div((int64_t)m * rowlen, kBlock);
    const int grid = tile_grid(nt, g_xcd_remap);
    switch (order) {
    case 1: blur_axis_kernel<V, 1><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, g_blur_ablate); break;
    case 2: blur_axis_kernel<V, 2><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, g_blur_ablate); break;
    case 3: blur_axis_kernel<V, 3><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, g_blur_ablate); break;
    default: blur_axis_kernel<V, 0><<<grid, kBlock, 0, stream>>>(cur, nxt, nb, m, mstride, rowlen, order, taps, nt, g_xcd_remap, g_blur_ablate); break;
    }
}

int blur_impl(plx_lattice *L, float *d_values, float *d_scratch, int vd, int *result_in_scratch,
              hipStream_t stream)
{
    const int m = (int)L->m, d1 = L->d + 1, order = L->order;
    const int vdp = values_stride(vd);
    const bool v1 = (vd == 1 && order >= 1 && order <= 3 && (g_blur_vpt == 2 || g_blur_vpt == 4));
    if (v1 && m <= kSmallM && g_blur_small) {
        // result goes where the per-axis path would leave it, so callers see no difference
        float *dst = (d1 & 1) ? d_scratch : d_values;
        const int *nb = L->nbr.as<int>();
        switch (order) {
        case 1: blur_small_kernel<1><<<1, 1024, 0, stream>>>(d_values, dst, nb, m, L->mstride, d1, L->taps); break;
        case 2: blur_small_kernel<2><<<1, 1024, 0, stream>>>(d_values, dst, nb, m, L->mstride, d1, L->taps); break;
        default: blur_small_kernel<3><<<1, 1024, 0, stream>>>(d_values, dst, nb, m, L->mstride, d1, L->taps); break;
        }
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
    for (int axis = 0; axis < d1; ++axis) {
        const int *nb = L->nbr.as<int>() + (size_t)axis * 2 * order * L->mstride;
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
        } else if (v1) {
            switch (order) {
            case 1: launch_blur_v1<1>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            case 2: launch_blur_v1<2>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            default: launch_blur_v1<3>(cur, nxt, nb, m, L->mstride, L->taps, stream); break;
            }
        } else if (vd == 1) {
            launch_blur_general<float>(cur, nxt, nb, m, L->mstride, 1, order, L->taps, stream);
        } else if (order >= 1 && order <= 3 && vdp / 4 <= 4 && g_blur_narrow && (int64_t)d1 * 2 * order * L->mstride < (1ll << 32)) {
            const float4 *c4 = reinterpret_cast<const float4 *>(cur);
            float4 *n4 = reinterpret_cast<float4 *>(nxt);
            // nb is already offset to this axis: plane offsets inside the kernel stay below 2 * order * mstride
            switch (order) {
            case 1: launch_blur_narrow<1>(c4, n4, nb, m, L->mstride, vdp / 4, L->taps, stream, g_xcd_remap); break;
            case 2: launch_blur_narrow<2>(c4, n4, nb, m, L->mstride, vdp / 4, L->taps, stream, g_xcd_remap); break;
            default: launch_blur_narrow<3>(c4, n4, nb, m, L->mstride, vdp / 4, L->taps, stream, g_xcd_remap); break;
            }
        } else if (order >= 1 && order <= 3 && g_blur_multi && vdp / 4 >= 32) {   // narrower rows: no gain (vd 2..16 measured 0-30 % slower)
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
        } else {
            launch_blur_general<float4>(reinterpret_cast<const float4 *>(cur), reinterpret_cast<float4 *>(nxt), nb, m,
                                        L->mstride, vdp / 4, order, L->taps, stream);
        }
        float *t = cur; cur = nxt; nxt = t;
    }
    tmark(L, stream);
    *result_in_scratch = (cur == d_scratch) ? 1 : 0;
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

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

// vd > 1: one thread per (point, 16-byte chunk)
__global__ __launch_bounds__(kBlock) void slice_vec_kernel(const int *__restrict__ evid,
                                                           const float *__restrict__ ew,
                                                           const uint32_t *__restrict__ perm, int n, int own_begin,
                                                           int n_own, int d1, const float4 *__restrict__ values,
                                                           int nch, int vd, float rden, float *__restrict__ out,
                                                           int ntiles, int remap, const float *__restrict__ affine,
                                                           const float *__restrict__ src)
{
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    const int64_t item = (int64_t)tile * kBlock + threadIdx.x;
    if (item >= (int64_t)n_own * nch) return;
    const int pl = (int)(item / nch), ch = (int)(item - (int64_t)pl * nch);
    const int p = own_begin + pl;
    float4 acc = f4_zero();
    for (int r = 0; r < d1; ++r) {
        const int v = evid[(size_t)r * n + p];
        const float w = ew[(size_t)r * n + p];
        const float4 g = values[(size_t)v * nch + ch];
        acc.x += w * g.x * rden; acc.y += w * g.y * rden; acc.z += w * g.z * rden; acc.w += w * g.w * rden;
    }
    const size_t row = perm ? (size_t)((int)perm[p] - own_begin) : (size_t)pl;
    float *o = out + row * vd + 4 * ch;
    const int left = vd - 4 * ch;
    if (affine) {
        const float a = affine[0], b = affine[1];
        const float *sp = src + row * vd + 4 * ch;
        acc.x = a * acc.x + b * sp[0];
        if (left > 1) acc.y = a * acc.y + b * sp[1];
        if (left > 2) acc.z = a * acc.z + b * sp[2];
        if (left > 3) acc.w = a * acc.w + b * sp[3];
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

int slice_impl(plx_lattice *L, const float *d_values, int vd, float *d_out, hipStream_t stream, const float *d_affine,
               const float *d_src)
{
    const int n_own = (int)(L->own_end - L->own_begin);
    if (n_own == 0) return PLX_OK;
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
    } else {
        const int nch = values_stride(vd) / 4;
        const int nt = ceil_div((int64_t)n_own * nch, kBlock);
        slice_vec_kernel<<<tile_grid(nt, g_xcd_remap), kBlock, 0, stream>>>(
            evid, ew, perm, n, ob, n_own, L->d + 1, reinterpret_cast<const float4 *>(d_values), nch, vd,
            1.0f / L->slice_denom, d_out, nt, g_xcd_remap, d_affine, d_src);
    }
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ----------------------------------------------------------------------------
// fused position gradient (py:113-123): splat of the stacked matrix straight from
// the packed records, the usual blur, then slice and contraction in one kernel --
// neither the stacked matrix nor its filtered image ever reach memory.

// rec[i] = [ g | s | x | 0 | 1 | pad ] of the i-th point in lattice order
__global__ __launch_bounds__(kBlock) void backward_pack_kernel(const float *__restrict__ g, const float *__restrict__ s,
                                                               const float *__restrict__ x,
                                                               const uint32_t *__restrict__ perm, int own_begin,
                                                               int n_own, int L, int d, int recw, float *__restrict__ rec)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= (int64_t)n_own * recw) return;
    const int i = (int)(item / recw), c = (int)(item - (int64_t)i * recw);
    const size_t row = perm ? (size_t)((int)perm[own_begin + i] - own_begin) : (size_t)i;
    float v = 0.f;
    if (c < L) v = g[row * L + c];
    else if (c < 2 * L) v = s[row * L + (c - L)];
    else if (c < 2 * L + d) v = x[row * d + (c - 2 * L)];
    else if (c == 2 * L + d + 1) v = 1.f;
    rec[item] = v;
}

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

int backward_impl(plx_lattice *lat, const float *d_g, const float *d_src, const float *d_x, int L, float *d_grad_x,
                  float *d_grad_src, hipStream_t stream)
{
    const int d = lat->d, W = 2 * L * (1 + d), vdp = values_stride(W), nch = vdp / 4;
    const int n_own = (int)(lat->own_end - lat->own_begin);
    const int recw = (2 * L + d + 2 + 3) & ~3;
    PLX_TRY(ensure(lat->rec, (size_t)n_own * recw * 4));
    PLX_TRY(ensure(lat->val_a, (size_t)lat->m * vdp * 4));
    PLX_TRY(ensure(lat->val_b, (size_t)lat->m * vdp * 4));
    const int nnz = (int)lat->nnz, nwide = ceil_div(nnz, kWideChunk), nwt = ceil_div(nwide, kBlock / 64);
    PLX_TRY(ensure(lat->head_partial, (size_t)nwide * vdp * 4));
    PLX_TRY(ensure(lat->tail_partial, (size_t)nwide * vdp * 4));
    const uint32_t *perm = lat->lattice_rows ? nullptr : lat->perm.as<uint32_t>();
    float *rec = lat->rec.as<float>();
    lat->tev_n = 0;
    tmark(lat, stream);
    backward_pack_kernel<<<ceil_div((int64_t)n_own * recw, kBlock), kBlock, 0, stream>>>(
        d_g, d_src, d_x, perm, (int)lat->own_begin, n_own, L, d, recw, rec);
    // splat
    float *va = lat->val_a.as<float>(), *vb = lat->val_b.as<float>();
    float4 *v4 = reinterpret_cast<float4 *>(va);
    float4 *h4 = reinterpret_cast<float4 *>(lat->head_partial.as<float>()), *t4 = reinterpret_cast<float4 *>(lat->tail_partial.as<float>());
    const int *pt = lat->csr_pt.as<int>(), *vid = lat->sort_keys_out.as<int>();
    const StackSource stack{rec, recw, L, d};
    const size_t wide_lds = (size_t)(kBlock / 64) * 64 * recw * 4;   // <= 64 KB: plx_apply_backward bounds recw
    const int grid = tile_grid(nwt, g_xcd_remap);
    if (nch <= 64)
        splat_wide_kernel<1, StackSource><<<grid, kBlock, wide_lds, stream>>>(pt, lat->csr_w.as<float>(), vid, stack, nch, nnz, v4, h4, t4, nwt, g_xcd_remap);
    else
        splat_wide_kernel<2, StackSource><<<grid, kBlock, wide_lds, stream>>>(pt, lat->csr_w.as<float>(), vid, stack, nch, nnz, v4, h4, t4, nwt, g_xcd_remap);
    splat_fixup_kernel<<<ceil_div((int64_t)nwide * vdp, kBlock), kBlock, 0, stream>>>(
        pt, vid, nwide, kWideChunk, nnz, vdp, lat->head_partial.as<float>(), lat->tail_partial.as<float>(), va);
    tmark(lat, stream);
    PLX_HIP_TRY(hipGetLastError());
    int in_b = 0;
    PLX_TRY(blur_impl(lat, va, vb, W, &in_b, stream));
    const float4 *res = reinterpret_cast<const float4 *>(in_b ? vb : va);
    const int nt = ceil_div(n_own, kBlock / 64);
    const int sgrid = tile_grid(nt, g_xcd_remap);
    if (nch <= 64)
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

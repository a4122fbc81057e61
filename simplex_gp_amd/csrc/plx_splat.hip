// plx_splat.hip -- gather-in and the splat kernels of the per-MVM path: segmented scan (1 chunk), lane groups (2..16 chunks), corner-streaming wide rows (>= 17 chunks, also the stacked source of the fused position gradient).  Reference: h:478-479.
// Overview of the per-MVM path, value-row layout and shared helpers: plx_kernels.h.

#include "plx_kernels.h"

namespace plx {

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

// the same, one thread per (row, 16-byte chunk): up to four 4-byte loads of the caller's row (its rows are vd floats: not
// 16-byte aligned unless vd is a multiple of 4), one aligned 16-byte store -- a quarter of the store instructions and
// of the index arithmetic of the per-float form (N = 4e6, vd = 11: 164 -> see DESIGN.md 4)
__global__ __launch_bounds__(kBlock) void gather_in_rows_kernel(const float *__restrict__ src,
                                                                const uint32_t *__restrict__ perm, int own_begin,
                                                                int n_own, int vd, int nch, float4 *__restrict__ ssrc)
{
    const int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= (int64_t)n_own * nch) return;
    const int i = (int)(item / nch), ch = (int)(item - (int64_t)i * nch);
    const int row = perm ? (int)perm[own_begin + i] - own_begin : i;
    const float *sp = src + (size_t)row * vd + 4 * ch;
    const int left = vd - 4 * ch;
    float4 r;
    if (left >= 4 && (vd & 3) == 0) {
        r = *reinterpret_cast<const float4 *>(sp);
    } else {
        r.x = sp[0];
        r.y = left > 1 ? sp[1] : 0.f;
        r.z = left > 2 ? sp[2] : 0.f;
        r.w = left > 3 ? sp[3] : 0.f;
    }
    ssrc[item] = r;
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
    ablate = PLX_DIAG_VALUE(ablate);                   // diagnostics are compiled into libplx_diag.so only
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
// splat for WIDE rows (nch >= 17 chunks, i.e. vd >= 65: the backward pass, py:113-119, and the evaluation's 101 columns).  The scan kernel tiles the
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
    ablate = PLX_DIAG_VALUE(ablate);                   // diagnostics are compiled into libplx_diag.so only
    using O = VecOps<V>;
    constexpr int G = 64 / NCHP;                       // groups per wave
    constexpr int WC = G * RUN;                  // corners per wave = one chunk of the partial protocol
    constexpr int RS = RUN + 1;                  // LDS words per group region
    constexpr int U = 8;                               // source rows in flight per lane (16: no faster)
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
    // (all loads of the wave issued before the first LDS store: with a load -> store loop a wave had three
    // 256-byte requests in flight at a time and the staging alone took a third of the kernel)
    constexpr int NST = WC / 64 + 1;                   // r = lane + 64 it covers 0 .. WC
    int st_pt[NST], st_vid[NST];
    float st_w[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int r = lane + 64 * it, e = k0 + r;
        const bool in = e < nnz && r < WC;
        st_pt[it] = (e < nnz && r <= WC) ? csr_pt[e] : (int)0x80000000;   // past the data: reads as a row head
        st_w[it] = in ? csr_w[e] : 0.f;
        st_vid[it] = in ? csr_vid[e] : 0;
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int r = lane + 64 * it;
        const int rg = r / RUN, rj = r - rg * RUN;
        if (r < WC) { lds_pt[wave][rg * RS + rj] = st_pt[it]; lds_w[wave][rg * RS + rj] = st_w[it]; lds_vid[wave][rg * RS + rj] = st_vid[it]; }
        if (r <= WC && rj == 0 && rg > 0) lds_pt[wave][(rg - 1) * RS + RUN] = st_pt[it];
    }
    __builtin_amdgcn_wave_barrier();                   // LDS traffic of one wave is in order; keep the compiler from moving it

    // NCHP need not divide 64 (3 chunks: 21 groups, lane 63 idles): lanes past the last group walk group 0's
    // region with no corners of their own and never store
    const int graw = lane / NCHP;
    const bool lane_on = graw < G;
    const int g = lane_on ? graw : 0, cl = lane - graw * NCHP;
    const bool col = cl < nch && lane_on;
    const int len = lane_on ? min(RUN, nnz - (k0 + g * RUN)) : 0;   // <= 0: this group has no corners
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

// splat of nb ONE-HOT columns (column b = the unit vector of point cand[b]): the d + 1 corners of every such point put their
// barycentric weight into column b of their vertex row, everything else is zero -- the general splat of such a right-hand
// side streams all nnz corners to add up nb (d + 1) numbers (the rows of K a pivoted Cholesky asks for, plx_pcg.hip)
__global__ void splat_onehot_kernel(const int *__restrict__ evid, const float *__restrict__ ew, const int *__restrict__ cand,
                                    int nb, int d1, int n, int vdp, float *__restrict__ values)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nb * d1) return;
    const int b = x / d1, r = x - b * d1;
    const int p = cand[b];
    if (p < 0 || p >= n) return;
    values[(size_t)evid[(size_t)r * n + p] * vdp + b] = ew[(size_t)r * n + p];     // the corners of one point are distinct vertices
}

int splat_onehot_impl(plx_lattice *L, const int *d_cand, int nb, int vd, float *d_values, hipStream_t stream)
{
    const int vdp = values_stride(vd), d1 = L->d + 1;
    PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)L->m * vdp * 4, stream));
    splat_onehot_kernel<<<ceil_div((int64_t)nb * d1, 64), 64, 0, stream>>>(L->evid.as<int>(), splat_weights(L), d_cand, nb, d1, (int)L->n,
                                                                           vdp, d_values);
    L->kn_splat = "splat_onehot_kernel";
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

int splat_impl(plx_lattice *L, const float *d_src, int vd, float *d_values, hipStream_t stream)
{
    const int64_t m = L->m;
    const int vdp = values_stride(vd);
    const int n_own = (int)(L->own_end - L->own_begin);
    if (L->nnz == 0) {
        PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)m * vdp * 4, stream));
        L->kn_splat = "";
        tmark(L, stream);
        return PLX_OK;
    }
    bool splat_blocks = false, slice_blocks = false;
    PLX_TRY(choose_paths(L, vd, stream, &splat_blocks, &slice_blocks));
    if (splat_blocks) return splat_block_impl(L, d_src, d_values, stream);
    if (vd == 1) {
        PLX_TRY(ensure_first(L, stream));
        if (L->use_first) return splat_first_impl(L, d_src, d_values, stream);
    }
    PLX_TRY(ensure_csr(L, stream));
    const bool all_rows_touched = (L->n_shards == 1 && !L->partial_cover);
    if (!all_rows_touched) PLX_HIP_TRY(hipMemsetAsync(d_values, 0, (size_t)m * vdp * 4, stream));
    PLX_TRY(ensure(L->head_partial, (size_t)L->nchunks * vdp * 4));
    PLX_TRY(ensure(L->tail_partial, (size_t)L->nchunks * vdp * 4));
    PLX_TRY(ensure(L->ssrc, (size_t)n_own * vdp * 4));
    const uint32_t *perm = L->lattice_rows ? nullptr : L->perm.as<uint32_t>();
    const float *ss = L->ssrc.as<float>();
    const int *pt = L->csr_pt.as<int>();
    bool gathered = false;
    const bool direct = g_splat_direct == 2 || (g_splat_direct == 1 && L->nnz <= 2000000);
    if (vd == 1 && (L->lattice_rows || direct)) {
        // single column: no padding needed, so gather straight from the caller's buffer -- through the
        // lattice-order indices when the rows are in lattice order, else through the caller-row indices
        ss = d_src;
        if (!L->lattice_rows) pt = L->csr_row.as<int>();
    } else if (vd == 1) {
        gathered = true;
        gather_in_v1_kernel<<<ceil_div(n_own, kBlock), kBlock, 0, stream>>>(d_src, perm, (int)L->own_begin, n_own,
                                                                            L->ssrc.as<float>());
    } else if (L->lattice_rows && vd == vdp && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0) {
        // rows already in lattice order and whole 16-byte vectors: splat straight from the caller's buffer
        ss = d_src;
    } else {
        gathered = true;
        if (g_perm_rows && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0)
            gather_in_rows_kernel<<<ceil_div((int64_t)n_own * (vdp / 4), kBlock), kBlock, 0, stream>>>(
                d_src, perm, (int)L->own_begin, n_own, vd, vdp / 4, reinterpret_cast<float4 *>(L->ssrc.as<float>()));
        else
            gather_in_kernel<<<ceil_div((int64_t)n_own * vdp, kBlock), kBlock, 0, stream>>>(
                d_src, perm, (int)L->own_begin, n_own, vd, vdp, L->ssrc.as<float>());
    }
    const float *w = L->csr_w.as<float>();
    const int *vid = L->csr_vid.as<int>();   // sorted vertex id of every corner
    float *hp = L->head_partial.as<float>(), *tp = L->tail_partial.as<float>();
    const int nnz = (int)L->nnz, nch_total = vdp / 4, nchunks = (int)L->nchunks;
    if (vd == 1) {
        // (one lane per run of corners -- splat_group_kernel<float, 1, 16> -- was measured 18-75 % slower here:
        // the scan kernel's 16-byte index loads and coalesced stores win on single-column rows)
        splat_scan_kernel<float, 1><<<tile_grid(nchunks, g_xcd_remap), kSplatBlock, 0, stream>>>(pt, w, vid, ss, 1, nnz, d_values, hp, tp, PLX_DIAG_VALUE(g_splat_ablate), nchunks, g_xcd_remap);
    } else {
        const float4 *s4 = reinterpret_cast<const float4 *>(ss);
        float4 *v4 = reinterpret_cast<float4 *>(d_values), *h4 = reinterpret_cast<float4 *>(hp), *t4 = reinterpret_cast<float4 *>(tp);
        // measured: 16 chunks 8 % slower than the group kernel, 50 chunks 1.8x faster than column tiles; 17..31 chunks (round 6,
        // tools/ab_splat_mid_r6.py: the evaluation's 101 columns; 6-26 of the wave's lanes idle): the MVM 6.91 -> 5.28 ms at
        // m = 1.9e6 against 9-13 column tiles through splat_scan_kernel.  splat_wide = 2: the gate of rounds 1-5 (32 chunks)
        if (nch_total >= (g_splat_wide >= 2 ? 32 : 17) && nch_total <= 128 && g_splat_wide) {
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
            L->kn_splat = gathered ? "gather_in_kernel+splat_wide_kernel+splat_fixup_kernel" : "splat_wide_kernel+splat_fixup_kernel";
            tmark(L, stream);
            PLX_HIP_TRY(hipGetLastError());
            return PLX_OK;
        }
        // one chunk (vd 2..4): the scan kernel is 25 % faster; two and more: the group kernel by 5 % .. 4x
        if (nch_total >= 2 && nch_total <= 16 && g_splat_group && (int64_t)n_own * nch_total < (1ll << 32) && (int64_t)m * nch_total < (1ll << 32)) {
            const int nchp = nch_total <= 2 ? 2 : (nch_total == 3 ? 3 : (nch_total <= 4 ? 4 : (nch_total <= 8 ? 8 : 16)));
            const int wc = (64 / nchp) * kGroupRun;
            const int nwchunks = ceil_div(nnz, wc), nt = ceil_div(nwchunks, kBlock / 64);
            PLX_TRY(ensure(L->head_partial, (size_t)nwchunks * vdp * 4));
            PLX_TRY(ensure(L->tail_partial, (size_t)nwchunks * vdp * 4));
            h4 = reinterpret_cast<float4 *>(L->head_partial.as<float>());
            t4 = reinterpret_cast<float4 *>(L->tail_partial.as<float>());
            const int grid = tile_grid(nt, g_xcd_remap);
            switch (nchp) {
            case 2: splat_group_kernel<float4, 2, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, PLX_DIAG_VALUE(g_splat_ablate)); break;
            case 3: splat_group_kernel<float4, 3, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, PLX_DIAG_VALUE(g_splat_ablate)); break;
            case 4: splat_group_kernel<float4, 4, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, PLX_DIAG_VALUE(g_splat_ablate)); break;
            case 8: splat_group_kernel<float4, 8, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, PLX_DIAG_VALUE(g_splat_ablate)); break;
            default: splat_group_kernel<float4, 16, kGroupRun><<<grid, kBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, nt, g_xcd_remap, PLX_DIAG_VALUE(g_splat_ablate)); break;
            }
            splat_fixup_kernel<<<ceil_div((int64_t)nwchunks * vdp, kBlock), kBlock, 0, stream>>>(
                pt, vid, nwchunks, wc, nnz, vdp, L->head_partial.as<float>(), L->tail_partial.as<float>(), d_values);
            L->kn_splat = gathered ? "gather_in_kernel+splat_group_kernel+splat_fixup_kernel" : "splat_group_kernel+splat_fixup_kernel";
            tmark(L, stream);
            PLX_HIP_TRY(hipGetLastError());
            return PLX_OK;
        }
        // up to 3 chunks (12 columns) per workgroup in registers; wider rows take more column tiles
        const int nch = nch_total <= 3 ? nch_total : (nch_total % 3 == 0 ? 3 : (nch_total % 2 == 0 ? 2 : 3));
        dim3 grid((unsigned)tile_grid(nchunks, g_xcd_remap), (unsigned)ceil_div(nch_total, nch));
        switch (nch) {
        case 1: splat_scan_kernel<float4, 1><<<grid, kSplatBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, PLX_DIAG_VALUE(g_splat_ablate), nchunks, g_xcd_remap); break;
        case 2: splat_scan_kernel<float4, 2><<<grid, kSplatBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, PLX_DIAG_VALUE(g_splat_ablate), nchunks, g_xcd_remap); break;
        default: splat_scan_kernel<float4, 3><<<grid, kSplatBlock, 0, stream>>>(pt, w, vid, s4, nch_total, nnz, v4, h4, t4, PLX_DIAG_VALUE(g_splat_ablate), nchunks, g_xcd_remap); break;
        }
    }
    splat_fixup_kernel<<<ceil_div((int64_t)nchunks * vdp, kBlock), kBlock, 0, stream>>>(pt, vid, nchunks, kSplatChunk, nnz, vdp,
                                                                                         hp, tp, d_values);
    L->kn_splat = vd == 1 ? (gathered ? "gather_in_v1_kernel+splat_scan_kernel+splat_fixup_kernel" : "splat_scan_kernel+splat_fixup_kernel")
                          : (gathered ? "gather_in_kernel+splat_scan_kernel+splat_fixup_kernel" : "splat_scan_kernel+splat_fixup_kernel");
    tmark(L, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}


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

// The splat half of the fused position gradient: pack the (g, src, x) records in lattice order, then splat
// the stacked matrix [ g | g (x) x | src | src (x) x ] formed from them on the fly (StackSource) into d_values.
int splat_stack_impl(plx_lattice *lat, const float *d_g, const float *d_src, const float *d_x, int L, float *d_values,
                     hipStream_t stream)
{
    const int d = lat->d, W = 2 * L * (1 + d), vdp = values_stride(W), nch = vdp / 4;
    const int n_own = (int)(lat->own_end - lat->own_begin);
    const int recw = backward_record_width(L, d);
    PLX_TRY(ensure_csr(lat, stream));
    const int nnz = (int)lat->nnz, nwide = ceil_div(nnz, kWideChunk), nwt = ceil_div(nwide, kBlock / 64);
    PLX_TRY(ensure(lat->rec, (size_t)n_own * recw * 4));
    PLX_TRY(ensure(lat->head_partial, (size_t)nwide * vdp * 4));
    PLX_TRY(ensure(lat->tail_partial, (size_t)nwide * vdp * 4));
    const uint32_t *perm = lat->lattice_rows ? nullptr : lat->perm.as<uint32_t>();
    float *rec = lat->rec.as<float>();
    backward_pack_kernel<<<ceil_div((int64_t)n_own * recw, kBlock), kBlock, 0, stream>>>(
        d_g, d_src, d_x, perm, (int)lat->own_begin, n_own, L, d, recw, rec);
    float4 *v4 = reinterpret_cast<float4 *>(d_values);
    float4 *h4 = reinterpret_cast<float4 *>(lat->head_partial.as<float>()), *t4 = reinterpret_cast<float4 *>(lat->tail_partial.as<float>());
    const int *pt = lat->csr_pt.as<int>(), *vid = lat->csr_vid.as<int>();
    const StackSource stack{rec, recw, L, d};
    const size_t wide_lds = (size_t)(kBlock / 64) * 64 * recw * 4;   // <= 64 KB: plx_apply_backward bounds recw
    const int grid = tile_grid(nwt, g_xcd_remap);
    if (nch <= 64)
        splat_wide_kernel<1, StackSource><<<grid, kBlock, wide_lds, stream>>>(pt, lat->csr_w.as<float>(), vid, stack, nch, nnz, v4, h4, t4, nwt, g_xcd_remap);
    else
        splat_wide_kernel<2, StackSource><<<grid, kBlock, wide_lds, stream>>>(pt, lat->csr_w.as<float>(), vid, stack, nch, nnz, v4, h4, t4, nwt, g_xcd_remap);
    splat_fixup_kernel<<<ceil_div((int64_t)nwide * vdp, kBlock), kBlock, 0, stream>>>(
        pt, vid, nwide, kWideChunk, nnz, vdp, lat->head_partial.as<float>(), lat->tail_partial.as<float>(), d_values);
    lat->kn_splat = "backward_pack_kernel+splat_wide_kernel+splat_fixup_kernel";
    tmark(lat, stream);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

// plx_build.hip -- lattice construction on gfx950.
//
// Replaces the structural half of the reference's splat() and every hash lookup
// of its blur() (cpp/permutohedral.h:395-475, 482-484, 539-545; "h" below).
// Nothing here is a translation of the reference's CUDA kernels: the reference
// re-hashes every neighbour on every MVM and patches racy duplicate inserts
// afterwards (cuda/permutohedral_cuda_kernel.cu:298-332); here the lattice is
// built once, duplicate-free by construction, numbered exactly like the CPU
// reference (first touch, h:73-79), and turned into gather tables.
//
// Pipeline (all arrays "entry"-indexed are SoA [r][p], r = simplex corner,
// p = point IN LATTICE ORDER, so that every wave access is a contiguous run):
//   order    per point: rounded lattice coordinates -> 64-bit key (shard id, then
//            the coordinates lexicographically); radix sort -> perm.  Points that
//            share or neighbour a simplex become neighbours in memory, which is
//            what makes the splat / slice / blur gathers hit the same cache lines
//            (measured: warm MVM 159 -> 118 us at N=1e6, d=8)
//   embed    per point: elevate, round, rank, barycentric (registers only)
//            -> the 32-byte point record prec[p] (vertex coordinates + ranks: any corner key follows), weights ew[r][p]
//   insert   per corner: open addressing on a uint32 table whose slot value is
//            the SMALLEST reference entry index e = p*(d+1)+r seen with that
//            key (CAS to claim, atomicMin to lower)  -> eslot[r][p]
//   number   per point: corner is "first touch" iff table[slot] == e; counts are
//            scanned in (p, r) order => vertex ids are the first-touch order of
//            the reference (h:73-79) applied to the lattice-ordered points:
//            deterministic, identical on every rank, spatially coherent
//   ids      per corner: evid[r][p] = id(table[eslot])
//   neighbours  per (vertex, axis): hash the 2r neighbour keys once -> nbr table
//   csr      stable radix sort of corners by vertex id -> splat CSR
//
// This file is compiled with -ffp-contract=off: the embedding must round
// exactly like the reference's FMA-free x86-64 build or points on a rounding
// boundary pick a different simplex.

#include "plx_internal.h"
#include "plx_kernels.h"

#include <math.h>

#include <string.h>

#include <algorithm>
#include <chrono>
#include <utility>

namespace plx {

// ----------------------------------------------------------------------------
// small device helpers

// ---- the table hash (round 5; the 64-bit mix of the packed key words it replaced was removed in round 6): LINEAR in the
// key coordinates before one multiplicative mix.
//   s(key) = sum_c key[c] * kHashMul[c]  (mod 2^32);   slot bits = top 3 bits of s, then the top bits of (s ^ (s >> 15)) * kHashMix
// A blur neighbour's key is the vertex's own plus a constant vector (h:541-542), so its s is the vertex's s plus a constant
// per (axis, tap): one add instead of re-hashing 16 bytes -- which is what lets every XCD look at every lookup and serve
// only those whose slot falls into its own eighth of the table (neighbor_sliced_kernel): the owning eighth is (s + delta) >> 29.
// Fingerprint: the top byte of a second product of the same s ^ (s >> 15).  Which slot a key lands in is internal: vertex ids
// come from first touch (h:73-79), so the structure does not depend on the hash
// (tests/test_hip_parity.py::test_structure_bit_exact_round5_build_paths).
#define PLX_HASH_MULS                                                                                          \
    0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u,    \
    0x7FEB352Du, 0x846CA68Bu, 0x2C1B3C6Du, 0x297A2D39u, 0x9C06FAF5u, 0xCC9E2D51u, 0x1B873593u, 0xE6546B65u,    \
    0x63D83595u, 0x4CF5AD43u, 0xA54FF53Bu, 0x510E527Fu, 0x9B05688Du, 0x1F83D9ABu, 0x5BE0CD19u, 0xBB67AE85u,    \
    0x3C6EF373u, 0xCBBB9D5Du, 0x629A292Bu, 0x9159015Bu, 0x152FECD9u, 0x67332667u, 0x8EB44A87u, 0xDB0C2E0Du
__device__ constexpr uint32_t kHashMul[PLX_MAX_DIM] = {PLX_HASH_MULS};
static const uint32_t kHashMulHost[PLX_MAX_DIM] = {PLX_HASH_MULS};      // the host's copy (neighbour deltas)
static_assert(PLX_MAX_DIM == 32, "one odd multiplier per key coordinate");
constexpr uint32_t kHashMix = 0x2545F491u, kHashFp = 0x9E3779B9u;

__device__ __forceinline__ uint32_t lin_mix(uint32_t s) { return (s ^ (s >> 15)) * kHashMix; }
__device__ __forceinline__ uint32_t lin_fp(uint32_t s) { return ((s ^ (s >> 15)) * kHashFp) >> 24; }
// the value whose top log2(cap) bits are the slot: the top 3 bits -- the eighth of the table, i.e. the XCD that serves the
// lookup -- are those of s itself (one add and one shift decide whose lookup it is), the bits below them come from the
// mix.  (s alone as the slot was measured on the CPU: same probe lengths as a random hash at l = 0.25, 16 % more probes per
// insert at l = 0.5 -- the arithmetic progressions of a linear hash cluster a little; tools/slot_locality_study.py.)
__device__ __forceinline__ uint32_t lin_slotbits(uint32_t s) { return (s & 0xE0000000u) | (lin_mix(s) >> 3); }
// map nibble of an occupied slot: 1 + (fingerprint byte scaled to 0..14)
__device__ __forceinline__ uint32_t fp_nibble(uint32_t fp8) { return 1u + ((fp8 * 15u) >> 8); }

template <int D>
__device__ __forceinline__ uint32_t lin_hash_coords(const int (&key)[D])
{
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) s += (uint32_t)key[c] * kHashMul[c];
    return s;
}

template <int D>
__device__ __forceinline__ uint32_t lin_hash_packed(const uint32_t (&kw)[(D + 1) / 2])
{
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) s += (uint32_t)(int)(int16_t)((kw[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu) * kHashMul[c];
    return s;
}

// what the table kernels need to know about the table: slot = x >> shift (the top log2(cap) bits), probing wraps with mask
struct HashSel { uint32_t mask; int shift; };

template <int D>
__device__ __forceinline__ uint32_t key_hash(const uint32_t (&kw)[(D + 1) / 2], const HashSel &)
{
    return lin_slotbits(lin_hash_packed<D>(kw));
}
__device__ __forceinline__ uint32_t hash_slot(uint32_t x, const HashSel &hs) { return x >> hs.shift; }

// table word of a numbered vertex: the id, and (fp_on: m < 2^24 - 1) eight fingerprint bits of the key above it.
// An id field of all ones never occurs, so kEmpty stays distinguishable.
constexpr int kFpShift = 24;
template <int D>
__device__ __forceinline__ uint32_t table_word(uint32_t id, const uint32_t (&kw)[(D + 1) / 2], int fp_on)
{
    return fp_on ? (id | (lin_fp(lin_hash_packed<D>(kw)) << kFpShift)) : id;
}

template <int DW>
__device__ __forceinline__ void load_key(const uint32_t *__restrict__ base, size_t idx, uint32_t (&k)[DW])
{
    const uint32_t *p = base + idx * DW;
    if constexpr (DW % 4 == 0) {
#pragma unroll
        for (int j = 0; j < DW / 4; ++j) {
            uint4 v = reinterpret_cast<const uint4 *>(p)[j];
            k[4 * j] = v.x; k[4 * j + 1] = v.y; k[4 * j + 2] = v.z; k[4 * j + 3] = v.w;
        }
    } else if constexpr (DW % 2 == 0) {
#pragma unroll
        for (int j = 0; j < DW / 2; ++j) {
            uint2 v = reinterpret_cast<const uint2 *>(p)[j];
            k[2 * j] = v.x; k[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < DW; ++j) k[j] = p[j];
    }
}

template <int DW>
__device__ __forceinline__ void store_key(uint32_t *__restrict__ base, size_t idx, const uint32_t (&k)[DW])
{
    uint32_t *p = base + idx * DW;
    if constexpr (DW % 4 == 0) {
#pragma unroll
        for (int j = 0; j < DW / 4; ++j)
            reinterpret_cast<uint4 *>(p)[j] = make_uint4(k[4 * j], k[4 * j + 1], k[4 * j + 2], k[4 * j + 3]);
    } else if constexpr (DW % 2 == 0) {
#pragma unroll
        for (int j = 0; j < DW / 2; ++j)
            reinterpret_cast<uint2 *>(p)[j] = make_uint2(k[2 * j], k[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < DW; ++j) p[j] = k[j];
    }
}

template <int DW>
__device__ __forceinline__ bool key_equal(const uint32_t (&a)[DW], const uint32_t (&b)[DW])
{
    bool eq = true;
#pragma unroll
    for (int j = 0; j < DW; ++j) eq = eq && (a[j] == b[j]);
    return eq;
}

// exclusive scan of one int per thread over a 256-thread workgroup
__device__ __forceinline__ int block_exclusive_scan(int val, int *total)
{
    __shared__ int wave_sums[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = val;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        int s = wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - val;
}

__global__ void iota_kernel(uint32_t *__restrict__ out, int n)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

// ----------------------------------------------------------------------------
// read_back: a few ints from device memory to the host WITHOUT synchronising the stream.  A one-wave kernel stores them
// into the lattice's mailbox (coherent pinned host memory) and then, release-ordered at system scope, a sequence number;
// the host spins on that word.  A hipMemcpyAsync + hipStreamSynchronize pair costs a copy submission and a wake-up
// (~20-30 us each on this platform; a build has three to four of them); the spin sees the value a few microseconds after
// the kernel wrote it.  If the word does not arrive within a few milliseconds (a long queue ahead of the build on this
// stream) the host falls back to waiting for the stream.
__global__ void mailbox_kernel(const int *__restrict__ src, int count, int *mail, int seq)
{
    const int t = threadIdx.x;
    if (t < count) __hip_atomic_store(mail + 1 + t, src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int read_back(plx_lattice *L, const int *d_src, int count, int *h_dst, hipStream_t stream)
{
    if (count < 1 || count > 62) { set_error("read_back: %d values", count); return PLX_ERR_INVALID; }
    // a capturing stream only records the mailbox kernel: the host would spin for a word that never arrives and then
    // invalidate the capture by synchronising.  Everything that reads back (builds, the first MVM's tables) has to run
    // before the capture starts: plx_prepare.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("this call has to build lattice tables (a host read-back) and the stream is being captured: call "
                  "plx_prepare(lat, vd, stream) before the capture");
        return PLX_ERR_STATE;
    }
    const int seq = ++L->mail_seq;
    mailbox_kernel<<<1, 64, 0, stream>>>(d_src, count, L->h_mail, seq);
    PLX_HIP_TRY(hipGetLastError());
    volatile int *mail = L->h_mail;
    if (!g_readback_spin) PLX_HIP_TRY(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (__atomic_load_n(const_cast<int *>(mail), __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 1023) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3) {
            PLX_HIP_TRY(hipStreamSynchronize(stream));
            break;
        }
    }
    if (__atomic_load_n(const_cast<int *>(mail), __ATOMIC_ACQUIRE) != seq) {
        set_error("read_back: the mailbox word did not arrive");
        return PLX_ERR_HIP;
    }
    for (int i = 0; i < count; ++i) h_dst[i] = mail[1 + i];
    return PLX_OK;
}

// h:398-402, same association order as the reference expression
template <int D>
__device__ __forceinline__ void elevate(const float (&pos)[D], const ScaleArgs &sf, float (&el)[D + 1])
{
    el[D] = (float)(-D) * pos[D - 1] * sf.v[D - 1];
#pragma unroll
    for (int i = D - 1; i > 0; --i)
        el[i] = (el[i + 1] - (float)i * pos[i - 1] * sf.v[i - 1] + (float)(i + 2) * pos[i] * sf.v[i]);
    el[0] = el[1] + 2.0f * pos[0] * sf.v[0];
}

// ----------------------------------------------------------------------------
// order: sort key of a point = (shard, rounded lattice coordinates bit-interleaved along their Z-curve, or
// lexicographically).  The keys are COMPACT: a first pass finds the range of every rounded coordinate, the host reads
// the 2 (d+1) numbers back through the mailbox (read_back: no stream synchronisation) and gives every coordinate exactly
// the bits its range needs, so the radix sort runs over the significant bits only (N = 1e6, d = 8, l = 1: 36 bits
// instead of 63).  Outliers cost bits, never correctness: past 62 key bits the widest coordinates lose low bits.

constexpr int kMaxOrderCoords = 16;
struct OrderArgs {
    int n_shards, ncoord;            // ncoord = min(d+1, 16) leading coordinates are used
    int zcurve;                      // g_order_zcurve
    int maxbits;                     // widest coordinate
    long long base, extra;           // shard layout: first `extra` shards have base+1 rows
    int lo[kMaxOrderCoords], bits[kMaxOrderCoords], drop[kMaxOrderCoords];   // per coordinate: smallest value, key bits, low bits dropped
};

// rounded lattice coordinates of a point (the cell of its nearest zero-colour vertex, in units of d+1)
template <int D>
__device__ __forceinline__ void order_coords(const float *__restrict__ x, int p, const ScaleArgs &sf, int zcurve, int (&q)[D + 1])
{
    constexpr int D1 = D + 1;
    float pos[D], el[D1];
#pragma unroll
    for (int i = 0; i < D; ++i) pos[i] = x[(size_t)p * D + i];
    elevate<D>(pos, sf, el);
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        float c = rintf(el[i] * (1.0f / (float)D1));
        c = fminf(fmaxf(c, -1.0e6f), 1.0e6f);                // NaN -> -1e6 (fmaxf), rejected later by embed
        q[i] = (int)c;
    }
    if (zcurve == 2) {
        // the vertices' own curve (renumber_vertices): blur-axis coordinates a_i = q_d - q_i, i < d.  A point and the
        // vertices of its simplex differ by at most one step in every a_i, so points and vertices that meet in splat and
        // slice are close on the same curve.  Measured against mode 1 (N = 1e6, d = 8): +-3 % per MVM (l = 1.0 93.7 vs
        // 91.4 us, l = 0.5 299 vs 309, CG iteration 34.8 vs 35.0 ms): not the default.
#pragma unroll
        for (int i = 0; i < D; ++i) q[i] = q[D] - q[i];
    }
}

// range[2c] = max q_c, range[2c+1] = max -q_c over all points (range[] preset to a very negative number)
template <int D>
__global__ __launch_bounds__(kBlock) void coord_range_kernel(const float *__restrict__ x, int n, ScaleArgs sf, int zcurve,
                                                             int ncoord, int *__restrict__ range, int stride)
{
    constexpr int D1 = D + 1;
    __shared__ int red[kBlock / 64][2 * kMaxOrderCoords];
    // stride > 1: the range of every stride-th point.  The order keys clamp a coordinate to its range (sortkey_kernel), so a
    // point beyond the sampled extremes sorts with the outermost cell: the order only places points in memory, and the
    // extreme cells of a cloud hold a handful of them.  (The full pass read all of x a second time: 23 us at N = 1e6.)
    const long long pl = ((long long)blockIdx.x * kBlock + threadIdx.x) * stride;
    const int p = (int)(pl < n ? pl : n - 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int q[D1];
    order_coords<D>(x, min(p, n - 1), sf, zcurve, q);          // a padding thread repeats the last point: no effect on the range
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        if (i < kMaxOrderCoords) {
            int hi = q[i], lo = -q[i];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                hi = max(hi, __shfl_xor(hi, off));
                lo = max(lo, __shfl_xor(lo, off));
            }
            if (lane == 0) { red[wave][2 * i] = hi; red[wave][2 * i + 1] = lo; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * ncoord) {
        int v = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) v = max(v, red[w][threadIdx.x]);
        // (a plain read first: after the first few workgroups hardly any extreme still moves, and 4,000 workgroups
        // adding to the same 18 words one after the other were 50 us of a 0.8 ms build)
        if (v > __hip_atomic_load(&range[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&range[threadIdx.x], v);
    }
}

// bit widths of a compact interleaved key from per-coordinate ranges; returns the number of key bits below `head_bits`
// reserved bits (shard id), at most 62 in all
static int layout_key_bits(const int *range, int ncoord, int head_bits, int *lo, int *bits, int *drop, int *maxbits)
{
    int total = 0;
    for (int c = 0; c < ncoord; ++c) {
        const long long hi = range[2 * c], lw = -(long long)range[2 * c + 1];
        lo[c] = (int)lw;
        long long span = hi - lw;                               // values 0 .. span
        int b = 0;
        while (span >> b) ++b;
        bits[c] = b;
        drop[c] = 0;
        total += b;
    }
    while (total + head_bits > 62) {                           // outliers: the widest coordinate gives up its lowest bit
        int w = 0;
        for (int c = 1; c < ncoord; ++c)
            if (bits[c] > bits[w]) w = c;
        if (bits[w] == 0) break;
        --bits[w]; ++drop[w]; --total;
    }
    *maxbits = 0;
    for (int c = 0; c < ncoord; ++c) *maxbits = std::max(*maxbits, bits[c]);
    return total;
}

template <int D>
__global__ __launch_bounds__(kBlock) void sortkey_kernel(const float *__restrict__ x, int n, ScaleArgs sf,
                                                         OrderArgs oa, unsigned long long *__restrict__ keys,
                                                         uint32_t *__restrict__ iota)
{
    constexpr int D1 = D + 1;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    int q[D1];
    order_coords<D>(x, p, sf, oa.zcurve, q);
    const long long split = (oa.base + 1) * oa.extra;
    const unsigned long long shard =
        (oa.n_shards <= 1) ? 0ull
        : (unsigned long long)((p < split) ? p / (oa.base + 1) : oa.extra + (p - split) / (oa.base > 0 ? oa.base : 1));
    unsigned long long key = shard;
#pragma unroll
    for (int i = 0; i < D1; ++i)
        if (i < kMaxOrderCoords && i < oa.ncoord) {
            const int v = (q[i] - oa.lo[i]) >> oa.drop[i];
            const int top = (1 << oa.bits[i]) - 1;
            q[i] = v < 0 ? 0 : (v > top ? top : v);
        }
    if (oa.zcurve) {
        // Z-order: bit b of every coordinate (that has one) before bit b-1 of any.  Every blur axis changes all d+1
        // coordinates, so under the lexicographic order each neighbour is about a slab of the leading
        // coordinate away (median 13k-320k vertex ids at N = 1e6, d = 8, l = 0.69); along the Z-curve
        // the medians are 6k-16k and the medium-regime blur is 3-5 % faster (tools/ab_order.py).
        // Ordering by the blur-axis coordinates q_i - q_d instead was measured too: no gain.
        for (int b = oa.maxbits - 1; b >= 0; --b)
#pragma unroll
            for (int i = 0; i < D1; ++i)
                if (i < kMaxOrderCoords && i < oa.ncoord && oa.bits[i] > b) key = (key << 1) | (unsigned long long)((q[i] >> b) & 1);
    } else {
#pragma unroll
        for (int i = 0; i < D1; ++i)
            if (i < kMaxOrderCoords && i < oa.ncoord) key = (key << oa.bits[i]) | (unsigned long long)q[i];
    }
    keys[p] = key;
    iota[p] = (uint32_t)p;
}

// ----------------------------------------------------------------------------
// The point record (round 5): what the embedding knows about one point in a few words -- the first d coordinates of its
// (fixed-up) nearest zero-colour vertex as packed int16 (DW words) and the rank of every coordinate as one byte each
// ((d+2)/4... words) -- from which the key of any of its d+1 corners follows with one select and one add per coordinate
// (h:468-471).  It replaced the d+1 packed corner keys per point of rounds 1-4 (144 bytes per point at d = 8 against 32): the
// embedding wrote 172 MB that the insert fetched back (522 MB by the counters) only to hash every key once.
template <int D> struct Rec {
    static constexpr int D1 = D + 1, DW = (D + 1) / 2, WR = (D1 + 3) / 4;
    static constexpr int W = (DW + WR + 3) & ~3;                   // words per record: whole 16-byte vectors
};

template <int D>
__device__ __forceinline__ void rec_load(const uint32_t *__restrict__ prec, size_t p, int (&gr)[D], int (&rk)[D + 1])
{
    using R = Rec<D>;
    uint32_t w[R::W];
#pragma unroll
    for (int j = 0; j < R::W / 4; ++j) {
        const uint4 v = reinterpret_cast<const uint4 *>(prec + p * R::W)[j];
        w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) gr[i] = (int)(int16_t)((w[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);
#pragma unroll
    for (int i = 0; i <= D; ++i) rk[i] = (int)((w[R::DW + (i >> 2)] >> ((i & 3) * 8)) & 0xFFu);
}

// key of corner r (packed int16), h:468-471
template <int D>
__device__ __forceinline__ void rec_key(const int (&gr)[D], const int (&rk)[D + 1], int r, uint32_t (&kw)[(D + 1) / 2])
{
#pragma unroll
    for (int j = 0; j < (D + 1) / 2; ++j) kw[j] = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int c = gr[i] + ((rk[i] <= D - r) ? r : (r - (D + 1)));
        kw[i >> 1] |= ((uint32_t)c & 0xFFFFu) << ((i & 1) * 16);
    }
}

// ----------------------------------------------------------------------------
// embed: h:395-471 for one point per thread, everything in registers

template <int D>
__global__ __launch_bounds__(kBlock) void embed_kernel(const float *__restrict__ x,
                                                       const uint32_t *__restrict__ perm, int n, ScaleArgs sf,
                                                       float *__restrict__ ew, int *__restrict__ counters,
                                                       uint32_t *__restrict__ prec, int *__restrict__ vrange)
{
    constexpr int D1 = D + 1;
    __shared__ int vred[2 * kMaxOrderCoords];
    const bool valid = blockIdx.x * kBlock + threadIdx.x < n;
    const int p = valid ? blockIdx.x * kBlock + threadIdx.x : n - 1;      // (a padding thread repeats the last point and stores nothing)

    const size_t row = perm[p];          // original row of the p-th point in lattice order
    float pos[D];
#pragma unroll
    for (int i = 0; i < D; ++i) pos[i] = x[row * D + i];

    float el[D1];
    elevate<D>(pos, sf, el);

    // h:405-423
    constexpr float scale = 1.0f / (float)D1;
    constexpr float limit = 32767.0f - 2.0f * (float)D1;   // room for the fix-up and the canonical offsets
    int gr[D1];
    int sum = 0;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        float v = el[i] * scale;
        float up = ceilf(v) * (float)D1;
        float down = floorf(v) * (float)D1;
        float g = (up - el[i] < el[i] - down) ? up : down;
        if (!(fabsf(g) <= limit)) { bad = true; g = 0.f; }   // also catches NaN / Inf
        gr[i] = (int)g;
        sum += gr[i];
    }
    sum = (int)((float)sum * scale);   // int *= float, h:423

    // h:427-433
    int rk[D1];
#pragma unroll
    for (int i = 0; i < D1; ++i) rk[i] = 0;
    {
        float df[D1];
#pragma unroll
        for (int i = 0; i < D1; ++i) df[i] = el[i] - (float)gr[i];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i + 1; j <= D; ++j) {
                bool lt = df[i] < df[j];
                rk[i] += lt ? 1 : 0;
                rk[j] += lt ? 0 : 1;
            }
    }

    // h:435-457
    if (sum > 0) {
#pragma unroll
        for (int i = 0; i < D1; ++i) {
            bool wrap = rk[i] >= D1 - sum;
            gr[i] -= wrap ? D1 : 0;
            rk[i] += wrap ? (sum - D1) : sum;
        }
    } else if (sum < 0) {
#pragma unroll
        for (int i = 0; i < D1; ++i) {
            bool wrap = rk[i] < -sum;
            gr[i] += wrap ? D1 : 0;
            rk[i] += wrap ? (D1 + sum) : sum;
        }
    }

    // h:460-465.  rank is a permutation of 0..d, so every barycentric cell gets
    // exactly one "+=" and one "-=": bary[k] = delta[rank == d-k] - delta[rank == d+1-k].
    // Select instead of indexing by rank (a runtime index would spill to scratch).
    float sd[D1];   // delta ordered by rank
#pragma unroll
    for (int k = 0; k < D1; ++k) sd[k] = 0.f;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        float delta = (el[i] - (float)gr[i]) * scale;
#pragma unroll
        for (int k = 0; k < D1; ++k) sd[k] = (rk[i] == k) ? delta : sd[k];
    }
    float bary[D1];
#pragma unroll
    for (int k = 1; k <= D; ++k) bary[k] = sd[D - k] - sd[D1 - k];
    bary[0] = sd[D] + (1.0f + (0.0f - sd[0]));

    if (bad) atomicOr(&counters[1], 1);

    // Range of the blur-axis coordinates a_c = (k_d - k_c) / (d+1), c < d, over this point's d+1 vertices: what the Morton
    // renumbering lays its codes out over (renumber_vertices; it used to find it with a pass over the vertex keys and a
    // read-back of its own).  At corner 0 the canonical offsets vanish, a_c(0) = (gr_d - gr_c) / (d+1); corner r adds
    // canonical[r][rank_d] - canonical[r][rank_c] = +(d+1) for the corners with rank_d <= d-r < rank_c, -(d+1) for those with
    // rank_c <= d-r < rank_d (h:364-369): the coordinate runs over [a_c(0), a_c(0) + 1] if rank_c > rank_d, else over
    // [a_c(0) - 1, a_c(0)].  Workgroup maxima through LDS, then one guarded atomicMax per workgroup and bound.
    if (vrange) {
        // thread 0's values seed the workgroup maxima; a thread then adds only what exceeds the word it reads (neighbouring
        // points of the lattice order differ by a step or two, so a handful of LDS atomics per workgroup remain; 256 threads
        // adding to the same 32 words one after the other cost 100 us per build)
        int hi[D < kMaxOrderCoords ? D : kMaxOrderCoords], nlo[D < kMaxOrderCoords ? D : kMaxOrderCoords];
#pragma unroll
        for (int c = 0; c < D; ++c)
            if (c < kMaxOrderCoords) {
                const int a0 = (gr[D] - gr[c]) / D1;                 // exact: all coordinates of a lattice point agree mod d+1
                hi[c] = a0 + (rk[c] > rk[D] ? 1 : 0);
                nlo[c] = -(a0 - (rk[c] < rk[D] ? 1 : 0));
                if (threadIdx.x == 0) { vred[2 * c] = hi[c]; vred[2 * c + 1] = nlo[c]; }
            }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < D; ++c)
            if (c < kMaxOrderCoords) {
                if (hi[c] > vred[2 * c]) atomicMax(&vred[2 * c], hi[c]);
                if (nlo[c] > vred[2 * c + 1]) atomicMax(&vred[2 * c + 1], nlo[c]);
            }
        __syncthreads();
        const int nred = 2 * (D < kMaxOrderCoords ? D : kMaxOrderCoords);
        if ((int)threadIdx.x < nred) {
            const int v = vred[threadIdx.x];
            if (v > __hip_atomic_load(&vrange[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&vrange[threadIdx.x], v);
        }
    }
    if (!valid) return;

    // the point record: greedy (first d coordinates) + the final ranks, one byte per coordinate
    if (prec) {
        using R = Rec<D>;
        uint32_t w[R::W];
#pragma unroll
        for (int j = 0; j < R::W; ++j) w[j] = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) w[i >> 1] |= ((uint32_t)gr[i] & 0xFFFFu) << ((i & 1) * 16);
#pragma unroll
        for (int i = 0; i < D1; ++i) w[R::DW + (i >> 2)] |= ((uint32_t)rk[i] & 0xFFu) << ((i & 3) * 8);
#pragma unroll
        for (int j = 0; j < R::W / 4; ++j)
            reinterpret_cast<uint4 *>(prec + (size_t)p * R::W)[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
    }

    // (the corner keys of h:468-471 -- greedy + canonical[r][rank] -- are rebuilt from the record by whoever needs one: rec_key)
#pragma unroll
    for (int r = 0; r < D1; ++r) ew[(size_t)r * n + p] = bary[r];
}

// ----------------------------------------------------------------------------
// insert.  Slot value = smallest reference entry index e = p*(d+1)+r among the corners that share the slot's key.
// Invariant: every value ever stored in a slot belongs to a corner with the same key, so a stale plain load can only cost
// an extra atomic, never a wrong match.  First-touch marks (flag_own): a slot's value only ever decreases, and every
// value that is replaced is replaced by exactly one atomicMin, whose return value names it.  "own" = this corner put its
// index into the slot (claimed it empty, or lowered it: bit 31 of eslot); "displaced" = a later atomicMin replaced it (disp).
// The first touch of a vertex is the one corner that owns and was never displaced: own & ~displaced, without reading the
// table again (flag_own_kernel).
// (Rounds 1-5 also carried a one-corner-per-thread form over packed corner keys (144 bytes per point at d = 8): bound by the
// latency of its dependent accesses (table word -> owner's key -> atomic -> atomic) times the waves the chip holds, 15 %
// slower than the form below on every lattice measured; removed in round 6.  So was a workgroup form
// that de-duplicated a block's corners in LDS first: 203 -> 194 us at N = 1e6, 428 -> 591 us in the fine regime.)

// One thread per POINT (round 5), keys from the point records, a fast path and a parked slow path.
// A one-corner-per-thread kernel is bound by the latency of its dependent accesses (table word -> owner's key ->
// atomic -> atomic) times the waves the chip holds: a wave lives until its SLOWEST lane is through, and with 64 lanes there
// nearly always is one that has to claim an empty slot or lower an index (two or three dependent atomics, ~10 us), while on
// a lattice whose corners share vertices 90 % of the lanes only need to see that their key is already in the table under a
// smaller index -- two plain loads.  Here a thread rebuilds its point's corner keys in registers (one select + add per
// coordinate; no 144-byte key block per point is written or read), the lanes holding equal keys of the same corner index
// elect a leader by ballots over 10 bits of the key hash (confirmed by comparing the key itself), and insert_chains(d) corners go through the FAST path together: table word, owner's
// record, compare.  Whatever needs an atomic or another probe is PARKED in an LDS queue (leader, its follower lanes, the
// slot reached) and the workgroup works the queue off at the end with dense lanes; a full queue (lattices where every
// corner claims a slot) sends the rest through the same loop on the spot.  Same table, same first-touch marks.
constexpr int insert_chains(int d) { return d <= 10 ? 3 : (d <= 20 ? 2 : 1); }      // (two records of d = 32 are 56 registers)
constexpr int kInsertQueue = 512;

// the probe loop of one corner from slot h on: returns the slot of its key; *own = this corner put its index there
template <int D>
__device__ __forceinline__ uint32_t probe_insert(const uint32_t *__restrict__ prec, uint32_t *__restrict__ table, uint32_t mask,
                                                 uint32_t *__restrict__ disp, const uint32_t (&k)[(D + 1) / 2], uint32_t e,
                                                 uint32_t h, bool *own)
{
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    *own = false;
    for (;;) {
        uint32_t o = table[h];
        if (o == kEmpty) {
            o = atomicCAS(&table[h], kEmpty, e);
            if (o == kEmpty) { *own = true; return h; }      // claimed an empty slot
        }
        if (o == e) return h;
        int ogr[D], ork[D1];
        rec_load<D>(prec, (size_t)(o / D1), ogr, ork);
        uint32_t ko[DW];
        rec_key<D>(ogr, ork, (int)(o % D1), ko);
        if (key_equal<DW>(k, ko)) {
            if (e < o) {
                if (disp) {
                    const uint32_t old = atomicMin(&table[h], e);
                    if (old > e) {
                        *own = true;
                        const uint32_t pd = old / D1, rd = old - pd * D1;
                        atomicOr(&disp[2 * (size_t)pd + (rd >> 5)], 1u << (rd & 31));
                    }
                } else {
                    atomicMin(&table[h], e);
                }
            }
            return h;
        }
        h = (h + 1) & mask;
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void insert_point_kernel(const uint32_t *__restrict__ prec, int n,
                                                              uint32_t *__restrict__ table, HashSel hs,
                                                              uint32_t *__restrict__ eslot, int dedupe,
                                                              uint32_t *__restrict__ disp, int ntiles, int remap)
{
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    constexpr int G = insert_chains(D) < D1 ? insert_chains(D) : D1;
    __shared__ uint32_t q_pbase[kInsertQueue], q_h[kInsertQueue], q_who[kInsertQueue];      // who = corner r | leader lane << 8
    __shared__ unsigned long long q_peers[kInsertQueue];                                    // lanes of the wave holding this key
    __shared__ int q_count;
    // XCD-aware tile order (tile_index; plx_tune insert_xcd bit 0, off): every XCD inserts one contiguous eighth of the
    // lattice-ordered points, so that the ~22 corners sharing a vertex meet its table line in ONE L2.  Measured SLOWER
    // (N = 1e6, l = 1: 183 -> 231 us): eight ranges running side by side claim their shared vertices out of index order
    // and lower them afterwards, and the in-order sweep loses the "smaller index already in place" fast path it lives on.
    // The id lookup (bit 1, on) has no such order to lose: 39 -> 34 us.
    const int tile = tile_index(ntiles, remap);
    if (tile < 0) return;
    if (threadIdx.x == 0) q_count = 0;
    __syncthreads();
    const int p = tile * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = p < n;
    int gr[D], rk[D1];
    rec_load<D>(prec, (size_t)(valid ? p : 0), gr, rk);
    const uint32_t mask = hs.mask;
    int q_full = 0;                                     // this thread has seen the queue full: no more LDS atomics on it

    for (int r0 = 0; r0 < D1; r0 += G) {
        uint32_t k[G][DW], h[G], e[G], o[G];
        unsigned long long peers[G];
        int leader[G];
        bool pend[G], own[G], parked[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = r0 + g;
            const bool live = valid && r < D1;
            rec_key<D>(gr, rk, r < D1 ? r : 0, k[g]);
            const uint32_t hk = key_hash<D>(k[g], hs);
            e[g] = (uint32_t)p * D1 + r;
            leader[g] = lane;
            peers[g] = 1ull << lane;
            if (dedupe) {
                // the lanes of the wave with this key, the lowest of them its leader: candidates by 10 hash bits, confirmed
                // on the key itself (a lane whose candidate group holds another key probes for itself)
                unsigned long long grp = __ballot(live);
#pragma unroll
                for (int b = 0; b < 10; ++b) {
                    const bool bit = (hk >> (22 + b)) & 1u;
                    const unsigned long long mb = __ballot(bit);
                    grp &= bit ? mb : ~mb;
                }
                const int cand = live ? __ffsll((long long)grp) - 1 : lane;
                bool same = true;
#pragma unroll
                for (int j = 0; j < DW; ++j) same = same && (__shfl(k[g][j], cand) == k[g][j]);
                leader[g] = same ? cand : lane;
                // the followers of a leader are the lanes of its group that confirmed it
                const unsigned long long conf = __ballot(live && same);
                peers[g] = same ? (grp & conf) : (1ull << lane);
            }
            h[g] = hash_slot(hk, hs);
            pend[g] = live && leader[g] == lane;
            own[g] = false;
            parked[g] = false;
        }
        // fast path: the slot's word, its owner's record, one compare
#pragma unroll
        for (int g = 0; g < G; ++g) o[g] = pend[g] ? table[h[g]] : 0u;
        int ogr[G][D], ork[G][D1];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (pend[g] && o[g] != kEmpty) rec_load<D>(prec, (size_t)(o[g] / D1), ogr[g], ork[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!pend[g]) continue;
            bool done = false;
            if (o[g] != kEmpty) {
                uint32_t ko[DW];
                rec_key<D>(ogr[g], ork[g], (int)(o[g] % D1), ko);
                if (key_equal<DW>(k[g], ko)) {
                    if (o[g] <= e[g]) {
                        done = true;                                     // the key is there under a smaller index: nothing to do
                    } else if (!disp) {
                        // ... under a LARGER one (a later point's workgroup got here first): lower it.  Without the own /
                        // displaced marks nobody needs the value the atomic returns, so the wave does not wait for it
                        atomicMin(&table[h[g]], e[g]);
                        done = true;
                    }
                }
            }
            pend[g] = false;
            if (!done) {
                const int pos = (q_full == 0) ? atomicAdd(&q_count, 1) : kInsertQueue;
                if (pos < kInsertQueue) {
                    q_pbase[pos] = (uint32_t)(p - lane); q_h[pos] = h[g]; q_who[pos] = (uint32_t)(r0 + g) | ((uint32_t)lane << 8);
                    q_peers[pos] = peers[g];
                    parked[g] = true;
                } else {
                    q_full = 1;
                    pend[g] = true;                                     // queue full: this corner probes here
                }
            }
        }
        // the corners that found the queue full (lattices where every corner claims a slot): their probe loops advance
        // together, table loads back to back, then the owners' records
        for (;;) {
            bool any = false;
#pragma unroll
            for (int g = 0; g < G; ++g) any = any || pend[g];
            if (!any) break;
#pragma unroll
            for (int g = 0; g < G; ++g) o[g] = pend[g] ? table[h[g]] : 0u;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (pend[g] && o[g] == kEmpty) {
                    o[g] = atomicCAS(&table[h[g]], kEmpty, e[g]);
                    if (o[g] == kEmpty) { own[g] = true; pend[g] = false; }      // claimed an empty slot
                }
                if (pend[g] && o[g] == e[g]) pend[g] = false;
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (pend[g]) rec_load<D>(prec, (size_t)(o[g] / D1), ogr[g], ork[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!pend[g]) continue;
                uint32_t ko[DW];
                rec_key<D>(ogr[g], ork[g], (int)(o[g] % D1), ko);
                if (key_equal<DW>(k[g], ko)) {
                    if (e[g] < o[g]) {
                        if (disp) {
                            const uint32_t old = atomicMin(&table[h[g]], e[g]);
                            if (old > e[g]) {
                                own[g] = true;
                                const uint32_t pd = old / D1, rd = old - pd * D1;
                                atomicOr(&disp[2 * (size_t)pd + (rd >> 5)], 1u << (rd & 31));
                            }
                        } else {
                            atomicMin(&table[h[g]], e[g]);
                        }
                    }
                    pend[g] = false;
                } else {
                    h[g] = (h[g] + 1) & mask;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = r0 + g;
            uint32_t hh = h[g];
            int pk = parked[g] ? 1 : 0;
            if (dedupe) { hh = __shfl(hh, leader[g]); pk = __shfl(pk, leader[g]); }
            if (valid && r < D1 && !pk) eslot[(size_t)r * n + p] = (disp && own[g]) ? (hh | 0x80000000u) : hh;
        }
    }
    __syncthreads();
    const int nq = min(q_count, kInsertQueue);
    for (int q = threadIdx.x; q < nq; q += kBlock) {
        const uint32_t pbase = q_pbase[q], who = q_who[q];
        const int r = (int)(who & 0xFFu), ll = (int)(who >> 8);
        const uint32_t pl = pbase + (uint32_t)ll;
        int lgr[D], lrk[D1];
        rec_load<D>(prec, (size_t)pl, lgr, lrk);
        uint32_t kq[DW];
        rec_key<D>(lgr, lrk, r, kq);
        bool ownq;
        const uint32_t hq = probe_insert<D>(prec, table, mask, disp, kq, pl * D1 + (uint32_t)r, q_h[q], &ownq);
        unsigned long long pm = q_peers[q];
        while (pm) {
            const int fl = __ffsll((long long)pm) - 1;
            pm &= pm - 1;
            eslot[(size_t)r * n + pbase + fl] = (fl == ll && disp && ownq) ? (hq | 0x80000000u) : hq;
        }
    }
}

// ----------------------------------------------------------------------------
// number: first-touch flags per point, counted per workgroup ...

template <int D1>
__global__ __launch_bounds__(kBlock) void flag_kernel(const uint32_t *__restrict__ eslot,
                                                      const uint32_t *__restrict__ table, int n,
                                                      uint32_t *__restrict__ flagmask,
                                                      int *__restrict__ blockcnt)
{
    // d+1 compiled in: all slot loads, then all table gathers, are in flight together (with a runtime trip count every
    // corner waited for its own slot and then for its own table word: 2 (d+1) dependent round trips per thread)
    const int p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t bits = 0, bits_hi = 0;
    if (p < n) {
        uint32_t slot[D1], owner[D1];
#pragma unroll
        for (int r = 0; r < D1; ++r) slot[r] = eslot[(size_t)r * n + p];
#pragma unroll
        for (int r = 0; r < D1; ++r) owner[r] = table[slot[r]];
#pragma unroll
        for (int r = 0; r < D1; ++r) {
            const bool first = owner[r] == (uint32_t)p * D1 + r;
            if (r < 32) bits |= (first ? 1u : 0u) << r;
            else bits_hi |= (first ? 1u : 0u) << (r - 32);
        }
        flagmask[2 * (size_t)p] = bits;
        flagmask[2 * (size_t)p + 1] = bits_hi;
    }
    int total;
    block_exclusive_scan(__popc(bits) + __popc(bits_hi), &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

// flag_own: the same flags from the insert's own marks (bit 31 of eslot = the corner put its index into the slot; disp =
// corners whose index was replaced later): no table gathers.  disp and flagmask may be the same array.
template <int D1>
__global__ __launch_bounds__(kBlock) void flag_own_kernel(const uint32_t *__restrict__ eslot, const uint32_t *disp, int n,
                                                          uint32_t *flagmask, int *__restrict__ blockcnt)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t bits = 0, bits_hi = 0;
    if (p < n) {
        uint32_t slot[D1];
#pragma unroll
        for (int r = 0; r < D1; ++r) slot[r] = eslot[(size_t)r * n + p];
#pragma unroll
        for (int r = 0; r < D1; ++r) {
            if (r < 32) bits |= (slot[r] >> 31) << r;
            else bits_hi |= (slot[r] >> 31) << (r - 32);
        }
        bits &= ~disp[2 * (size_t)p];
        bits_hi &= ~disp[2 * (size_t)p + 1];
        flagmask[2 * (size_t)p] = bits;
        flagmask[2 * (size_t)p + 1] = bits_hi;
    }
    int total;
    block_exclusive_scan(__popc(bits) + __popc(bits_hi), &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

// ... scanned by one 1024-thread workgroup, 16 counts per thread and step (nblocks <= ~16k for n = 4e6: one step; the
// 256-thread, one-count-per-thread form took 16 dependent steps, 10.8 us, for the 3,907 counts of N = 1e6) ...
constexpr int kWideScanT = 1024, kWideScanIpt = 16;
__global__ __launch_bounds__(kWideScanT) void scan_blocks_kernel(int *__restrict__ blockcnt, int nblocks,
                                                                 int *__restrict__ counters)
{
    __shared__ int wsum[kWideScanT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < nblocks; base += kWideScanT * kWideScanIpt) {
        const int i0 = base + (int)threadIdx.x * kWideScanIpt;
        int v[kWideScanIpt], sum = 0;
#pragma unroll
        for (int k = 0; k < kWideScanIpt; ++k) { v[k] = (i0 + k < nblocks) ? blockcnt[i0 + k] : 0; sum += v[k]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry + incl - sum, total = 0;
#pragma unroll
        for (int w = 0; w < kWideScanT / 64; ++w) { const int t = wsum[w]; if (w < wave) before += t; total += t; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kWideScanIpt; ++k) {
            if (i0 + k < nblocks) blockcnt[i0 + k] = before;
            before += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) counters[0] = carry;   // m
}

// ... and turned into vertex ids: id order == (p, r) order == reference first touch.
template <int D>
__global__ __launch_bounds__(kBlock) void assign_kernel(const uint32_t *__restrict__ flagmask,
                                                        const int *__restrict__ blockoff,
                                                        const uint32_t *__restrict__ eslot,
                                                        const uint32_t *__restrict__ prec, int n,
                                                        uint32_t *__restrict__ table,
                                                        uint32_t *__restrict__ vkeys, uint32_t *__restrict__ vslot, int fp_on,
                                                        int *__restrict__ evid, uint32_t *__restrict__ vs0,
                                                        uint32_t *__restrict__ vowner)
{
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    // The ids a workgroup hands out are one contiguous range, but a lane owns ~d+1 consecutive ones: storing the
    // id-indexed words straight from the lanes leaves a wave at a stride of d+1 ids -- one 32-byte sector per 4-byte store
    // (1.43 GB written for 0.29 GB of payload at m = 8.9e6).  The 4-byte arrays (slot, pre-mix hash, owner) are staged in
    // LDS at their rank in the workgroup and leave as coalesced runs; the 16-byte keys go out from the lanes (a sector pair
    // each: half wasted, not seven eighths).  STAGE: the workgroup's 256 (d+1) words x 3 fit LDS comfortably up to d = 16.
    constexpr bool STAGE = D1 <= 17;
    constexpr int CAP = STAGE ? kBlock * D1 : 1;
    __shared__ uint32_t st_slot[CAP], st_s0[CAP], st_own[CAP];
    const int p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t bits = 0, bits_hi = 0;
    if (p < n) { bits = flagmask[2 * (size_t)p]; bits_hi = flagmask[2 * (size_t)p + 1]; }
    int total;
    const int base = blockoff[blockIdx.x];
    int id = base + block_exclusive_scan(__popc(bits) + __popc(bits_hi), &total);
    if (p < n && (bits | bits_hi) != 0) {
        int gr[D], rk[D1];
        rec_load<D>(prec, (size_t)p, gr, rk);
#pragma unroll
        for (int r = 0; r < D1; ++r) {
            bool first = (r < 32) ? ((bits >> r) & 1u) : ((bits_hi >> (r - 32)) & 1u);
            if (first) {
                const size_t idx = (size_t)r * n + p;
                const uint32_t slot = eslot[idx] & 0x7FFFFFFFu;
                uint32_t k[DW];
                rec_key<D>(gr, rk, r, k);
                table[slot] = table_word<D>((uint32_t)id, k, fp_on);
                store_key<DW>(vkeys, (size_t)id, k);
                if (evid) evid[idx] = id;            // the numbering is final (no renumbering follows): assign_evid
                if (STAGE) {
                    st_slot[id - base] = slot;
                    if (vs0) st_s0[id - base] = lin_hash_packed<D>(k);
                    if (vowner) st_own[id - base] = (uint32_t)p * D1 + r;
                } else {
                    vslot[id] = slot;
                    if (vs0) vs0[id] = lin_hash_packed<D>(k);
                    if (vowner) vowner[id] = (uint32_t)p * D1 + r;
                }
                ++id;
            }
        }
    }
    if (STAGE) {
        __syncthreads();
        for (int j = threadIdx.x; j < total; j += kBlock) {
            vslot[base + j] = st_slot[j];
            if (vs0) vs0[base + j] = st_s0[j];
            if (vowner) vowner[base + j] = st_own[j];
        }
    }
}

__global__ __launch_bounds__(kBlock) void ids_kernel(const uint32_t *__restrict__ eslot,
                                                     const uint32_t *__restrict__ table, int n,
                                                     int *__restrict__ evid, uint32_t idmask, int ntiles, int remap)
{
    const int tile = tile_index(ntiles, remap);      // (XCD-aware: the table lines of a vertex are read by one L2)
    if (tile < 0) return;
    const int p = tile * kBlock + threadIdx.x;
    if (p >= n) return;
    const size_t idx = (size_t)blockIdx.y * n + p;
    evid[idx] = (int)(table[eslot[idx] & 0x7FFFFFFFu] & idmask);
}

// assign_evid: the first-touch corners already have their ids (assign_kernel); the others look theirs up.  On the lattices
// that keep first-touch numbering (m >= 0.9 nnz) that is 1 % of the corners instead of 9e6 random reads of the table.
__global__ __launch_bounds__(kBlock) void ids_rest_kernel(const uint32_t *__restrict__ eslot, const uint32_t *__restrict__ flagmask,
                                                          const uint32_t *__restrict__ table, int n, int d1,
                                                          int *__restrict__ evid, uint32_t idmask)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t lo = flagmask[2 * (size_t)p], hi = flagmask[2 * (size_t)p + 1];
    for (int r = 0; r < d1; ++r) {
        const bool first = (r < 32) ? ((lo >> r) & 1u) : ((hi >> (r - 32)) & 1u);
        if (!first) {
            const size_t idx = (size_t)r * n + p;
            evid[idx] = (int)(table[eslot[idx] & 0x7FFFFFFFu] & idmask);
        }
    }
}

// Neighbours known from the embedding.  The corners of one point's simplex are consecutive blur neighbours: going from
// corner r to corner r+1 (mod d+1) adds 1 to every key coordinate except the one whose rank is d-r, which loses d
// (h:364-369) -- which is the blur step nid = -1 along that coordinate (h:541-542).  So a vertex that is corner r of point
// p has corner r-1 as its +1 neighbour along the coordinate of rank d-r+1 (rank 0 for r = 0) and corner r+1 as its -1
// neighbour along the coordinate of rank d-r.  Every vertex has exactly one first-touch corner (vowner); one thread per
// vertex writes the vertex's WHOLE row of the table -- these two ids, -1 everywhere else -- as coalesced full-line stores,
// and records the +1 axis in vaxis[v] (bits 0-5; 0xFF: none), which tells the sliced lookups not to search it.  On the lattices where almost every
// corner owns its vertex these are nearly ALL the neighbours that exist (11 % of the slots at l = 0.25), i.e. nearly all the
// lookups that would have to touch the table and a key; what is left to the hash table are proofs of absence.  (First
// version: a memset of the table, then one thread per POINT scattering the two ids of each first-touch corner: 27 M 4-byte
// stores that each became a 32-byte write, 699 MB for 81 MB of payload, 420 us on top of the 100 us memset.)
template <int D>
__global__ __launch_bounds__(kBlock) void nbr_rows_init_kernel(const uint32_t *__restrict__ vowner, const int *__restrict__ evid,
                                                               const uint32_t *__restrict__ prec, int n, int m, int64_t mstride,
                                                               int order, int *__restrict__ nbr, uint8_t *__restrict__ vaxis)
{
    constexpr int D1 = D + 1;
    constexpr int WR = (D1 + 3) / 4;
    using R = Rec<D>;
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= mstride) return;
    int axis_plus = 255, axis_minus = 255, u_plus = -1, u_minus = -1;
    if (v < m) {
        const uint32_t e = vowner[v];
        const uint32_t p = e / D1;
        const int r = (int)(e - p * D1);
        uint32_t rw[WR];
#pragma unroll
        for (int j = 0; j < WR; ++j) rw[j] = prec[(size_t)p * R::W + R::DW + j];
        const int want_plus = (r == 0) ? 0 : D - r + 1, want_minus = D - r;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
            const int rk = (int)((rw[i >> 2] >> ((i & 3) * 8)) & 0xFFu);
            axis_plus = (rk == want_plus) ? i : axis_plus;
            axis_minus = (rk == want_minus) ? i : axis_minus;
        }
        u_plus = evid[(size_t)(r == 0 ? D : r - 1) * n + p];
        u_minus = evid[(size_t)(r == D ? 0 : r + 1) * n + p];
        // The mirror of the +1 entry -- "v is u_plus's -1 neighbour along this axis" -- is what u_plus's own thread writes
        // when u_plus has its first touch in this same simplex.  Otherwise bit 7 asks the lookup kernel to store it (it
        // skips this lookup, so nobody else would).  The mirror of the -1 entry needs nothing: if u_minus has its first
        // touch elsewhere, its +1 lookup along this axis is not skipped and finds v through the table.
        const bool mirror = vowner[u_plus] / D1 != p;
        vaxis[v] = (uint8_t)(axis_plus | (mirror ? 0x80 : 0));
    }
#pragma unroll
    for (int axis = 0; axis < D1; ++axis) {
        int *plane = nbr + (size_t)axis * 2 * order * mstride + v;
        for (int sidx = 0; sidx < 2 * order; ++sidx) {
            int val = -1;
            if (sidx == order && axis == axis_plus) val = u_plus;            // tap nid = +1
            if (sidx == order - 1 && axis == axis_minus) val = u_minus;      // tap nid = -1
            plane[(size_t)sidx * mstride] = val;
        }
    }
}

// ----------------------------------------------------------------------------
// neighbours: h:539-545 evaluated once per lattice instead of once per MVM.
// nbr[(axis*2r + s)*mstride + i]; s enumerates nid = -r..-1, 1..r.

// SYMMETRIC = true: only the positive taps are looked up; a hit j = nbr(i, +t) also fills
// nbr(j, -t) = i (the relation is symmetric), the planes having been preset to -1.
// One bit per hash slot: occupied.  On large, sparse lattices most neighbour lookups are for vertices that do not
// exist (88 % at l = 0.25, 43 % at l = 0.69: SURVEY 6.3), and an absent key ends its probe sequence at the first empty
// slot: with the bit tested first that slot is never fetched -- a random 4-byte read of a 134 MB table becomes a read of
// a 4 MB bitmap.  Same probe sequence, same result, bit for bit.  Measured (round 4, N = 1e6, d = 8, tools/ab_fine_r4.py):
// neighbours 3.07 -> 2.92 ms at l = 0.25 (m = 8.9e6), 1.90 -> 1.92 at l = 0.4 (7.2e6), 0.38 -> 0.48 at l = 0.69 (1.7e6: the
// pass that builds the bitmap reads the whole table) -- the lookups are bound by the RATE of random requests the
// fabric serves (~40 G/s), which a bitmap that does not fit the 4 MB L2s next to the kernel's other streams does not
// lower; it is used from m = 2^22 up, where it is at least not slower.
__global__ __launch_bounds__(kBlock) void slotmap_kernel(const uint32_t *__restrict__ table, uint64_t cap,
                                                         unsigned long long *__restrict__ bits)
{
    const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool occ = h < cap && table[h] != kEmpty;
    const unsigned long long b = __ballot(occ);
    if ((threadIdx.x & 63) == 0 && h < cap) bits[h >> 6] = b;
}

// Under the Morton numbering (renumber_vertices) vertex i's id IS its rank among the sorted codes of the blur-axis
// coordinates (a_0 .. a_{d-1}), and a blur step along axis j < d changes a_j alone (by -nid: see vertex_axis_coords).  The
// neighbour's code is therefore vertex i's code with the bits of coordinate j replaced, it lies on the SAME side of i for
// every vertex, and -- the codes being sorted and, when no coordinate had to drop bits, unique -- a binary search in a
// window of the code array next to i decides the lookup whenever the window brackets the target: present (the id is the
// position: no key compare, no table) or PROVEN absent.  Only targets outside the window (and axis d, which moves every
// coordinate) go to the hash table.  Where the bits of (coordinate c, bit b) sit in a code, and each coordinate's
// measured range, come from the layout the renumbering used.
struct NbrCode {
    int nbits[kMaxOrderCoords], lo[kMaxOrderCoords], hi[kMaxOrderCoords];
    unsigned char pos[kMaxOrderCoords][16];
};

template <int D, bool SYMMETRIC>
__global__ __launch_bounds__(kBlock) void neighbor_kernel(const uint32_t *__restrict__ vkeys, int m,
                                                          int64_t mstride, int order,
                                                          const uint32_t *__restrict__ table,
                                                          HashSel hs, uint32_t idmask, int *__restrict__ nbr, int plane_fast,
                                                          const uint32_t *__restrict__ slotmap,
                                                          const unsigned long long *__restrict__ vcode, NbrCode nc, int window)
{
    const uint32_t mask = hs.mask;
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    const int i = (plane_fast ? blockIdx.y : blockIdx.x) * kBlock + threadIdx.x;
    const int axis = plane_fast ? blockIdx.x : blockIdx.y;
    if (i >= m) return;
    uint32_t kw[DW];
    load_key<DW>(vkeys, (size_t)i, kw);
    int key[D];
#pragma unroll
    for (int c = 0; c < D; ++c) key[c] = (int)(int16_t)((kw[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu);
    const bool windowed = vcode != nullptr && axis < D && axis < kMaxOrderCoords;
    unsigned long long ci = 0;
    int aj = 0;
    if (windowed) {
        ci = vcode[i];
        int kd = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) kd -= key[c];
        int ka = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) ka = (c == axis) ? key[c] : ka;
        aj = (kd - ka) / D1;
    }

    int *plane = nbr + (size_t)axis * 2 * order * mstride;
    for (int s = SYMMETRIC ? order : 0; s < 2 * order; ++s) {
        const int nid = (s < order) ? (s - order) : (s - order + 1);
        if (windowed) {
            const int ajn = aj - nid;
            int found_w = -1;
            bool resolved = false;
            if (ajn < nc.lo[axis] || ajn > nc.hi[axis]) {
                resolved = true;                                   // no vertex has that coordinate value
            } else {
                unsigned long long T = ci;
                const int q = ajn - nc.lo[axis];
                for (int b = 0; b < nc.nbits[axis]; ++b) {
                    const int pb = nc.pos[axis][b];
                    T = (T & ~(1ull << pb)) | ((unsigned long long)((q >> b) & 1) << pb);
                }
                const int lo_i = T < ci ? max(0, i - window) : i + 1;
                const int hi_i = T < ci ? i : min(m, i + 1 + window);
                if (lo_i < hi_i && vcode[lo_i] <= T && T <= vcode[hi_i - 1]) {
                    int a = lo_i, e = hi_i;
                    while (a < e) {
                        const int mid = (a + e) >> 1;
                        if (vcode[mid] < T) a = mid + 1; else e = mid;
                    }
                    found_w = (a < hi_i && vcode[a] == T) ? a : -1;
                    resolved = true;
                }
            }
            if (resolved) {
                plane[(size_t)s * mstride + i] = found_w;
                if (SYMMETRIC && found_w >= 0) plane[(size_t)(order - nid) * mstride + found_w] = i;
                continue;
            }
        }
        uint32_t nk[DW];
#pragma unroll
        for (int j = 0; j < DW; ++j) nk[j] = 0;
        bool in_range = true;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            // neighbour[k] = key[k] - nid, neighbour[axis] = key[axis] + nid*d  (h:541-542)
            int v = key[c] - nid + ((c == axis) ? nid * D1 : 0);
            in_range = in_range && (v >= -32768) && (v <= 32767);
            nk[c >> 1] |= ((uint32_t)v & 0xFFFFu) << ((c & 1) * 16);
        }
        int found = -1;
        if (in_range) {
            uint32_t h = hash_slot(key_hash<D>(nk, hs), hs);
            for (;;) {
                if (slotmap && !((slotmap[h >> 5] >> (h & 31)) & 1u)) break;
                uint32_t v = table[h];
                if (v == kEmpty) break;
                v &= idmask;
                uint32_t kv[DW];
                load_key<DW>(vkeys, (size_t)v, kv);
                if (key_equal<DW>(nk, kv)) { found = (int)v; break; }
                h = (h + 1) & mask;
            }
        }
        plane[(size_t)s * mstride + i] = found;
        if (SYMMETRIC && found >= 0) plane[(size_t)(order - nid) * mstride + found] = i;   // slot of tap -nid
    }
}


// ----------------------------------------------------------------------------
// XCD-sliced neighbour lookups (round 5; hash_v = 2).  Where most neighbours do not exist (88 % at l = 0.25) a lookup is a
// proof of absence, and every random 4-byte read of the 134 MB table, of its 4 MB occupancy bitmap or of a 16-byte key costs
// a 128-byte line fill: neighbor_kernel<8, true> moved 15.6 GB to write a 643 MB table (profiles/r04_summary.md), at the
// line-fill bandwidth of the fabric.  Here the probing runs on a MAP of four bits per slot -- 0 = empty, else 1 +
// fingerprint % 15 -- and every lookup is served by the XCD that owns the slot's eighth of that map: workgroup b serves
// slice b % 8 of vertex tile b / 8 (workgroups b and b + 8 share an XCD, MI355X_MICROARCH.md "Workgroup dispatch"), so an
// XCD only ever reads 1/8 of the map (2 MB at 2^25 slots), which stays in its 4 MB L2.  The price is that all 8 XCDs look
// at every (vertex, axis): that has to cost a handful of instructions, which the linear hash provides -- the neighbour's
// pre-mix hash is the vertex's own plus a constant per (axis, tap), and the owning slice is the top 3 bits of one product.
// The table and a key are touched only when a nibble matches the lookup's own (the neighbour exists, or 1 in 15 of the
// occupied slots met on the way).  Placement is a speed assumption only: each (tile, slice) pair is one workgroup wherever
// it runs.  Positive taps only, hits mirrored (the SYMMETRIC form of neighbor_kernel); same table, bit for bit.
__global__ __launch_bounds__(kBlock) void nibmap_kernel(const uint32_t *__restrict__ table, uint64_t nwords, int fp_on,
                                                        uint32_t *__restrict__ nib)
{
    const uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (w >= nwords) return;
    const uint4 a = reinterpret_cast<const uint4 *>(table)[2 * w], b = reinterpret_cast<const uint4 *>(table)[2 * w + 1];
    const uint32_t t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t nb = t[j] == kEmpty ? 0u : (fp_on ? fp_nibble(t[j] >> kFpShift) : 1u);
        out |= nb << (4 * j);
    }
    nib[w] = out;
}

struct NbrDelta { uint32_t d[PLX_MAX_DIM + 1]; };       // s(neighbour along axis, nid = +1) - s(vertex)

constexpr int kSlicedQueue = 1024;  // nibble matches a workgroup can park in LDS before it resolves them
// consecutive vertices per thread of neighbor_sliced_kernel: their d+1 ownership bits share one 64-bit mask
constexpr int sliced_vpt(int d1) { return d1 <= 16 ? 4 : (d1 <= 32 ? 2 : 1); }

// vs0[v] = the pre-mix hash s(key of v), stored by the numbering passes: a thread starts from 4 bytes per vertex instead of
// unpacking a 16-byte key and eight multiplies -- and it does so once per XCD.  A thread takes V consecutive vertices and
// decides with an add, a shift and a compare per (vertex, axis) which lookups its XCD owns (the top 3 bits of s + delta),
// collecting them in ONE 64-bit mask; it then pops them two at a time, so that a map load has most lanes active and two
// are in flight per lane.  A lookup walks the nibbles from its home slot: an empty one is the proof of absence (98 % end
// there without touching anything but the map); a nibble equal to its own fingerprint's is PARKED in an LDS queue, and the
// workgroup resolves the parked ones together at the end -- table word, fingerprint byte, key compare, dense lanes.
// History (m = 8.9e6, l = 0.25): one vertex per thread, key hashed by all 8 XCDs, d+1 predicated map loads: 1.15 ms at 40 %
// issue utilisation; four vertices per thread with matches resolved where they occur: 1.21 ms (two dependent HBM round
// trips in most loop iterations of every wave); + the parking queue: 0.97 ms, now bound by instruction issue (2,900 VALU
// instructions per wave: a multiply per ownership test, per-vertex 64-bit masks with select chains in the pop).
template <int D>
__device__ __forceinline__ int sliced_resolve(const uint32_t *__restrict__ vkeys, const uint32_t *__restrict__ table,
                                              const uint32_t *__restrict__ nib, uint32_t mask, uint32_t idmask, int fp_on, int i,
                                              int axis, int t, uint32_t sn, uint32_t h)
{
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    uint32_t kw[DW];
    load_key<DW>(vkeys, (size_t)i, kw);
    uint32_t nk[DW];
#pragma unroll
    for (int j = 0; j < DW; ++j) nk[j] = 0;
    bool in_range = true;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const int kc = (int)(int16_t)((kw[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu);
        const int v = kc - t + ((c == axis) ? t * D1 : 0);        // h:541-542
        in_range = in_range && v >= -32768 && v <= 32767;
        nk[c >> 1] |= ((uint32_t)v & 0xFFFFu) << ((c & 1) * 16);
    }
    if (!in_range) return -1;
    const uint32_t fpn = lin_fp(sn), mynib = fp_nibble(fpn);
    uint32_t w = nib[h >> 3];
    for (;;) {
        const uint32_t nb = (w >> ((h & 7u) * 4)) & 15u;
        if (nb == 0) return -1;
        if (!fp_on || nb == mynib) {
            const uint32_t tw = table[h];
            if (!fp_on || (tw >> kFpShift) == fpn) {
                const uint32_t v = tw & idmask;
                uint32_t kv[DW];
                load_key<DW>(vkeys, (size_t)v, kv);
                if (key_equal<DW>(nk, kv)) return (int)v;
            }
        }
        h = (h + 1) & mask;
        if ((h & 7u) == 0) w = nib[h >> 3];
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void neighbor_sliced_kernel(const uint32_t *__restrict__ vs0, const uint32_t *__restrict__ vkeys,
                                                                 int m, int64_t mstride, int order,
                                                                 const uint32_t *__restrict__ table, HashSel hs, uint32_t idmask,
                                                                 int fp_on, const uint32_t *__restrict__ nib, NbrDelta nd,
                                                                 int *__restrict__ nbr, const uint8_t *__restrict__ vaxis)
{
    constexpr int D1 = D + 1;
    constexpr int V = sliced_vpt(D1);
    constexpr int B = 64 / V;                          // mask bits per vertex (>= d+1)
    __shared__ uint32_t delta[D1];
    __shared__ uint32_t q_sn[kSlicedQueue], q_h[kSlicedQueue];
    __shared__ uint32_t q_who[kSlicedQueue];          // vertex index within the workgroup's tile (12 bits) | axis << 12 | tap << 20
    __shared__ int q_count;
    if (threadIdx.x < D1) delta[threadIdx.x] = nd.d[threadIdx.x];
    if (threadIdx.x == 0) q_count = 0;
    __syncthreads();
    const uint32_t slice = blockIdx.x & 7u;
    const int tile0 = (int)(blockIdx.x >> 3) * kBlock * V;
    const int i0 = tile0 + threadIdx.x * V;
    const bool live = i0 < m;
    uint32_t s0[V], known[V];                           // (vs0 and vaxis are padded to whole vectors)
#pragma unroll
    for (int j = 0; j < V; ++j) { s0[j] = 0; known[j] = 0xFFu; }
    if (live) {
        if constexpr (V == 4) {
            const uint4 s4 = *reinterpret_cast<const uint4 *>(vs0 + i0);
            s0[0] = s4.x; s0[1] = s4.y; s0[2] = s4.z; s0[3] = s4.w;
            if (vaxis) {
                const uint32_t k4 = *reinterpret_cast<const uint32_t *>(vaxis + i0);
#pragma unroll
                for (int j = 0; j < 4; ++j) known[j] = (k4 >> (8 * j)) & 0xFFu;
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) { s0[j] = vs0[i0 + j]; if (vaxis) known[j] = vaxis[i0 + j]; }
        }
    }
    const uint32_t mask = hs.mask;
    const int shift = hs.shift;
    auto write_hit = [&](int i, int axis, int t, int found) {
        const size_t plane = (size_t)axis * 2 * order * mstride;
        nbr[plane + (size_t)(order + t - 1) * mstride + i] = found;
        nbr[plane + (size_t)(order - t) * mstride + found] = i;
    };
    for (int t = 1; t <= order; ++t) {
        // which (vertex, axis) lookups of this tap belong to this XCD's eighth of the map: bit j * B + axis
        unsigned long long own = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            unsigned long long oj = 0;
#pragma unroll
            for (int axis = 0; axis < D1; ++axis)
                oj |= (unsigned long long)(((s0[j] + (uint32_t)t * nd.d[axis]) >> 29) == slice) << axis;
            // tap +1 along axis (known & 63) came from the embedding (nbr_rows_init_kernel); bit 7: its mirror is ours to store
            if (t == 1 && known[j] != 0xFFu) {
                oj &= ~(1ull << (known[j] & 63u));
                if ((known[j] & 0x80u) && slice == 0 && live && i0 + j < m) {
                    const size_t plane = (size_t)(known[j] & 63u) * 2 * order * mstride;
                    const int u = nbr[plane + (size_t)order * mstride + i0 + j];
                    nbr[plane + (size_t)(order - 1) * mstride + u] = i0 + j;
                }
            }
            if (!live || i0 + j >= m) oj = 0;
            own |= oj << (j * B);
        }
        // pop them two at a time (any vertex, any axis); walk the nibbles from the home slot
        while (own) {
            uint32_t sn[2], h[2], w[2], who[2];
            bool valid[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                valid[u] = own != 0;
                const int bit = valid[u] ? __ffsll((long long)own) - 1 : 0;
                own &= own - 1;                               // (0 stays 0)
                const int j = bit / B, axis = bit % B;
                uint32_t sj = s0[0];
#pragma unroll
                for (int q = 1; q < V; ++q) sj = (q == j) ? s0[q] : sj;
                sn[u] = sj + (uint32_t)t * delta[axis];
                h[u] = lin_slotbits(sn[u]) >> shift;
                who[u] = (uint32_t)(threadIdx.x * V + j) | ((uint32_t)axis << 12) | ((uint32_t)t << 20);
            }
            w[0] = nib[h[0] >> 3];
            w[1] = valid[1] ? nib[h[1] >> 3] : 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint32_t hh = h[u], ww = w[u];
                uint32_t nb = (ww >> ((hh & 7u) * 4)) & 15u;
                if (!valid[u] || nb == 0) continue;                      // an empty home slot: the proof of absence
                const uint32_t mynib = fp_nibble(lin_fp(sn[u]));
                for (;;) {
                    if (!fp_on || nb == mynib) {
                        const int pos = atomicAdd(&q_count, 1);
                        if (pos < kSlicedQueue) {
                            q_sn[pos] = sn[u]; q_h[pos] = hh; q_who[pos] = who[u];
                        } else {                                           // queue full: resolve it here
                            const int i = tile0 + (int)(who[u] & 0xFFFu), axis = (int)((who[u] >> 12) & 0xFFu);
                            const int found = sliced_resolve<D>(vkeys, table, nib, mask, idmask, fp_on, i, axis, t, sn[u], hh);
                            if (found >= 0) write_hit(i, axis, t, found);
                        }
                        break;
                    }
                    hh = (hh + 1) & mask;
                    if ((hh & 7u) == 0) ww = nib[hh >> 3];
                    nb = (ww >> ((hh & 7u) * 4)) & 15u;
                    if (nb == 0) break;
                }
            }
        }
    }
    __syncthreads();
    const int nq = min(q_count, kSlicedQueue);
    for (int k = threadIdx.x; k < nq; k += kBlock) {
        const uint32_t who = q_who[k];
        const int i = tile0 + (int)(who & 0xFFFu), axis = (int)((who >> 12) & 0xFFu), t = (int)(who >> 20);
        const int found = sliced_resolve<D>(vkeys, table, nib, mask, idmask, fp_on, i, axis, t, q_sn[k], q_h[k]);
        if (found >= 0) write_hit(i, axis, t, found);
    }
}

// ----------------------------------------------------------------------------
// compaction of the neighbour table for sparse lattices.  A "quad" is 4 consecutive
// vertices (what one blur thread handles); bit j*2r+s of its mask says neighbour s
// of vertex j exists.  Ids are stored densely in (quad, bit) order; a wave of the
// blur kernel finds its ids at cbase[wave] + (prefix of popcounts over its lanes).

// the 4 x taps2 ids of a quad, one 16-byte load per tap plane (i0 is a multiple of 4, mstride of 64), and their mask
__device__ __forceinline__ uint32_t quad_load(const int *__restrict__ nbr_axis, int64_t mstride, int taps2, int i0, int m,
                                              int (&ids)[6][4])
{
    uint32_t mask = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s) {                                  // (compile-time indices: the ids stay in registers)
        if (s < taps2) {
            const int4 v = *reinterpret_cast<const int4 *>(nbr_axis + (size_t)s * mstride + i0);
            ids[s][0] = v.x; ids[s][1] = v.y; ids[s][2] = v.z; ids[s][3] = v.w;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < m && ids[s][j] >= 0) mask |= 1u << (j * taps2 + s);
        }
    }
    return mask;
}

__global__ __launch_bounds__(kBlock) void compact_count_kernel(const int *__restrict__ nbr, int m, int64_t mstride,
                                                               int taps2, int64_t nquads, int64_t nqwaves,
                                                               uint32_t *__restrict__ cmask,
                                                               uint32_t *__restrict__ cbase)
{
    const int axis = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int *nb = nbr + (size_t)axis * taps2 * mstride;
    uint32_t mask = 0;
    if (q < nquads) {
        int ids[6][4];
        mask = quad_load(nb, mstride, taps2, (int)(q * 4), m, ids);
        cmask[(size_t)axis * nquads + q] = mask;
    }
    int cnt = __popc(mask);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    const int64_t w = q >> 6;
    if ((threadIdx.x & 63) == 0 && w < nqwaves) cbase[(size_t)axis * (nqwaves + 1) + w] = (uint32_t)cnt;
}

// exclusive scan of the wave counts of every axis (one 1024-thread workgroup per axis, 16 counts per thread and step: the
// 256-thread, one-count-per-thread form took 103 us for 35 k counts -- 137 dependent steps); totals -> counters[2 + axis]
constexpr int kScanT = 1024, kScanIpt = 16;
__global__ __launch_bounds__(kScanT) void compact_scan_kernel(uint32_t *__restrict__ cbase, int64_t nqwaves,
                                                              int *__restrict__ counters)
{
    __shared__ int wsum[kScanT / 64];
    uint32_t *base = cbase + (size_t)blockIdx.x * (nqwaves + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int64_t b = 0; b < nqwaves; b += (int64_t)kScanT * kScanIpt) {
        const int64_t i0 = b + (int64_t)threadIdx.x * kScanIpt;
        int v[kScanIpt], sum = 0;
#pragma unroll
        for (int k = 0; k < kScanIpt; ++k) { v[k] = (i0 + k < nqwaves) ? (int)base[i0 + k] : 0; sum += v[k]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry + incl - sum, total = 0;
#pragma unroll
        for (int w = 0; w < kScanT / 64; ++w) { const int t = wsum[w]; if (w < wave) before += t; total += t; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanIpt; ++k) {
            if (i0 + k < nqwaves) base[i0 + k] = (uint32_t)before;
            before += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) { base[nqwaves] = (uint32_t)carry; counters[2 + blockIdx.x] = carry; }
}

struct AxisOffsets { long long off[PLX_MAX_DIM + 2]; };

__global__ __launch_bounds__(kBlock) void compact_fill_kernel(const int *__restrict__ nbr, int m, int64_t mstride,
                                                              int taps2, int64_t nquads, int64_t nqwaves,
                                                              const uint32_t *__restrict__ cmask,
                                                              const uint32_t *__restrict__ cbase, AxisOffsets ao,
                                                              int *__restrict__ cids)
{
    const int axis = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int *nb = nbr + (size_t)axis * taps2 * mstride;
    const uint32_t mask = (q < nquads) ? cmask[(size_t)axis * nquads + q] : 0u;
    int incl = __popc(mask);
    const int cnt = incl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (q >= nquads) return;
    const int64_t w = q >> 6;
    int64_t pos = ao.off[axis] + cbase[(size_t)axis * (nqwaves + 1) + w] + (incl - cnt);
    if (mask == 0) return;
    int ids[6][4];
    (void)quad_load(nb, mstride, taps2, (int)(q * 4), m, ids);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 6; ++s)
            if (s < taps2 && (mask & (1u << (j * taps2 + s)))) cids[pos++] = ids[s][j];
}

// ----------------------------------------------------------------------------
// csr: corners of the owned points sorted (stably) by vertex id

__global__ __launch_bounds__(kBlock) void csr_keys_kernel(const int *__restrict__ evid, int n, int own_begin,
                                                          int n_own, uint32_t *__restrict__ keys,
                                                          uint32_t *__restrict__ vals)
{
    const int pl = blockIdx.x * kBlock + threadIdx.x;
    if (pl >= n_own) return;
    const int r = blockIdx.y;
    const size_t src = (size_t)r * n + own_begin + pl;
    const size_t dst = (size_t)r * n_own + pl;   // corner-major: coalesced; inside a vertex row the stable sort keeps (r, p) order
    keys[dst] = (uint32_t)evid[src];
    vals[dst] = (uint32_t)src;
}

__global__ __launch_bounds__(kBlock) void csr_finalize_kernel(const uint32_t *__restrict__ skeys,
                                                              const uint32_t *__restrict__ svals,
                                                              const float *__restrict__ ew, int n,
                                                              int own_begin, int nnz, int m,
                                                              const uint32_t *__restrict__ perm,
                                                              int *__restrict__ csr_pt,
                                                              int *__restrict__ csr_row,
                                                              float *__restrict__ csr_w)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nnz) return;
    const uint32_t v = skeys[k];
    const uint32_t idx = svals[k];
    const uint32_t r = idx / (uint32_t)n;
    const int prev = (k == 0) ? -1 : (int)skeys[k - 1];
    // sign bit of csr_pt marks the first entry of a vertex row (segment head)
    const uint32_t head = (prev != (int)v) ? 0x80000000u : 0u;
    const uint32_t p = idx - r * (uint32_t)n;                  // point, lattice order
    csr_pt[k] = (int)((p - (uint32_t)own_begin) | head);
    csr_row[k] = (int)((perm[p] - (uint32_t)own_begin) | head);   // the same point as the caller numbers its rows
    csr_w[k] = ew[idx];
}

// row_ptr[u] = first k with skeys[k] >= u (rows that no owned point touches are empty).  No kernel on
// the MVM path needs it (row heads are flagged in csr_pt), so it is produced only when exported;
// filling it inside csr_finalize_kernel made the thread behind a gap walk it alone -- 8 ms per build
// for one rank of an 8-rank job, whose own rows touch less than half of the merged vertex set.
__global__ __launch_bounds__(kBlock) void row_ptr_kernel(const uint32_t *__restrict__ skeys, int nnz, int m,
                                                         int *__restrict__ row_ptr)
{
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u > m) return;
    int lo = 0, hi = nnz;                          // lower bound of u in the sorted vertex ids
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skeys[mid] < (uint32_t)u) lo = mid + 1; else hi = mid;
    }
    row_ptr[u] = lo;
}

// ----------------------------------------------------------------------------
// host side

// h:203-219, fp32 throughout
static float taps_variance(const float *c, int R)
{
    float mom0 = 0, mom1 = 0.f, mom2 = 0.f;
    for (int i = 0; i < R; ++i) {
        mom0 += c[i];
        mom1 += i * c[i];
        mom2 += i * i * c[i];
    }
    float mean = mom1 / mom0;
    return mom2 / mom0 - mean * mean;
}

// ----------------------------------------------------------------------------
// Vertex order.  First-touch numbering follows the points, so the vertices of one point block are consecutive -- good for
// splat / slice -- but along a blur axis the neighbour of vertex v is ~15,000 ids away (every axis step changes all d+1
// key coordinates).  In the basis of the blur directions, a_i = (k_d - k_i) / (d+1) for i < d (k_d = -sum k_i), a step along
// axis i < d changes a_i ALONE by +-1 (axis d changes all of them).  Numbering the vertices along the Morton curve of
// (a_0 .. a_{d-1}) puts the axis-i neighbour a median 2^(d-1-i) ids away for i < d and roughly halves the share of
// neighbours beyond any cache-sized window (N = 1e6, d = 8, l = 0.69: farther than 87k ids 32 % -> 16 %, than 262k ids
// 20 % -> 8.5 %; tools/vertex_order_study.py).  The numbering depends on the vertex set only, so every rank of a sharded
// job derives the same one.  Measured (N = 1e6, d = 8, vd = 1, us per MVM, first touch -> Morton): l = 1.0 114 -> 94
// (two-axis blur pass 10.8 -> 6.4), l = 0.69 211 -> 170, l = 0.5 415 -> 305, l = 0.4 438 -> 357, l = 0.35 373 -> 363,
// l = 0.3 307 -> 357, l = 0.25 280 -> 346: where nearly every corner has a vertex of its own (m > 0.9 nnz) the blur has
// few neighbours to find and splat / slice lose the point-order locality of first touch, so those keep first touch.
// Cost: one 64-bit sort of m codes + two passes over the keys, +0.1 ms of 1.25 (l = 1.0), +0.24 of 1.9 (l = 0.69).
// Dispatch order of the per-plane kernels.  1: the d+1 corner planes of one run of 256 points (insert) / one run of 256
// vertices (neighbours) are adjacent workgroups, so the workgroups that meet at a table slot run close together in time:
// a later corner finds the entry, its key and usually the smaller index already in the L2s, and the atomicMin that the
// plane-major order needs for most shared vertices (plane 0 of a later point runs long before plane 5 of an earlier
// one) becomes rare.  N = 1e6, d = 8, l = 1: insert 335 -> 215 us, neighbours 93 -> 84 us (build 1.37 -> 1.25 ms);
// l = 0.69: build 2.15 -> 2.05 ms; l = 0.25: 4.94 -> 4.91 ms.  The id lookup (ids_kernel) is faster plane-major (35 vs
// 41 us) and stays so.  Giving every XCD one contiguous eighth of the (point block, plane) pairs instead: no better
// (1.275 ms).  With only the first corner of every vertex per 256-point block probing (a workgroup-level LDS
// de-duplication would achieve that) the plane-major insert took 219 us, with only the 4e5 first-touch corners 67 us:
// the duplicate lookups are what costs, and time locality removes most of that without the LDS stage.
constexpr int kMortonMinVertices = 65536;

// blur-axis coordinates a_c = (k_d - k_c) / (d+1), c < d, of a vertex (k_d = -sum k_c; exact: all coordinates of a lattice
// point agree mod d+1)
template <int D>
__device__ __forceinline__ void vertex_axis_coords(const uint32_t *__restrict__ vkeys, int v, int (&a)[D])
{
    constexpr int DW = (D + 1) / 2;
    uint32_t kw[DW];
    load_key<DW>(vkeys, (size_t)v, kw);
    int key[D], kd = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        key[c] = (int)(int16_t)((kw[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu);
        kd -= key[c];
    }
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = (kd - key[c]) / (D + 1);
}

// range[2c] = max a_c, range[2c+1] = max -a_c over the vertices (preset to a very negative number); c < min(d, 16)
template <int D>
__global__ __launch_bounds__(kBlock) void vertex_range_kernel(const uint32_t *__restrict__ vkeys, int m, int ncoord,
                                                              int *__restrict__ range)
{
    __shared__ int red[kBlock / 64][2 * kMaxOrderCoords];
    const int v = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int a[D];
    vertex_axis_coords<D>(vkeys, min(v, m - 1), a);
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (c < kMaxOrderCoords) {
            int hi = a[c], lo = -a[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                hi = max(hi, __shfl_xor(hi, off));
                lo = max(lo, __shfl_xor(lo, off));
            }
            if (lane == 0) { red[wave][2 * c] = hi; red[wave][2 * c + 1] = lo; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * ncoord) {
        int r = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) r = max(r, red[w][threadIdx.x]);
        if (r > __hip_atomic_load(&range[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&range[threadIdx.x], r);
    }
}

struct CodeArgs { int ncoord, maxbits; int lo[kMaxOrderCoords], bits[kMaxOrderCoords], drop[kMaxOrderCoords]; };

// Morton code over exactly the bits every coordinate's range needs (coordinates past the 16th do not take part)
template <int D>
__global__ __launch_bounds__(kBlock) void vertex_code_kernel(const uint32_t *__restrict__ vkeys, int m, CodeArgs ca,
                                                             unsigned long long *__restrict__ code, uint32_t *__restrict__ ids)
{
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= m) return;
    int a[D];
    vertex_axis_coords<D>(vkeys, v, a);
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < kMaxOrderCoords) {
            const int q = (a[c] - ca.lo[c]) >> ca.drop[c];
            const int top = (1 << ca.bits[c]) - 1;
            a[c] = q < 0 ? 0 : (q > top ? top : q);
        }
    unsigned long long z = 0;
    for (int b = ca.maxbits - 1; b >= 0; --b)
#pragma unroll
        for (int c = 0; c < D; ++c)
            if (c < kMaxOrderCoords && ca.bits[c] > b) z = (z << 1) | (unsigned long long)((a[c] >> b) & 1);
    code[v] = z;
    ids[v] = (uint32_t)v;
}

// order[new] = old  ->  keys into the new order, and the key -> id table follows
template <int D>
__global__ __launch_bounds__(kBlock) void vertex_permute_kernel(const uint32_t *__restrict__ order, int m,
                                                                const uint32_t *__restrict__ vkeys_old,
                                                                const uint32_t *__restrict__ vslot_old,
                                                                uint32_t *__restrict__ vkeys_new, uint32_t *__restrict__ vslot_new,
                                                                uint32_t *__restrict__ table, int fp_on,
                                                                uint32_t *__restrict__ vs0)
{
    constexpr int DW = (D + 1) / 2;
    const int nw = blockIdx.x * kBlock + threadIdx.x;
    if (nw >= m) return;
    const uint32_t old = order[nw];
    uint32_t k[DW];
    load_key<DW>(vkeys_old, (size_t)old, k);
    store_key<DW>(vkeys_new, (size_t)nw, k);
    if (vs0) vs0[nw] = lin_hash_packed<D>(k);
    const uint32_t slot = vslot_old[old];
    vslot_new[nw] = slot;
    table[slot] = table_word<D>((uint32_t)nw, k, fp_on);
}

// does this build renumber its m vertices along the Morton curve?  (renumber_vertices; asked before the numbering pass too,
// which stores final ids itself when the answer is no)
static bool will_renumber(const plx_lattice *L, int64_t m, int64_t corners)
{
    if (g_vertex_order == 0 || m < 2) return false;
    if (g_vertex_order == 1 && (L->single_use || m < kMortonMinVertices || 10 * m > 9 * corners)) return false;
    return true;
}

template <int D>
static int renumber_vertices(plx_lattice *L, int64_t corners, hipStream_t stream, const int *h_range_known = nullptr)
{
    // corners: how many (point, vertex) incidences stand behind the m vertices -- n (d+1) of the whole job, or the sum of
    // the per-rank vertex counts when only those are known (plx_build_merge); the same number on every rank
    constexpr int DW = (D + 1) / 2;
    const int m = (int)L->m;
    L->vertex_order = 0;
    L->vcode = nullptr;
    if (!will_renumber(L, m, corners)) return PLX_OK;
    L->vertex_order = 1;
    CodeArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.ncoord = D < kMaxOrderCoords ? D : kMaxOrderCoords;
    PLX_TRY(ensure(L->sort_temp, radix_temp_bytes(m)));
    PLX_TRY(ensure(L->sortkey_in, (size_t)m * 8));
    PLX_TRY(ensure(L->sortkey_out, (size_t)m * 8));
    PLX_TRY(ensure(L->iota, (size_t)m * 4));
    PLX_TRY(ensure(L->vorder, (size_t)m * 4 + 16));
    PLX_TRY(ensure(L->vkeys_alt, (size_t)m * DW * 4 + 16));
    PLX_TRY(ensure(L->vslot_alt, (size_t)m * 4 + 16));
    const int nb = ceil_div(m, kBlock);
    // range of every blur-axis coordinate -> exactly the code bits it needs (counters[32 ..] preset to a very negative int)
    int h_range[2 * kMaxOrderCoords];
    if (h_range_known) {
        // (the embedding already found the range of every blur-axis coordinate: it came back with m)
        memcpy(h_range, h_range_known, sizeof(h_range));
    } else {
        int *range = L->counters.as<int>() + 32;
        PLX_HIP_TRY(hipMemsetAsync(range, 0x80, 2 * kMaxOrderCoords * sizeof(int), stream));
        vertex_range_kernel<D><<<nb, kBlock, 0, stream>>>(L->vkeys.as<uint32_t>(), m, ca.ncoord, range);
        PLX_TRY(read_back(L, range, 2 * ca.ncoord, h_range, stream));
    }
    const int key_bits = layout_key_bits(h_range, ca.ncoord, 0, ca.lo, ca.bits, ca.drop, &ca.maxbits);
    vertex_code_kernel<D><<<nb, kBlock, 0, stream>>>(L->vkeys.as<uint32_t>(), m, ca, L->sortkey_in.as<unsigned long long>(),
                                                     L->iota.as<uint32_t>());
    int second = 0;
    PLX_TRY(radix_sort_pairs64(L->sort_temp.p, L->sortkey_in.as<uint64_t>(), L->sortkey_out.as<uint64_t>(),
                               L->iota.as<uint32_t>(), L->vorder.as<uint32_t>(), m, key_bits, &second, stream));
    if (!second) std::swap(L->iota, L->vorder);
    // the sorted codes stay where the sort left them until this build's neighbour lookups have used them (the window search
    // of neighbor_kernel); exact = every coordinate kept all its bits, so codes are unique and ids are ranks
    L->vcode = second ? L->sortkey_out.as<unsigned long long>() : L->sortkey_in.as<unsigned long long>();
    L->vcode_exact = D <= kMaxOrderCoords;
    for (int c = 0; c < ca.ncoord; ++c) L->vcode_exact = L->vcode_exact && ca.drop[c] == 0;
    {
        int at = key_bits;                                        // bits are appended from the top: the first one is the MSB
        for (int b = ca.maxbits - 1; b >= 0; --b)
            for (int c = 0; c < ca.ncoord; ++c)
                if (ca.bits[c] > b) L->vcode_pos[c][b] = (unsigned char)(--at);
        for (int c = 0; c < ca.ncoord; ++c) {
            L->vcode_bits[c] = ca.bits[c];
            L->vcode_lo[c] = ca.lo[c];
            L->vcode_hi[c] = h_range[2 * c];
        }
    }
    vertex_permute_kernel<D><<<nb, kBlock, 0, stream>>>(L->vorder.as<uint32_t>(), m, L->vkeys.as<uint32_t>(),
                                                        L->vslot.as<uint32_t>(), L->vkeys_alt.as<uint32_t>(),
                                                        L->vslot_alt.as<uint32_t>(), L->table.as<uint32_t>(),
                                                        L->table_idmask != 0xFFFFFFFFu ? 1 : 0,
                                                        L->vs0_valid ? L->vs0.as<uint32_t>() : nullptr);
    std::swap(L->vkeys, L->vkeys_alt);
    std::swap(L->vslot, L->vslot_alt);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

static HashSel hash_sel(const plx_lattice *L)
{
    HashSel hs;
    hs.mask = L->table_mask;
    hs.shift = 32 - L->table_bits;
    return hs;
}

// table geometry + the hash in force for a table of `cap` slots (a power of two, 2^10 .. 2^32)
static void set_table(plx_lattice *L, uint64_t cap)
{
    L->table_mask = (uint32_t)(cap - 1);
    L->table_bits = 0;
    while ((1ull << L->table_bits) < cap) ++L->table_bits;
    L->table_idmask = 0xFFFFFFFFu;
}

// fingerprints ride in the table words of numbered vertices when the ids leave the top byte free
static int table_fp_on(plx_lattice *L, int64_t m)
{
    const int on = m < (1ll << kFpShift) - 1 ? 1 : 0;
    L->table_idmask = on ? ((1u << kFpShift) - 1u) : 0xFFFFFFFFu;
    return on;
}

// ---- stage 1: everything that depends only on this lattice's own points --------------------
// order, embed, hashed insert, first-touch numbering, per-corner vertex ids.  Leaves L->m (the
// number of vertices THESE points touch), vkeys, evid, ew, perm and the key -> id table.
template <int D>
static int stage_local(plx_lattice *L, const float *d_ref, hipStream_t stream, int *evi)
{
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    const int n = (int)L->n;
    const int64_t E = (int64_t)n * D1;
    const int nblocks = ceil_div(n, kBlock);
    auto mark = [&]() { if (L->timing) (void)hipEventRecord(L->ev[(*evi)++], stream); };

    // scale factors exactly as the reference computes them (h:372-390)
    ScaleArgs sf;
    for (int i = 0; i < D; ++i) {
        sf.v[i] = 1.0f / (sqrtf((float)(i + 1) * (i + 2)));
        float sigma_blur = taps_variance(L->taps.c, L->ntaps);
        sf.v[i] *= (D + 1) * sqrtf(sigma_blur + 1.0f / 6.0f);
    }
    L->slice_denom = 1 + powf(2, -D);

    // table capacity: power of two >= 2E  (load factor <= 0.5)
    uint64_t cap = 1024;
    while (cap < 2ull * (uint64_t)E) cap <<= 1;
    set_table(L, cap);

    // ---- point order
    OrderArgs oa;
    memset(&oa, 0, sizeof(oa));
    oa.n_shards = L->n_shards;
    oa.zcurve = g_order_zcurve;
    oa.ncoord = D1 < kMaxOrderCoords ? D1 : kMaxOrderCoords;
    if (oa.zcurve == 2) oa.ncoord = D < kMaxOrderCoords ? D : kMaxOrderCoords;
    int shard_bits = 0;
    while ((1 << shard_bits) < L->n_shards) ++shard_bits;
    oa.base = L->n / L->n_shards;
    oa.extra = L->n % L->n_shards;
    PLX_TRY(ensure(L->perm, (size_t)n * 4));
    PLX_TRY(ensure(L->iota, (size_t)n * 4));
    PLX_TRY(ensure(L->sortkey_in, (size_t)n * 8));
    PLX_TRY(ensure(L->sortkey_out, (size_t)n * 8));
    PLX_TRY(ensure(L->sort_temp, radix_temp_bytes(n)));

    PLX_TRY(ensure(L->ew, (size_t)E * 4));
    PLX_TRY(ensure(L->eslot, (size_t)E * 4));
    PLX_TRY(ensure(L->evid, (size_t)E * 4));
    PLX_TRY(ensure(L->flagmask, (size_t)n * 8));
    const bool want_rank = g_nbr_seed != 0;      // the neighbour seeding reads the ranks after the build
    PLX_TRY(ensure(L->prank, (size_t)n * Rec<D>::W * 4 + 16));      // the point records: insert and numbering rebuild keys from them
    L->prank_valid = false;
    PLX_TRY(ensure(L->blockcnt, (size_t)(nblocks + 1) * 4));
    PLX_TRY(ensure(L->table, (size_t)cap * 4));
    PLX_TRY(ensure(L->counters, 256));

    PLX_HIP_TRY(hipMemsetAsync(L->counters.p, 0, 256, stream));
    PLX_HIP_TRY(hipMemsetAsync(L->table.p, 0xFF, (size_t)cap * 4, stream));

    mark();
    // plx_set_reuse_order: the positions are the previous build's, re-scaled (a lengthscale that moved): the point order --
    // a locality device, nothing in the structure depends on it -- is kept, and the range pass, its read-back, the key pass and
    // the radix passes over the points are skipped.  Only for the same rows on the same shard; one shot.
    const bool keep_order = L->reuse_order && g_sort_points && L->order_n == L->n && L->order_d == D &&
                            L->order_shard == L->shard_index && L->order_shards == L->n_shards;
    L->reuse_order = false;
    if (keep_order) {
        ++L->order_age;
    } else if (g_sort_points) {
        int key_bits;
        if (g_order_compact) {
            // range of every rounded coordinate -> exactly the key bits it needs (counters[32 ..] preset to a very negative int)
            int *range = L->counters.as<int>() + 32;
            PLX_HIP_TRY(hipMemsetAsync(range, 0x80, 2 * kMaxOrderCoords * sizeof(int), stream));
            const int rstride = (g_order_sample > 1 && n >= (1 << 16)) ? g_order_sample : 1;
            coord_range_kernel<D><<<ceil_div(ceil_div(n, rstride), kBlock), kBlock, 0, stream>>>(d_ref, n, sf, oa.zcurve, oa.ncoord, range,
                                                                                                 rstride);
            int h_range[2 * kMaxOrderCoords];
            PLX_TRY(read_back(L, range, 2 * oa.ncoord, h_range, stream));
            key_bits = shard_bits + layout_key_bits(h_range, oa.ncoord, shard_bits, oa.lo, oa.bits, oa.drop, &oa.maxbits);
        } else {
            int b = (64 - shard_bits) / oa.ncoord;
            if (b > 8) b = 8;
            for (int c = 0; c < oa.ncoord; ++c) { oa.lo[c] = -(1 << (b - 1)); oa.bits[c] = b; oa.drop[c] = 0; }
            oa.maxbits = b;
            key_bits = shard_bits + b * oa.ncoord;
        }
        sortkey_kernel<D><<<nblocks, kBlock, 0, stream>>>(d_ref, n, sf, oa, L->sortkey_in.as<unsigned long long>(),
                                                          L->iota.as<uint32_t>());
        int second = 0;
        PLX_TRY(radix_sort_pairs64(L->sort_temp.p, L->sortkey_in.as<uint64_t>(), L->sortkey_out.as<uint64_t>(),
                                   L->iota.as<uint32_t>(), L->perm.as<uint32_t>(), n, key_bits, &second, stream));
        if (!second) std::swap(L->iota, L->perm);          // the sorted point ids are wherever the last pass left them
        L->order_n = L->n; L->order_d = D; L->order_shard = L->shard_index; L->order_shards = L->n_shards; L->order_age = 0;
    } else {
        iota_kernel<<<nblocks, kBlock, 0, stream>>>(L->perm.as<uint32_t>(), n);
        L->order_n = 0;
    }
    // counters[30] = m, [31] = key-range error flag, [32 .. 63] = range of the vertices' blur-axis coordinates: one read-back
    int *cnt = L->counters.as<int>() + 30;
    const bool embed_range = g_embed_vrange != 0 && !L->for_merge && g_vertex_order != 0;
    if (embed_range) PLX_HIP_TRY(hipMemsetAsync(cnt + 2, 0x80, 2 * kMaxOrderCoords * sizeof(int), stream));
    embed_kernel<D><<<nblocks, kBlock, 0, stream>>>(d_ref, L->perm.as<uint32_t>(), n, sf, L->ew.as<float>(), cnt,
                                                    L->prank.as<uint32_t>(), embed_range ? cnt + 2 : nullptr);
    mark();
    // flag_own: the insert marks owners (bit 31 of eslot: the table must not need that bit) and displaced corners (flagmask
    // doubles as the displaced mask until flag_own_kernel turns it into the first-touch mask in place); a table of 2^32
    // slots (more than 2^30 corners) has no bit to spare and takes the table-gather form (flag_kernel)
    const bool flag_own = L->table_bits <= 31;
    if (flag_own) PLX_HIP_TRY(hipMemsetAsync(L->flagmask.p, 0, (size_t)n * 8, stream));
    insert_point_kernel<D><<<tile_grid(nblocks, g_insert_xcd & 1), kBlock, 0, stream>>>(
        L->prank.as<uint32_t>(), n, L->table.as<uint32_t>(), hash_sel(L), L->eslot.as<uint32_t>(), g_insert_dedupe,
        flag_own ? L->flagmask.as<uint32_t>() : nullptr, nblocks, g_insert_xcd & 1);
    mark();
    L->flags_valid = !L->for_merge;      // the first-touch bits of this build's points stay in flagmask (plx_first.hip reads them)
    if (flag_own)
        flag_own_kernel<D1><<<nblocks, kBlock, 0, stream>>>(L->eslot.as<uint32_t>(), L->flagmask.as<uint32_t>(), n,
                                                            L->flagmask.as<uint32_t>(), L->blockcnt.as<int>());
    else
        flag_kernel<D1><<<nblocks, kBlock, 0, stream>>>(L->eslot.as<uint32_t>(), L->table.as<uint32_t>(), n,
                                                        L->flagmask.as<uint32_t>(), L->blockcnt.as<int>());
    scan_blocks_kernel<<<1, kWideScanT, 0, stream>>>(L->blockcnt.as<int>(), nblocks, cnt);
    int h_cnt[2 + 2 * kMaxOrderCoords];
    PLX_TRY(read_back(L, cnt, embed_range ? 2 + 2 * kMaxOrderCoords : 2, h_cnt, stream));   // m sizes everything below
    if (h_cnt[1] != 0) {
        set_error("a lattice coordinate left the int16 key range (|x/lengthscale| too large, NaN or Inf)");
        return PLX_ERR_KEY_RANGE;
    }
    const int m = h_cnt[0];
    L->m = m;
    PLX_TRY(ensure(L->vkeys, (size_t)m * DW * 4 + 16));
    PLX_TRY(ensure(L->vslot, (size_t)m * 4 + 16));
    L->vs0_valid = !L->for_merge;
    if (L->vs0_valid) PLX_TRY(ensure(L->vs0, ((size_t)m + 8) * 4));
    if (want_rank) PLX_TRY(ensure(L->vowner, ((size_t)m + 8) * 4));
    const int fp_on = table_fp_on(L, m);
    // the ids handed out here are final unless a renumbering follows (a local stage's are final as LOCAL ids: the merge maps them)
    const bool ids_final = L->for_merge || !will_renumber(L, m, E);
    const bool assign_evid = g_assign_evid != 0 && ids_final;
    assign_kernel<D><<<nblocks, kBlock, 0, stream>>>(L->flagmask.as<uint32_t>(), L->blockcnt.as<int>(),
                                                     L->eslot.as<uint32_t>(), L->prank.as<uint32_t>(), n,
                                                     L->table.as<uint32_t>(), L->vkeys.as<uint32_t>(), L->vslot.as<uint32_t>(), fp_on,
                                                     assign_evid ? L->evid.as<int>() : nullptr,
                                                     (L->vs0_valid && ids_final) ? L->vs0.as<uint32_t>() : nullptr,
                                                     (want_rank && ids_final) ? L->vowner.as<uint32_t>() : nullptr);
    if (!L->for_merge) PLX_TRY(renumber_vertices<D>(L, E, stream, embed_range ? h_cnt + 2 : nullptr));   // (a job built from local rows renumbers the union, after the merge)
    mark();
    if (assign_evid)
        ids_rest_kernel<<<nblocks, kBlock, 0, stream>>>(L->eslot.as<uint32_t>(), L->flagmask.as<uint32_t>(), L->table.as<uint32_t>(), n, D1,
                                                        L->evid.as<int>(), L->table_idmask);
    else
        ids_kernel<<<dim3(tile_grid(nblocks, g_insert_xcd >> 1), D1), kBlock, 0, stream>>>(L->eslot.as<uint32_t>(), L->table.as<uint32_t>(), n,
                                                                                           L->evid.as<int>(), L->table_idmask, nblocks,
                                                                                           g_insert_xcd >> 1);
    L->prank_valid = want_rank && !L->for_merge;
    mark();
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// ---- stage 2 (sharded jobs only): merge the vertex sets of all ranks ---------------------------
// all_keys = every rank's local vertex keys, concatenated in rank order, each block in that
// rank's local id order.  First occurrence wins, ids in order of first occurrence: exactly the
// first-touch numbering of the shard-major point order, the same on every rank.

template <int D>
__global__ __launch_bounds__(kBlock) void merge_insert_kernel(const uint32_t *__restrict__ keys, int M,
                                                              uint32_t *__restrict__ table, HashSel hs,
                                                              uint32_t *__restrict__ slot)
{
    constexpr int DW = (D + 1) / 2;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= M) return;
    uint32_t k[DW];
    load_key<DW>(keys, (size_t)idx, k);
    const uint32_t mask = hs.mask;
    uint32_t h = hash_slot(key_hash<D>(k, hs), hs);
    for (;;) {
        uint32_t o = table[h];
        if (o == kEmpty) {
            o = atomicCAS(&table[h], kEmpty, (uint32_t)idx);
            if (o == kEmpty) break;
        }
        if (o == (uint32_t)idx) break;
        uint32_t ko[DW];
        load_key<DW>(keys, (size_t)o, ko);
        if (key_equal<DW>(k, ko)) {
            if ((uint32_t)idx < o) atomicMin(&table[h], (uint32_t)idx);
            break;
        }
        h = (h + 1) & mask;
    }
    slot[idx] = h;
}

__global__ __launch_bounds__(kBlock) void merge_flag_kernel(const uint32_t *__restrict__ slot,
                                                            const uint32_t *__restrict__ table, int M,
                                                            uint32_t *__restrict__ flags, int *__restrict__ blockcnt)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    int first = 0;
    if (idx < M) {
        first = table[slot[idx]] == (uint32_t)idx ? 1 : 0;
        flags[idx] = (uint32_t)first;
    }
    int total;
    block_exclusive_scan(first, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

template <int D>
__global__ __launch_bounds__(kBlock) void merge_assign_kernel(const uint32_t *__restrict__ flags,
                                                              const int *__restrict__ blockoff,
                                                              const uint32_t *__restrict__ slot,
                                                              const uint32_t *__restrict__ keys, int M,
                                                              uint32_t *__restrict__ table,
                                                              uint32_t *__restrict__ gkeys, uint32_t *__restrict__ vslot, int fp_on,
                                                              uint32_t *__restrict__ vs0)
{
    constexpr int DW = (D + 1) / 2;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const int first = (idx < M) ? (int)flags[idx] : 0;
    int total;
    const int gid = blockoff[blockIdx.x] + block_exclusive_scan(first, &total);
    if (idx < M && first) {
        uint32_t k[DW];
        load_key<DW>(keys, (size_t)idx, k);
        table[slot[idx]] = table_word<D>((uint32_t)gid, k, fp_on);
        vslot[gid] = slot[idx];
        store_key<DW>(gkeys, (size_t)gid, k);
        if (vs0) vs0[gid] = lin_hash_packed<D>(k);
    }
}

// evid[i] = global id of local vertex evid[i]  (its key sits at all_keys[my_off + local id])
__global__ __launch_bounds__(kBlock) void merge_remap_kernel(int *__restrict__ evid, int64_t E,
                                                             const uint32_t *__restrict__ slot,
                                                             const uint32_t *__restrict__ table, int my_off, uint32_t idmask)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < E) evid[i] = (int)(table[slot[my_off + evid[i]]] & idmask);
}

template <int D>
static int stage_merge(plx_lattice *L, const uint32_t *d_all_keys, const int64_t *h_counts, int n_ranks, int my_rank,
                       hipStream_t stream)
{
    constexpr int D1 = D + 1;
    constexpr int DW = (D + 1) / 2;
    int64_t M = 0, my_off = 0;
    for (int r = 0; r < n_ranks; ++r) {
        if (r == my_rank) my_off = M;
        M += h_counts[r];
    }
    if (h_counts[my_rank] != L->m) {
        set_error("plx_build_merge: rank %d announced %lld local vertices, the lattice has %lld", my_rank,
                  (long long)h_counts[my_rank], (long long)L->m);
        return PLX_ERR_STATE;
    }
    if (M >= (1ll << 31) - 1024) { set_error("plx_build_merge: %lld keys in total", (long long)M); return PLX_ERR_TOO_LARGE; }
    uint64_t cap = 1024;
    while (cap < 2ull * (uint64_t)M) cap <<= 1;
    const int nblocks = ceil_div(M, kBlock);
    PLX_TRY(ensure(L->table, (size_t)cap * 4));
    PLX_TRY(ensure(L->merge_slot, (size_t)M * 4 + 16));
    PLX_TRY(ensure(L->merge_flags, (size_t)M * 4 + 16));
    PLX_TRY(ensure(L->blockcnt, (size_t)(nblocks + 1) * 4));
    set_table(L, cap);
    PLX_HIP_TRY(hipMemsetAsync(L->table.p, 0xFF, (size_t)cap * 4, stream));
    merge_insert_kernel<D><<<nblocks, kBlock, 0, stream>>>(d_all_keys, (int)M, L->table.as<uint32_t>(), hash_sel(L),
                                                           L->merge_slot.as<uint32_t>());
    merge_flag_kernel<<<nblocks, kBlock, 0, stream>>>(L->merge_slot.as<uint32_t>(), L->table.as<uint32_t>(), (int)M,
                                                      L->merge_flags.as<uint32_t>(), L->blockcnt.as<int>());
    scan_blocks_kernel<<<1, kWideScanT, 0, stream>>>(L->blockcnt.as<int>(), nblocks, L->counters.as<int>());
    int h_cnt[2];
    PLX_TRY(read_back(L, L->counters.as<int>(), 2, h_cnt, stream));
    const int m = h_cnt[0];
    PLX_TRY(ensure(L->vkeys, (size_t)m * DW * 4 + 16));   // local keys are no longer needed: all_keys holds them
    PLX_TRY(ensure(L->vslot, (size_t)m * 4 + 16));
    L->vs0_valid = true;
    if (L->vs0_valid) PLX_TRY(ensure(L->vs0, ((size_t)m + 8) * 4));
    const int fp_on = table_fp_on(L, m);
    merge_assign_kernel<D><<<nblocks, kBlock, 0, stream>>>(L->merge_flags.as<uint32_t>(), L->blockcnt.as<int>(),
                                                           L->merge_slot.as<uint32_t>(), d_all_keys, (int)M,
                                                           L->table.as<uint32_t>(), L->vkeys.as<uint32_t>(), L->vslot.as<uint32_t>(), fp_on,
                                                           L->vs0_valid ? L->vs0.as<uint32_t>() : nullptr);
    L->m = m;
    PLX_TRY(renumber_vertices<D>(L, L->merge_total_points > 0 ? L->merge_total_points * D1 : M, stream));
    const int64_t E = L->n * D1;
    merge_remap_kernel<<<ceil_div(E, kBlock), kBlock, 0, stream>>>(L->evid.as<int>(), E, L->merge_slot.as<uint32_t>(),
                                                                   L->table.as<uint32_t>(), (int)my_off, L->table_idmask);
    L->m = m;
    L->prank_valid = false;      // (the first-touch flags of the local stage name local vertices)
    L->partial_cover = true;     // this rank's points do not touch every vertex: splat zero-fills first
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// The vertex-sorted splat CSR of the current build (csr_pt / csr_row / csr_w, sorted vertex ids in csr_vid):
// stable radix sort of the owned corners by vertex id.  Needed by the multi-column splat kernels, the fused
// backward, the structure exports and every lattice without block tables; built once per lattice build, on demand.
int ensure_csr(plx_lattice *L, hipStream_t stream)
{
    if (L->csr_ready) return PLX_OK;
    PLX_TRY(refuse_under_capture(stream, "the vertex-sorted corner table of this lattice"));
    const int n = (int)L->n, n_own = (int)(L->own_end - L->own_begin), m = (int)L->m, D1 = L->d + 1;
    PLX_TRY(ensure(L->csr_pt, (size_t)L->nnz * 4 + 64));   // slack: 16-byte loads at the tail
    PLX_TRY(ensure(L->csr_w, (size_t)L->nnz * 4 + 64));
    PLX_TRY(ensure(L->csr_row, (size_t)L->nnz * 4 + 64));
    PLX_TRY(ensure(L->csr_vid, (size_t)L->nnz * 4 + 16));   // its own buffer: the block build recycles the sort scratch
    if (L->nnz > 0) {
        int end_bit = 1;
        while ((1ll << end_bit) < (int64_t)m) ++end_bit;
        PLX_TRY(ensure(L->sort_keys_in, (size_t)L->nnz * 4 + 16));
        PLX_TRY(ensure(L->sort_vals_in, (size_t)L->nnz * 4));
        PLX_TRY(ensure(L->sort_vals_out, (size_t)L->nnz * 4));
        PLX_TRY(ensure(L->sort_temp, radix_temp_bytes(L->nnz)));
        csr_keys_kernel<<<dim3(ceil_div(n_own, kBlock), D1), kBlock, 0, stream>>>(
            L->evid.as<int>(), n, (int)L->own_begin, n_own, L->sort_keys_in.as<uint32_t>(),
            L->sort_vals_in.as<uint32_t>());
        int second = 0;
        PLX_TRY(radix_sort_pairs32(L->sort_temp.p, L->sort_keys_in.as<uint32_t>(), L->csr_vid.as<uint32_t>(),
                                   L->sort_vals_in.as<uint32_t>(), L->sort_vals_out.as<uint32_t>(), L->nnz, end_bit, &second,
                                   stream));
        if (!second) { std::swap(L->sort_keys_in, L->csr_vid); std::swap(L->sort_vals_in, L->sort_vals_out); }
        csr_finalize_kernel<<<ceil_div(L->nnz, kBlock), kBlock, 0, stream>>>(
            L->csr_vid.as<uint32_t>(), L->sort_vals_out.as<uint32_t>(), splat_weights(L), n,
            (int)L->own_begin, (int)L->nnz, m, L->perm.as<uint32_t>(), L->csr_pt.as<int>(), L->csr_row.as<int>(),
            L->csr_w.as<float>());
    }
    PLX_HIP_TRY(hipGetLastError());
    L->csr_ready = true;
    return PLX_OK;
}

// ---- stage 3: gather tables over the final vertex numbering -------------------------------------
template <int D>
static int stage_tables(plx_lattice *L, hipStream_t stream, int *evi)
{
    constexpr int D1 = D + 1;
    const int n_own = (int)(L->own_end - L->own_begin);
    const int m = (int)L->m;
    const int order = L->order;
    auto mark = [&]() { if (L->timing) (void)hipEventRecord(L->ev[(*evi)++], stream); };
    L->mstride = ((int64_t)m + 63) & ~63ll;
    L->nnz = (int64_t)n_own * D1;
    L->nchunks = ceil_div(L->nnz, kSplatChunk);

    PLX_TRY(ensure(L->nbr, (size_t)D1 * 2 * order * L->mstride * 4 + 4));

    if (order > 0) {
        const int nplane_fast = (g_insert_plane_fast != 0 && ceil_div(m, kBlock) <= 65535) ? 1 : 0;
        dim3 ngrid(ceil_div(m, kBlock), D1);
        if (nplane_fast) ngrid = dim3(D1, ceil_div(m, kBlock));
        // Morton-numbered lattices: most lookups are decided by a binary search in a window of the sorted codes
        NbrCode nc;
        memset(&nc, 0, sizeof(nc));
        const unsigned long long *vcode = nullptr;
        if (g_nbr_window > 0 && L->vertex_order == 1 && L->vcode && L->vcode_exact) {
            vcode = L->vcode;
            for (int c = 0; c < kMaxOrderCoords; ++c) {
                nc.nbits[c] = L->vcode_bits[c]; nc.lo[c] = L->vcode_lo[c]; nc.hi[c] = L->vcode_hi[c];
                memcpy(nc.pos[c], L->vcode_pos[c], 16);
            }
        }
        const bool sliced = L->vs0_valid && g_nbr_symmetric && ceil_div(m, kBlock) < (1 << 28) &&
                            (g_nbr_sliced == 2 || (g_nbr_sliced == 1 && vcode == nullptr && m >= (1 << 20)));
        const uint32_t *slotmap = nullptr;
        if (!sliced && (g_nbr_bitmap == 2 || (g_nbr_bitmap == 1 && m >= (1 << 22)))) {
            const uint64_t cap = (uint64_t)L->table_mask + 1u;       // a power of two >= 1024
            PLX_TRY(ensure(L->slotmap, (size_t)cap / 8 + 8));
            slotmap_kernel<<<ceil_div(cap, kBlock), kBlock, 0, stream>>>(L->table.as<uint32_t>(), cap,
                                                                         L->slotmap.as<unsigned long long>());
            slotmap = L->slotmap.as<uint32_t>();
        }
        if (sliced) {
            const uint64_t cap = (uint64_t)L->table_mask + 1u;
            PLX_TRY(ensure(L->nibmap, (size_t)cap / 2 + 16));
            const int fp_on = L->table_idmask != 0xFFFFFFFFu ? 1 : 0;
            nibmap_kernel<<<ceil_div(cap / 8, kBlock), kBlock, 0, stream>>>(L->table.as<uint32_t>(), cap / 8, fp_on,
                                                                            L->nibmap.as<uint32_t>());
            NbrDelta nd;
            uint32_t sum = 0;
            for (int c = 0; c < D; ++c) sum += kHashMulHost[c];
            for (int a = 0; a < D; ++a) nd.d[a] = (uint32_t)D1 * kHashMulHost[a] - sum;
            nd.d[D] = 0u - sum;
            const uint8_t *vaxis = nullptr;
            if (g_nbr_seed != 0 && L->prank_valid && L->vertex_order == 0 && !L->partial_cover) {
                // (first-touch numbering only: vowner and evid then name the same, final ids)
                PLX_TRY(ensure(L->vaxis, (size_t)L->mstride + 16));
                PLX_HIP_TRY(hipMemsetAsync(L->vaxis.p, 0xFF, (size_t)L->mstride + 16, stream));
                nbr_rows_init_kernel<D><<<ceil_div(L->mstride, kBlock), kBlock, 0, stream>>>(
                    L->vowner.as<uint32_t>(), L->evid.as<int>(), L->prank.as<uint32_t>(), (int)L->n, m, L->mstride, order,
                    L->nbr.as<int>(), L->vaxis.as<uint8_t>());
                vaxis = L->vaxis.as<uint8_t>();
            } else {
                PLX_HIP_TRY(hipMemsetAsync(L->nbr.p, 0xFF, (size_t)D1 * 2 * order * L->mstride * 4, stream));
            }
            neighbor_sliced_kernel<D><<<8u * (unsigned)ceil_div(m, kBlock * sliced_vpt(D1)), kBlock, 0, stream>>>(
                L->vs0.as<uint32_t>(), L->vkeys.as<uint32_t>(), m, L->mstride, order, L->table.as<uint32_t>(), hash_sel(L),
                L->table_idmask, fp_on, L->nibmap.as<uint32_t>(), nd, L->nbr.as<int>(), vaxis);
        } else if (g_nbr_symmetric) {
            PLX_HIP_TRY(hipMemsetAsync(L->nbr.p, 0xFF, (size_t)D1 * 2 * order * L->mstride * 4, stream));
            neighbor_kernel<D, true><<<ngrid, kBlock, 0, stream>>>(L->vkeys.as<uint32_t>(), m, L->mstride, order,
                                                                    L->table.as<uint32_t>(), hash_sel(L), L->table_idmask,
                                                                    L->nbr.as<int>(), nplane_fast, slotmap, vcode, nc, g_nbr_window);
        } else {
            neighbor_kernel<D, false><<<ngrid, kBlock, 0, stream>>>(L->vkeys.as<uint32_t>(), m, L->mstride, order,
                                                                     L->table.as<uint32_t>(), hash_sel(L), L->table_idmask,
                                                                     L->nbr.as<int>(), nplane_fast, slotmap, vcode, nc, g_nbr_window);
        }
        L->vcode = nullptr;                                       // (the sort buffers are free for their next user)
    }
    PLX_TRY(replay_patch_tables(L, stream));   // reference_growth: invisible vertices, the blur-time miss (before anything is derived from the rows)
    PLX_TRY(build_blur_pairs(L, stream));      // composite neighbours for the two-axes-per-launch blur (coarse lattices)
    // compacted copy for sparse lattices (used by the vd = 1 blur when under a quarter of the neighbours exist)
    // Neighbourhoods are only sparse when most corners created a vertex of their own (measured: m/E = 0.19 ->
    // 57 % of the slots exist, 0.8 -> ~30 %, 0.91 -> ~20 %, 0.99 -> 12 %), so the count + host sync is skipped for
    // denser lattices.  What the copy buys per MVM against what it costs per build (N = 1e6, d = 8): l = 0.5 / 0.4
    // (m/E 0.53 / 0.80) nothing for 0.4-0.5 ms; l = 0.35 (0.91) 10 us for 0.52 ms; l = 0.3 / 0.25 / 0.2 (>= 0.97) 20 / 19 /
    // 26 us for 0.56 ms, i.e. it pays from ~28 MVMs per build on: built for m/E >= 0.85 and fill < 0.25, and never for a
    // lattice that serves one MVM (plx_filter).
    L->use_compact = false;
    const bool maybe_sparse = g_compact_nbr == 2 || ((double)m >= 0.85 * (double)L->n * D1 && !L->single_use);
    if (order >= 1 && order <= 3 && g_compact_nbr != 0 && maybe_sparse) {
        const int taps2 = 2 * order;
        L->nquads = ((int64_t)m + 3) / 4;
        L->nqwaves = (L->nquads + 63) / 64;
        PLX_TRY(ensure(L->cmask, (size_t)D1 * L->nquads * 4));
        PLX_TRY(ensure(L->cbase, (size_t)D1 * (L->nqwaves + 1) * 4));
        dim3 cgrid((unsigned)ceil_div(L->nqwaves * 64, kBlock), D1);
        compact_count_kernel<<<cgrid, kBlock, 0, stream>>>(L->nbr.as<int>(), m, L->mstride, taps2, L->nquads,
                                                           L->nqwaves, L->cmask.as<uint32_t>(), L->cbase.as<uint32_t>());
        compact_scan_kernel<<<D1, kScanT, 0, stream>>>(L->cbase.as<uint32_t>(), L->nqwaves, L->counters.as<int>());
        int h_axis[2 + PLX_MAX_DIM + 1];
        PLX_TRY(read_back(L, L->counters.as<int>(), 2 + D1, h_axis, stream));
        AxisOffsets ao;
        int64_t total = 0;
        for (int a = 0; a < D1; ++a) { ao.off[a] = total; L->compact_off[a] = total; total += h_axis[2 + a]; }
        L->compact_off[D1] = total;
        const double fill = (double)total / ((double)m * taps2 * D1);
        if (g_compact_nbr == 2 || fill < 0.25) {
            PLX_TRY(ensure(L->cids, (size_t)total * 4 + 64));
            compact_fill_kernel<<<cgrid, kBlock, 0, stream>>>(L->nbr.as<int>(), m, L->mstride, taps2, L->nquads,
                                                              L->nqwaves, L->cmask.as<uint32_t>(),
                                                              L->cbase.as<uint32_t>(), ao, L->cids.as<int>());
            L->use_compact = true;
        }
    }
    mark();

    // splat / slice tables over the owned points (block tables, their vertex-sorted half, the vertex-sorted CSR): each is
    // built by its first user (plx_prepare, or the first MVM that needs it -- ensure_blocks / ensure_s2 / ensure_csr)
    L->csr_ready = false;
    L->first_ready = false;
    L->use_first = false;
    L->blocks_ready = false;
    L->s2_ready = false;
    L->inv_perm_ready = false;
    L->use_blocks = false;
    mark();
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

template <int D>
static int build_typed(plx_lattice *L, const float *d_ref, hipStream_t stream)
{
    int evi = 0;
    L->partial_cover = false;
    PLX_TRY(stage_local<D>(L, d_ref, stream, &evi));
    L->replay = plx_lattice::Replay();
    if (g_reference_growth) PLX_TRY(replay_simulate(L, stream));      // the reference CPU path's table-growth quirk (plx_replay.hip)
    PLX_TRY(stage_tables<D>(L, stream, &evi));
    if (L->timing) {
        PLX_HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < 6; ++i) (void)hipEventElapsedTime(&L->build_ms[i], L->ev[i], L->ev[i + 1]);
    }
    return PLX_OK;
}

template <int D>
static int local_typed(plx_lattice *L, const float *d_ref, hipStream_t stream)
{
    int evi = 0;
    const bool t = L->timing;
    L->timing = false;
    L->partial_cover = false;
    L->replay = plx_lattice::Replay();
    int rc = stage_local<D>(L, d_ref, stream, &evi);
    L->timing = t;
    return rc;
}

template <int D>
static int merge_typed(plx_lattice *L, const uint32_t *d_all_keys, const int64_t *h_counts, int n_ranks, int my_rank,
                       hipStream_t stream)
{
    int evi = 0;
    const bool t = L->timing;
    L->timing = false;
    int rc = stage_merge<D>(L, d_all_keys, h_counts, n_ranks, my_rank, stream);
    if (rc == PLX_OK) rc = stage_tables<D>(L, stream, &evi);
    L->timing = t;
    return rc;
}

#define PLX_DIM_SWITCH(CALL)                                                                                  \
    switch (L->d) {                                                                                           \
        PLX_CASE(1) PLX_CASE(2) PLX_CASE(3) PLX_CASE(4) PLX_CASE(5) PLX_CASE(6) PLX_CASE(7) PLX_CASE(8)       \
        PLX_CASE(9) PLX_CASE(10) PLX_CASE(11) PLX_CASE(12) PLX_CASE(13) PLX_CASE(14) PLX_CASE(15) PLX_CASE(16) \
        PLX_CASE(17) PLX_CASE(18) PLX_CASE(19) PLX_CASE(20) PLX_CASE(21) PLX_CASE(22) PLX_CASE(23) PLX_CASE(24) \
        PLX_CASE(25) PLX_CASE(26) PLX_CASE(27) PLX_CASE(28) PLX_CASE(29) PLX_CASE(30) PLX_CASE(31) PLX_CASE(32) \
    default:                                                                                                  \
        set_error("d = %d outside 1..%d", L->d, PLX_MAX_DIM);                                                 \
        return PLX_ERR_DIM;                                                                                   \
    }

int build_impl(plx_lattice *L, const float *d_ref, hipStream_t stream)
{
#define PLX_CASE(D) case D: return build_typed<D>(L, d_ref, stream);
    PLX_DIM_SWITCH()
#undef PLX_CASE
}

int build_local_impl(plx_lattice *L, const float *d_ref, hipStream_t stream)
{
#define PLX_CASE(D) case D: return local_typed<D>(L, d_ref, stream);
    PLX_DIM_SWITCH()
#undef PLX_CASE
}

int build_merge_impl(plx_lattice *L, const uint32_t *d_all_keys, const int64_t *h_counts, int n_ranks, int my_rank,
                     hipStream_t stream)
{
#define PLX_CASE(D) case D: return merge_typed<D>(L, d_all_keys, h_counts, n_ranks, my_rank, stream);
    PLX_DIM_SWITCH()
#undef PLX_CASE
}

int export_row_ptr(plx_lattice *L, hipStream_t stream)
{
    const int m = (int)L->m;
    PLX_TRY(ensure_csr(L, stream));
    PLX_TRY(ensure(L->row_ptr, (size_t)(m + 1) * 4));
    row_ptr_kernel<<<ceil_div((int64_t)m + 1, kBlock), kBlock, 0, stream>>>(L->csr_vid.as<uint32_t>(), (int)L->nnz, m,
                                                                            L->row_ptr.as<int>());
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

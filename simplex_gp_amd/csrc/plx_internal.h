// plx_internal.h -- shared declarations of libplx (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "plx.h"

namespace plx {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;   // empty hash slot
constexpr int kBlock = 256;                // threads per workgroup for every kernel here
constexpr int kSplatBlock = 256;           // threads per splat-scan workgroup (one wave per chunk measured 15 % slower)
constexpr int kSplatChunk = 4 * kSplatBlock; // CSR corners per splat workgroup (4 per thread; 8 measured slower)

void set_error(const char *fmt, ...);

#define PLX_HIP_TRY(expr)                                                            \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            plx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                           __FILE__, __LINE__);                                      \
            return PLX_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

#define PLX_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != PLX_OK) return _rc; \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipStream_t owner = nullptr;   // stream the buffer was allocated on (stream-ordered: hipMallocAsync)
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ScaleArgs { float v[PLX_MAX_DIM]; };              // h:372-390 scale factors
struct TapArgs { float c[2 * PLX_MAX_ORDER + 1]; };      // blur weights, h:546

// Kernel-variant switches (plx_tune).  The process-wide defaults live in g_tune_defaults; every lattice takes a snapshot
// when a build starts (plx_build / plx_build_local / plx_filter) and every entry point serves its lattice under that
// snapshot (tl_tune), so a plx_tune call never changes what a built lattice does -- it takes effect at the next build.
struct Tune {
    int sort_points = 1;   // 0 keeps the caller's point order (A/B only)
    int readback_spin = 1;   // read_back: 1 = spin on the mailbox word, 0 = wait for the stream
    int order_compact = 1;   // 1: point-order keys over exactly the bits every coordinate's range needs; 0: a fixed 7 / 8 bits per coordinate
    int order_zcurve = 1;   // 1: points along the Z-curve of their rounded lattice coordinates; 0: lexicographically; 2: Z-curve of the blur-axis coordinates
    int nbr_symmetric = 1;   // neighbour build looks up the positive taps only and mirrors the hits (fine regime 7.2 -> 4.9 ms)
    int insert_dedupe = 2;   // hashed insert: 2 = every key of a wave probes once, 1 = runs of equal neighbouring lanes probe once, 0 = every lane probes
    int compact_nbr = 1;   // 0 never, 1 when under a quarter of the neighbour slots exist, 2 always
    int insert_plane_fast = 1;   // the d+1 corner planes of a run of points are adjacent workgroups of the hashed insert / neighbour lookups
    int vertex_order = 1;   // 0: first touch; 1: Morton order where it pays; 2: always
    int blur_vpt = 4;   // vertices per thread in the vd = 1 blur (2 or 4)
    int blur_small = 1;   // all blur passes in one workgroup when m <= 16384 (vd = 1)
    int xcd_remap = 1;   // workgroup b works on tile (b % 8) * ceil(nb/8) + b / 8: every XCD owns one contiguous slice of the lattice
    int splat_direct = 1;   // vd = 1 CSR splat gathers from d_src through caller-row indices: 0 never, 1 for <= 2e6 corners, 2 always
    int blur_narrow = 1;   // vd 2..16 blur: row length compiled in, branch-free
    int blur_multi = 1;   // vd > 1 blur: 4 items per thread (and, on sparse lattices, only the rows that change) on rows of >= 17 chunks; 2: >= 32 (rounds 1-5)
    int splat_group = 1;   // vd 2..64: lane-group streaming splat
    int splat_wide = 1;   // row-parallel splat for rows of 17..128 chunks; 2: 32..128 (rounds 1-5)
    int blur_fuse = 1;   // two blur axes per launch (vd = 1): 0 never, 1 on cache-resident lattices, 2 always
    int blur_fuse_vec = 1;   // two blur axes per launch for rows of 2..4 chunks
    int block_path = 1;   // 0 never, 1 when the lattice qualifies (see build_blocks), 2 whenever representable
    int scatter_store = 0;   // slice's row-scattered output stores: 0 plain, 1 non-temporal, 2 agent-scope
    int unpermute_gather = 1;   // caller row order out of slice: 1 = lattice-ordered scratch + a gather pass, 0 = scatter from the slice kernel
    int block_e = 0;   // corners per thread of the block kernels: 0 = per lattice (choose_block_e), 16 or 24
    int block_dense_combine = 1;   // combine numbers the vertices by counting row ends when every vertex has block rows
    int perm_rows = 1;   // multi-column row permutations: 1 = 16-byte chunks in, LDS-transposed whole-line stores out; 0 = the round-3 per-float / per-chunk kernels
    int nbr_window = 512;   // Morton-numbered lattices: neighbour lookups first search this many sorted codes next to the vertex (0 = hash only)
    int nbr_bitmap = 1;   // neighbour lookups test a slot-occupancy bitmap before they touch the hash table: 0 never, 1 when m >= 2^22, 2 always
    int nbr_sliced = 1;   // neighbour lookups served by the XCD that owns the slot's eighth of a 4-bit-per-slot map (L2 resident): 0 never, 1 when the lattice keeps first-touch numbering and m >= 2^20, 2 always
    int insert_xcd = 2;   // XCD-aware tile order (each XCD one contiguous eighth of the points): bit 0 the point-per-thread insert (measured slower), bit 1 the id lookup
    int order_sample = 8;   // point-order key layout from the coordinate ranges of every k-th point (1: of all points); from 65,536 points up
    int embed_vrange = 0;   // 1: the embedding finds the range of the vertices' blur-axis coordinates (Morton renumbering) itself -- no pass over the vertex keys, no read-back of its own; measured: saves 22 us there, costs the embedding 30 (l = 1) to 70 us (l = 0.25): off
    int reference_growth = 0;   // 1: replay the reference CPU path's hash-table-growth quirk (plx_replay.hip): literal parity with cpp/permutohedral.h where its table doubles; plain single-process builds only, O(m) host work per build (the event form); 2: the lookup-by-lookup form, O(N (d+1)) (the checker of 1)
    int blk_sort = 15;   // per-block LDS sort of the block tables: 0 = (vertex, corner) pairs, 4 bits per pass; 4 / 5 / 6 = corner index packed under the vertex id, keys only, that many bits per pass, 256 threads; 15 = 5 bits with 512 threads
    int assign_evid = 1;   // the numbering pass stores the vertex id of every first-touch corner itself when the numbering is final; the id lookup then serves the other corners only
    int nbr_seed = 1;   // sliced neighbour lookups: the +1 neighbour that is a corner of the vertex's own first-touch simplex comes from the embedding, no lookup
    int blur_active = 1;   // wide rows on sparse lattices (centre tap 1): a blur pass touches only the vertices that have a neighbour on its axis, in place (0 never, 1 when under kActiveShare of the vertices are, 2 whenever representable)
    int contract_v = 1;   // fused backward, slice + contraction: 1 = corner count compiled in (all rows in flight, all-lane contraction), 0 = the run-time form
    int splat_first = 1;   // vd = 1 splat on lattices where almost every corner owns its vertex: first-touch corners store, the rest add (0 never, 1 when m >= 0.9 nnz, 2 whenever representable, 3 = 2 without the contiguous-range store)
    // diagnostic ablations: the members always exist (one layout for both libraries), but only libplx_diag.so knows
    // their names and compiles the branches behind them (PLX_DIAG_VALUE)
    int splat_ablate = 0;   // libplx_diag.so only: 1 no value gather, 2 no stores, 4 no row-id loads
    int blur_ablate = 0;   // libplx_diag.so only: 1 no neighbour gathers, 2 no neighbour-id loads either
    int block_ablate = 0;   // libplx_diag.so only: 1 combine without the partial gathers, 2 without idx loads too, 4 without stores
};
extern Tune g_tune_defaults;
extern thread_local const Tune *tl_tune;   // the snapshot of the lattice this thread is serving (the defaults outside any entry point)

}  // namespace plx

// the names the kernel files use for the switches
#define g_sort_points (plx::tl_tune->sort_points)
#define g_readback_spin (plx::tl_tune->readback_spin)
#define g_order_compact (plx::tl_tune->order_compact)
#define g_order_zcurve (plx::tl_tune->order_zcurve)
#define g_nbr_symmetric (plx::tl_tune->nbr_symmetric)
#define g_insert_dedupe (plx::tl_tune->insert_dedupe)
#define g_compact_nbr (plx::tl_tune->compact_nbr)
#define g_insert_plane_fast (plx::tl_tune->insert_plane_fast)
#define g_vertex_order (plx::tl_tune->vertex_order)
#define g_blur_vpt (plx::tl_tune->blur_vpt)
#define g_blur_small (plx::tl_tune->blur_small)
#define g_xcd_remap (plx::tl_tune->xcd_remap)
#define g_splat_direct (plx::tl_tune->splat_direct)
#define g_blur_narrow (plx::tl_tune->blur_narrow)
#define g_blur_multi (plx::tl_tune->blur_multi)
#define g_splat_group (plx::tl_tune->splat_group)
#define g_splat_wide (plx::tl_tune->splat_wide)
#define g_blur_fuse (plx::tl_tune->blur_fuse)
#define g_blur_fuse_vec (plx::tl_tune->blur_fuse_vec)
#define g_block_path (plx::tl_tune->block_path)
#define g_scatter_store (plx::tl_tune->scatter_store)
#define g_unpermute_gather (plx::tl_tune->unpermute_gather)
#define g_block_e (plx::tl_tune->block_e)
#define g_block_dense_combine (plx::tl_tune->block_dense_combine)
#define g_nbr_bitmap (plx::tl_tune->nbr_bitmap)
#define g_nbr_window (plx::tl_tune->nbr_window)
#define g_perm_rows (plx::tl_tune->perm_rows)
#define g_splat_first (plx::tl_tune->splat_first)
#define g_insert_xcd (plx::tl_tune->insert_xcd)
#define g_order_sample (plx::tl_tune->order_sample)
#define g_embed_vrange (plx::tl_tune->embed_vrange)
#define g_reference_growth (plx::tl_tune->reference_growth)
#define g_blk_sort (plx::tl_tune->blk_sort)
#define g_assign_evid (plx::tl_tune->assign_evid)
#define g_nbr_seed (plx::tl_tune->nbr_seed)
#define g_nbr_sliced (plx::tl_tune->nbr_sliced)
#define g_contract_v (plx::tl_tune->contract_v)
#define g_blur_active (plx::tl_tune->blur_active)
#define g_splat_ablate (plx::tl_tune->splat_ablate)
#define g_blur_ablate (plx::tl_tune->blur_ablate)
#define g_block_ablate (plx::tl_tune->block_ablate)

struct plx_lattice {
    int device = 0;
    plx::Tune tn;                // the switches this lattice was built under (snapshot of the process defaults)
    bool built = false;
    bool timing = false;
    bool partial_cover = false;  // built by plx_build_merge: this rank's points do not touch every vertex
    bool local_ready = false;    // plx_build_local done, waiting for plx_build_merge
    int vertex_order = 0;        // 0 first touch, 1 Morton order of the blur-axis coordinates (this build)
    // the sorted Morton codes of this build's vertices and their bit layout, between the renumbering and the neighbour lookups
    const unsigned long long *vcode = nullptr;
    bool vcode_exact = false;
    int vcode_bits[16] = {}, vcode_lo[16] = {}, vcode_hi[16] = {};
    unsigned char vcode_pos[16][16] = {};
    int64_t merge_total_points = 0;   // plx_build_merge: points of all ranks (<= 0: unknown)
    bool single_use = false;     // built by plx_filter for one MVM: no vertex renumbering, no axis-pair tables
    bool for_merge = false;      // the local stage of a sharded build is running (vertex renumbering waits for the merge)
    bool lattice_rows = false;   // d_src / d_out rows are in lattice order (plx_set_row_order)
    bool reuse_order = false;    // plx_set_reuse_order: the NEXT build keeps the point order of the previous one (one shot)
    int64_t order_n = 0;         // what L->perm was computed for: rows, dimension, shard (0 rows: no order yet)
    int order_d = 0, order_shard = 0, order_shards = 0;
    int order_age = 0;           // builds since the order was computed from the positions themselves
    float build_ms[6] = {0, 0, 0, 0, 0, 0};

    // problem
    int64_t n = 0;
    int shard_index = 0, n_shards = 1;
    int64_t own_begin = 0, own_end = 0;   // rows of this shard; identical in original and sorted order
    int d = 0, order = 0, ntaps = 0;
    plx::TapArgs taps;
    float slice_denom = 1.f;       // 1 + 2^-d, h:507
    int64_t m = 0;                 // vertices
    int64_t mstride = 0;           // m rounded up to 64 (plane stride of the neighbour table)
    int64_t nnz = 0;               // owned CSR entries = n_own * (d+1)
    int64_t nchunks = 0;           // ceil(nnz / kSplatChunk)
    uint32_t table_mask = 0;       // capacity - 1
    int table_bits = 0;            // log2(capacity)
    uint32_t table_idmask = 0xFFFFFFFFu;   // id bits of a numbered vertex's table word (0x00FFFFFF when the word carries a fingerprint)

    // point order: perm[i] = original row of the i-th point in lattice order (shard-major, then
    // lexicographic in the rounded lattice coordinates); every per-point array below is in that order
    plx::DevBuf perm;         // uint32 [n]
    plx::DevBuf sortkey_in, sortkey_out, iota;   // uint64 [n], uint64 [n], uint32 [n]

    // build scratch
    plx::DevBuf eslot;      // uint32 [d+1][n]       hash slot of every corner
    plx::DevBuf flagmask;   // uint32 [n]            bit r set: corner r is the first touch of its vertex
    plx::DevBuf blockcnt;   // int32  [nblocks+1]    per-workgroup first-touch counts, then offsets
    plx::DevBuf table;      // uint32 [capacity]     slot -> min entry index, later slot -> vertex id
    plx::DevBuf merge_slot, merge_flags;   // uint32 [sum of all ranks' local vertex counts] (sharded build)
    plx::DevBuf counters;   // int32  [8]            {m, error flag, ...}
    plx::DevBuf sort_keys_in, sort_vals_in, sort_vals_out, sort_temp;
    plx::DevBuf slotmap;    // uint32 [capacity / 32] one bit per hash slot: occupied (neighbour lookups of large lattices)
    plx::DevBuf prank;      // uint32 [n][W]         the point records (plx_build.hip, Rec<D>): packed greedy coordinates + one rank byte per coordinate (h:427-457), lattice order
    plx::DevBuf vowner;     // uint32 [m]            first-touch corner e = p (d+1) + r of every vertex (first-touch numbering; neighbour seeding)
    plx::DevBuf vs0;        // uint32 [m]            pre-mix hash of every vertex key (sliced neighbour lookups)
    bool vs0_valid = false;
    plx::DevBuf vaxis;      // uint8  [m]            blur axis whose +1 neighbour was taken from the vertex's first-touch simplex (255: none)
    bool prank_valid = false;      // prank / flagmask describe THIS build's points and final vertex ids (plain single-process builds)
    plx::DevBuf nibmap;     // uint32 [capacity / 8]  four bits per hash slot: 0 empty, else 1 + fingerprint % 15 (XCD-sliced neighbour lookups)

    // structure
    plx::DevBuf vkeys;      // uint32 [m][DW]        packed vertex keys, in vertex id order
    plx::DevBuf vslot;      // uint32 [m]            hash-table slot of every vertex (renumbering)
    plx::DevBuf vkeys_alt, vslot_alt, vorder;   // the other halves of the renumbering pass: keys / slots in the new order, order[new] = old
    plx::DevBuf ew;         // float  [d+1][n]       barycentric weights
    plx::DevBuf evid;       // int32  [d+1][n]       vertex ids
    plx::DevBuf nbr;        // int32  [d+1][2r][mstride]
    // compacted neighbour table (vd = 1 blur on sparse lattices): per axis, per quad of 4
    // vertices a bit mask of the existing neighbours, per wave (256 vertices) the offset of
    // its first id, and the existing ids only, in (vertex, tap) order
    bool use_compact = false;
    int64_t nquads = 0, nqwaves = 0;             // ceil(m/4), ceil(nquads/64)
    int64_t compact_off[PLX_MAX_DIM + 2] = {};   // first id of each axis in cids
    plx::DevBuf cmask;      // uint32 [d+1][nquads]
    plx::DevBuf cbase;      // uint32 [d+1][nqwaves + 1]
    plx::DevBuf cids;       // int32  [total existing neighbours]
    // sparse lattices, wide rows (plx_blur.hip, round 6): per axis the vertices that have at least one neighbour on it -- the
    // only rows a pass changes when the centre tap is 1
    bool active_ready = false;
    int64_t active_off[PLX_MAX_DIM + 2] = {};    // first entry of each axis in active_list; [d+1] = total
    int64_t active_max = 0;                      // longest per-axis list
    plx::DevBuf active_list;                     // int32 [active_off[d+1]] vertex ids, ascending within an axis
    plx::DevBuf active_cnt;                      // int32 [d+1][blocks] per-block counts -> exclusive offsets
    bool pairs_ready = false, use_pairs = false;      // pair_nbr holds the composite neighbours of the axis pairs (0,1), (2,3), ... (plx_blur.hip)
    plx::DevBuf pair_nbr;   // int32  [(d+1)/2][8][mstride]  nbr_i(nbr_j(v, b), a) without the centre; -1 absent
    plx::DevBuf csr_pt;     // int32  [nnz]          local (owned) point index, sorted by vertex
    plx::DevBuf csr_row;    // int32  [nnz]          the same points numbered as the caller's rows (vd = 1 splat
                            //                       gathers straight from d_src, no sorted copy)
    plx::DevBuf csr_w;      // float  [nnz]
    plx::DevBuf csr_vid;    // int32  [nnz]          sorted vertex id of every owned corner (splat reads it at row ends)
    plx::DevBuf row_ptr;    // int32  [m+1]          produced on demand by plx_export (no kernel reads it)
    bool csr_ready = false;      // csr_* hold the vertex-sorted splat CSR of this build (built on first
                                 // use: only the multi-column kernels, the exports and lattices without block tables
                                 // need it -- plx::ensure_csr)

    // block tables (plx_block.hip): vd = 1 splat / slice on lattices whose corners share vertices.  The owned
    // points, in lattice order, are cut into blocks of blk_P points; a block's corners are sorted by vertex.
    bool use_blocks = false;
    // The block tables are built by their first user (ensure_blocks: any single-column MVM, multi-column MVMs on
    // lattices where the tables pay); the vertex-sorted half that only the block splat reads by ITS first user (ensure_s2).
    bool blocks_ready = false;   // build_blocks has run for the current build (use_blocks is decided)
    bool s2_ready = false;       // s2_* hold the block rows of the current build sorted by vertex
    int blk_P = 0, blk_E = 0, blk_cpb = 0;     // points per block, corners per thread (16 / 24), corners per full block
    int blk_max_rows = 0;                      // most distinct vertices in one block (LDS rows of the slice kernel)
    int64_t nblocks = 0, n_brows = 0;          // blocks, block rows (sum over blocks of distinct vertices)
    int64_t srow_stride = 0;                   // plane stride of srow (n_own rounded up to 8)
    plx::DevBuf bc_pt;      // uint16 [nnz]          block-local point of every corner, (block, vertex) order; bit 15: last of its row
    plx::DevBuf bc_w;       // float  [nnz]          its barycentric weight
    plx::DevBuf srow;       // uint16 [d+1][srow_stride]  block-local row of corner r of owned point p (slice)
    plx::DevBuf brow_ptr;   // int32  [nblocks+1]    first block row of every block
    plx::DevBuf brow_vid;   // int32  [n_brows]      vertex of every block row
    plx::DevBuf s2_idx;     // int32  [n_brows]      block rows sorted by vertex; bit 31: last row of its vertex
    plx::DevBuf s2_ptr;     // int32  [m+1]          block rows of vertex v: s2_idx[s2_ptr[v] .. s2_ptr[v+1])
    plx::DevBuf s2_vid;     // int32  [n_brows]      vertex of every sorted block row (read at row ends)
    plx::DevBuf s2_wave;    // int32  [n_s2waves+1]  first block row of every combine wave
    plx::DevBuf s2_wave_v;  // int32  [n_s2waves+1]  vertex of that row
    int64_t n_s2waves = 0;
    plx::DevBuf partial;    // float  [n_brows]      per-MVM block-row sums
    plx::DevBuf inv_perm;   // uint32 [n_own]        lattice-order position of every caller row of the shard
    bool inv_perm_ready = false;
    // first-touch splat (plx_first.hip): vd = 1 on lattices where almost every corner owns its vertex
    bool flags_valid = false;    // flagmask holds the first-touch bits of THIS build's points (plain single-process builds)
    bool first_ready = false, use_first = false;
    int64_t n_extra = 0;         // corners that are not the first touch of their vertex (nnz - m)
    plx::DevBuf ex_vid, ex_pt, ex_w, ex_keys;   // int32 / int32 / float [n_extra] the extras sorted by vertex; sort scratch

    // plx_tune("reference_growth", 1): what the host replay of the reference's table layout found (plx_replay.hip)
    struct Replay {
        bool active = false;          // this build was replayed
        int64_t m_reference = 0;      // entries the reference's table holds (hashTable.size(): duplicates included)
        int grows = 0;                // doublings of the reference's table (splat + blur)
        int n_dropped = 0;            // (point, corner) lookups whose splat contribution the reference loses
        int n_invisible = 0;          // vertices no blur-time lookup finds
        bool blur_miss = false;       // blur()'s first lookup fell on a doubling and read an existing neighbour as absent
        int blur_miss_vertex = 0;
        bool inexact = false;         // a combination this representation cannot express (see plx_replay.hip)
    } replay;
    plx::DevBuf ew_splat;         // float [d+1][n]  the splat's copy of ew with the dropped lookups zeroed (replay mode)
    plx::DevBuf replay_vat, replay_list, replay_invisible;   // int32 scratch / lists of the replay
    plx::DevBuf replay_keys;      // the event form's sort buffers: uint32 [4][m] (first lookup, vertex; ping-pong) + uint64 [m] hashes + a counter

    // plx_filter_onehot (plx_onehot.hip): frontier of the non-zero vertex rows
    plx::DevBuf oh_pos, oh_list, oh_cnt;      // int32 [m] position of a vertex in the list or -1; int32 [m] the list; int32 counters

    // apply workspace
    plx::DevBuf head_partial, tail_partial;   // float [nchunks][vd]
    plx::DevBuf val_a, val_b;                 // float [m][vdp]   (vdp = value row stride, plx_values_stride)
    plx::DevBuf ssrc;                         // float [n_own][vdp] right-hand side in lattice order
    plx::DevBuf rec;                          // float [n_own][2L+d+2 rounded up to 4] packed (g, src, x) records (backward)

    // kernels launched by the last splat / blur / slice on this lattice (names as rocprofv3 shows them, '+'-joined)
    const char *kn_splat = "", *kn_blur = "", *kn_slice = "";

    int32_t *h_pinned = nullptr;   // pinned host staging (exports)
    int *h_mail = nullptr;         // mailbox of read_back: coherent pinned host memory, word 0 = sequence number, then up to 62 values
    int mail_seq = 0;
    hipEvent_t ev[8] = {};
    // per-launch timing of plx_apply (timing on): events between consecutive launches
    hipEvent_t tev[PLX_MAX_DIM + 8] = {};
    int tev_n = 0;
};

namespace plx {

int ensure(DevBuf &b, size_t bytes);
void release(DevBuf &b);
// PLX_ERR_STATE (with the plx_prepare hint) when `stream` is being captured into a graph: for the table builders
int refuse_under_capture(hipStream_t stream, const char *what);

// plx_build.hip
int build_impl(plx_lattice *L, const float *d_ref, hipStream_t stream);
int build_local_impl(plx_lattice *L, const float *d_ref, hipStream_t stream);
int build_merge_impl(plx_lattice *L, const uint32_t *d_all_keys, const int64_t *h_counts, int n_ranks, int my_rank,
                     hipStream_t stream);
int read_back(plx_lattice *L, const int *d_src, int count, int *h_dst, hipStream_t stream);   // a few device ints to the host, no stream synchronisation
int export_row_ptr(plx_lattice *L, hipStream_t stream);   // fills L->row_ptr on demand (plx_export only)
int ensure_csr(plx_lattice *L, hipStream_t stream);       // vertex-sorted splat CSR of the current build, built once on demand
// plx_block.hip (block tables + the vd = 1 kernels that use them)
int build_blocks(plx_lattice *L, hipStream_t stream);
int ensure_blocks(plx_lattice *L, hipStream_t stream);    // the block tables of the current build (built by their first user)
int ensure_s2(plx_lattice *L, hipStream_t stream);
int ensure_inv_perm(plx_lattice *L, hipStream_t stream);  // caller row -> lattice position (gather-out passes of slice)
int unpermute_rows(plx_lattice *L, const float *d_tmp, int vd, float *d_out, const float *d_affine, const float *d_src,
                   hipStream_t stream);              // out[j][0..vd) = tmp[inv_perm[j]][0..vd) (+ affine epilogue)        // ... and their vertex-sorted half (block splat only)
int choose_paths(plx_lattice *L, int vd, hipStream_t stream, bool *splat_blocks, bool *slice_blocks);
int prepare_tables(plx_lattice *L, int vd, hipStream_t stream);
int splat_block_impl(plx_lattice *L, const float *d_src, float *d_values, hipStream_t stream);
// plx_replay.hip (plx_tune "reference_growth")
int replay_simulate(plx_lattice *L, hipStream_t stream);
int replay_patch_tables(plx_lattice *L, hipStream_t stream);
int replay_nocentre_fix(plx_lattice *L, const float *d_old, float *d_new, int vdp, hipStream_t stream);
// barycentric weights as the SPLAT side reads them (the slice side always reads ew)
inline const float *splat_weights(const plx_lattice *L) { return L->replay.active ? L->ew_splat.as<float>() : L->ew.as<float>(); }
// plx_first.hip
int ensure_first(plx_lattice *L, hipStream_t stream);
int splat_first_impl(plx_lattice *L, const float *d_src, float *d_values, hipStream_t stream);
int slice_block_impl(plx_lattice *L, const float *d_values, float *d_out, hipStream_t stream, const float *d_affine,
                     const float *d_src);
// plx_sort.hip: plx::radix (plx_radix.h) behind plain functions -- stable LSD sort of bits [0, end_bit), ping-ponging between
// the two buffer pairs; *in_second: 0 = result in keys_a / vals_a, 1 = in keys_b / vals_b
size_t radix_temp_bytes(int64_t n);
int radix_sort_pairs64(void *temp, uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n, int end_bit,
                       int *in_second, hipStream_t stream);
// first_keys (optional): the first pass reads its keys from there (left untouched) with the values 0, 1, 2, ... implied
int radix_sort_pairs32(void *temp, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n, int end_bit,
                       int *in_second, hipStream_t stream, const uint32_t *first_keys = nullptr);
int selftest_sort(int64_t n, int key_bytes, int end_bit, uint64_t seed, hipStream_t stream, int64_t *mismatches);
// block tables built in LDS, one workgroup per block (256-thread blocks): sort by vertex + every per-corner record;
// the block's vertex list lands in rows_tmp[b * cpb + row] and is compacted once the row offsets are scanned
int sort_fill_blocks_lds(const int *evid, const float *ew, int n, int own_begin, int n_own, int P, int d1, int cpb, int vbits,
                         int64_t m_vertices, int ipt, int64_t nblocks, uint16_t *bc_pt, float *bc_w, uint16_t *srow, int64_t sstride, int *rows_tmp,
                         int *rows, hipStream_t stream);
int compact_block_rows(const int *rows_tmp, const int *brow_ptr, int cpb, int64_t nblocks, int *brow_vid, hipStream_t stream);
// plx_splat.hip / plx_blur.hip / plx_slice.hip
int splat_impl(plx_lattice *L, const float *d_src, int vd, float *d_values, hipStream_t stream);
int splat_onehot_impl(plx_lattice *L, const int *d_cand, int nb, int vd, float *d_values, hipStream_t stream);
// plx_onehot.hip: K e_p for nb points (splat, blur, slice of one-hot columns); sparse != 0: on the frontier of the non-zeros where the lattice allows
int filter_onehot_impl(plx_lattice *L, const int *d_cand, int nb, int vd, float *d_values, float *d_scratch, float *d_out,
                       int sparse, int *d_frontier, hipStream_t stream);
int blur_impl(plx_lattice *L, float *d_values, float *d_scratch, int vd, int *result_in_scratch,
              hipStream_t stream);
int build_blur_pairs(plx_lattice *L, hipStream_t stream);   // composite neighbour tables of the two-axes-per-launch blur
int ensure_blur_pairs(plx_lattice *L, hipStream_t stream);  // ... built on first use (multi-column blurs, plx_filter_onehot)
int slice_impl(plx_lattice *L, const float *d_values, int vd, float *d_out, hipStream_t stream,
               const float *d_affine = nullptr, const float *d_src = nullptr, float *d_dot_partial = nullptr);
// plx_linalg.hip: out[c] = sum over nblocks of partial[k * vd + c], fixed order
int coldot_final(const float *d_partial, int nblocks, int vd, float *d_out, hipStream_t stream);
int backward_impl(plx_lattice *L, const float *d_g, const float *d_src, const float *d_x, int nrhs, float *d_grad_x,
                  float *d_grad_src, hipStream_t stream);
int splat_stack_impl(plx_lattice *L, const float *d_g, const float *d_src, const float *d_x, int nrhs, float *d_values,
                     hipStream_t stream);
// floats per packed (g, src, x, 0, 1) record of the fused position gradient, a whole number of 16-byte vectors
inline int backward_record_width(int nrhs, int d) { return (2 * nrhs + d + 2 + 3) & ~3; }

// record the next apply-timing event (no-op unless timing is on)
inline void tmark(plx_lattice *L, hipStream_t stream)
{
    if (L->timing && L->tev_n < (int)(sizeof(L->tev) / sizeof(L->tev[0])))
        (void)hipEventRecord(L->tev[L->tev_n++], stream);
}

// name -> member of Tune (plx_tune)
struct Tunable { const char *name; int Tune::*member; };
const Tunable *tunables();

// value-row stride in floats: 1 for a single column, else the column count rounded up to 4 so
// that every row is a whole number of 16-byte vectors
inline int values_stride(int vd) { return vd == 1 ? 1 : (vd + 3) & ~3; }

// contiguous near-equal row blocks: the first n % shards blocks get one extra row
inline void shard_range(int64_t n, int shards, int index, int64_t *lo, int64_t *hi)
{
    const int64_t base = n / shards, extra = n % shards;
    *lo = index * base + (index < extra ? index : extra);
    *hi = *lo + base + (index < extra ? 1 : 0);
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace plx

// plx_api.hip -- extern "C" entry points declared in include/plx.h.
#include "plx_internal.h"

#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace plx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// The stream of the entry point that is running on this thread: buffers grow stream-ordered on it (hipFreeAsync +
// hipMallocAsync), so the first MVM that needs a wider workspace does not synchronise the device the way hipFree does.
static thread_local hipStream_t g_entry_stream = nullptr;
// ... and the switches of the lattice it serves (Tune): a build takes a fresh snapshot of the process defaults first.
struct EntryScope {
    hipStream_t prev;
    const Tune *prev_tune;
    EntryScope(plx_lattice *L, void *s) : prev(g_entry_stream), prev_tune(tl_tune)
    {
        g_entry_stream = (hipStream_t)s;
        tl_tune = L ? &L->tn : &g_tune_defaults;
    }
    ~EntryScope() { g_entry_stream = prev; tl_tune = prev_tune; }
};

int refuse_under_capture(hipStream_t stream, const char *what)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("%s has to be built or grown (an allocation, for tables also a host read-back) and the stream is being "
                  "captured: call plx_prepare(lat, vd, stream) and run one MVM of this width before the capture", what);
        return PLX_ERR_STATE;
    }
    return PLX_OK;
}

int ensure(DevBuf &b, size_t bytes)
{
    if (bytes == 0) bytes = 4;
    if (b.cap >= bytes) return PLX_OK;
    const hipStream_t s = g_entry_stream;
    // growing a buffer inside a capture would hand the lattice an address that belongs to the graph (a captured
    // hipMallocAsync is a memory node: the memory exists while the graph runs, not afterwards)
    PLX_TRY(refuse_under_capture(s, "a device buffer of this lattice"));
    if (b.p) {
        // work queued on another stream may still read the old buffer.  The stored handle is only COMPARED, never used:
        // its owner may have destroyed that stream since (and the runtime may have recycled the handle), so the wait is
        // for the device.  Growth across streams is rare -- a lattice normally lives on one stream.
        if (b.owner != s) PLX_HIP_TRY(hipDeviceSynchronize());
        PLX_HIP_TRY(hipFreeAsync(b.p, s));
        b.p = nullptr;
        b.cap = 0;
    }
    PLX_HIP_TRY(hipMallocAsync(&b.p, bytes, s));
    b.cap = bytes;
    b.owner = s;
    return PLX_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

static std::vector<DevBuf *> all_bufs(plx_lattice *L)
{
    return {&L->eslot, &L->flagmask, &L->blockcnt, &L->table, &L->counters,
            &L->sort_keys_in, &L->slotmap, &L->nibmap, &L->prank, &L->vaxis, &L->vs0, &L->vowner, &L->ew_splat, &L->replay_vat, &L->replay_list, &L->replay_invisible, &L->replay_keys, &L->active_list, &L->active_cnt, &L->oh_pos, &L->oh_list, &L->oh_cnt, &L->ex_vid, &L->ex_pt, &L->ex_w, &L->ex_keys, &L->sort_vals_in, &L->sort_vals_out, &L->sort_temp,
            &L->vkeys, &L->ew, &L->evid, &L->nbr, &L->csr_pt, &L->csr_row, &L->csr_w, &L->csr_vid, &L->row_ptr,
            &L->head_partial, &L->tail_partial, &L->val_a, &L->val_b, &L->ssrc, &L->rec, &L->perm, &L->iota, &L->cmask, &L->cbase, &L->cids, &L->merge_slot, &L->merge_flags,
            &L->sortkey_in, &L->sortkey_out,
            &L->bc_pt, &L->bc_w, &L->srow, &L->brow_ptr, &L->brow_vid, &L->s2_idx, &L->s2_ptr, &L->s2_vid, &L->s2_wave, &L->s2_wave_v, &L->partial, &L->pair_nbr, &L->inv_perm, &L->vslot, &L->vkeys_alt, &L->vslot_alt, &L->vorder};
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace plx

using namespace plx;

extern "C" {

const char *plx_strerror(int code)
{
    switch (code) {
    case PLX_OK: return "ok";
    case PLX_ERR_INVALID: return "invalid argument";
    case PLX_ERR_HIP: return "HIP runtime error";
    case PLX_ERR_KEY_RANGE: return "lattice coordinate outside the int16 key range";
    case PLX_ERR_DIM: return "dimension or order outside the compiled range";
    case PLX_ERR_STATE: return "lattice not built or size mismatch";
    case PLX_ERR_TOO_LARGE: return "n*(d+1) exceeds the 31-bit entry index";
    default: return "unknown error";
    }
}

const char *plx_last_error(void) { return g_err; }

/* minor = the round that last extended the C ABI */
const char *plx_version(void) { return "libplx 0.8.0 gfx950"; }

int plx_create(int device, plx_lattice **out)
{
    if (!out) { set_error("plx_create: out is NULL"); return PLX_ERR_INVALID; }
    DeviceGuard g(device);
    if (!g.ok) { set_error("plx_create: cannot select device %d", device); return PLX_ERR_HIP; }
    plx_lattice *L = new plx_lattice();
    L->device = device;
    if (hipHostMalloc((void **)&L->h_pinned, 256, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&L->h_mail, 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        set_error("plx_create: hipHostMalloc failed");
        if (L->h_pinned) (void)hipHostFree(L->h_pinned);
        delete L;
        return PLX_ERR_HIP;
    }
    memset(L->h_mail, 0, 256);
    // on failure plx_destroy releases whatever was created so far (events not yet created are null)
    for (auto &e : L->ev)
        if (hipEventCreate(&e) != hipSuccess) { e = nullptr; plx_destroy(L); set_error("plx_create: hipEventCreate failed"); return PLX_ERR_HIP; }
    for (auto &e : L->tev)
        if (hipEventCreate(&e) != hipSuccess) { e = nullptr; plx_destroy(L); set_error("plx_create: hipEventCreate failed"); return PLX_ERR_HIP; }
    *out = L;
    return PLX_OK;
}

void plx_destroy(plx_lattice *L)
{
    if (!L) return;
    DeviceGuard g(L->device);
    for (DevBuf *b : all_bufs(L)) release(*b);
    if (L->h_pinned) (void)hipHostFree(L->h_pinned);
    if (L->h_mail) (void)hipHostFree(L->h_mail);
    for (auto &e : L->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : L->tev) if (e) (void)hipEventDestroy(e);
    delete L;
}

// single_use: the lattice serves ONE MVM (plx_filter, the reference's one-shot contract): what only pays back over several
// MVMs is left out of the build -- the vertex renumbering (+0.16 ms for -17 us per MVM at N = 1e6), the axis-pair tables
// and, for a multi-column MVM, the block tables.
static int build_entry(plx_lattice *L, const float *d_ref, int64_t n, int d, const float *h_taps, int ntaps,
                       int shard_index, int n_shards, void *stream, bool single_use)
{
    EntryScope sc(L, stream);
    if (!L || !d_ref || !h_taps) { set_error("plx_build: NULL argument"); return PLX_ERR_INVALID; }
    if (n <= 0) { set_error("plx_build: n = %lld must be positive", (long long)n); return PLX_ERR_INVALID; }
    if (d < 1 || d > PLX_MAX_DIM) { set_error("plx_build: d = %d outside 1..%d", d, PLX_MAX_DIM); return PLX_ERR_DIM; }
    if (ntaps < 1 || (ntaps % 2) == 0) { set_error("plx_build: tap count %d must be odd", ntaps); return PLX_ERR_INVALID; }
    if (ntaps / 2 > PLX_MAX_ORDER) { set_error("plx_build: order %d > %d", ntaps / 2, PLX_MAX_ORDER); return PLX_ERR_DIM; }
    if (n_shards < 1 || n_shards > 256 || shard_index < 0 || shard_index >= n_shards) {
        set_error("plx_build: shard %d of %d is not a valid shard (1..256 shards)", shard_index, n_shards);
        return PLX_ERR_INVALID;
    }
    if (n * (int64_t)(d + 1) >= (1ll << 31) - 1024) {
        set_error("plx_build: n*(d+1) = %lld does not fit the 31-bit entry index", (long long)(n * (d + 1)));
        return PLX_ERR_TOO_LARGE;
    }
    DeviceGuard g(L->device);
    if (!g.ok) { set_error("plx_build: cannot select device %d", L->device); return PLX_ERR_HIP; }
    L->tn = g_tune_defaults;       // the snapshot of the process defaults this build (and every later call on it) runs under:
                                   // taken only once the arguments are accepted -- a rejected call leaves a built lattice as it was
    L->built = false;
    L->local_ready = false;
    L->single_use = single_use;
    L->n = n; L->d = d; L->ntaps = ntaps; L->order = ntaps / 2;
    L->shard_index = shard_index; L->n_shards = n_shards;
    shard_range(n, n_shards, shard_index, &L->own_begin, &L->own_end);
    memset(&L->taps, 0, sizeof(L->taps));
    for (int i = 0; i < ntaps; ++i) L->taps.c[i] = h_taps[i];
    int rc = build_impl(L, d_ref, (hipStream_t)stream);
    if (rc == PLX_OK) L->built = true;
    return rc;
}

int plx_build(plx_lattice *L, const float *d_ref, int64_t n, int d, const float *h_taps, int ntaps,
              int shard_index, int n_shards, void *stream)
{
    return build_entry(L, d_ref, n, d, h_taps, ntaps, shard_index, n_shards, stream, false);
}

// ---- sharded build: local stage, key exchange by the caller, merge stage ------------------------

int plx_build_local(plx_lattice *L, const float *d_ref_local, int64_t n_local, int d, const float *h_taps, int ntaps,
                    void *stream)
{
    EntryScope sc(L, stream);
    if (!L || !d_ref_local || !h_taps) { set_error("plx_build_local: NULL argument"); return PLX_ERR_INVALID; }
    if (n_local <= 0) { set_error("plx_build_local: n_local = %lld must be positive", (long long)n_local); return PLX_ERR_INVALID; }
    if (d < 1 || d > PLX_MAX_DIM) { set_error("plx_build_local: d = %d outside 1..%d", d, PLX_MAX_DIM); return PLX_ERR_DIM; }
    if (ntaps < 1 || (ntaps % 2) == 0) { set_error("plx_build_local: tap count %d must be odd", ntaps); return PLX_ERR_INVALID; }
    if (ntaps / 2 > PLX_MAX_ORDER) { set_error("plx_build_local: order %d > %d", ntaps / 2, PLX_MAX_ORDER); return PLX_ERR_DIM; }
    if (n_local * (int64_t)(d + 1) >= (1ll << 31) - 1024) {
        set_error("plx_build_local: n*(d+1) = %lld does not fit the 31-bit entry index", (long long)(n_local * (d + 1)));
        return PLX_ERR_TOO_LARGE;
    }
    DeviceGuard g(L->device);
    if (!g.ok) { set_error("plx_build_local: cannot select device %d", L->device); return PLX_ERR_HIP; }
    L->tn = g_tune_defaults;
    L->built = false;
    L->local_ready = false;
    L->n = n_local; L->d = d; L->ntaps = ntaps; L->order = ntaps / 2;
    L->shard_index = 0; L->n_shards = 1;
    L->own_begin = 0; L->own_end = n_local;
    memset(&L->taps, 0, sizeof(L->taps));
    for (int i = 0; i < ntaps; ++i) L->taps.c[i] = h_taps[i];
    L->for_merge = true;
    L->single_use = false;
    int rc = build_local_impl(L, d_ref_local, (hipStream_t)stream);
    L->for_merge = false;
    if (rc == PLX_OK) L->local_ready = true;
    return rc;
}

int plx_key_words(int d) { return (d >= 1 && d <= PLX_MAX_DIM) ? (d + 1) / 2 : -1; }

int64_t plx_local_vertices(const plx_lattice *L) { return (L && (L->local_ready || L->built)) ? L->m : -1; }

int plx_copy_local_keys(plx_lattice *L, void *d_dst, void *stream)
{
    EntryScope sc(L, stream);
    if (!L || !d_dst) { set_error("plx_copy_local_keys: NULL argument"); return PLX_ERR_INVALID; }
    if (!L->local_ready) { set_error("plx_copy_local_keys: call plx_build_local first"); return PLX_ERR_STATE; }
    DeviceGuard g(L->device);
    PLX_HIP_TRY(hipMemcpyAsync(d_dst, L->vkeys.p, (size_t)L->m * plx_key_words(L->d) * 4, hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
    return PLX_OK;
}

int plx_build_merge(plx_lattice *L, const void *d_all_keys, const int64_t *h_counts, int n_ranks, int my_rank, int64_t total_points,
                    void *stream)
{
    EntryScope sc(L, stream);
    if (!L || !d_all_keys || !h_counts) { set_error("plx_build_merge: NULL argument"); return PLX_ERR_INVALID; }
    if (!L->local_ready) { set_error("plx_build_merge: call plx_build_local first"); return PLX_ERR_STATE; }
    if (n_ranks < 1 || my_rank < 0 || my_rank >= n_ranks) {
        set_error("plx_build_merge: rank %d of %d", my_rank, n_ranks);
        return PLX_ERR_INVALID;
    }
    DeviceGuard g(L->device);
    if (!g.ok) { set_error("plx_build_merge: cannot select device %d", L->device); return PLX_ERR_HIP; }
    L->merge_total_points = total_points;
    int rc = build_merge_impl(L, (const uint32_t *)d_all_keys, h_counts, n_ranks, my_rank, (hipStream_t)stream);
    L->local_ready = false;
    if (rc == PLX_OK) L->built = true;
    return rc;
}

int64_t plx_num_points(const plx_lattice *L) { return L ? L->n : -1; }
int64_t plx_num_owned(const plx_lattice *L) { return L ? L->own_end - L->own_begin : -1; }
int64_t plx_num_vertices(const plx_lattice *L) { return (L && L->built) ? L->m : -1; }

int plx_reference_growth_info(const plx_lattice *L, int64_t *h_out6)
{
    if (!L || !L->built || !h_out6) { set_error("plx_reference_growth_info: lattice not built"); return PLX_ERR_STATE; }
    const auto &rp = L->replay;
    h_out6[0] = rp.active ? 1 : 0;
    h_out6[1] = rp.active ? rp.m_reference : L->m;
    h_out6[2] = rp.n_dropped;
    h_out6[3] = rp.n_invisible;
    h_out6[4] = rp.blur_miss ? 1 : 0;
    h_out6[5] = rp.inexact ? 1 : 0;
    return PLX_OK;
}
int plx_dim(const plx_lattice *L) { return L ? L->d : -1; }
int plx_order(const plx_lattice *L) { return L ? L->order : -1; }

int plx_set_row_order(plx_lattice *L, int lattice_order)
{
    if (!L) { set_error("plx_set_row_order: NULL lattice"); return PLX_ERR_INVALID; }
    L->lattice_rows = lattice_order != 0;
    return PLX_OK;
}

int plx_set_reuse_order(plx_lattice *L, int on)
{
    if (!L) { set_error("plx_set_reuse_order: NULL lattice"); return PLX_ERR_INVALID; }
    L->reuse_order = on != 0;
    return PLX_OK;
}

int plx_order_age(const plx_lattice *L) { return L ? (L->order_n > 0 ? L->order_age : -1) : -1; }

int plx_values_stride(int vd) { return vd >= 1 ? values_stride(vd) : -1; }

int64_t plx_device_bytes(const plx_lattice *L)
{
    if (!L) return -1;
    int64_t total = 0;
    for (DevBuf *b : all_bufs(const_cast<plx_lattice *>(L))) total += (int64_t)b->cap;
    return total;
}

static int check_apply(const plx_lattice *L, const void *a, const void *b, int vd, const char *who)
{
    if (!L || !a || !b) { set_error("%s: NULL argument", who); return PLX_ERR_INVALID; }
    if (!L->built) { set_error("%s: lattice not built", who); return PLX_ERR_STATE; }
    if (vd < 1) { set_error("%s: vd = %d must be positive", who, vd); return PLX_ERR_INVALID; }
    if ((int64_t)L->m * values_stride(vd) >= (1ll << 31) ||
        (int64_t)(L->own_end - L->own_begin) * values_stride(vd) >= (1ll << 31)) {
        set_error("%s: m*vd or n*vd exceeds 2^31 elements; split the columns", who);
        return PLX_ERR_TOO_LARGE;
    }
    return PLX_OK;
}

int plx_splat(plx_lattice *L, const float *d_src, int vd, float *d_values, void *stream)
{
    EntryScope sc(L, stream);
    // a rank that owns no rows has no source block: only the accumulator is required
    if (L && L->built && L->own_end == L->own_begin && !d_src) d_src = d_values;
    PLX_TRY(check_apply(L, d_src, d_values, vd, "plx_splat"));
    DeviceGuard g(L->device);
    return splat_impl(L, d_src, vd, d_values, (hipStream_t)stream);
}

int plx_splat_onehot(plx_lattice *L, const int32_t *d_points, int nb, int vd, float *d_values, void *stream)
{
    EntryScope sc(L, stream);
    PLX_TRY(check_apply(L, d_points, d_values, vd, "plx_splat_onehot"));
    if (nb < 1 || nb > vd) { set_error("plx_splat_onehot: %d one-hot columns in %d", nb, vd); return PLX_ERR_INVALID; }
    if (L->n_shards != 1 || L->partial_cover) { set_error("plx_splat_onehot: single-shard lattices only"); return PLX_ERR_STATE; }
    DeviceGuard g(L->device);
    return splat_onehot_impl(L, d_points, nb, vd, d_values, (hipStream_t)stream);
}

int plx_filter_onehot(plx_lattice *L, const int32_t *d_points, int nb, int vd, float *d_values, float *d_scratch, float *d_out,
                      int sparse, int32_t *d_frontier, void *stream)
{
    EntryScope sc(L, stream);
    PLX_TRY(check_apply(L, d_values, d_scratch, vd, "plx_filter_onehot"));
    if (!d_points || !d_out) { set_error("plx_filter_onehot: NULL argument"); return PLX_ERR_INVALID; }
    if (d_values == d_scratch) { set_error("plx_filter_onehot: d_values and d_scratch must be different buffers"); return PLX_ERR_INVALID; }
    if (nb < 1 || nb > vd || nb > 16) { set_error("plx_filter_onehot: %d one-hot columns in %d (at most 16)", nb, vd); return PLX_ERR_INVALID; }
    if (L->n_shards != 1 || L->partial_cover) { set_error("plx_filter_onehot: single-shard lattices only"); return PLX_ERR_STATE; }
    DeviceGuard g(L->device);
    return filter_onehot_impl(L, d_points, nb, vd, d_values, d_scratch, d_out, sparse, d_frontier, (hipStream_t)stream);
}

int plx_blur(plx_lattice *L, float *d_values, float *d_scratch, int vd, int *result_in_scratch, void *stream)
{
    EntryScope sc(L, stream);
    PLX_TRY(check_apply(L, d_values, d_scratch, vd, "plx_blur"));
    if (!result_in_scratch) { set_error("plx_blur: result_in_scratch is NULL"); return PLX_ERR_INVALID; }
    DeviceGuard g(L->device);
    return blur_impl(L, d_values, d_scratch, vd, result_in_scratch, (hipStream_t)stream);
}

int plx_slice(plx_lattice *L, const float *d_values, int vd, float *d_out, void *stream)
{
    EntryScope sc(L, stream);
    if (L && L->built && L->own_end == L->own_begin && !d_out) d_out = const_cast<float *>(d_values);
    PLX_TRY(check_apply(L, d_values, d_out, vd, "plx_slice"));
    DeviceGuard g(L->device);
    return slice_impl(L, d_values, vd, d_out, (hipStream_t)stream);
}

static int apply_common(plx_lattice *L, const float *d_src, int vd, float *d_out, const float *d_affine, void *stream,
                        const char *who);

int plx_apply_affine(plx_lattice *L, const float *d_src, int vd, float *d_out, const float *d_scale_shift, void *stream)
{
    if (!d_scale_shift) { set_error("plx_apply_affine: NULL scale/shift"); return PLX_ERR_INVALID; }
    if (d_src == d_out) { set_error("plx_apply_affine: d_out must not alias d_src (rows are written in a different order)"); return PLX_ERR_INVALID; }
    return apply_common(L, d_src, vd, d_out, d_scale_shift, stream, "plx_apply_affine");
}

int plx_apply(plx_lattice *L, const float *d_src, int vd, float *d_out, void *stream)
{
    return apply_common(L, d_src, vd, d_out, nullptr, stream, "plx_apply");
}

static int affine_dot_tiles(const plx_lattice *L, int vd)
{
    const int64_t n_own = L->own_end - L->own_begin;
    return ceil_div(n_own * (values_stride(vd) / 4), kBlock);
}

int64_t plx_affine_dot_work_floats(const plx_lattice *L, int vd)
{
    if (!L || !L->built || vd < 2 || vd > 256) return -1;
    return (int64_t)affine_dot_tiles(L, vd) * values_stride(vd);
}

int plx_apply_affine_dot(plx_lattice *L, const float *d_src, int vd, float *d_out, const float *d_scale_shift,
                         float *d_dot, float *d_work, void *stream)
{
    EntryScope sc(L, stream);
    if (!d_scale_shift || !d_work) { set_error("plx_apply_affine_dot: NULL argument"); return PLX_ERR_INVALID; }
    if (d_src == d_out) { set_error("plx_apply_affine_dot: d_out must not alias d_src"); return PLX_ERR_INVALID; }
    if (vd < 2 || vd > 256) { set_error("plx_apply_affine_dot: vd = %d outside 2..256 (use plx_apply_affine + plx_coldot)", vd); return PLX_ERR_INVALID; }
    PLX_TRY(check_apply(L, d_src, d_out, vd, "plx_apply_affine_dot"));
    DeviceGuard g(L->device);
    const int vdp = values_stride(vd);
    PLX_TRY(ensure(L->val_a, (size_t)L->m * vdp * 4));
    PLX_TRY(ensure(L->val_b, (size_t)L->m * vdp * 4));
    hipStream_t s = (hipStream_t)stream;
    L->tev_n = 0;
    tmark(L, s);
    PLX_TRY(splat_impl(L, d_src, vd, L->val_a.as<float>(), s));
    int in_b = 0;
    PLX_TRY(blur_impl(L, L->val_a.as<float>(), L->val_b.as<float>(), vd, &in_b, s));
    PLX_TRY(slice_impl(L, in_b ? L->val_b.as<float>() : L->val_a.as<float>(), vd, d_out, s, d_scale_shift, d_src, d_work));
    if (!d_dot) return PLX_OK;          // the per-tile partial sums stay in d_work (plx_cg_step_update_fused adds them up itself)
    return coldot_final(d_work, affine_dot_tiles(L, vd), vdp, d_dot, s);
}

int plx_affine_dot_tiles(const plx_lattice *L, int vd)
{
    if (!L || !L->built || vd < 2 || vd > 256) return -1;
    return affine_dot_tiles(L, vd);
}

static int apply_common(plx_lattice *L, const float *d_src, int vd, float *d_out, const float *d_affine, void *stream,
                        const char *who)
{
    EntryScope sc(L, stream);
    PLX_TRY(check_apply(L, d_src, d_out, vd, who));
    DeviceGuard g(L->device);
    PLX_TRY(ensure(L->val_a, (size_t)L->m * values_stride(vd) * 4));
    PLX_TRY(ensure(L->val_b, (size_t)L->m * values_stride(vd) * 4));
    hipStream_t s = (hipStream_t)stream;
    L->tev_n = 0;
    tmark(L, s);
    PLX_TRY(splat_impl(L, d_src, vd, L->val_a.as<float>(), s));
    int in_b = 0;
    PLX_TRY(blur_impl(L, L->val_a.as<float>(), L->val_b.as<float>(), vd, &in_b, s));
    return slice_impl(L, in_b ? L->val_b.as<float>() : L->val_a.as<float>(), vd, d_out, s, d_affine, d_src);
}

int plx_apply_backward(plx_lattice *L, const float *d_g, const float *d_src, const float *d_ref, int nrhs,
                       float *d_grad_ref, float *d_grad_src, void *stream)
{
    EntryScope sc(L, stream);
    if (!L) { set_error("plx_apply_backward: NULL lattice"); return PLX_ERR_INVALID; }
    if (!L->built) { set_error("plx_apply_backward: lattice not built"); return PLX_ERR_STATE; }
    if (L->n_shards != 1 || L->partial_cover) {
        set_error("plx_apply_backward: single-shard lattices only (a sharded caller needs the vertex all-reduce between splat and blur)");
        return PLX_ERR_STATE;
    }
    if (!d_g || !d_src || !d_ref || !d_grad_ref) { set_error("plx_apply_backward: NULL argument"); return PLX_ERR_INVALID; }
    const int nch = values_stride(2 * nrhs * (1 + L->d)) / 4;
    if (nrhs < 1 || nch < 32 || nch > 128 || 2 * nrhs + L->d + 2 > 64) {
        set_error("plx_apply_backward: nrhs = %d, d = %d (%d columns) is outside the fused kernels' range (125..512 "
                  "columns, 2*nrhs + d <= 62); use plx_backward_stack + plx_apply + plx_backward_contract", nrhs, L->d,
                  2 * nrhs * (1 + L->d));
        return PLX_ERR_INVALID;
    }
    DeviceGuard g(L->device);
    return backward_impl(L, d_g, d_src, d_ref, nrhs, d_grad_ref, d_grad_src, (hipStream_t)stream);
}

int plx_filter(plx_lattice *scratch, const float *d_src, const float *d_ref, int64_t n, int d, int vd,
               const float *h_taps, int ntaps, float *d_out, void *stream)
{
    plx_lattice *L = scratch;
    if (!L) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { set_error("plx_filter: hipGetDevice failed"); return PLX_ERR_HIP; }
        PLX_TRY(plx_create(dev, &L));
    }
    int rc = build_entry(L, d_ref, n, d, h_taps, ntaps, 0, 1, stream, true);
    if (rc == PLX_OK) rc = plx_apply(L, d_src, vd, d_out, stream);
    if (!scratch) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        plx_destroy(L);
    }
    return rc;
}

int64_t plx_export_bytes(const plx_lattice *L, int which)
{
    if (!L || !L->built) return -1;
    const int64_t n = L->n, m = L->m, d1 = L->d + 1;
    switch (which) {
    case PLX_ARRAY_KEYS: return m * L->d * 2;
    case PLX_ARRAY_ENTRY_VERTEX: return d1 * n * 4;
    case PLX_ARRAY_ENTRY_WEIGHT: return d1 * n * 4;
    case PLX_ARRAY_NEIGHBORS: return d1 * 2 * L->order * m * 4;
    case PLX_ARRAY_ROW_PTR: return (m + 1) * 4;
    case PLX_ARRAY_CSR_POINT: return L->nnz * 4;
    case PLX_ARRAY_CSR_WEIGHT: return L->nnz * 4;
    case PLX_ARRAY_POINT_PERM: return n * 4;
    default: return -1;
    }
}

int plx_export(plx_lattice *L, int which, void *h_dst, int64_t bytes, void *stream)
{
    EntryScope sc(L, stream);
    if (!L || !h_dst) { set_error("plx_export: NULL argument"); return PLX_ERR_INVALID; }
    if (!L->built) { set_error("plx_export: lattice not built"); return PLX_ERR_STATE; }
    const int64_t want = plx_export_bytes(L, which);
    if (want < 0) { set_error("plx_export: unknown array %d", which); return PLX_ERR_INVALID; }
    if (want != bytes) { set_error("plx_export: array %d is %lld bytes, caller gave %lld", which, (long long)want, (long long)bytes); return PLX_ERR_INVALID; }
    if (bytes == 0) return PLX_OK;
    DeviceGuard g(L->device);
    hipStream_t s = (hipStream_t)stream;
    const int64_t m = L->m;
    switch (which) {
    case PLX_ARRAY_KEYS: {
        // device keys are packed int16 pairs padded to DW words per vertex
        const int dw = (L->d + 1) / 2;
        std::vector<uint32_t> tmp((size_t)m * dw);
        PLX_HIP_TRY(hipMemcpyAsync(tmp.data(), L->vkeys.p, tmp.size() * 4, hipMemcpyDeviceToHost, s));
        PLX_HIP_TRY(hipStreamSynchronize(s));
        int16_t *dst = (int16_t *)h_dst;
        for (int64_t i = 0; i < m; ++i)
            for (int c = 0; c < L->d; ++c)
                dst[i * L->d + c] = (int16_t)((tmp[i * dw + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFFu);
        return PLX_OK;
    }
    case PLX_ARRAY_NEIGHBORS: {
        // strip the plane padding (mstride -> m)
        const int planes = (L->d + 1) * 2 * L->order;
        PLX_HIP_TRY(hipMemcpy2DAsync(h_dst, (size_t)m * 4, L->nbr.p, (size_t)L->mstride * 4, (size_t)m * 4,
                                     planes, hipMemcpyDeviceToHost, s));
        break;
    }
    case PLX_ARRAY_ENTRY_VERTEX: PLX_HIP_TRY(hipMemcpyAsync(h_dst, L->evid.p, bytes, hipMemcpyDeviceToHost, s)); break;
    case PLX_ARRAY_ENTRY_WEIGHT: PLX_HIP_TRY(hipMemcpyAsync(h_dst, L->ew.p, bytes, hipMemcpyDeviceToHost, s)); break;
    case PLX_ARRAY_ROW_PTR:
        PLX_TRY(export_row_ptr(L, s));
        PLX_HIP_TRY(hipMemcpyAsync(h_dst, L->row_ptr.p, bytes, hipMemcpyDeviceToHost, s));
        break;
    case PLX_ARRAY_CSR_POINT: {
        PLX_TRY(ensure_csr(L, s));
        PLX_HIP_TRY(hipMemcpyAsync(h_dst, L->csr_pt.p, bytes, hipMemcpyDeviceToHost, s));
        PLX_HIP_TRY(hipStreamSynchronize(s));
        int32_t *dst = (int32_t *)h_dst;   // strip the segment-head flag kept in the sign bit
        for (int64_t i = 0; i < bytes / 4; ++i) dst[i] &= 0x7FFFFFFF;
        return PLX_OK;
    }
    case PLX_ARRAY_CSR_WEIGHT:
        PLX_TRY(ensure_csr(L, s));
        PLX_HIP_TRY(hipMemcpyAsync(h_dst, L->csr_w.p, bytes, hipMemcpyDeviceToHost, s));
        break;
    case PLX_ARRAY_POINT_PERM: PLX_HIP_TRY(hipMemcpyAsync(h_dst, L->perm.p, bytes, hipMemcpyDeviceToHost, s)); break;
    }
    PLX_HIP_TRY(hipStreamSynchronize(s));
    return PLX_OK;
}

int plx_tune(const char *key, int value)
{
    if (!key) return PLX_ERR_INVALID;
    for (const Tunable *t = tunables(); t->name; ++t)
        if (strcmp(t->name, key) == 0) { g_tune_defaults.*(t->member) = value; return PLX_OK; }
    set_error("plx_tune: unknown key %s", key);
    return PLX_ERR_INVALID;
}

int plx_lattice_tune(plx_lattice *L, const char *key, int value)
{
    if (!L || !key) { set_error("plx_lattice_tune: NULL argument"); return PLX_ERR_INVALID; }
    for (const Tunable *t = tunables(); t->name; ++t)
        if (strcmp(t->name, key) == 0) { L->tn.*(t->member) = value; return PLX_OK; }
    set_error("plx_lattice_tune: unknown key %s", key);
    return PLX_ERR_INVALID;
}

int plx_copy_point_perm(plx_lattice *L, void *d_dst, void *stream)
{
    if (!L || !d_dst) { set_error("plx_copy_point_perm: NULL argument"); return PLX_ERR_INVALID; }
    if (!L->built) { set_error("plx_copy_point_perm: lattice not built"); return PLX_ERR_STATE; }
    DeviceGuard g(L->device);
    PLX_HIP_TRY(hipMemcpyAsync(d_dst, L->perm.p, (size_t)L->n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PLX_OK;
}

int plx_set_timing(plx_lattice *L, int on)
{
    if (!L) return PLX_ERR_INVALID;
    L->timing = on != 0;
    return PLX_OK;
}

int plx_apply_times(plx_lattice *L, float *h_ms, int cap, int *count)
{
    if (!L || !h_ms || !count) return PLX_ERR_INVALID;
    *count = 0;
    if (!L->timing || L->tev_n < 2) return PLX_OK;
    DeviceGuard g(L->device);
    PLX_HIP_TRY(hipEventSynchronize(L->tev[L->tev_n - 1]));
    const int k = L->tev_n - 1;
    for (int i = 0; i < k && i < cap; ++i) PLX_HIP_TRY(hipEventElapsedTime(&h_ms[i], L->tev[i], L->tev[i + 1]));
    *count = k < cap ? k : cap;
    return PLX_OK;
}

int plx_last_kernels(const plx_lattice *L, char *buf, int cap)
{
    if (!L || !buf || cap < 1) return PLX_ERR_INVALID;
    snprintf(buf, (size_t)cap, "splat=%s;blur_axis=%s;slice=%s;vertex_order=%s", L->kn_splat, L->kn_blur, L->kn_slice,
             L->vertex_order ? "morton" : "first_touch");
    return PLX_OK;
}

int64_t plx_block_rows(const plx_lattice *L)
{
    return (L && L->built && L->blocks_ready && L->use_blocks) ? L->n_brows : 0;
}

int plx_prepare(plx_lattice *L, int vd, void *stream)
{
    EntryScope sc(L, stream);
    if (!L) { set_error("plx_prepare: NULL lattice"); return PLX_ERR_INVALID; }
    if (!L->built) { set_error("plx_prepare: lattice not built"); return PLX_ERR_STATE; }
    if (vd < 1) { set_error("plx_prepare: vd = %d must be positive", vd); return PLX_ERR_INVALID; }
    DeviceGuard g(L->device);
    return prepare_tables(L, vd, (hipStream_t)stream);
}

int plx_selftest_sort(int64_t n, int key_bytes, int end_bit, uint64_t seed, void *stream, int64_t *mismatches)
{
    if (n < 1 || n > (1ll << 31) - 8192 || (key_bytes != 4 && key_bytes != 8) || end_bit < 1 || end_bit > 8 * key_bytes || !mismatches) {
        set_error("plx_selftest_sort: n = %lld, %d-byte keys, %d bits", (long long)n, key_bytes, end_bit);
        return PLX_ERR_INVALID;
    }
    return selftest_sort(n, key_bytes, end_bit, seed, (hipStream_t)stream, mismatches);
}

int plx_build_times(const plx_lattice *L, float *h_ms6)
{
    if (!L || !h_ms6) return PLX_ERR_INVALID;
    for (int i = 0; i < 6; ++i) h_ms6[i] = L->build_ms[i];
    return PLX_OK;
}

}  // extern "C"

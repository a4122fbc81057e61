/*
 * plx.h -- C ABI of the MI355X permutohedral-lattice filter (libplx.so).
 *
 * This is the drop-in boundary for the reference's one native entry point,
 *
 *     filter(src[N,vd], ref[N,d], coeffs[R]) -> out[N,vd]
 *         gpytorch_lattice_kernel/cpp/lattice.cpp:6-16            (CPU)
 *         gpytorch_lattice_kernel/cuda/permutohedral_cuda.cpp:12-22 (CUDA)
 *
 * restated as plain pointers and sizes (no torch types), plus the staged form
 * the reference fuses into that call (PermutohedralLattice ctor + splat, blur,
 * slice: cpp/permutohedral.h:346-392, 395-486, 513-572, 497-510), so that one
 * lattice can serve every MVM on the same inputs (all conjugate-gradient
 * iterations, the backward pass) and so that a sharded job can put one RCCL
 * all-reduce between splat and blur.
 *
 * Conventions
 *   - every data pointer is DEVICE memory (fp32, row-major, contiguous) on the
 *     lattice's device unless the name starts with h_ (host);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it.  A build waits on the host for a few small counts (coordinate ranges, the vertex
 *     count m that sizes the lattice, ...): each arrives through a pinned mailbox the host spins on, the stream
 *     itself is not synchronised.  plx_splat / plx_blur / plx_slice / plx_apply never synchronise and never
 *     allocate once the lattice's tables exist (plx_prepare, or the first MVM of that width) and their workspace
 *     has reached its high-water mark, so they are graph-capturable from then on.  Buffers grow stream-ordered
 *     (hipMallocAsync) on the stream of the call that needs them; a lattice may move between streams as long as
 *     the caller orders the calls;
 *   - every function returns PLX_OK (0) or an error code; nothing calls exit()
 *     (reference: cuda/permutohedral_cuda_kernel.cu:24-32 does).
 *     plx_last_error() returns a thread-local detail string;
 *   - a plx_lattice is not safe for concurrent use from two threads; different lattices may be used from
 *     different threads at the same time (the library keeps no other mutable state than the plx_tune defaults).
 */
#ifndef PLX_H
#define PLX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plx_lattice plx_lattice;

enum {
    PLX_OK = 0,
    PLX_ERR_INVALID = 1,     /* bad argument (shape, NULL, even tap count, ...)        */
    PLX_ERR_HIP = 2,         /* a HIP runtime call failed (see plx_last_error)         */
    PLX_ERR_KEY_RANGE = 3,   /* a lattice coordinate left the int16 key range
                                (the reference wraps silently, permutohedral.h:417-419) */
    PLX_ERR_DIM = 4,         /* d or order outside the compiled range                  */
    PLX_ERR_STATE = 5,       /* lattice not built / wrong sizes for this lattice        */
    PLX_ERR_TOO_LARGE = 6    /* n*(d+1) does not fit the 31-bit entry index             */
};

#define PLX_MAX_DIM 32       /* position dimension d, 1..PLX_MAX_DIM                    */
#define PLX_MAX_ORDER 8      /* taps R = 2*order+1, order 0..PLX_MAX_ORDER              */

/* names for plx_export() */
enum {
    PLX_ARRAY_KEYS = 0,          /* int16  [m][d]        vertex keys, first-touch order (h:73-79)   */
    PLX_ARRAY_ENTRY_VERTEX = 1,  /* int32  [d+1][n]      vertex id of simplex corner r of point p   */
    PLX_ARRAY_ENTRY_WEIGHT = 2,  /* float  [d+1][n]      barycentric weight (h:460-465)             */
    PLX_ARRAY_NEIGHBORS = 3,     /* int32  [d+1][2r][m]  blur neighbour ids, -1 absent (h:539-544)  */
    PLX_ARRAY_ROW_PTR = 4,       /* int32  [m+1]         splat CSR row pointers (owned points)      */
    PLX_ARRAY_CSR_POINT = 5,     /* int32  [nnz]         splat CSR point index                      */
    PLX_ARRAY_CSR_WEIGHT = 6,    /* float  [nnz]         splat CSR weight                           */
    PLX_ARRAY_POINT_PERM = 7     /* uint32 [n]           caller's row of the i-th point in lattice order;
                                    ENTRY_* and CSR_POINT are indexed in lattice order             */
};

const char *plx_strerror(int code);
const char *plx_last_error(void);
/* "libplx <version> gfx950" -- lets a host check which build it loaded */
const char *plx_version(void);

/* A lattice object owns its device buffers (grow-only); re-building it for new
 * inputs reuses them.  `device` is a HIP device ordinal. */
int plx_create(int device, plx_lattice **out);
void plx_destroy(plx_lattice *lat);

/*
 * Build the lattice structure for positions d_ref[n][d] (already divided by the
 * lengthscale, bilateral_kernel.py:198) and stencil taps h_taps[ntaps]
 * (ntaps odd; they set the embedding scale through variance(), h:203-219,
 * h:372-390, and are the blur weights, h:546).
 *
 * Replaces: PermutohedralLattice ctor (h:346-392), the structural half of
 * splat() (h:395-475, 482-484: embedding, hashed vertex creation, replay
 * entries), and every hashTable.lookup() of blur() (h:541-545), which becomes a
 * neighbour table.
 *
 * Internally the points are visited in "lattice order" (shard by shard, inside a
 * shard lexicographically by their rounded lattice coordinates) so that points
 * sharing simplices are adjacent in memory; vertex ids are the reference's
 * first-touch numbering (h:73-79) applied to that order.  The order is an
 * implementation detail: d_src / d_out rows are always in the caller's order.
 *
 * Sharded jobs: the n rows are split into n_shards contiguous near-equal blocks
 * (the first n % n_shards blocks have one extra row); this process splats and
 * slices block `shard_index` only.  Every rank passes the SAME d_ref and
 * n_shards and obtains the same vertex numbering without communication.  A single
 * GPU uses shard_index = 0, n_shards = 1.
 */
int plx_build(plx_lattice *lat, const float *d_ref, int64_t n, int d,
              const float *h_taps, int ntaps,
              int shard_index, int n_shards, void *stream);

/*
 * Sharded build without replicated work (alternative to calling plx_build with the
 * full d_ref on every rank): every rank passes ONLY ITS OWN rows.
 *   1. plx_build_local     embeds / inserts / numbers the rank's own points;
 *   2. the caller exchanges the per-rank vertex keys: m_r = plx_local_vertices(),
 *      keys = [m_r][plx_key_words(d)] uint32 (plx_copy_local_keys), all-gathered in
 *      rank order (RCCL all_gather through torch.distributed in this repo);
 *   3. plx_build_merge     numbers the union (first occurrence in rank order, then
 *      local order -- or along the Morton curve of the vertices, see "vertex_order": either way
 *      the same ids plx_build's shard-major numbering produces),
 *      relabels the local corners, builds the neighbour table over the union and the
 *      splat CSR / slice tables over the rank's rows.
 * Afterwards the lattice behaves like one built by plx_build for that shard:
 * plx_num_points / plx_num_owned = n_local, plx_num_vertices = size of the union.
 */
int plx_build_local(plx_lattice *lat, const float *d_ref_local, int64_t n_local, int d,
                    const float *h_taps, int ntaps, void *stream);
int plx_key_words(int d);                               /* uint32 words per packed vertex key: ceil(d/2) */
int64_t plx_local_vertices(const plx_lattice *lat);     /* m_r after plx_build_local                     */
int plx_copy_local_keys(plx_lattice *lat, void *d_dst, void *stream);   /* d_dst: [m_r][key_words] uint32 */
int plx_build_merge(plx_lattice *lat, const void *d_all_keys, const int64_t *h_counts, int n_ranks,
                    int my_rank, int64_t total_points, void *stream);
/* total_points: points of ALL ranks together (the row counts ride along with the key counts), the same value on every
 * rank; it only feeds the choice of vertex numbering ("vertex_order" under plx_tune), which must fall the same way on
 * every rank.  <= 0: unknown, the choice is made from the key counts alone. */

int64_t plx_num_points(const plx_lattice *lat);    /* n                              */
int64_t plx_num_owned(const plx_lattice *lat);     /* rows of this shard             */
int64_t plx_num_vertices(const plx_lattice *lat);  /* m = hashTable.size(), h:44     */
/* plx_tune("reference_growth", 1) -- LITERAL parity with the reference CPU path where its hash table doubles.  lookup()
 * (cpp/permutohedral.h:104-106) hashes with the capacity in force before lookupOffset() (:58-63) grows the table, so the one
 * lookup that triggers each doubling probes from a stale bucket: a duplicate entry for a key that exists, or a key that
 * later lookups do not find; grow() (:125-161) then re-places entries in old-position order.  The default build is the
 * duplicate-free lattice (which is also what the reference's CUDA path builds: its table never grows).  With the switch
 * on, plx_build / plx_filter replay the reference's table LAYOUT on the host (entry positions only; the EVENTS that shape it
 * -- the m first-touch creations, the stale probe behind each doubling, every lookup of the few keys such a probe touched:
 * O(m) host work, 45-60 ms at N = 1e6, d = 8, m = 1.7e6; value 2 = all N (d+1) lookups one by one, the checker of the
 * event form, 250 ms there) and patch the built structure so that every MVM equals the reference's filter(): the splat
 * contributions the reference loses are dropped, keys its blur-time lookups cannot find read as absent.  Plain
 * single-process builds only (ignored by plx_build_local / plx_build_merge).  h_out6 = {replayed (0/1), entries the
 * reference's table holds (= plx_num_vertices when nothing was replayed or nothing went wrong), dropped (point, corner)
 * contributions, invisible vertices, blur-time missed neighbour (0/1), inexact (0/1: a combination the patched structure
 * cannot express -- never observed)}. */
int plx_reference_growth_info(const plx_lattice *lat, int64_t *h_out6);
int plx_dim(const plx_lattice *lat);               /* d                              */
int plx_order(const plx_lattice *lat);             /* (ntaps-1)/2                    */
/* Row order of d_src / d_out for plx_splat / plx_slice / plx_apply on this lattice:
 * 0 (default) the caller's order -- rows are permuted into lattice order on the way
 * in and back on the way out; 1 the rows ARE in lattice order (row i of this shard
 * = caller row PLX_ARRAY_POINT_PERM[own_begin + i] - own_begin), no permutation
 * work per MVM.  A CG solve permutes its right-hand side once, iterates in
 * lattice order and permutes the solution back once. */
int plx_set_row_order(plx_lattice *lat, int lattice_order);

/* Warm start of the point order (round 6).  on != 0: the NEXT plx_build on this lattice keeps the lattice order of its
 * points from the previous build instead of computing it (coordinate ranges + read-back, sort keys, four radix passes:
 * 0.15 ms of a 1.8 ms build at N = 1e6, d = 8).  For a caller who KNOWS that the positions are the previous build's,
 * re-scaled a little -- a GP training loop whose lengthscale moved: the order is a locality device only (which points sit
 * next to each other in memory), no vertex, weight or neighbour depends on it, so the result is the cold build's up to
 * the order of the fp32 sums inside a vertex row; an order computed for other positions would merely be a slow one.
 * One shot (cleared by the build); ignored unless the row count, dimension and shard are those of the order at hand.
 * plx_order_age: builds since the order was computed from the positions themselves (0 = by the last build, -1 = none). */
int plx_set_reuse_order(plx_lattice *lat, int on);
int plx_order_age(const plx_lattice *lat);

/* Floats per vertex row of a values buffer for vd value columns: 1 for vd = 1,
 * otherwise vd rounded up to a multiple of 4 (rows are whole 16-byte vectors;
 * the padding columns hold zeros).  d_values / d_scratch below are [m][stride]. */
int plx_values_stride(int vd);
/* bytes of device memory currently held by the lattice */
int64_t plx_device_bytes(const plx_lattice *lat);

/*
 * Stage 1 -- splat (h:478-479): d_values[m][stride] = S^T d_src, where d_src holds
 * this shard's rows only, [plx_num_owned][vd], in the caller's row order.  Every row of d_values is
 * written (vertices no owned point touches get 0).  Deterministic: no float
 * atomics.
 */
int plx_splat(plx_lattice *lat, const float *d_src, int vd, float *d_values, void *stream);

/* plx_splat of nb one-hot columns without streaming the corners: column b (< nb <= vd) of d_values is S^T e_p for
 * p = d_points[b], a point index in LATTICE order (device int32; PLX_ARRAY_POINT_PERM maps it to the caller's row); the
 * other columns and rows are zero.  What a pivoted Cholesky of the operator asks for: rows of K = slice(blur(this)).
 * Single-shard lattices. */
int plx_splat_onehot(plx_lattice *lat, const int32_t *d_points, int nb, int vd, float *d_values, void *stream);
/* The whole filter of nb <= 16 one-hot columns: d_out [n][vd] (rows as plx_set_row_order says) = K [e_p0 .. e_p(nb-1) 0 ..],
 * p_b = d_points[b] in LATTICE order as above -- plx_splat_onehot + plx_blur + plx_slice in one call.  sparse != 0: the
 * three stages run on the FRONTIER of the columns' non-zero vertex rows (d + 1 rows per column after the splat, at most
 * 2 r + 1 times as many after every blur axis) instead of streaming all m rows d + 1 times; same operations in the same
 * order as the dense kernels, so the same numbers.  The frontier path needs vd == 1 or vd % 4 == 0 with a 16-byte
 * aligned d_out and a lattice built without "reference_growth"; otherwise (and with sparse == 0) the call runs the three
 * dense stages.  d_values / d_scratch: [m][plx_values_stride(vd)] each, distinct, clobbered.  *d_frontier (device int32,
 * optional): vertex rows the last blur axis worked on (m for the dense stages) -- a caller that batches such calls reads
 * it back to see when the frontier stops being small.  Single-shard lattices. */
int plx_filter_onehot(plx_lattice *lat, const int32_t *d_points, int nb, int vd, float *d_values, float *d_scratch,
                      float *d_out, int sparse, int32_t *d_frontier, void *stream);

/*
 * Stage 2 -- blur (h:513-572): d+1 Jacobi passes over the neighbour table,
 * ping-ponging between d_values and d_scratch (both [m][stride]).  On return
 * *result_in_scratch tells which of the two holds the result (d+1 odd => 1).
 */
int plx_blur(plx_lattice *lat, float *d_values, float *d_scratch, int vd,
             int *result_in_scratch, void *stream);

/*
 * Stage 3 -- slice (h:497-510): d_out[plx_num_owned][vd] = S d_values / (1 + 2^-d)
 * for this shard's rows, in the caller's row order.
 */
int plx_slice(plx_lattice *lat, const float *d_values, int vd, float *d_out, void *stream);

/* splat -> blur -> slice on the lattice's own workspace (single-GPU MVM). */
int plx_apply(plx_lattice *lat, const float *d_src, int vd, float *d_out, void *stream);

/*
 * The reference's one-shot call (cpp:6-10 -> h:259-340): build a lattice for
 * d_ref, apply it to d_src, leave nothing behind.  `scratch` may be NULL or a
 * lattice object whose buffers are reused (avoids hipMalloc in steady state; it
 * is left built and usable).  The build knows it serves ONE MVM and leaves out
 * the stages that only pay back over several (vertex renumbering, axis-pair
 * tables of the blur): callers with more than a few MVMs per lattice use
 * plx_build + plx_apply.
 */
int plx_filter(plx_lattice *scratch, const float *d_src, const float *d_ref,
               int64_t n, int d, int vd, const float *h_taps, int ntaps,
               float *d_out, void *stream);

/*
 * Column-wise dot products of two row-major device matrices, d_out[c] = sum_r
 * a[r][c] * b[r][c]: the reduction a batched-CG caller needs per iteration
 * (GPyTorch's mBCG does it with torch ops).  Deterministic (fixed reduction
 * order).  d_work: plx_coldot_work_floats(vd) floats of device scratch.
 */
int plx_coldot(const float *d_a, const float *d_b, int64_t n, int vd, float *d_out, float *d_work, void *stream);
int64_t plx_coldot_work_floats(int vd);
/* Position gradient of one filter call, fused (replaces LatticeFilterGeneral.backward with a reference gradient,
 * bilateral_kernel.py:113-123): `lat` is built on d_ref [n][d] with the DERIVATIVE taps; d_g (the incoming gradient)
 * and d_src (the forward right-hand side) are [n][nrhs].  Writes d_grad_ref [n][d] and, unless NULL,
 * d_grad_src [n][nrhs].  The 2*nrhs*(1+d)-column stacked matrix and its filtered image are never stored: the splat
 * forms the stack from packed per-point records, slice and contraction are one kernel.  Column range 125..512
 * and 2*nrhs + d <= 62 (PLX_ERR_INVALID otherwise: use the three-call form below), single-shard lattices only. */
int plx_apply_backward(plx_lattice *lat, const float *d_g, const float *d_src, const float *d_ref, int nrhs,
                       float *d_grad_ref, float *d_grad_src, void *stream);
/* The two elementwise ends of the position gradient (bilateral_kernel.py:113-122), one pass each, row-major fp32:
 *   plx_backward_stack:    d_out[n][2L(1+d)] = [ g | g (x) x | src | src (x) x ]   (the matrix the filter is applied to)
 *   plx_backward_contract: d_grad_x[n][d] = -2 sum_l ( src x wg - src wgx + g x ws - g wsx ) from the filtered stack */
int plx_backward_stack(const float *d_g, const float *d_src, const float *d_x, int64_t n, int L, int d,
                       float *d_out, void *stream);
int plx_backward_contract(const float *d_g, const float *d_src, const float *d_x, const float *d_filtered,
                          int64_t n, int L, int d, float *d_grad_x, void *stream);
/* out = a * (K src) + b * src with (a, b) = d_scale_shift[0..1] read on the device: the (s K + sigma^2 I) v of a GP
 * solve in one call (the two scalars live in device memory so that no host synchronisation is needed to pass
 * hyper-parameters that are device tensors).  d_out must not alias d_src. */
int plx_apply_affine(plx_lattice *lat, const float *d_src, int vd, float *d_out, const float *d_scale_shift, void *stream);
/* plx_apply_affine that also returns d_dot[c] = <src[:, c], out[:, c]> (c < plx_values_stride(vd); padding entries are 0):
 * the p^T A p of a CG iteration comes out of the slice kernel's registers instead of a second pass over both
 * matrices.  2 <= vd <= 256.  d_work: plx_affine_dot_work_floats(lat, vd) floats of scratch. */
int64_t plx_affine_dot_work_floats(const plx_lattice *lat, int vd);
int plx_apply_affine_dot(plx_lattice *lat, const float *d_src, int vd, float *d_out, const float *d_scale_shift,
                         float *d_dot, float *d_work, void *stream);
/* d_dot may be NULL: the slice kernel's per-tile partial sums are then left in d_work -- plx_affine_dot_tiles(lat, vd) rows
 * of plx_values_stride(vd) floats -- for plx_cg_step_update_fused, which adds them up itself (one launch less). */
int plx_affine_dot_tiles(const plx_lattice *lat, int vd);
/* One batched-CG iteration's vector work with the coefficients formed on the device (all small arrays are float [vd],
 * `active` holds 1.0 / 0.0):
 *   plx_cg_step_update:    alpha = active ? rs / max(pAp, tiny) : 0;  X += alpha P;  R -= alpha AP;  rs_new = |R|^2
 *   plx_cg_step_direction: beta = active ? rs_new / max(rs, tiny) : 0;  P = R + beta P;
 *                          active_out = active and sqrt(rs_new) / b_norm > tol     (active_out != active) */
int plx_cg_step_update(float *d_x, float *d_r, const float *d_p, const float *d_ap, const float *d_rs, const float *d_pap,
                       const float *d_active, int64_t n, int vd, float *d_rs_new, float *d_alpha, float *d_work,
                       void *stream);
/* The same iteration without its two stand-alone reductions (round 6; vd = 4, 8, 12 or 16 -- rows of whole 16-byte chunks,
 * the widths solvers.khat_solve pads to; every pointer 16-byte aligned):
 *   plx_cg_step_update_fused     pAp comes as the PARTIAL sums plx_apply_affine_dot(d_dot = NULL) left behind (d_pap_partial:
 *                                ntiles = plx_affine_dot_tiles rows of vd floats); every workgroup adds them up itself, in
 *                                a fixed order.  |R|^2 leaves as partial sums in d_work (plx_cg_fused_work_floats(vd) floats).
 *   plx_cg_step_direction_fused  adds those up (every workgroup, fixed order), stores rs_new (d_rs_new != d_rs), beta and
 *                                active_out, and updates P.
 * Same arithmetic as the pair above except for the association of the column sums; deterministic.  Two launches fewer per
 * iteration: measured 11 % faster at n = 2e4, neutral at 1e5 ... 3e5, 0.3-0.5 % slower at n = 1e6 (every workgroup re-reads all
 * partial sums): a caller picks by size (simplex_gp_amd.solvers: up to 65,536 rows). */
int64_t plx_cg_fused_work_floats(int vd);
int plx_cg_step_update_fused(float *d_x, float *d_r, const float *d_p, const float *d_ap, const float *d_rs,
                             const float *d_pap_partial, int ntiles, const float *d_active, int64_t n, int vd,
                             float *d_alpha, float *d_work, void *stream);
int plx_cg_step_direction_fused(float *d_p, const float *d_r, const float *d_work, const float *d_rs, const float *d_active,
                                const float *d_b_norm, float tol, int64_t n, int vd, float *d_rs_new, float *d_beta,
                                float *d_active_out, void *stream);
/* ... and of the preconditioned iteration: <R, Z> as the partial sums plx_pcg_apply(d_rz = NULL) left in its d_work
 * (plx_pcg_rz_partial_rows(n, factor_type) rows of vd floats at float offset plx_pcg_rz_partial_offset(kp)), |R|^2 as
 * plx_cg_step_update_fused's partial sums; stores rz_new (!= d_rz), rr = |R|^2, beta, active_out; P = Z + beta P. */
int64_t plx_pcg_rz_partial_offset(int kp);
int plx_pcg_rz_partial_rows(int64_t n, int factor_type);
int plx_pcg_step_direction_fused(float *d_p, const float *d_z, const float *d_rz_partial, int nrz, const float *d_rr_partial,
                                 const float *d_rz, const float *d_active, const float *d_b_norm, float tol, int64_t n, int vd,
                                 float *d_rz_new, float *d_rr, float *d_beta, float *d_active_out, void *stream);
int plx_cg_step_direction(float *d_p, const float *d_r, const float *d_rs_new, const float *d_rs, const float *d_active,
                          const float *d_b_norm, float tol, int64_t n, int vd, float *d_beta, float *d_active_out,
                          void *stream);
/*
 * One Lanczos step with full re-orthogonalisation next to its MVM -- the variance cache of the reference's evaluation
 * (gpytorch.settings.fast_pred_var + max_root_decomposition_size(lanc_iter), experiments/train_simplexgp.py:63-72; GPyTorch
 * runs the recurrence with torch ops).  d_q: the basis, float [rows >= i + 2][ld] row-major, rows 0..i orthonormal; d_w:
 * A q_i on entry ([n], overwritten: on return the re-orthogonalised, un-normalised vector).  Writes d_alphas[i] =
 * q_i . A q_i, d_betas[i] = the norm of w after its components along rows i-1 and i (the alpha q_i and beta q_{i-1} terms
 * of the three-term recurrence) and then, in one classical Gram-Schmidt pass, along all rows 0..i have been removed, and
 * row i + 1 of d_q = w / max(beta_i, 1e-30).  Four launches, two streams of the basis, deterministic (fixed-order sums, no atomics).
 * i + 1 <= plx_lanczos_max_rows() (256); d_work: float [plx_lanczos_work_floats(n)] (< 0: n is larger than the 2,097,152
 * rows the step serves); ld >= n, a multiple of 4, d_q 16-byte aligned (columns n..ld-1 of the basis are never written).
 */
int plx_lanczos_max_rows(void);
int64_t plx_lanczos_work_floats(int64_t n);
int plx_lanczos_step(float *d_q, int64_t ld, float *d_w, int64_t n, int i, float *d_alphas, float *d_betas, float *d_work,
                     void *stream);
/* The two vector updates of a batched CG iteration, one pass each (row-major [n][vd], per-column scalars on the
 * device):  plx_cg_update: X += P*alpha, R -= AP*alpha, d_rs_new[c] = sum_r R[r][c]^2 (d_work as for plx_coldot);
 *           plx_cg_direction: P = R + P*beta. */
int plx_cg_update(float *d_x, float *d_r, const float *d_p, const float *d_ap, const float *d_alpha,
                  int64_t n, int vd, float *d_rs_new, float *d_work, void *stream);
int plx_cg_direction(float *d_p, const float *d_r, const float *d_beta, int64_t n, int vd, void *stream);

/*
 * Preconditioned batched CG: the reference trains with gpytorch.settings.max_preconditioner_size(100)
 * (experiments/train_simplexgp.py:36, configs/simplexgp.yml), i.e. every solve of (s K + sigma^2 I) is preconditioned by
 * P = L L^T + sigma^2 I, L [n][k] the rank-k pivoted Cholesky factor of s K (GPyTorch builds and applies it with torch
 * ops).  These entry points are that work as native passes.  The factor is handed over TRANSPOSED:
 *   d_lt   float [kp][ld] row-major = L^T; kp = k rounded up to a multiple of 16 (the extra rows zero), ld = n rounded up
 *          to a multiple of 64 (the tail of every row zero), 16-byte aligned; the n dimension in the same row order as
 *          the CG vectors (solvers.py keeps everything in lattice row order, so nothing is permuted per iteration);
 *   t      columns of the CG vectors, 1..16, row-major [n][t];
 *   d_work plx_pcg_work_floats(n, kp, t) floats of scratch.
 * One application Z = P^-1 R = (R - L C^-1 L^T R) / sigma^2, C = sigma^2 I + L^T L, is two streaming passes over L^T:
 *   plx_pcg_project   T [kp][16] = C^-1 (L^T R): the products on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32
 *                     products, fp32 accumulation), per-workgroup partial sums added in fp64 in a fixed order, then the
 *                     kp x kp solve with d_cinv = C^-1 in fp64 [kp][kp] (symmetric; identity / sigma^2 on the padding);
 *   plx_pcg_apply     Z = (d_scale[0] R - L T) d_scale[1] (k = columns of L actually used) and, unless NULL,
 *                     d_rz[c] = <R[:, c], Z[:, c]> from the same registers.  d_scale: two floats in device memory.
 *   plx_pcg_step_direction   beta = active ? rz_new / rz : 0; P = Z + beta P; active_out = active and
 *                     sqrt(rr) / b_norm > tol (rr = |R|^2 as plx_cg_step_update returns it: the TRUE residual).
 * plx_cg_step_update is used unchanged with rs := rz.  Deterministic (no atomics).
 * factor_type: the factor's storage, PLX_FACTOR_F32 or PLX_FACTOR_F16 (IEEE half, same [kp][ld] layout; made from the fp32
 * factor by plx_pcg_factor_to_half).  Both passes are bound by streaming the factor, so the half-width copy halves them.
 * A preconditioner only has to be symmetric positive definite and the same matrix wherever it is used: a caller that
 * stores L in fp16 must form C = sigma^2 I + L^T L, the log-determinant and its probe vectors from that SAME rounded L
 * (solvers.LatticePreconditioner does); products and accumulation are fp32 either way.
 */
enum { PLX_FACTOR_F32 = 0, PLX_FACTOR_F16 = 1 };
int64_t plx_pcg_work_floats(int64_t n, int kp, int t);
int plx_pcg_project(const void *d_lt, int factor_type, int64_t ld, int kp, const float *d_r, int64_t n, int t,
                    const double *d_cinv, float *d_t, float *d_work, void *stream);
int plx_pcg_apply(const void *d_lt, int factor_type, int64_t ld, int kp, int k, const float *d_r, int64_t n, int t,
                  const float *d_t, const float *d_scale, float *d_z, float *d_rz, float *d_work, void *stream);
int plx_pcg_factor_to_half(const float *d_lt, int64_t ld, int kp, void *d_lt_half, void *stream);
int plx_pcg_step_direction(float *d_p, const float *d_z, const float *d_rz_new, const float *d_rz, const float *d_rr,
                           const float *d_active, const float *d_b_norm, float tol, int64_t n, int vd, float *d_beta,
                           float *d_active_out, void *stream);
/*
 * The factor itself, built in batches of speculated pivots.  The sequential algorithm (GPyTorch's pivoted_cholesky, one
 * kernel row = one single-column MVM per pivot) is reproduced exactly, but nb <= 16 pivots share ONE nb-column MVM:
 *   plx_pchol_select        d_cand[0..nb) = the nb largest entries of the residual diagonal d_diag [n], ties by LOWER
 *                           d_rank[i] (NULL: by lower i) -- pass the lattice's point permutation so that ties fall as
 *                           torch.argmax breaks them on the caller-order vector; the largest entry that is NOT a
 *                           candidate is kept in d_work (it bounds every non-candidate for the whole batch: steps only
 *                           lower entries).  May run as soon as the previous batch's plx_pchol_factor_batch is enqueued;
 *   plx_pchol_onehot        d_rhs [n][t] = the nb one-hot columns (t >= nb; the caller runs the MVM on it);
 *   plx_pchol_factor_batch  d_rows [n][t] = K d_rhs.  Panel update against the m_done finished columns
 *                           (d_scale[0] d_rows - L L[cand]^T, one pass over them), then the in-batch steps in pivot
 *                           order: column m_done + b = panel row of the step's pivot / sqrt(pivot) (zero if the pivot is
 *                           <= tol_abs), residual diagonal updated.  Which candidate a step takes is the argmax of the
 *                           updated diagonal (the sequential algorithm may take the candidates in another order than
 *                           their values at the start of the batch suggest).  The steps whose argmax is CERTAIN from the
 *                           candidates' own entries -- the largest unused candidate still lies above the largest entry
 *                           outside the batch, which a step can only lower -- are planned on the nb x nb block of the
 *                           panel and written in ONE pass over the n-vectors ("planned" steps, >= 1).  exact_steps != 0:
 *                           the remaining candidates are then tried one launch per step, the argmax looked up on the
 *                           device among them; the batch ends when the argmax is an entry whose kernel row is not in
 *                           it.  exact_steps == 0: the batch ends with the plan.  d_accepted (device int32 [2]) =
 *                           {columns written, planned steps}; the caller reads it back once per batch, carries on from
 *                           m_done + accepted and passes exact_steps = 0 while whole batches come out planned (sparse
 *                           kernel rows: a column touches few other candidates).
 * d_work: plx_pchol_work_bytes(ld, kp) bytes, the same buffer for the three calls of a batch.
 */
int64_t plx_pchol_work_bytes(int64_t ld, int kp);
int plx_pchol_select(const float *d_diag, const uint32_t *d_rank, int64_t n, int nb, int64_t ld, int kp, int32_t *d_cand,
                     void *d_work, void *stream);
int plx_pchol_onehot(const int32_t *d_cand, int nb, int64_t n, int t, float *d_rhs, void *stream);
int plx_pchol_factor_batch(float *d_lt, int64_t ld, int kp, int m_done, const float *d_rows, int t, const float *d_scale,
                           const int32_t *d_cand, int nb, float *d_diag, const uint32_t *d_rank, int64_t n, float tol_abs,
                           int exact_steps, int32_t *d_accepted, void *d_work, void *stream);

/* Copy one structure array to host memory (parity tests, debugging).
 * h_dst must hold `bytes` bytes, which must equal the array's size. */
int plx_export(plx_lattice *lat, int which, void *h_dst, int64_t bytes, void *stream);
/* PLX_ARRAY_POINT_PERM device to device (uint32 [n]): lets a caller permute its vectors into lattice row
 * order on the GPU without a host round trip. */
int plx_copy_point_perm(plx_lattice *lat, void *d_dst, void *stream);
/* size in bytes of an exportable array, or -1 */
int64_t plx_export_bytes(const plx_lattice *lat, int which);

/* Select a kernel variant by name: for A/B measurements in one process -- the defaults are the shipped configuration.
 * plx_tune sets the PROCESS DEFAULT of a switch.  A lattice never reads the defaults while it works: every build
 * (plx_build, plx_build_local, plx_filter) starts by copying them into the lattice, and every later call on that lattice
 * (its merge, tables, MVMs, exports) runs under that copy -- so a plx_tune call changes nothing for lattices that are
 * already built, two lattices built under different settings keep their own, and a thread that tunes while another
 * thread is inside a plx_* call on a built lattice does not disturb it (concurrent plx_tune and build calls still need
 * the caller's own ordering: the copy is a plain struct copy).  plx_lattice_tune changes a switch in ONE lattice's copy,
 * effective from its next call until its next build (A/B of MVM-side variants over the same tables).  Keys (default):
 *   "sort_points" (1; 0 keeps the caller's point order), "order_zcurve" (1; 0 = lexicographic point order, 2 = Z-curve of
 *   the blur-axis coordinates), "order_compact" (1 = point-order keys laid out over exactly the bits each coordinate's
 *   range needs; 0 = a fixed 7 bits per coordinate), "readback_spin" (1 = counts come back through the mailbox; 0 = stream
 *   synchronisation), "vertex_order" (1: vertices numbered along the Morton curve of their blur-axis coordinates where
 *   that pays, 65536 <= m <= 0.9 n (d+1); 0: always by first touch; 2: always Morton -- vertex ids are internal, the
 *   PLX_ARRAY_* exports are in whichever numbering the build used), "insert_dedupe" (2 = every key of a wave probes the table once: lanes with equal keys are grouped by hash ballots; 1 = runs of
 *   equal NEIGHBOURING lanes probe once; 0 = every lane probes), "insert_plane_fast" (1 = the
 *   d+1 corner planes of a run of points are adjacent workgroups of the hashed insert / neighbour lookups; 0 =
 *   plane-major launch order), "nbr_symmetric" (1), "compact_nbr" (1 = when under a quarter of the neighbour slots exist;
 *   0 never, 2 always), "blur_vpt" (4; vertices per thread at vd = 1: 2 or 4, anything else selects the general kernel),
 *   "blur_small" (1), "blur_narrow" (1), "blur_multi" (1 = the wide-row blur kernels from 17 chunks per row; 2 = from 32, the
 *   gate of rounds 1-5), "blur_fuse" (1), "blur_fuse_vec" (1 = two blur axes per launch
 *   for rows of 2..4 chunks; 0 = one), "splat_direct" (1), "splat_group" (1), "splat_wide" (1 = the row-parallel splat from 17 chunks per
 *   row; 2 = from 32), "xcd_remap" (1),
 *   "block_path" (1 = block tables for vd = 1 when corners share vertices; 0 never, 2 whenever representable),
 *   "block_e" (0 = corners per thread of the block kernels chosen per lattice; 16 or 24: a block holds 256 * e corners),
 *   "block_dense_combine" (1), "scatter_store" (0), "unpermute_gather" (1), "nbr_window" (512: on Morton-numbered lattices a neighbour lookup first binary-searches this
 *   many sorted vertex codes next to the vertex -- found, or proven absent, without touching the hash table when the window
 *   brackets the target; 0 = hash table only), "nbr_bitmap" (1 = neighbour lookups test a
 *   slot-occupancy bitmap before the hash table when m >= 2^22; 0 never, 2 always), "splat_first" (1 = single-column splat by
 *   first-touch stores + a short extras list when m >= 0.9 nnz; 0 never, 2 whenever representable, 3 = 2 with scattered
 *   stores), "perm_rows" (1 = multi-column row permutations by 16-byte chunks / LDS-transposed whole-line stores; 0 = the
 *   per-float forms), "reference_growth" (0; 1 = replay the reference CPU path's table-growth quirk: plx_reference_growth_info; 2 = the same by
 *   running every lookup, the checker of 1), "blur_active" (1: wide rows on sparse lattices -- centre tap 1 -- blur only the vertices with a neighbour on the axis, in place; 0 never, 2 whenever representable),
 *   "contract_v" (1 = the fused backward's slice + contraction with the corner count
 *   compiled in; 0 = the run-time form),
 *   and the round-5 build switches "nbr_sliced" (1), "nbr_seed" (1), "assign_evid" (1), "insert_xcd" (2), "order_sample" (8),
 *   "embed_vrange" (0), "blk_sort" (15): DESIGN.md 2.  (Round 6 removed "hash_v", "table_fp", "flag_own" and "insert_v" with the
 *   measured-loser code paths behind them: the linear hash, fingerprints, own-mark flags and the point-per-thread insert
 *   are what every build uses.)
 * The diagnostic ablations "splat_ablate" / "blur_ablate" / "block_ablate" exist only in libplx_diag.so (make diag).
 * Unknown keys return PLX_ERR_INVALID. */
int plx_tune(const char *key, int value);
int plx_lattice_tune(plx_lattice *lat, const char *key, int value);

/* Names of the kernels the last plx_splat / plx_blur / plx_slice (or plx_apply) on this lattice launched, as
 * "splat=a+b;blur_axis=c;slice=d;vertex_order=morton|first_touch" -- the names rocprofv3 --kernel-trace shows (without
 * template arguments) and the vertex numbering of the last build, so that a bench line can name what actually ran. */
int plx_last_kernels(const plx_lattice *lat, char *buf, int cap);
/* The splat / slice tables of a built lattice (block tables, their vertex-sorted half, the vertex-sorted CSR) are built
 * by their first user: the first plx_splat / plx_slice / plx_apply that needs them, on that call's stream -- a CG-only
 * caller never pays for single-column tables and vice versa.  plx_prepare builds, on `stream`, everything an MVM with
 * vd columns will read, so that the first MVM costs what every later one does (and so that a profile can time the
 * tables apart from the MVM).  Idempotent. */
int plx_prepare(plx_lattice *lat, int vd, void *stream);
/* Block rows of the lattice's block tables (coarse lattices: the owned points are cut into blocks, a block row = one
 * distinct vertex of one block), or 0 when the tables have not been built (see plx_prepare) or the lattice uses the
 * vertex-sorted CSR path instead.  A pure query: no launch, no synchronisation. */
int64_t plx_block_rows(const plx_lattice *lat);

/* Self test of the build's stable radix sort (plx_radix.h) on the current device: n (key, position) pairs with many
 * duplicate keys of key_bytes (4 or 8) bytes and end_bit significant bits are sorted and checked on the device --
 * ascending, equal keys in input order, every value still with its key.  *mismatches = offending positions (0 = pass).
 * Synchronises the stream.  For tests; no lattice needed. */
int plx_selftest_sort(int64_t n, int key_bytes, int end_bit, uint64_t seed, void *stream, int64_t *mismatches);

/* Per-stage device time of the last plx_build on this lattice, in ms, in the
 * order {order+embed, insert, number, ids, neighbours, csr}; 0 when timing is off.
 * plx_set_timing(lat, 1) turns hipEvent timing on (adds event records only). */
int plx_set_timing(plx_lattice *lat, int on);
int plx_build_times(const plx_lattice *lat, float *h_ms6);
/* Per-stage device time of the last plx_apply (timing on), in ms: {splat (scan +
 * fix-up launch), blur (all d+1 launches), slice}.  One hipEvent pair per stage
 * on the apply stream, so a stage time divided by its launch count is directly
 * comparable with rocprofv3's per-kernel average.  Synchronises on the last event. */
int plx_apply_times(plx_lattice *lat, float *h_ms, int cap, int *count);

#ifdef __cplusplus
}
#endif
#endif /* PLX_H */

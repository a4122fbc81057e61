/*
 * lattice_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's permutohedral
 * lattice filter (splat -> blur -> slice), written to be the parity oracle for
 * the HIP path in simplex_gp_amd/csrc.  Nothing in the product path may link,
 * import or call this file: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Every function cites the reference lines it follows; "h" is
 * /root/reference/gpytorch_lattice_kernel/cpp/permutohedral.h.
 *
 * Parity status: PINNED.  the .npz files under tests/golden/ hold inputs and outputs produced
 * by the reference's own CPU extension (built from /root/reference by
 * oracle/build_ref.py into oracle/_ref/, script tests/golden/make_golden.py);
 * tests/test_oracle_golden.py checks this file against every one of them
 * (bit-for-bit on outputs, identical vertex count m, and stage by stage -- keys,
 * greedy / rank, values after splat and after blur -- on the stage dumps).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the reference's JIT
 * build emits no FMA on baseline x86-64, so contraction must stay off for the
 * discrete front end -- rounding, ranks -- to agree bit for bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int key_idx;   /* offset of the key in keys[] (= vertex * d), -1 if empty */
    int val_idx;   /* vertex id (the reference stores vertex * vd)          */
} plxo_entry;

typedef struct plxo_lattice {
    int d;                 /* position dimension                             */
    long n;                /* number of points                               */
    /* hash table of lattice vertices, h:28-175 */
    size_t capacity;
    size_t filled;         /* = m, number of vertices                        */
    long grow_lookups;     /* lookups that triggered a grow (quirk sites)    */
    plxo_entry *entries;
    short *keys;           /* [capacity/2][d], first-touch order             */
    /* per-point simplex structure ("replay"), h:581-584 */
    int *entry_vertex;     /* [n*(d+1)] vertex id                            */
    float *entry_weight;   /* [n*(d+1)] barycentric weight                   */
    /* per-point discrete front end, kept for stage checks */
    short *greedy;         /* [n*(d+1)] after the sum fix-up                 */
    signed char *rank;     /* [n*(d+1)] after the sum fix-up                 */
    float *scale_factor;   /* [d]                                            */
    short *canonical;      /* [(d+1)*(d+1)]                                  */
} plxo_lattice;

/* ------------------------------------------------------------------ hash */

/* h:114-121: base conversion with multiplier 2531011 in size_t; the short
 * is sign-extended when added. */
static size_t plxo_hash(const short *key, int d)
{
    size_t k = 0;
    for (int i = 0; i < d; i++) {
        k += (size_t)(long)key[i];
        k *= 2531011;
    }
    return k;
}

/* h:125-161: double the capacity and re-insert every entry by linear probing */
static void plxo_grow(plxo_lattice *L)
{
    size_t old_cap = L->capacity;
    L->capacity *= 2;
    L->keys = (short *)realloc(L->keys, sizeof(short) * L->d * (L->capacity / 2));
    plxo_entry *ne = (plxo_entry *)malloc(sizeof(plxo_entry) * L->capacity);
    for (size_t i = 0; i < L->capacity; i++) { ne[i].key_idx = -1; ne[i].val_idx = -1; }
    for (size_t i = 0; i < old_cap; i++) {
        if (L->entries[i].key_idx == -1) continue;
        size_t h = plxo_hash(L->keys + L->entries[i].key_idx, L->d) % L->capacity;
        while (ne[h].key_idx != -1) { h++; if (h == L->capacity) h = 0; }
        ne[h] = L->entries[i];
    }
    free(L->entries);
    L->entries = ne;
}

/* h:58-94 + h:104-111: lookup-or-create; returns the vertex id or -1.
 * Vertex ids are handed out in first-touch order (h:73-79). */
/* Reference quirk (h:105-106 then h:61-63): the bucket is computed with the
 * capacity in force BEFORE lookupOffset() grows the table, so the one lookup
 * that triggers a grow probes the new table from a stale bucket.  When the
 * key's new home is in the upper half, a key that is already present is not
 * found: with create=true a duplicate vertex is appended (m grows by one and
 * that one point/vertex pair talks to an orphan); with create=false (the first
 * blur lookup after a grow) one neighbour reads as absent.  At most one lookup
 * per doubling is affected.  plxo_exact_mode = 1 (default) reproduces this bit
 * for bit; 0 re-hashes after the grow, which is the duplicate-free lattice the
 * HIP path builds. */
static int plxo_exact_mode = 1;
void plxo_set_exact_mode(int on) { plxo_exact_mode = on; }

static int plxo_lookup(plxo_lattice *L, const short *key, int create)
{
    const int d = L->d;
    size_t h = plxo_hash(key, d) % L->capacity;             /* h:105 */
    if (L->filled >= (L->capacity / 2) - 1) {               /* h:61-63 */
        plxo_grow(L);
        if (!plxo_exact_mode) h = plxo_hash(key, d) % L->capacity;
        L->grow_lookups++;
    }
    for (;;) {
        plxo_entry e = L->entries[h];
        if (e.key_idx == -1) {                              /* h:69-80 */
            if (!create) return -1;
            for (int i = 0; i < d; i++) L->keys[L->filled * d + i] = key[i];
            e.key_idx = (int)(L->filled * d);
            e.val_idx = (int)L->filled;
            L->entries[h] = e;
            L->filled++;
            return e.val_idx;
        }
        int match = 1;                                      /* h:83-87 */
        for (int i = 0; i < d && match; i++) match = L->keys[e.key_idx + i] == key[i];
        if (match) return e.val_idx;
        h++;                                                /* h:90-92 */
        if (h == L->capacity) h = 0;
    }
}

/* ------------------------------------------------------------ front end */

/* h:203-219: index-moment variance of the taps, all in fp32 */
float plxo_variance(const float *coeffs, int R)
{
    float mom0 = 0, mom1 = 0.f, mom2 = 0.f;
    for (int i = 0; i < R; ++i) {
        float c = coeffs[i];
        mom0 += c;
        mom1 += i * c;
        mom2 += i * i * c;
    }
    float mean = mom1 / mom0;
    return mom2 / mom0 - mean * mean;
}

/* h:372-390: scaleFactor[i] = 1/sqrt((i+1)(i+2)) * (d+1) * sqrt(var + 1/6) */
void plxo_scale_factors(int d, const float *coeffs, int R, float *sf)
{
    for (int i = 0; i < d; i++) {
        sf[i] = 1.0f / (sqrtf((float)(i + 1) * (i + 2)));
        float sigma_blur = plxo_variance(coeffs, R);
        sf[i] *= (d + 1) * sqrtf(sigma_blur + 1.0f / 6.0f);
    }
}

/* h:395-465: elevate, round to the nearest zero-colour vertex, rank, fix up,
 * barycentric weights.  Outputs greedy[d+1], rank[d+1], bary[d+2]. */
void plxo_embed_point(int d, const float *sf, const float *position,
                      short *greedy, signed char *rank, float *bary, float *elevated)
{
    /* h:398-402 */
    elevated[d] = -d * position[d - 1] * sf[d - 1];
    for (int i = d - 1; i > 0; i--)
        elevated[i] = (elevated[i + 1] - i * position[i - 1] * sf[i - 1] +
                       (i + 2) * position[i] * sf[i]);
    elevated[0] = elevated[1] + 2 * position[0] * sf[0];

    /* h:405-423 */
    float scale = 1.0f / (d + 1);
    int sum = 0;
    for (int i = 0; i <= d; i++) {
        float v = elevated[i] * scale;
        float up = ceilf(v) * (d + 1);
        float down = floorf(v) * (d + 1);
        if (up - elevated[i] < elevated[i] - down) greedy[i] = (short)up;
        else greedy[i] = (short)down;
        sum += greedy[i];
    }
    sum *= scale;   /* int * float -> float -> truncated back to int, h:423 */

    /* h:427-433 */
    memset(rank, 0, sizeof(signed char) * (d + 1));
    for (int i = 0; i < d; i++)
        for (int j = i + 1; j <= d; j++)
            if (elevated[i] - greedy[i] < elevated[j] - greedy[j]) rank[i]++;
            else rank[j]++;

    /* h:435-457 */
    if (sum > 0) {
        for (int i = 0; i <= d; i++) {
            if (rank[i] >= d + 1 - sum) { greedy[i] -= d + 1; rank[i] += sum - (d + 1); }
            else rank[i] += sum;
        }
    } else if (sum < 0) {
        for (int i = 0; i <= d; i++) {
            if (rank[i] < -sum) { greedy[i] += d + 1; rank[i] += (d + 1) + sum; }
            else rank[i] += sum;
        }
    }

    /* h:460-465 */
    memset(bary, 0, sizeof(float) * (d + 2));
    for (int i = 0; i <= d; i++) {
        bary[d - rank[i]] += (elevated[i] - greedy[i]) * scale;
        bary[d + 1 - rank[i]] -= (elevated[i] - greedy[i]) * scale;
    }
    bary[0] += 1.0f + bary[d + 1];
}

/* ------------------------------------------------------------- lifecycle */

void plxo_free(plxo_lattice *L)
{
    if (!L) return;
    free(L->entries); free(L->keys); free(L->entry_vertex); free(L->entry_weight);
    free(L->greedy); free(L->rank); free(L->scale_factor); free(L->canonical);
    free(L);
}

/* h:346-392 (constructor) + the structural half of splat, h:395-486, run
 * over all points in order (h:294-296).  Values are not touched here: the
 * reference fuses value accumulation into splat; plxo_splat() below replays
 * it in the same order so the sums are bit-identical. */
plxo_lattice *plxo_build(const float *ref, long n, int d, const float *coeffs, int R)
{
    plxo_lattice *L = (plxo_lattice *)calloc(1, sizeof(plxo_lattice));
    L->d = d; L->n = n;
    L->capacity = 1 << 15;                                   /* h:35 */
    L->filled = 0;
    L->entries = (plxo_entry *)malloc(sizeof(plxo_entry) * L->capacity);
    for (size_t i = 0; i < L->capacity; i++) { L->entries[i].key_idx = -1; L->entries[i].val_idx = -1; }
    L->keys = (short *)malloc(sizeof(short) * d * (L->capacity / 2));
    L->entry_vertex = (int *)malloc(sizeof(int) * n * (d + 1));
    L->entry_weight = (float *)malloc(sizeof(float) * n * (d + 1));
    L->greedy = (short *)malloc(sizeof(short) * n * (d + 1));
    L->rank = (signed char *)malloc(sizeof(signed char) * n * (d + 1));
    L->scale_factor = (float *)malloc(sizeof(float) * d);
    L->canonical = (short *)malloc(sizeof(short) * (d + 1) * (d + 1));

    /* h:364-369 */
    for (int i = 0; i <= d; i++) {
        for (int j = 0; j <= d - i; j++) L->canonical[i * (d + 1) + j] = i;
        for (int j = d - i + 1; j <= d; j++) L->canonical[i * (d + 1) + j] = i - (d + 1);
    }
    plxo_scale_factors(d, coeffs, R, L->scale_factor);

    float *elevated = (float *)malloc(sizeof(float) * (d + 1));
    float *bary = (float *)malloc(sizeof(float) * (d + 2));
    short *key = (short *)malloc(sizeof(short) * (d + 1));
    for (long p = 0; p < n; p++) {
        short *greedy = L->greedy + p * (d + 1);
        signed char *rank = L->rank + p * (d + 1);
        plxo_embed_point(d, L->scale_factor, ref + p * d, greedy, rank, bary, elevated);
        /* h:468-485 */
        for (int r = 0; r <= d; r++) {
            for (int i = 0; i < d; i++)
                key[i] = greedy[i] + L->canonical[r * (d + 1) + rank[i]];
            int v = plxo_lookup(L, key, 1);
            L->entry_vertex[p * (d + 1) + r] = v;
            L->entry_weight[p * (d + 1) + r] = bary[r];
        }
    }
    free(elevated); free(bary); free(key);
    return L;
}

long plxo_num_vertices(const plxo_lattice *L) { return (long)L->filled; }
long plxo_grow_lookups(const plxo_lattice *L) { return L->grow_lookups; }
const short *plxo_keys(const plxo_lattice *L) { return L->keys; }
const int *plxo_entry_vertex(const plxo_lattice *L) { return L->entry_vertex; }
const float *plxo_entry_weight(const plxo_lattice *L) { return L->entry_weight; }
const short *plxo_greedy(const plxo_lattice *L) { return L->greedy; }
const signed char *plxo_rank(const plxo_lattice *L) { return L->rank; }
const float *plxo_scale(const plxo_lattice *L) { return L->scale_factor; }

/* ----------------------------------------------------------------- stages */

/* h:478-479, in point order then vertex order r = 0..d: values[m*vd] += w*v */
void plxo_splat(const plxo_lattice *L, const float *src, int vd, float *values)
{
    const int d = L->d;
    memset(values, 0, sizeof(float) * vd * L->filled);
    for (long p = 0; p < L->n; p++)
        for (int r = 0; r <= d; r++) {
            float *val = values + (size_t)L->entry_vertex[p * (d + 1) + r] * vd;
            float w = L->entry_weight[p * (d + 1) + r];
            for (int i = 0; i < vd; i++) val[i] += (w * src[p * vd + i]);
        }
}

/* h:539-544: ids of the 2r neighbours of every vertex on every axis,
 * nbr[(j*2r + s)*m + i]; slot s enumerates nid = -r..-1, 1..r; -1 = absent. */
void plxo_neighbors(plxo_lattice *L, int R, int *nbr)
{
    const int d = L->d, order = R / 2;
    const long m = (long)L->filled;
    short *neighbor = (short *)malloc(sizeof(short) * (d + 1));
    for (int j = 0; j <= d; j++)
        for (long i = 0; i < m; i++) {
            const short *key = L->keys + i * d;
            int s = 0;
            for (int nid = -order; nid <= order; ++nid) {
                if (nid == 0) continue;
                for (int k = 0; k < d; k++) neighbor[k] = key[k] - nid;
                if (j < d) neighbor[j] = key[j] + nid * d;   /* j == d: only d coords are hashed, h:541-544 */
                nbr[((size_t)j * 2 * order + s) * m + i] = plxo_lookup(L, neighbor, 0);
                s++;
            }
        }
    free(neighbor);
}

/* h:513-572: for each axis, Jacobi pass new[i] = sum_nid c[nid+r] * old[nbr],
 * accumulated from zero in tap order nid = -r..r; values updated in place. */
void plxo_blur(plxo_lattice *L, float *values, int vd, const float *coeffs, int R)
{
    const int d = L->d, order = R / 2;
    const long m = (long)L->filled;
    short *neighbor = (short *)malloc(sizeof(short) * (d + 1));
    float *new_value = (float *)calloc((size_t)vd * m + 1, sizeof(float));
    float *old_value = values;
    for (int j = 0; j <= d; j++) {
        for (long i = 0; i < m; i++) {
            const short *key = L->keys + i * d;
            float *nv = new_value + i * vd;
            for (int k = 0; k < vd; k++) nv[k] = 0;
            for (int nid = -order; nid <= order; ++nid) {
                for (int k = 0; k < d; k++) neighbor[k] = key[k] - nid;
                if (j < d) neighbor[j] = key[j] + nid * d;
                int v = plxo_lookup(L, neighbor, 0);
                float c = coeffs[nid + order];
                if (v >= 0) {
                    const float *val = old_value + (size_t)v * vd;
                    for (int k = 0; k < vd; k++) nv[k] += c * val[k];
                } else {
                    for (int k = 0; k < vd; k++) nv[k] += c * 0.0f;   /* "zero" vector, h:545 */
                }
            }
        }
        float *tmp = new_value; new_value = old_value; old_value = tmp;
    }
    if (old_value != values) {                                /* h:559-563 */
        memcpy(values, old_value, sizeof(float) * vd * m);
        free(old_value);
    } else {
        free(new_value);
    }
    free(neighbor);
}

/* h:497-510: out[p][c] = sum_r w_r * values[v_r][c] / (1 + 2^-d) */
void plxo_slice(const plxo_lattice *L, const float *values, int vd, float *out)
{
    const int d = L->d;
    for (long p = 0; p < L->n; p++) {
        float *col = out + p * vd;
        for (int j = 0; j < vd; j++) col[j] = 0;
        for (int i = 0; i <= d; i++) {
            float w = L->entry_weight[p * (d + 1) + i];
            const float *base = values + (size_t)L->entry_vertex[p * (d + 1) + i] * vd;
            for (int j = 0; j < vd; j++)
                col[j] += w * base[j] / (1 + powf(2, -d));
        }
    }
}

/* h:259-340: the whole filter; returns the vertex count through m_out. */
int plxo_filter(const float *src, const float *ref, long n, int d, int vd,
                const float *coeffs, int R, float *out, long *m_out)
{
    if (n <= 0 || d <= 0 || vd <= 0 || R <= 0 || (R % 2) == 0) return 1;
    plxo_lattice *L = plxo_build(ref, n, d, coeffs, R);
    float *values = (float *)calloc((size_t)vd * L->filled + 1, sizeof(float));
    plxo_splat(L, src, vd, values);
    plxo_blur(L, values, vd, coeffs, R);
    plxo_slice(L, values, vd, out);
    if (m_out) *m_out = (long)L->filled;
    free(values);
    plxo_free(L);
    return 0;
}

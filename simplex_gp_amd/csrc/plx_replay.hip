// plx_replay.hip -- plx_tune("reference_growth", 1): the reference CPU path's hash-table-growth quirk, replayed.
//
// What the reference does (cpp/permutohedral.h, "h").  lookup() hashes a key with the capacity in force BEFORE
// lookupOffset() doubles the table (h:104-106 against h:58-63), so the ONE lookup that triggers each doubling probes the
// new table from a stale bucket.  It can then miss a key that is there and create a second entry for it (h:69-79: one
// vertex more than the lattice has keys), or create a new key where later lookups, which start from the right bucket, do
// not find it.  grow() re-places every entry in old-position order (h:146-156), so after the NEXT doubling it is the
// misplaced entry that lookups meet first and the regular one that is orphaned.  A doubling can also fall on the first
// lookup of blur() (h:545), which then reads one existing neighbour as absent.
//
// What that does to the OUTPUT follows from one observation: blur() finds every operand by KEY (h:539-545), the centre tap
// included (nid = 0 is a lookup of the vertex's own key).  All entries of one key therefore get the same blurred value, and
// the value splatted into an entry that lookups do not resolve to is never read by anything.  So the reference's filter is
// the duplicate-free lattice's, except that
//   (1) the splat contributions of the (point, corner) pairs whose lookup returned an entry other than the one blur-time
//       lookups resolve the key to are DROPPED (their slice reads are unchanged);
//   (2) a key whose entries are all unreachable at blur time is INVISIBLE: its neighbours read it as absent, its own centre
//       tap reads zero in every pass, all its splat contributions are dropped;
//   (3) the blur-time doubling can make vertex 0's first neighbour lookup (axis 0, tap -order) read absent.
// Verified on the CPU against the reference-exact oracle before this was written (numpy restatement of (1)-(3) over the
// clean lattice: 0.0 rel-L2 on every quirk case tried, duplicates and invisible vertices alike).
//
// This file replays ONLY THE TABLE LAYOUT of h:58-161 on the host -- entry positions, no values: the vertex id of every
// (point, corner) in the caller's order and the vertex keys are copied from the built (duplicate-free) lattice, the
// N (d+1) lookups of splat() are run against a layout-exact model of the reference's table (same hash h:114-121, same
// probing, same growth condition and migration order), then blur()'s first lookup and one resolving lookup per key.  The
// result is a short list of dropped corners, invisible vertices and at most one missed neighbour, which three small
// kernels apply to the device structure: a copy of the barycentric weights for the SPLAT side with the dropped corners
// zeroed, -1 in the neighbour rows that point to an invisible vertex, and a centre-tap correction after each blur axis.
// Product code: nothing here touches oracle/.  An opt-in parity mode, off by default -- the default lattice is the
// duplicate-free one, which is also what the reference's CUDA path builds (its table never grows, cu:61).
//
// Cost (round 6).  The table's layout changes only when an entry is CREATED; a lookup that finds its entry leaves no trace.
// A key that was never touched by a stale probe has exactly one entry, placed by regular probing, and regular probing
// finds it for ever (linear probing without deletions; grow() re-places it regularly): its lookups need not be run.  So
// reference_growth = 1 replays EVENTS instead of lookups: the m first-touch creations in caller order (the GPU finds every
// vertex's first lookup with one atomicMin pass and sorts the vertices by it), the ONE lookup behind each creation that
// fills the table to its growth threshold (the stale probe, h:58-63), and from then on every lookup of the few keys a stale
// probe has touched ("affected" keys: they may own a misplaced or a second entry, so each of their lookups is run against
// the table as it stands at that moment -- it may create yet another entry).  O(m) host work with the probe targets
// prefetched, instead of O(N (d+1)) dependent cache misses: N = 1e6, d = 8, l = 0.6931: about a second -> tens of
// milliseconds.  reference_growth = 2 keeps the full lookup-by-lookup replay (the checker of the event form:
// tests/test_hip_parity.py::test_reference_growth_event_replay_equals_full_replay).

#include "plx_internal.h"
#include "plx_kernels.h"

#include <algorithm>
#include <queue>
#include <unordered_map>
#include <vector>

namespace plx {

// vat[e] = vertex of the e-th lookup of the reference's splat loop, e = caller row * (d+1) + corner (h:395-485)
__global__ __launch_bounds__(kBlock) void replay_vat_kernel(const int *__restrict__ evid, const uint32_t *__restrict__ perm, int n,
                                                            int d1, int *__restrict__ vat)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const size_t row = perm[p];
    for (int r = 0; r < d1; ++r) vat[row * d1 + r] = evid[(size_t)r * n + p];
}

// the same, and first[v] = the smallest lookup index e that names vertex v (first preset to 0xFFFFFFFF)
__global__ __launch_bounds__(kBlock) void replay_vat_first_kernel(const int *__restrict__ evid, const uint32_t *__restrict__ perm, int n,
                                                                  int d1, int *__restrict__ vat, uint32_t *__restrict__ first,
                                                                  uint32_t *__restrict__ ids, int m)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p < m) ids[p] = (uint32_t)p;
    if (p >= n) return;
    const size_t row = perm[p];
    for (int r = 0; r < d1; ++r) {
        const int v = evid[(size_t)r * n + p];
        vat[row * d1 + r] = v;
        atomicMin(&first[v], (uint32_t)(row * d1 + r));
    }
}

// every lookup index e with vat[e] == v, in any order (the host sorts the handful); count may exceed cap: the caller retries
__global__ __launch_bounds__(kBlock) void replay_match_kernel(const int *__restrict__ vat, int64_t E, int v, int *__restrict__ out,
                                                              int cap, int *__restrict__ count)
{
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E || vat[e] != v) return;
    const int k = atomicAdd(count, 1);
    if (k < cap) out[k] = (int)e;
}

// ew_splat[r][p] = 0 for the dropped lookups (given by their caller-order index e)
__global__ __launch_bounds__(kBlock) void replay_drop_kernel(const int *__restrict__ dropped, int count, const uint32_t *__restrict__ inv_perm,
                                                             int n, int d1, float *__restrict__ ew_splat)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const int e = dropped[k];
    const int row = e / d1, r = e - row * d1;
    ew_splat[(size_t)r * n + inv_perm[row]] = 0.f;
}

// nobody finds an invisible vertex: the mirror entries of its own row become -1 (its own row stays: it finds its neighbours)
__global__ __launch_bounds__(kBlock) void replay_hide_kernel(const int *__restrict__ invisible, int count, int d1, int order,
                                                             int64_t mstride, int *__restrict__ nbr)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count * d1 * 2 * order) return;
    const int v = invisible[k / (d1 * 2 * order)];
    const int rest = k % (d1 * 2 * order), axis = rest / (2 * order), s = rest % (2 * order);
    int *plane = nbr + (size_t)axis * 2 * order * mstride;
    const int u = plane[(size_t)s * mstride + v];
    if (u >= 0) plane[(size_t)(2 * order - 1 - s) * mstride + u] = -1;      // tap -nid of u pointed at v
}

__global__ void replay_set_kernel(int *__restrict__ nbr, size_t index, int value) { nbr[index] = value; }

// after one blur axis: the centre tap of an invisible vertex read zero (h:545: the lookup of its own key fails)
__global__ __launch_bounds__(kBlock) void replay_nocentre_kernel(const float *__restrict__ old_values, float *__restrict__ new_values,
                                                                 const int *__restrict__ list, int count, int vdp, float c0)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count * vdp) return;
    const size_t at = (size_t)list[k / vdp] * vdp + k % vdp;
    new_values[at] -= c0 * old_values[at];
}

namespace {

// layout-exact model of HashTablePermutohedral (h:28-175) without the values: entries[position] = entry id or -1
struct RefTable {
    uint64_t cap = 1ull << 15;                     // h:35
    std::vector<int32_t> entries;                  // position -> entry id
    std::vector<int32_t> ekey;                     // entry id -> vertex (its key), creation order (h:73-79)
    const uint64_t *hk = nullptr;                  // reference hash of every vertex key
    int grows = 0;

    RefTable() : entries(cap, -1) {}

    void grow()                                    // h:125-161
    {
        std::vector<int32_t> old;
        old.swap(entries);
        cap *= 2;
        entries.assign(cap, -1);
        for (uint64_t pos = 0; pos < old.size(); ++pos) {
            const int32_t e = old[pos];
            if (e < 0) continue;
            uint64_t h = hk[ekey[e]] % cap;
            while (entries[h] != -1) { if (++h == cap) h = 0; }
            entries[h] = e;
        }
        ++grows;
    }

    // h:104-106 + h:58-95: hash with the capacity in force, THEN the growth check, then the probe from that (stale) bucket.
    // v = the vertex whose key is looked up, or -1 for a key no vertex has.
    int32_t lookup(uint64_t hash, int32_t v, bool create)
    {
        uint64_t h = hash % cap;
        if (ekey.size() >= cap / 2 - 1) grow();
        for (;;) {
            const int32_t e = entries[h];
            if (e == -1) {
                if (!create) return -1;
                entries[h] = (int32_t)ekey.size();
                ekey.push_back(v);
                return (int32_t)ekey.size() - 1;
            }
            if (v >= 0 && ekey[e] == v) return e;
            if (++h == cap) h = 0;
        }
    }
};

uint64_t ref_hash(const int16_t *key, int d)       // h:114-121 (size_t arithmetic, keys sign-extended)
{
    uint64_t k = 0;
    for (int i = 0; i < d; ++i) {
        k += (uint64_t)(int64_t)key[i];
        k *= 2531011ull;
    }
    return k;
}

}  // namespace

// ---- reference_growth = 1: the event form (header comment, "Cost") ---------------------------------------------------

// hk[v] = the reference's hash of vertex v's key (h:114-121), from the packed device keys
__global__ __launch_bounds__(kBlock) void replay_hash_kernel(const uint32_t *__restrict__ vkeys, int m, int d, int dw,
                                                             unsigned long long *__restrict__ hk)
{
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= m) return;
    unsigned long long k = 0;
    for (int c = 0; c < d; ++c) {
        const int16_t kc = (int16_t)((vkeys[(size_t)v * dw + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFFu);
        k += (unsigned long long)(long long)kc;
        k *= 2531011ull;
    }
    hk[v] = k;
}

namespace {

// The same table model holding LABELS instead of entry ids: label v < m = the first entry created for vertex v, label
// m + x = a further entry of vertex extra[x] (only keys a stale probe has touched ever own one).  Same hash, probing,
// growth condition and migration order as RefTable; entry ids are not needed because only the affected keys' lookups are
// compared entry by entry, and their entries are told apart by label.
struct EventTable {
    uint64_t cap = 1ull << 15;                     // h:35
    std::vector<int32_t> tab;                      // position -> label or -1
    std::vector<int32_t> extra;                    // label - m -> vertex
    std::vector<uint8_t> made;                     // vertex -> its first entry exists
    const uint64_t *hk = nullptr;
    int64_t m = 0, filled = 0;
    int grows = 0;

    EventTable(int64_t m_, const uint64_t *hk_) : tab(cap, -1), made((size_t)m_, 0), hk(hk_), m(m_) {}
    int32_t vertex_of(int32_t label) const { return label < m ? label : extra[(size_t)(label - m)]; }
    bool due() const { return (uint64_t)filled >= cap / 2 - 1; }       // h:61 (size_t arithmetic)

    void grow()                                    // h:125-161: re-place every entry in old-position order
    {
        std::vector<int32_t> old;
        old.swap(tab);
        cap *= 2;
        tab.assign(cap, -1);
        const size_t n_old = old.size();
        for (size_t pos = 0; pos < n_old; ++pos) {
            if (pos + 32 < n_old && old[pos + 32] >= 0) __builtin_prefetch(&hk[vertex_of(old[pos + 32])]);
            const int32_t l = old[pos];
            if (l < 0) continue;
            uint64_t h = hk[vertex_of(l)] & (cap - 1);   // cap is a power of two: hash % cap (h:104) is a mask.  (Old-position order visits the new homes in two rising runs: cache friendly)
            while (tab[h] != -1) { if (++h == cap) h = 0; }
            tab[h] = l;
        }
        ++grows;
    }

    // h:104-106 + h:58-95, as RefTable::lookup; *created tells whether the probe ended on an empty slot and made an entry
    int32_t lookup(int32_t v, bool create, bool *created)
    {
        uint64_t h = hk[v] & (cap - 1);             // = hash % cap with the capacity in force BEFORE the growth check
        if (due()) grow();
        for (;;) {
            const int32_t l = tab[h];
            if (l == -1) {
                if (!create) return -1;
                int32_t label = v;
                if (made[v]) { label = (int32_t)(m + (int64_t)extra.size()); extra.push_back(v); }
                made[v] = 1;
                tab[h] = label;
                ++filled;
                if (created) *created = true;
                return label;
            }
            if (l == v || (l >= m && extra[(size_t)(l - m)] == v)) return l;
            if (++h == cap) h = 0;
        }
    }
};

}  // namespace

static int replay_events(plx_lattice *L, hipStream_t stream, std::vector<int> &dropped, std::vector<int> &invisible)
{
    const int n = (int)L->n, d = L->d, d1 = d + 1, dw = (d + 1) / 2, order = L->order;
    const int64_t m = L->m, E = (int64_t)n * d1;
    if (E >= (1ll << 31)) { set_error("reference_growth: %lld lookups exceed the replay's 31-bit index", (long long)E); return PLX_ERR_TOO_LARGE; }
    // ---- device: lookup -> vertex in caller order, every vertex's first lookup, the vertices sorted by it, their hashes
    PLX_TRY(ensure(L->replay_vat, (size_t)E * 4));
    PLX_TRY(ensure(L->replay_keys, (size_t)m * 4 * 4 + (size_t)m * 8 + 64));
    PLX_TRY(ensure(L->sort_temp, radix_temp_bytes(m)));
    uint32_t *first_a = L->replay_keys.as<uint32_t>(), *first_b = first_a + m, *ids_a = first_b + m, *ids_b = ids_a + m;
    unsigned long long *d_hk = reinterpret_cast<unsigned long long *>(ids_b + m);     // (16 m bytes in: 8-byte aligned)
    int *d_count = reinterpret_cast<int *>(d_hk + m);
    PLX_HIP_TRY(hipMemsetAsync(first_a, 0xFF, (size_t)m * 4, stream));
    replay_vat_first_kernel<<<ceil_div(n > m ? (int64_t)n : m, kBlock), kBlock, 0, stream>>>(
        L->evid.as<int>(), L->perm.as<uint32_t>(), n, d1, L->replay_vat.as<int>(), first_a, ids_a, (int)m);
    replay_hash_kernel<<<ceil_div(m, kBlock), kBlock, 0, stream>>>(L->vkeys.as<uint32_t>(), (int)m, d, dw, d_hk);
    int ebits = 1;
    while ((1ll << ebits) < E) ++ebits;
    int second = 0;
    PLX_TRY(radix_sort_pairs32(L->sort_temp.p, first_a, first_b, ids_a, ids_b, m, ebits, &second, stream));
    std::vector<uint32_t> ce((size_t)m), cv((size_t)m);          // creation order: vertex cv[i] is first looked up at lookup ce[i]
    std::vector<uint64_t> hk((size_t)m);
    PLX_HIP_TRY(hipMemcpyAsync(ce.data(), second ? first_b : first_a, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
    PLX_HIP_TRY(hipMemcpyAsync(cv.data(), second ? ids_b : ids_a, (size_t)m * 4, hipMemcpyDeviceToHost, stream));
    PLX_HIP_TRY(hipMemcpyAsync(hk.data(), d_hk, (size_t)m * 8, hipMemcpyDeviceToHost, stream));
    PLX_HIP_TRY(hipStreamSynchronize(stream));
    if (ce[0] != 0u || ce[(size_t)m - 1] == 0xFFFFFFFFu) {
        set_error("reference_growth: a vertex without a lookup (first = %u .. %u)", ce[0], ce[(size_t)m - 1]);
        return PLX_ERR_STATE;
    }

    // small device reads the event loop needs now and then (each synchronises: a handful per build)
    int rc_dev = PLX_OK;
    auto vat_at = [&](int64_t e) -> int32_t {
        int32_t v = -1;
        if (hipMemcpyAsync(&v, L->replay_vat.as<int>() + e, 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) rc_dev = PLX_ERR_HIP;
        return v;
    };
    auto lookups_of = [&](int32_t v, std::vector<int32_t> &out) {
        size_t cap = L->replay_list.cap / 4;
        if (cap < (1u << 16)) { if (ensure(L->replay_list, (size_t)4 << 16) != PLX_OK) { rc_dev = PLX_ERR_HIP; return; } cap = L->replay_list.cap / 4; }
        for (;;) {
            int count = 0;
            if (hipMemsetAsync(d_count, 0, 4, stream) != hipSuccess) { rc_dev = PLX_ERR_HIP; return; }
            replay_match_kernel<<<ceil_div(E, kBlock), kBlock, 0, stream>>>(L->replay_vat.as<int>(), E, v, L->replay_list.as<int>(), (int)cap, d_count);
            if (hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { rc_dev = PLX_ERR_HIP; return; }
            if ((size_t)count <= cap) {
                out.resize((size_t)count);
                if (count > 0 && (hipMemcpyAsync(out.data(), L->replay_list.p, (size_t)count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                                  hipStreamSynchronize(stream) != hipSuccess)) { rc_dev = PLX_ERR_HIP; return; }
                std::sort(out.begin(), out.end());
                return;
            }
            if (ensure(L->replay_list, (size_t)count * 4 + 64) != PLX_OK) { rc_dev = PLX_ERR_HIP; return; }
            cap = L->replay_list.cap / 4;
        }
    };
    auto key_of = [&](int32_t v, int16_t *key) {
        uint32_t w[(PLX_MAX_DIM + 1) / 2];
        if (hipMemcpyAsync(w, L->vkeys.as<uint32_t>() + (size_t)v * dw, (size_t)dw * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { rc_dev = PLX_ERR_HIP; return; }
        for (int c = 0; c < d; ++c) key[c] = (int16_t)((w[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu);
    };

    // ---- splat(): creations in caller order + every lookup of the keys a stale probe has touched
    EventTable T(m, hk.data());
    std::vector<uint8_t> affected((size_t)m, 0);
    std::vector<int32_t> aff_list;                                              // the affected vertices, in the order they became so
    std::unordered_map<int32_t, std::vector<std::pair<int32_t, int32_t>>> hist;   // affected vertex -> (lookup, label it returned)
    using Ev = std::pair<int32_t, int32_t>;                                     // (lookup index, vertex)
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> pq;
    std::vector<int32_t> tmp;
    auto make_affected = [&](int32_t v, int64_t now) {      // from lookup `now` on, every lookup of v is run
        if (affected[v]) return;
        affected[v] = 1;
        aff_list.push_back(v);
        lookups_of(v, tmp);
        auto &h = hist[v];
        for (int32_t e : tmp) {
            if (e < now) h.emplace_back(e, v);              // earlier lookups found its one regular entry (label v)
            else pq.emplace(e, v);
        }
    };
    auto after_create = [&](int64_t e) {                    // the table is at its growth threshold: lookup e + 1 probes stale
        if (T.due() && e + 1 < E) make_affected(vat_at(e + 1), e + 1);
    };
    int64_t i = 0;
    constexpr int64_t kAhead = 32;
    for (;;) {
        while (i < m && affected[cv[i]]) ++i;               // an affected vertex's creation comes off the queue
        const int64_t tc = i < m ? (int64_t)ce[i] : INT64_MAX, th = pq.empty() ? INT64_MAX : (int64_t)pq.top().first;
        if (tc == INT64_MAX && th == INT64_MAX) break;
        bool created = false;
        if (tc < th) {
            if (i + kAhead < m) __builtin_prefetch(&T.tab[hk[cv[i + kAhead]] & (T.cap - 1)]);
            if (i + 2 * kAhead < m) __builtin_prefetch(&hk[cv[i + 2 * kAhead]]);
            const int32_t v = (int32_t)cv[i++];
            T.lookup(v, true, &created);
            if (created) after_create(tc);
        } else {
            const Ev ev = pq.top();
            pq.pop();
            const int32_t label = T.lookup(ev.second, true, &created);
            hist[ev.second].emplace_back(ev.first, label);
            if (created) after_create(ev.first);
        }
        if (rc_dev != PLX_OK) { set_error("reference_growth: a device read of the replay failed"); return rc_dev; }
    }
    auto &rp = L->replay;
    rp.active = true;
    rp.m_reference = T.filled;
    rp.grows = T.grows;

    // ---- blur(): its first lookup is entry 0's neighbour on axis 0 at tap nid = -order (h:526-545); if the table doubles
    // there, that lookup is the stale one (as in replay_full; keys of the few entries on the probe path come from the device)
    const int32_t v0 = (int32_t)cv[0];
    const int grows_before = T.grows;
    if (order >= 1) {
        int16_t key[PLX_MAX_DIM + 1], nk[PLX_MAX_DIM + 1], uk[PLX_MAX_DIM + 1];
        key_of(v0, key);
        const int nid = -order;
        bool in_range = true;
        for (int c = 0; c < d; ++c) {
            const int val = (int)key[c] - nid + (c == 0 ? nid * d1 : 0);
            in_range = in_range && val >= -32768 && val <= 32767;
            nk[c] = (int16_t)val;
        }
        const uint64_t hn = ref_hash(nk, d);
        auto find_by_key = [&](bool stale_first) -> int32_t {
            uint64_t h = hn % T.cap;
            if (stale_first && T.due()) T.grow();
            for (;;) {
                const int32_t l = T.tab[h];
                if (l == -1) return -1;
                key_of(T.vertex_of(l), uk);
                bool same = true;
                for (int c = 0; c < d && same; ++c) same = uk[c] == nk[c];
                if (same) return l;
                if (++h == T.cap) h = 0;
            }
        };
        const int32_t first = find_by_key(true);
        if (T.grows != grows_before && first < 0) {
            const int32_t again = find_by_key(false);
            if (again >= 0 && in_range) { rp.blur_miss = true; rp.blur_miss_vertex = v0; }
        }
    } else if (T.due()) {
        T.grow();
        rp.inexact = true;
    }
    if (rc_dev != PLX_OK) { set_error("reference_growth: a device read of the replay failed"); return rc_dev; }

    // ---- what blur-time lookups resolve the affected keys to (every other key: its one entry, always found)
    for (int32_t v : aff_list) {
        const int32_t F = T.lookup(v, false, nullptr);
        if (F < 0) invisible.push_back((int)v);
        for (const auto &el : hist[v])
            if (el.second != F) dropped.push_back((int)el.first);
    }
    std::sort(invisible.begin(), invisible.end());
    std::sort(dropped.begin(), dropped.end());
    if (rp.blur_miss && T.filled != m) {
        int copies = 1;
        for (int32_t u : T.extra) copies += (u == v0);
        if (copies > 1) rp.inexact = true;
    }
    rp.n_dropped = (int)dropped.size();
    rp.n_invisible = (int)invisible.size();
    return PLX_OK;
}

// reference_growth = 2: every one of the N (d+1) lookups of splat() run against the model, in the caller's order.  About a
// second at N = 1e6 (one dependent cache miss per lookup); kept as the checker of the event form below.
static int replay_full(plx_lattice *L, hipStream_t stream, std::vector<int> &dropped, std::vector<int> &invisible)
{
    const int n = (int)L->n, d = L->d, d1 = d + 1, dw = (d + 1) / 2, order = L->order;
    const int64_t m = L->m, E = (int64_t)n * d1;
    if (m <= 0 || E <= 0) return PLX_OK;
    if (E >= (1ll << 31)) { set_error("reference_growth: %lld lookups exceed the replay's 31-bit index", (long long)E); return PLX_ERR_TOO_LARGE; }
    PLX_TRY(ensure(L->replay_vat, (size_t)E * 4));
    replay_vat_kernel<<<ceil_div(n, kBlock), kBlock, 0, stream>>>(L->evid.as<int>(), L->perm.as<uint32_t>(), n, d1, L->replay_vat.as<int>());
    std::vector<int32_t> vat((size_t)E);
    std::vector<uint32_t> kw((size_t)m * dw);
    PLX_HIP_TRY(hipMemcpyAsync(vat.data(), L->replay_vat.p, (size_t)E * 4, hipMemcpyDeviceToHost, stream));
    PLX_HIP_TRY(hipMemcpyAsync(kw.data(), L->vkeys.p, (size_t)m * dw * 4, hipMemcpyDeviceToHost, stream));
    PLX_HIP_TRY(hipStreamSynchronize(stream));

    // reference hash of every vertex key
    std::vector<uint64_t> hk((size_t)m);
    std::vector<int16_t> key((size_t)d + 1);
    auto unpack = [&](int64_t v) {
        for (int c = 0; c < d; ++c) key[c] = (int16_t)((kw[(size_t)v * dw + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFFu);
    };
    for (int64_t v = 0; v < m; ++v) { unpack(v); hk[v] = ref_hash(key.data(), d); }

    // splat(): the N (d+1) lookups in the caller's order (h:395-485); R[e] = the entry lookup e returned
    RefTable T;
    T.hk = hk.data();
    std::vector<int32_t> R((size_t)E);
    for (int64_t e = 0; e < E; ++e) R[e] = T.lookup(hk[vat[e]], vat[e], true);
    auto &rp = L->replay;
    rp.active = true;
    rp.m_reference = (int64_t)T.ekey.size();
    rp.grows = T.grows;

    // blur(): its first lookup is entry 0's neighbour on axis 0 at tap nid = -order (h:526-545); if the table doubles
    // there, that lookup is the stale one
    const int v0 = T.ekey[0];
    const int grows_before = T.grows;
    if (order >= 1) {
        unpack(v0);
        const int nid = -order;
        std::vector<int16_t> nk((size_t)d);
        bool in_range = true;
        for (int c = 0; c < d; ++c) {
            const int val = (int)key[c] - nid + (c == 0 ? nid * d1 : 0);       // neighbor[k] = key[k] - nid; neighbor[0] = key[0] + nid * d
            in_range = in_range && val >= -32768 && val <= 32767;
            nk[c] = (int16_t)val;                                                // (the reference wraps: h:541 stores into a short)
        }
        const uint64_t hn = ref_hash(nk.data(), d);
        // which vertex, if any, has that key: probe the model for a matching KEY (entries hold vertex ids, so compare keys)
        auto find_by_key = [&](bool stale_first) -> int32_t {
            uint64_t h = hn % T.cap;
            if (stale_first && T.ekey.size() >= T.cap / 2 - 1) T.grow();        // h is now stale
            for (;;) {
                const int32_t e = T.entries[h];
                if (e == -1) return -1;
                bool same = true;
                const int32_t u = T.ekey[e];
                for (int c = 0; c < d && same; ++c)
                    same = (int16_t)((kw[(size_t)u * dw + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFFu) == nk[c];
                if (same) return e;
                if (++h == T.cap) h = 0;
            }
        };
        const int32_t first = find_by_key(true);
        if (T.grows != grows_before && first < 0) {
            const int32_t again = find_by_key(false);                            // a regular lookup of the same key
            if (again >= 0 && in_range) { rp.blur_miss = true; rp.blur_miss_vertex = v0; }
        }
    } else if (T.ekey.size() >= T.cap / 2 - 1) {
        // order 0: blur()'s first lookup is vertex 0's own key; a doubling there is not representable here
        T.grow();
        rp.inexact = true;
    }

    // what blur-time lookups resolve every key to (h:545 with create = false), and from that the dropped lookups
    std::vector<int32_t> F((size_t)m);
    for (int64_t v = 0; v < m; ++v) {
        F[v] = T.lookup(hk[v], (int32_t)v, false);
        if (F[v] < 0) invisible.push_back((int)v);
    }
    for (int64_t e = 0; e < E; ++e)
        if (R[e] != F[vat[e]]) dropped.push_back((int)e);
    if (rp.blur_miss && (int64_t)T.ekey.size() != m) {
        // vertex 0's key with several entries AND the blur-time miss: entry 0 and its twin would blur differently
        int copies = 0;
        for (int32_t u : T.ekey) copies += (u == v0);
        if (copies > 1) rp.inexact = true;
    }
    rp.n_dropped = (int)dropped.size();
    rp.n_invisible = (int)invisible.size();
    return PLX_OK;
}

// device side of both forms: the splat's own copy of the weights with the dropped lookups zeroed, the invisible list
static int replay_finish(plx_lattice *L, hipStream_t stream, const std::vector<int> &dropped, const std::vector<int> &invisible)
{
    const int n = (int)L->n, d1 = L->d + 1;
    const int64_t E = (int64_t)n * d1;
    PLX_TRY(ensure(L->ew_splat, (size_t)E * 4));
    PLX_HIP_TRY(hipMemcpyAsync(L->ew_splat.p, L->ew.p, (size_t)E * 4, hipMemcpyDeviceToDevice, stream));
    if (!dropped.empty()) {
        PLX_TRY(ensure_inv_perm(L, stream));
        PLX_TRY(ensure(L->replay_list, dropped.size() * 4));
        PLX_HIP_TRY(hipMemcpyAsync(L->replay_list.p, dropped.data(), dropped.size() * 4, hipMemcpyHostToDevice, stream));
        replay_drop_kernel<<<ceil_div((int64_t)dropped.size(), kBlock), kBlock, 0, stream>>>(
            L->replay_list.as<int>(), (int)dropped.size(), L->inv_perm.as<uint32_t>(), n, d1, L->ew_splat.as<float>());
        PLX_HIP_TRY(hipStreamSynchronize(stream));                                  // (`dropped` is pageable host memory)
    }
    PLX_TRY(ensure(L->replay_invisible, (invisible.size() + 1) * 4));
    if (!invisible.empty()) {
        PLX_HIP_TRY(hipMemcpyAsync(L->replay_invisible.p, invisible.data(), invisible.size() * 4, hipMemcpyHostToDevice, stream));
        PLX_HIP_TRY(hipStreamSynchronize(stream));
    }
    L->flags_valid = false;      // (the first-touch splat reads the slice's weights: not in this mode)
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// Runs between the structure build (vertex ids of every corner, vertex keys) and the gather tables.  Leaves the dropped
// lookups / invisible vertices / missed neighbour in the lattice (host side) and ew_splat on the device.
int replay_simulate(plx_lattice *L, hipStream_t stream)
{
    L->replay = plx_lattice::Replay();
    if (L->n_shards != 1 || L->for_merge || L->partial_cover) return PLX_OK;      // (plain single-process builds only)
    if (L->m <= 0 || L->n <= 0) return PLX_OK;
    std::vector<int> dropped, invisible;
    if (g_reference_growth == 2) PLX_TRY(replay_full(L, stream, dropped, invisible));
    else PLX_TRY(replay_events(L, stream, dropped, invisible));
    return replay_finish(L, stream, dropped, invisible);
}

// after the neighbour table is built, before anything is derived from it (axis pairs, compacted copy)
int replay_patch_tables(plx_lattice *L, hipStream_t stream)
{
    const auto &rp = L->replay;
    if (!rp.active || L->order < 1) return PLX_OK;
    const int d1 = L->d + 1, order = L->order;
    if (rp.n_invisible > 0) {
        const int64_t work = (int64_t)rp.n_invisible * d1 * 2 * order;
        replay_hide_kernel<<<ceil_div(work, kBlock), kBlock, 0, stream>>>(L->replay_invisible.as<int>(), rp.n_invisible, d1, order, L->mstride,
                                                                         L->nbr.as<int>());
    }
    if (rp.blur_miss)            // axis 0, tap nid = -order is plane s = 0
        replay_set_kernel<<<1, 1, 0, stream>>>(L->nbr.as<int>(), (size_t)rp.blur_miss_vertex, -1);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

// after every single-axis blur pass (blur_impl runs one axis per launch in this mode)
int replay_nocentre_fix(plx_lattice *L, const float *d_old, float *d_new, int vdp, hipStream_t stream)
{
    const auto &rp = L->replay;
    if (!rp.active || rp.n_invisible == 0) return PLX_OK;
    const int64_t work = (int64_t)rp.n_invisible * vdp;
    replay_nocentre_kernel<<<ceil_div(work, kBlock), kBlock, 0, stream>>>(d_old, d_new, L->replay_invisible.as<int>(), rp.n_invisible, vdp,
                                                                        L->taps.c[L->order]);
    PLX_HIP_TRY(hipGetLastError());
    return PLX_OK;
}

}  // namespace plx

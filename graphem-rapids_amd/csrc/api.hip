// C ABI of libgraphem_hip.so (include/graphem_hip.h): handle lifetime, host<->device
// copies, the iteration driver and the per-phase entry points.
#include "common.h"
#include "engine.h"

#include "torch_randperm.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <stdlib.h>
#include <string.h>

static thread_local std::string g_create_error;
void gh_set_create_error(const std::string &msg) { g_create_error = msg; }

// ---- timing ------------------------------------------------------------------------
gh_scope::gh_scope(gh_engine *h_, const char *name, hipStream_t on) : h(h_), stream(on ? on : h_->stream) {
    if (!h->timing) return;
    for (size_t i = 0; i < h->timers.size(); ++i)
        if (h->timers[i].name == name) slot = (int)i;
    if (slot < 0) {
        h->timers.emplace_back();
        h->timers.back().name = name;
        slot = (int)h->timers.size() - 1;
    }
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, stream);
}
gh_scope::~gh_scope() {
    if (slot < 0) return;
    (void)hipEventRecord(b, stream);
    h->timers[slot].pending.emplace_back(a, b);
}

static void resolve_timers(gh_engine *h) {
    for (auto &t : h->timers) {
        for (auto &p : t.pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(p.second);
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { t.total_ms += ms; t.launches += 1; }
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        t.pending.clear();
    }
}

// ---- helpers -----------------------------------------------------------------------
template <typename T>
static gh_status dev_alloc(gh_engine *h, T **p, size_t count, bool zero) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
    if (e != hipSuccess) {
        h->err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return GH_ERR_NOMEM;
    }
    if (zero) GH_HIP(hipMemsetAsync(*p, 0, count * sizeof(T), h->stream));
    return GH_OK;
}

#define GH_TRY(x)                        \
    do {                                 \
        gh_status st_ = (x);             \
        if (st_ != GH_OK) return st_;    \
    } while (0)

static gh_status check_handle(gh_engine *h) {
    if (!h) return GH_ERR_INVALID;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) {
        h->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return GH_ERR_HIP;
    }
    return GH_OK;
}

// Entry points of the float32 engine's internals have no meaning on a float64 engine (f64.hip).
static gh_status reject_f64(gh_engine *h, const char *what) {
    if (!h->f64) return GH_OK;
    h->err = std::string(what) + " is not available on a float64 engine";
    return GH_ERR_INVALID;
}

static void free_all(gh_engine *h) {
    gh_f64_free(h);
    gh_ivf_free(h);
    void *ptrs[] = {h->d_edges, h->d_rowptr, h->d_adj, h->d_pos, (h->d_gbuf || h->d_rows_all) ? (void *)h->d_new_own : (void *)h->d_new, h->d_gbuf, h->d_rows_all, h->d_rows_pk, h->d_stats_all, h->d_tmpF, h->d_tmpF2, h->d_io, h->d_acc,
                    h->d_tflag, h->d_touched, h->d_tcount, h->d_sampled, h->d_q, h->d_qscan, h->d_qA, h->d_qexact, h->d_order, h->d_long_rows, h->d_long_ownptr, h->d_long_ownadj, h->d_long_eptr, h->d_long_erow, h->d_long_terms, h->d_own_long, h->d_cand, h->d_cnt,
                    h->d_ovf, h->d_sel_redo, h->d_tq_count, h->d_tq_base, h->d_tq_touched, h->d_dbg_cnt, h->d_partial, h->d_merged, h->d_first_edge, h->d_own_eids, h->d_mid, h->d_Fs, h->d_gmin, h->d_sub_uv, h->d_stamps, h->d_tau_flag, h->d_wait_failed, h->d_grid_u32, h->d_grid_smid, h->d_grid_temp, h->d_iter, h->d_stats_comb, h->d_rows_packed, h->d_rare, h->d_cd_rows, h->d_cd_vbuf, h->d_cd_cmin, h->d_cd_stat, h->d_vblock, h->d_blockstats, (h->d_gbuf || h->d_rows_all) ? (void *)h->d_stats_own : (void *)h->d_stats, h->d_iscratch, h->d_stream_ids};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    for (hipEvent_t e : h->ring_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

// ---- lifetime ----------------------------------------------------------------------
extern "C" gh_status gh_create(gh_handle *out, int device_id, int64_t n, int32_t D, int64_t E, const int32_t *edges,
                               const gh_params *params, const gh_partition *part) {
    if (!out) return GH_ERR_INVALID;
    *out = nullptr;
    auto fail = [&](gh_status st, const std::string &msg) { g_create_error = msg; return st; };
    if (n <= 0) return fail(GH_ERR_INVALID, "Adjacency matrix cannot be empty");
    if (D <= 0) return fail(GH_ERR_INVALID, "Number of components must be positive, got " + std::to_string(D));
    if (!params) return fail(GH_ERR_INVALID, "params is NULL");
    if (params->k_attr < 0) return fail(GH_ERR_INVALID, "Attractive force constant k_attr must be non-negative");
    if (E < 0 || (E > 0 && !edges)) return fail(GH_ERR_INVALID, "edges is NULL");
    if (params->n_neighbors < 0 || params->sample_size < 0) return fail(GH_ERR_INVALID, "negative n_neighbors / sample_size");
    if (E >= ((int64_t)1 << 30) || n >= ((int64_t)1 << 31)) return fail(GH_ERR_INVALID, "graph too large for int32 ids");
    if ((int64_t)params->n_neighbors + 1 > GH_SEL_BUF - GH_SEL_CHUNK)
        return fail(GH_ERR_INVALID, "n_neighbors too large for the HIP backend (max 2047)");
    for (int64_t e = 0; e < E; ++e) {
        const int32_t u = edges[2 * e], v = edges[2 * e + 1];
        if (u < 0 || v < 0 || u >= n || v >= n) return fail(GH_ERR_INVALID, "edge endpoint out of range");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GH_ERR_HIP, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(GH_ERR_RUNTIME, "invalid device ordinal " + std::to_string(device_id));

    gh_engine *h = new (std::nothrow) gh_engine();
    if (!h) return fail(GH_ERR_NOMEM, "out of host memory");
    h->device = device_id;
    h->n = n; h->E = E; h->D = D; h->LD = gh_ld(D);
    h->prm = *params;
    h->k = params->n_neighbors; h->K = h->k + 1;
    h->S = std::min<int64_t>(params->sample_size, E);
    const bool auto_method = h->prm.knn_method != GH_KNN_SCAN && h->prm.knn_method != GH_KNN_GRID && h->prm.knn_method != GH_KNN_IVF;
    if (h->prm.knn_method != GH_KNN_SCAN && h->prm.knn_method != GH_KNN_GRID && h->prm.knn_method != GH_KNN_IVF) {   // AUTO (and anything unknown)
        // exact methods only.  Whole graph, up to 8 components, thousands of queries: the inverted file in its exact mode
        // (rr1m, scan / exact IVF us per iteration: D = 3 S = 4096 1063 / 591, 16384 3691 / 766 (grid 1926); D = 6 S = 16384
        // 5059 / 1401; D = 8 5264 / 2135; 100 K vertices D = 3 S = 4096 384 / 191; profiles/r03/knn_method_sweep.log); else the
        // grid for <= 3 components from 12288 queries on; else the scan
        const bool ivf = !part && params->knn_distance == GH_DIST_EXACT && D >= 2 && D <= 8 && h->S >= (D <= 4 ? 4096 : 8192) && E >= 262144;
        h->prm.knn_method = ivf ? GH_KNN_IVF : (D <= 3 && h->S >= 12288) ? GH_KNN_GRID : GH_KNN_SCAN;
        if (ivf) { h->prm.ivf_probes = -1; h->prm.ivf_lists = 0; }
    }
    if (params->knn_distance != GH_DIST_EXACT && params->knn_distance != GH_DIST_CDIST) { delete h; return fail(GH_ERR_INVALID, "unknown knn_distance"); }
    h->cdist = params->knn_distance == GH_DIST_CDIST;
    h->cd_part = h->cdist && part != nullptr;
    if (h->cdist && !auto_method && params->knn_method != GH_KNN_SCAN) {   // (an explicit request must not be dropped silently)
        delete h;
        return fail(GH_ERR_INVALID, "knn_distance = GH_DIST_CDIST re-values the candidates of GH_KNN_SCAN: it cannot be combined with GH_KNN_GRID / GH_KNN_IVF");
    }
    if (h->cdist) h->prm.knn_method = GH_KNN_SCAN;   // AUTO: the other searches know exact distances only
    h->Ksel = h->K + (h->cdist ? 1 : 0);
    if (part) h->part = *part;
    else h->part = gh_partition{0, n, 0, E, GH_EDGES_RANGE};
    if (h->part.edge_rule == GH_EDGES_HASHED) h->part.edge_lo = h->part.edge_hi = 0;  // not used by this rule
    if (h->part.row_lo < 0 || h->part.row_hi > n || h->part.row_lo > h->part.row_hi || h->part.edge_lo < 0 ||
        h->part.edge_hi > E || h->part.edge_lo > h->part.edge_hi ||
        (h->part.edge_rule != GH_EDGES_RANGE && h->part.edge_rule != GH_EDGES_HASHED)) {
        delete h;
        return fail(GH_ERR_INVALID, "partition out of range");
    }
    h->rows = h->part.row_hi - h->part.row_lo;

    auto bail = [&](gh_status st) { g_create_error = h->err; free_all(h); delete h; return st; };
    if (hipSetDevice(device_id) != hipSuccess) return bail(GH_ERR_HIP);
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return bail(GH_ERR_HIP); }
    h->stream = h->own_stream;
    h->pos_rows = n + GH_POS_PAD_ROWS;
    // Diagnostic switches, read once here (include/graphem_hip.h lists them): GRAPHEM_HIP_NO_PRESETUP keeps the next
    // iteration's KNN set-up out of the normalise launch (so that per-query flags survive a step for inspection),
    // GRAPHEM_HIP_GRAPH=1 replays iterations from a hipGraph.
    h->opt_no_presetup = getenv("GRAPHEM_HIP_NO_PRESETUP") != nullptr;
    { const char *e = getenv("GRAPHEM_HIP_GRAPH"); h->opt_graph = e && atoi(e) != 0; }

    // Internal vertex order (include/graphem_hip.h GH_REORDER_*): breadth-first numbers, components in
    // order of their smallest vertex, children in pull-list (= edge id) order.
    {
        const bool l2_miss = (double)n * h->LD * sizeof(float) > 3.0 * 1024 * 1024;
        const bool hashed_ok = !part || part->edge_rule == GH_EDGES_HASHED;
        int mode = params->reorder;
        if (const char *e = getenv("GRAPHEM_HIP_REORDER")) mode = atoi(e);  // tests / A-B runs: 1 off, 2 BFS
        const bool reorder = hashed_ok && E > 0 && (mode == GH_REORDER_BFS || (mode == GH_REORDER_AUTO && l2_miss));
        if (reorder) {
            std::vector<int64_t> off((size_t)n + 1, 0);
            for (int64_t e = 0; e < E; ++e) { off[(size_t)edges[2 * e] + 1]++; off[(size_t)edges[2 * e + 1] + 1]++; }
            for (int64_t i = 0; i < n; ++i) off[(size_t)i + 1] += off[(size_t)i];
            std::vector<int32_t> nb((size_t)off[(size_t)n]);
            {
                std::vector<int64_t> cur(off.begin(), off.end() - 1);
                for (int64_t e = 0; e < E; ++e) {
                    const int32_t u = edges[2 * e], v = edges[2 * e + 1];
                    nb[(size_t)cur[(size_t)u]++] = v;
                    nb[(size_t)cur[(size_t)v]++] = u;
                }
            }
            h->order_host.assign((size_t)n, -1);
            std::vector<int32_t> queue((size_t)n);
            int64_t head = 0, tail = 0, next = 0;
            for (int64_t root = 0; root < n; ++root) {
                if (h->order_host[(size_t)root] >= 0) continue;
                h->order_host[(size_t)root] = (int32_t)next++;
                queue[(size_t)tail++] = (int32_t)root;
                while (head < tail) {
                    const int32_t x = queue[(size_t)head++];
                    for (int64_t j = off[(size_t)x]; j < off[(size_t)x + 1]; ++j) {
                        const int32_t y = nb[(size_t)j];
                        if (h->order_host[(size_t)y] < 0) { h->order_host[(size_t)y] = (int32_t)next++; queue[(size_t)tail++] = y; }
                    }
                }
            }
            // Within blocks of 16384 consecutive breadth-first numbers, rows in order of falling degree: the lanes of a wave
            // walk their pull lists in lock-step, so a wave costs its LONGEST list.  G(n, p) at 1 M vertices (Poisson
            // degrees, mean 10): fused kernel 172.5 -> 163 us with blocks of 8 K - 32 K rows, 168 with 256, 170 - 172 with
            // 256 K or the whole graph (the breadth-first locality is gone); a regular graph is left as it is (stable sort).
            {
                const int64_t B = 16384;
                {
                    std::vector<int32_t> inv((size_t)n);
                    for (int64_t v = 0; v < n; ++v) inv[(size_t)h->order_host[(size_t)v]] = (int32_t)v;
                    for (int64_t b0 = 0; b0 < n; b0 += B) {
                        const int64_t b1 = std::min(n, b0 + B);
                        std::stable_sort(inv.begin() + b0, inv.begin() + b1, [&](int32_t a, int32_t c) {
                            return off[(size_t)a + 1] - off[(size_t)a] > off[(size_t)c + 1] - off[(size_t)c];
                        });
                    }
                    for (int64_t i = 0; i < n; ++i) h->order_host[(size_t)inv[(size_t)i]] = (int32_t)i;
                }
            }
            h->edges_internal.resize((size_t)E * 2);
            for (int64_t i = 0; i < 2 * E; ++i) h->edges_internal[(size_t)i] = h->order_host[(size_t)edges[i]];
            edges = h->edges_internal.data();  // everything below works on internal vertex numbers
            if (!part) h->part = gh_partition{0, n, 0, 0, GH_EDGES_HASHED};
        }
    }

    // Hubs (common.h GH_LONG_DEG): degrees over the WHOLE graph, so that every rank sees the same set.
    // A graph that has any takes the flagged ownership rule (it lets the short endpoint of a hub's
    // edge own it, so that no row owns more than a workgroup's tile).
    std::vector<int32_t> deg((size_t)n, 0);
    for (int64_t i = 0; i < 2 * E; ++i) deg[(size_t)edges[i]]++;
    bool has_long = false;
    h->long_deg = gh_long_degree(n, E);
    const int long_deg = h->long_deg;
    for (int64_t i = 0; i < n && !has_long; ++i) has_long = deg[(size_t)i] > long_deg;
    if (has_long && part && part->edge_rule != GH_EDGES_HASHED) has_long = false;  // range partitions: as before
    // A whole-graph engine always takes the hashed rule: under "endpoint 0 owns" vertex i of a u < v edge list owns its
    // edges to higher-numbered neighbours only -- 8 for the first vertices of an 8-regular graph, 0 for the last -- so the
    // fused workgroups at the end of the vertex range held 1024 rows for a few hundred owned edges and took 31 us where the
    // median workgroup took 18 (tools/stamp_probe.py, 100 K vertices): they were the length of the kernel.
    if (!part) h->part = gh_partition{0, n, 0, 0, GH_EDGES_HASHED};

    // Pull lists of the own rows in the reference's summation order (pt.py:633-634):
    // first the edges where the vertex is endpoint 0, then those where it is endpoint 1,
    // each in edge-list order.  Bit 31 of an entry marks the edges this row OWNS (emits the
    // midpoint of, and searches in the KNN phase).  Ownership rule GH_EDGES_RANGE: the edges
    // [edge_lo, edge_hi), each owned by its endpoint 0.  GH_EDGES_HASHED (partitioned engines):
    // a hash of the edge id picks the owning endpoint, so every rank owns ~E/world edges
    // whatever the vertex numbering (with endpoint-0 ownership the low-numbered ranks of a
    // u<v edge list hold most of the edges); an edge between a hub and a short row always belongs to
    // the short row.
    const bool hashed = h->part.edge_rule == GH_EDGES_HASHED;
    auto owner_is_v = [&](int64_t e) {
        if (has_long) {
            const int32_t du = deg[(size_t)edges[2 * e]], dv = deg[(size_t)edges[2 * e + 1]];
            const bool lu = du > long_deg, lv = dv > long_deg;
            if (lu != lv) return lu;
            // between two long rows the one with FEWER neighbours owns the edge (equal degrees: the hash): a hub then owns
            // edges to bigger hubs only, and no row owns more than a fused workgroup's tile (a 1045-degree hub of a graph whose
            // rows are all long owned 520 edges by the hash alone and forced the whole engine onto its unfused kernels)
            if (lu && du != dv) return du > dv;
        }
        uint32_t x = (uint32_t)e * 0x9E3779B1u;
        x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
        return (x >> 31) != 0;
    };
    std::vector<int32_t> rowptr((size_t)h->rows + 1, 0);
    const int64_t lo = h->part.row_lo, hi = h->part.row_hi;
    for (int64_t e = 0; e < E; ++e) {
        const int32_t u = edges[2 * e], v = edges[2 * e + 1];
        if (u >= lo && u < hi) rowptr[(size_t)(u - lo) + 1]++;
        if (v >= lo && v < hi) rowptr[(size_t)(v - lo) + 1]++;
    }
    for (int64_t i = 0; i < h->rows; ++i) rowptr[(size_t)i + 1] += rowptr[(size_t)i];
    h->adj_len = rowptr[(size_t)h->rows];
    std::vector<int32_t> adj((size_t)std::max<int64_t>(h->adj_len, 1));
    std::vector<int32_t> adj_eid(hashed ? (size_t)std::max<int64_t>(h->adj_len, 1) : 0);
    {
        std::vector<int32_t> cur(rowptr.begin(), rowptr.end() - 1);
        for (int64_t e = 0; e < E; ++e) {
            const int32_t u = edges[2 * e], v = edges[2 * e + 1];
            if (u >= lo && u < hi) {
                const size_t at = (size_t)cur[(size_t)(u - lo)]++;
                const bool own = hashed ? !owner_is_v(e) : (e >= h->part.edge_lo && e < h->part.edge_hi);
                adj[at] = (int32_t)((uint32_t)v | (own ? 0x80000000u : 0u));
                if (hashed) adj_eid[at] = (int32_t)e;
            }
        }
        for (int64_t e = 0; e < E; ++e) {
            const int32_t u = edges[2 * e], v = edges[2 * e + 1];
            if (v >= lo && v < hi) {
                const size_t at = (size_t)cur[(size_t)(v - lo)]++;
                const bool own = hashed && owner_is_v(e);
                adj[at] = (int32_t)((uint32_t)u | (own ? 0x80000000u : 0u));
                if (hashed) adj_eid[at] = (int32_t)e;
            }
        }
    }

    // d_first_edge[i]: where the midpoints of row i's owned edges go.  Range rule, edges sorted by
    // first endpoint (always true for the reference's CSR-order edge list): the owned edges of a
    // row are consecutive ids and the offset is the first of them.  Hashed rule: a prefix count
    // into the list own_eids of owned edge ids in (row, pull list) order.
    std::vector<int32_t> first_edge((size_t)h->rows + 1, 0);
    std::vector<int32_t> own_eids, long_rows, long_ownptr, long_ownadj, long_eptr, long_erow;
    std::vector<uint8_t> own_long;
    if (hashed) {
        own_eids.reserve((size_t)(E / std::max<int64_t>(1, n / std::max<int64_t>(h->rows, 1)) + 16));
        for (int64_t i = 0; i < h->rows; ++i) {
            first_edge[(size_t)i] = (int32_t)own_eids.size();
            for (int32_t j = rowptr[(size_t)i]; j < rowptr[(size_t)i + 1]; ++j)
                if ((uint32_t)adj[(size_t)j] >> 31) own_eids.push_back(adj_eid[(size_t)j]);
        }
        first_edge[(size_t)h->rows] = (int32_t)own_eids.size();
        h->own_count = (int64_t)own_eids.size();
        if (has_long) {  // the own hub rows and the (hub-hub) edges they own, for spring_long_kernel / spring_row
            long_ownptr.push_back(0);
            long_eptr.push_back(0);
            for (int64_t i = 0; i < h->rows; ++i) {
                if (rowptr[(size_t)i + 1] - rowptr[(size_t)i] <= long_deg) continue;
                long_rows.push_back((int32_t)i);
                long_eptr.push_back(long_eptr.back() + (rowptr[(size_t)i + 1] - rowptr[(size_t)i]));
                for (int32_t j = rowptr[(size_t)i]; j < rowptr[(size_t)i + 1]; ++j)
                    if ((uint32_t)adj[(size_t)j] >> 31) long_ownadj.push_back((int32_t)((uint32_t)adj[(size_t)j] & 0x7FFFFFFFu));
                long_ownptr.push_back((int32_t)long_ownadj.size());
            }
            h->nlong = (int)long_rows.size();
            h->long_entries = long_eptr.back();
            for (size_t r = 0; r + 1 < long_eptr.size(); ++r) h->long_max_deg = std::max(h->long_max_deg, (int)(long_eptr[r + 1] - long_eptr[r]));
            long_erow.resize((size_t)h->long_entries);   // list entry -> index of its long row (spares long_terms_kernel a binary search)
            for (size_t r = 0; r + 1 < long_eptr.size(); ++r)
                for (int32_t t = long_eptr[r]; t < long_eptr[r + 1]; ++t) long_erow[(size_t)t] = (int32_t)r;
            own_long.assign(own_eids.size() + 1, 0);   // owned-edge slots of the long rows
            for (int64_t i = 0; i < h->rows; ++i)
                if (rowptr[(size_t)i + 1] - rowptr[(size_t)i] > long_deg)
                    for (int32_t sl = first_edge[(size_t)i]; sl < (i + 1 < h->rows ? first_edge[(size_t)i + 1] : (int32_t)own_eids.size()); ++sl)
                        own_long[(size_t)sl] = 1;
        }
        h->mid_base = 0;
        h->fused_mid = true;
        std::vector<int32_t>().swap(adj_eid);
    } else {
        bool sorted = true;
        for (int64_t e = 1; e < E && sorted; ++e) sorted = edges[2 * e] >= edges[2 * (e - 1)];
        if (sorted) {
            int64_t e = 0;
            for (int64_t i = 0; i <= h->rows; ++i) {
                const int64_t x = lo + i;
                while (e < E && edges[2 * e] < x) ++e;
                first_edge[(size_t)i] = (int32_t)e;
            }
            h->fused_mid = first_edge[0] == h->part.edge_lo && first_edge[(size_t)h->rows] == h->part.edge_hi;
        }
        h->own_count = h->part.edge_hi - h->part.edge_lo;
        h->mid_base = h->part.edge_lo;
    }
    // AUTO on a partitioned engine: the same rule with the edges this rank owns (known only now)
    if (auto_method && part && !h->cdist && D >= 2 && D <= 8 && h->S >= (D <= 4 ? 4096 : 8192) && h->own_count >= 262144) {
        h->prm.knn_method = GH_KNN_IVF;
        h->prm.ivf_probes = -1;
        h->prm.ivf_lists = 0;
    }
    // Vertex ranges of the fused spring+scan workgroups: as many consecutive own rows as hold at
    // most TILE owned edges (and at most 1024 rows, 4 per thread).
    std::vector<int32_t> vblock;
    {
        const int tile = gh_fused_tile(h);
        const bool dim_ok = gh_dim_templated(D);
        bool ok = h->fused_mid && dim_ok;
        if (ok) {
            vblock.push_back(0);
            int64_t i = 0;
            while (i < h->rows && ok) {
                int64_t j = i, cnt = 0;
                while (j < h->rows && j - i < 1024) {
                    const int64_t own = first_edge[(size_t)j + 1] - first_edge[(size_t)j];
                    if (own > tile) { ok = false; break; }  // a single row owns more than a tile: unfused path
                    if (cnt + own > tile) break;
                    cnt += own;
                    ++j;
                }
                if (!ok) break;
                vblock.push_back((int32_t)j);
                i = j;
            }
        }
        h->fused_scan = ok;
        if (!ok) vblock.assign(1, 0);
        h->n_vblocks = (int)vblock.size() - 1;
    }

    const size_t nLD = (size_t)n * h->LD, S = (size_t)h->S;
    gh_status st;
#define GH_A(p, count, zero) if ((st = dev_alloc(h, &h->p, (count), (zero))) != GH_OK) return bail(st)
#define GH_A2(p, count) if ((st = dev_alloc(h, &h->p, (count), true)) != GH_OK) return bail(st)
    GH_A(d_edges, (size_t)E * 2, false);
    GH_A(d_rowptr, (size_t)h->rows + 1, false);
    GH_A(d_adj, (size_t)h->adj_len, false);
    GH_A(d_first_edge, (size_t)h->rows + 1, true);
    GH_A(d_mid, (size_t)h->own_count * h->LD, true);
    GH_A(d_Fs, (size_t)h->rows * h->LD, true);
    // candidate lists and the threshold subset exist only for the filtered scan (64 KiB per query)
    const bool scan_path = gh_knn_scan_path(h);
    GH_A(d_gmin, (size_t)gh_gmin_floats(h), true);  // also fixes the threshold subset (thr_stride, thr_M1)
    GH_A(d_sub_uv, (size_t)h->thr_M1 * 2, false);
    if (hashed) GH_A(d_own_eids, own_eids.size() + 1, true);
    if (h->nlong) {
        GH_A(d_long_rows, long_rows.size(), false);
        GH_A(d_long_ownptr, long_ownptr.size(), false);
        GH_A(d_long_ownadj, long_ownadj.size() + 1, true);
        GH_A(d_long_eptr, long_eptr.size(), false);
        GH_A(d_long_erow, long_erow.size() + 1, false);
        GH_A(d_long_terms, (size_t)h->long_entries * D, false);
        GH_A(d_own_long, own_long.size(), false);
    }
    GH_A(d_vblock, vblock.size(), false);
    GH_A(d_pos, (size_t)h->pos_rows * h->LD, true);
    GH_A(d_new, (size_t)h->rows * h->LD, true);
    GH_A(d_tmpF, nLD, true);
    GH_A(d_tmpF2, nLD, true);
    GH_A(d_io, (size_t)n * D, false);
    GH_A(d_acc, nLD, true);
    GH_A(d_tflag, (size_t)n, true);
    GH_A(d_touched, 4 * S * (size_t)h->k, false);
    GH_A(d_tcount, 1, true);
    GH_A(d_sampled, S, true);
    GH_A(d_q, S * (size_t)(h->LD + 4), true);
    GH_A(d_qscan, S * (size_t)(h->LD + 4), true);
    GH_A(d_qA, S * 16, true);
    GH_A(d_qexact, S + 1, true);
    if ((st = dev_alloc(h, &h->d_cand, scan_path ? S * GH_CAND_CAP : 1, false)) != GH_OK) {
        h->err = "hipMalloc of the KNN candidate lists failed: sample_size = " + std::to_string(h->S) + " needs " +
                 std::to_string((S * GH_CAND_CAP * sizeof(uint64_t)) >> 20) + " MiB (128 KiB per sampled midpoint)";
        return bail(st);
    }
    GH_A(d_cnt, S * GH_CNT_STRIDE, true);
    GH_A(d_ovf, S, true);
    GH_A(d_sel_redo, S, true);
    GH_A(d_tq_count, S >= 2048 ? S : 1, true);
    GH_A(d_tq_base, S >= 2048 ? S : 1, true);
    GH_A(d_tq_touched, S >= 2048 ? 4 * S * (size_t)std::max(h->k, 1) : 1, false);
    GH_A(d_dbg_cnt, 2 * S, true);
    GH_A(d_partial, S * (size_t)(h->K + (h->cd_part ? 2 : 0)), true);
    GH_A(d_merged, S * (size_t)h->K, true);
    GH_A(d_iscratch, S * (size_t)h->k * h->LD, false);
    h->nblocks_update = (int)((h->rows + 255) / 256);
    GH_A(d_blockstats, (size_t)std::max(std::max(h->nblocks_update, h->n_vblocks), 1) * 2 * h->LD, true);
    GH_A(d_stats, (size_t)(2 + 2 * gh_fix_blocks(h->LD)) * h->LD, true);
#undef GH_A
    h->d_sampled_cur = h->d_sampled;
    auto up = [&](void *dst, const void *src, size_t bytes) {
        return bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    };
    if (!up(h->d_edges, edges, sizeof(int32_t) * 2 * (size_t)E) ||
        !up(h->d_rowptr, rowptr.data(), sizeof(int32_t) * rowptr.size()) ||
        !up(h->d_adj, adj.data(), sizeof(int32_t) * (size_t)h->adj_len) ||
        !up(h->d_first_edge, first_edge.data(), sizeof(int32_t) * first_edge.size()) ||
        !up(h->d_vblock, vblock.data(), sizeof(int32_t) * vblock.size()) ||
        (hashed && !up(h->d_own_eids, own_eids.data(), sizeof(int32_t) * own_eids.size())) ||
        (h->nlong && (!up(h->d_long_rows, long_rows.data(), sizeof(int32_t) * long_rows.size()) ||
                      !up(h->d_long_ownptr, long_ownptr.data(), sizeof(int32_t) * long_ownptr.size()) ||
                      !up(h->d_long_eptr, long_eptr.data(), sizeof(int32_t) * long_eptr.size()) ||
                      !up(h->d_long_erow, long_erow.data(), sizeof(int32_t) * long_erow.size()) ||
                      !up(h->d_long_ownadj, long_ownadj.data(), sizeof(int32_t) * long_ownadj.size()) ||
                      !up(h->d_own_long, own_long.data(), own_long.size()))) ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
        h->err = "upload of the graph failed";
        return bail(GH_ERR_HIP);
    }
    if ((st = gh_grid_alloc(h)) != GH_OK) return bail(st);
    if ((st = gh_ivf_alloc(h)) != GH_OK) return bail(st);
    if ((st = gh_cdist_alloc(h)) != GH_OK) return bail(st);
    GH_A2(d_tau_flag, 1);
    GH_A2(d_iter, 1);
    GH_A2(d_wait_failed, 1);
    // Thresholds by the first workgroups of the fused launch (tau_core.h) where that launch is a single round of
    // workgroups or little more: there the iteration is a chain of launch latencies and this removes one (100 K vertices:
    // 64.9 -> 60.6 us).  A large graph gains nothing (1 M vertices: 175.7 -> 176.7 us, the first round of workgroups waits
    // ~3 us for producers that share their CUs with gathers) and keeps the launch of its own.
    // GRAPHEM_HIP_TAU_SEPARATE=1 / 0 forces either form.
    h->tau_embedded = h->n_vblocks <= 2048;
    if (const char *e = getenv("GRAPHEM_HIP_TAU_SEPARATE")) h->tau_embedded = atoi(e) == 0;
    if (getenv("GRAPHEM_HIP_STAMPS")) GH_A2(d_stamps, ((size_t)std::max(h->n_vblocks, 1) + GH_STAMP_EXTRA) * 8);
    if (h->thr_M1 > 0) {  // endpoints of the threshold subset: every thr_stride-th own edge
        std::vector<int32_t> sub((size_t)h->thr_M1 * 2);
        for (int64_t j = 0; j < h->thr_M1; ++j) {
            const int64_t e = hashed ? (int64_t)own_eids[(size_t)(j * h->thr_stride)] : h->part.edge_lo + j * h->thr_stride;
            sub[(size_t)(2 * j)] = edges[2 * e];
            sub[(size_t)(2 * j + 1)] = edges[2 * e + 1];
        }
        if (hipMemcpy(h->d_sub_uv, sub.data(), sizeof(int32_t) * sub.size(), hipMemcpyHostToDevice) != hipSuccess) {
            h->err = "upload of the threshold subset failed";
            return bail(GH_ERR_HIP);
        }
    }
    if (!h->order_host.empty()) {
        st = dev_alloc(h, &h->d_order, (size_t)n, false);
        if (st != GH_OK) return bail(st);
        if (hipMemcpy(h->d_order, h->order_host.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) {
            h->err = "upload of the vertex order failed";
            return bail(GH_ERR_HIP);
        }
    }
    std::vector<int32_t>().swap(h->edges_internal);
    *out = h;
    return GH_OK;
}

extern "C" void gh_destroy(gh_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    resolve_timers(h);
    gh_comm_free(h);
    free_all(h);
    delete h;
}

extern "C" const char *gh_last_error(gh_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// ---- positions ---------------------------------------------------------------------
extern "C" gh_status gh_set_positions(gh_handle h, const float *pos) {
    GH_TRY(check_handle(h));
    if (h->f64) return pos ? gh_f64_set_positions_f32(h, pos) : GH_ERR_INVALID;
    if (!pos) { h->err = "positions is NULL"; return GH_ERR_INVALID; }
    h->presetup_valid = false;
    GH_HIP(hipMemcpyAsync(h->d_io, pos, sizeof(float) * (size_t)h->n * h->D, hipMemcpyHostToDevice, h->stream));
    GH_TRY(gh_launch_pad(h, h->d_io, h->d_pos));
    GH_HIP(hipStreamSynchronize(h->stream));  // the host buffer may be released by the caller
    return GH_OK;
}

static gh_status download_padded(gh_engine *h, const float *d_src, float *host) {
    GH_TRY(gh_launch_unpad(h, d_src, h->d_io));
    GH_HIP(hipMemcpyAsync(host, h->d_io, sizeof(float) * (size_t)h->n * h->D, hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}

// A workgroup of a fused launch gave up waiting for that launch's thresholds (tau_core.h): whatever was computed since
// is not to be trusted.  Cannot happen while workgroups are started in index order; checked where the host synchronises.
static gh_status check_device_waits(gh_engine *h) {
    if (!h->tau_embedded || !h->d_wait_failed) return GH_OK;
    int32_t failed = 0;
    GH_HIP(hipMemcpyAsync(&failed, h->d_wait_failed, sizeof(failed), hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    if (failed) {
        // reported once: the flag is cleared and the engine goes on with the thresholds as a launch of their own (no
        // workgroup waits for another any more)
        GH_HIP(hipMemsetAsync(h->d_wait_failed, 0, sizeof(int32_t), h->stream));
        GH_HIP(hipStreamSynchronize(h->stream));
        h->tau_embedded = false;
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }   // captured with the other form
        if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
        h->err = "a workgroup of the fused spring+scan launch timed out waiting for the thresholds of its own launch; "
                 "results since the last successful gh_sync / gh_get_positions are invalid -- set the positions again. "
                 "The engine now computes the thresholds in a launch of their own (as GRAPHEM_HIP_TAU_SEPARATE=1 does); "
                 "please report this";
        return GH_ERR_RUNTIME;
    }
    return GH_OK;
}

extern "C" gh_status gh_get_positions(gh_handle h, float *pos) {
    GH_TRY(check_handle(h));
    if (h->f64) return pos ? gh_f64_get_positions_f32(h, pos) : GH_ERR_INVALID;
    if (!pos) { h->err = "positions is NULL"; return GH_ERR_INVALID; }
    GH_TRY(check_device_waits(h));
    return download_padded(h, h->d_pos, pos);
}

extern "C" float *gh_positions_device(gh_handle h) { return h && !h->f64 ? h->d_pos : nullptr; }
extern "C" gh_status gh_vertex_order(gh_handle h, int32_t *order) {
    GH_TRY(check_handle(h));
    if (!order) { h->err = "order is NULL"; return GH_ERR_INVALID; }
    for (int64_t i = 0; i < h->n; ++i) order[i] = h->order_host.empty() ? (int32_t)i : h->order_host[(size_t)i];
    return GH_OK;
}
extern "C" const float *gh_positions_unpadded_device(gh_handle h) {
    if (!h || h->f64 || hipSetDevice(h->device) != hipSuccess) return nullptr;
    if (gh_launch_unpad(h, h->d_pos, h->d_io) != GH_OK) return nullptr;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return nullptr;
    return h->d_io;
}
extern "C" int32_t gh_row_stride(gh_handle h) { return h ? h->LD : 0; }

// ---- the loop ----------------------------------------------------------------------
static bool whole_graph(gh_engine *h) {
    return h->part.row_lo == 0 && h->part.row_hi == h->n &&
           (h->part.edge_rule == GH_EDGES_HASHED || (h->part.edge_lo == 0 && h->part.edge_hi == h->E));
}
// gh_step / gh_run / the per-phase entry points merge with world = 1 and normalise with the own rows'
// statistics: on a row partition that would silently corrupt the positions.
static gh_status check_whole(gh_engine *h, const char *what) {
    if (whole_graph(h) && !h->d_gbuf && !h->g_world) return GH_OK;
    h->err = std::string(what) + " needs the whole graph on one rank; a partitioned engine runs gh_step_begin / "
             "gh_step_merge / gh_step_finish_gathered (or gh_run_partitioned)";
    return GH_ERR_INVALID;
}
static gh_status check_k(gh_engine *h) {
    if ((int64_t)h->K > h->E) {
        h->err = "selected index k out of range";  // torch.topk's message (pt.py:583)
        return GH_ERR_K_TOO_LARGE;
    }
    return GH_OK;
}

// Chooses this iteration's sample ids: caller's ids, arange (S >= E, pt.py:412) or the sampler.
static gh_status set_sample(gh_engine *h, const int32_t *host_ids, const int32_t *dev_ids) {
    h->sample_pending = false;
    if (h->S >= h->E) {  // no randomness consumed
        h->d_sampled_cur = h->d_sampled;
        h->sample_pending = true;  // produced inside the KNN setup kernel (or by gh_ensure_sample)
        h->sample_mode = 2;
        return GH_OK;
    }
    if (dev_ids) { h->d_sampled_cur = const_cast<int32_t *>(dev_ids); return GH_OK; }
    h->d_sampled_cur = h->d_sampled;
    if (host_ids) {
        for (int64_t i = 0; i < h->S; ++i)
            if (host_ids[i] < 0 || host_ids[i] >= h->E) { h->err = "sampled edge id out of range"; return GH_ERR_INVALID; }
        // a set-up done ahead (inside the last normalise launch) left ITS ids in d_sampled and built the query
        // records from them: overwriting the ids makes it stale
        h->presetup_valid = false;
        GH_HIP(hipMemcpyAsync(h->d_sampled, host_ids, sizeof(int32_t) * (size_t)h->S, hipMemcpyHostToDevice, h->stream));
        GH_HIP(hipStreamSynchronize(h->stream));
        return GH_OK;
    }
    h->sample_pending = true;
    h->sample_mode = 1;
    return GH_OK;
}

// Spring forces of the own rows -> d_Fs and this rank's K best keys per query -> d_partial.
// fuse_intersect: single-rank step, the KNN kernels also run the intersection phase.
static gh_status step_begin_launches(gh_engine *h, bool fuse_intersect);
static gh_status step_begin(gh_engine *h, bool fuse_intersect) {
    h->rows_early = false;
    GH_TRY(step_begin_launches(h, fuse_intersect));
    // form D: new0 = pos + Fs of the own rows is in their block -- written by the fused kernel, or (a rank too small for it;
    // every rank must send at the same point of the iteration) by a launch of its own -- and may travel now
    if (h->overlap) {
        if (!h->new0_ready) GH_TRY(gh_launch_new0(h));
        h->rows_early = true;
    }
    return GH_OK;
}
static gh_status step_begin_launches(gh_engine *h, bool fuse_intersect) {
    h->intersect_done = false;
    h->stats_reduced = false;
    h->new0_ready = false;
    if (h->S == 0 || h->k == 0) {  // nothing sampled / no neighbours asked for: spring forces only
        h->sample_pending = false;
        h->intersect_done = true;
        GH_HIP(hipMemsetAsync(h->d_tcount, 0, sizeof(int32_t), h->stream));
        return gh_launch_spring_mid(h);
    }
    if (gh_grid_path(h)) {  // sub-quadratic search: thresholds, own midpoints to memory, grid build + cell search
        GH_TRY(gh_knn_prepare(h));          // query records only: the thresholds come from the grid itself
        GH_TRY(gh_launch_spring_mid(h));
        GH_TRY(gh_grid_search(h));
        return gh_knn_finish(h, true, fuse_intersect);
    }
    if (gh_ivf_path(h)) {   // inverted-file search (approximate: probed lists only; ivf.hip)
        GH_TRY(gh_knn_prepare(h));          // query records only
        GH_TRY(gh_launch_spring_mid(h));
        GH_TRY(gh_ivf_search(h));
        return gh_knn_finish(h, true, fuse_intersect);
    }
    if (h->fused_scan && gh_knn_scan_path(h)) {
        GH_TRY(gh_knn_prepare(h));
        if (!h->tau_embedded) GH_TRY(gh_knn_thresholds(h));   // else: the first workgroups of the fused launch (tau_core.h)
        GH_TRY(gh_launch_spring_scan(h));
        return gh_knn_finish(h, false, fuse_intersect);
    }
    GH_TRY(gh_launch_spring_mid(h));
    return gh_knn_local(h, fuse_intersect);
}

static gh_status step_merge(gh_engine *h, const uint64_t *gathered, int world) {
    GH_TRY(gh_knn_merge(h, gathered, world));
    if (!h->intersect_done) GH_TRY(gh_launch_intersect(h));
    GH_TRY(gh_launch_integrate(h));
    return GH_OK;
}

// next_mode < 0: plain finish.  Otherwise the normalise launch of a single-rank step on the fused path
// also runs the next iteration's KNN set-up, for the sample source expected then (gh_launch_normalise);
// gh_knn_prepare falls back to its own kernel when the next step turns out different.
static gh_status step_finish(gh_engine *h, int next_mode = -1, int32_t *next_ids = nullptr) {
    const bool presetup = next_mode >= 0 && h->rows == h->n && !h->d_gbuf && gh_knn_scan_path(h) &&
                          (h->fused_scan || gh_grid_path(h) || gh_ivf_path(h)) && h->S > 0 && h->k > 0 && !h->opt_no_presetup;
    GH_TRY(gh_launch_normalise(h, true, presetup, next_mode, next_ids));  // also zeroes what the intersection phase touched
    h->iter += 1;
    return GH_OK;
}

extern "C" gh_status gh_step(gh_handle h, const int32_t *sampled) {
    GH_TRY(check_handle(h));
    if (h->f64) return gh_f64_step(h, sampled);
    GH_TRY(check_whole(h, "gh_step"));
    GH_TRY(check_k(h));
    GH_TRY(set_sample(h, sampled, nullptr));
    GH_TRY(step_begin(h, true));
    GH_TRY(step_merge(h, h->d_partial, 1));
    // a caller that drew this step's ids itself will do so again; otherwise prepare the next step's own draw
    return step_finish(h, sampled ? -1 : (h->S >= h->E ? 2 : 1));
}

// Uploads an (iters, S) host id stream for a run (validated); *d_ids = nullptr when the run draws its own ids
// (no stream given, or S >= E where the reference uses arange, pt.py:412).
gh_status gh_upload_sample_stream(gh_engine *h, int32_t iters, const int32_t *sample_stream, const int32_t **d_ids) {
    *d_ids = nullptr;
    if (!sample_stream || h->S >= h->E || iters <= 0) return GH_OK;
    const size_t cnt = (size_t)iters * (size_t)h->S;
    for (size_t i = 0; i < cnt; ++i)
        if (sample_stream[i] < 0 || sample_stream[i] >= h->E) { h->err = "sampled edge id out of range"; return GH_ERR_INVALID; }
    if (cnt > h->stream_ids_cap) {
        if (h->d_stream_ids) { GH_HIP(hipStreamSynchronize(h->stream)); GH_HIP(hipFree(h->d_stream_ids)); h->d_stream_ids = nullptr; }
        GH_TRY(dev_alloc(h, &h->d_stream_ids, cnt, false));
        h->stream_ids_cap = cnt;
    }
    GH_HIP(hipMemcpyAsync(h->d_stream_ids, sample_stream, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    *d_ids = h->d_stream_ids;
    return GH_OK;
}

// One iteration of a device-sampled run as the host enqueues it (set-up of the next iteration inside its normalise launch).
static gh_status run_one_device_sampled(gh_engine *h) {
    GH_TRY(set_sample(h, nullptr, nullptr));
    GH_TRY(step_begin(h, true));
    GH_TRY(step_merge(h, h->d_partial, 1));
    return step_finish(h, h->S >= h->E ? 2 : 1);
}

// Iterations 2.. of a device-sampled run replayed from a hipGraph (OPT-IN: GRAPHEM_HIP_GRAPH=1).  In steady state an
// iteration is the same four or five launches with the same arguments, except for the iteration number the sampler is
// keyed with -- that lives in device memory while replaying (d_iter: moved on by stats_fix_kernel, read by the set-up
// inside the following normalise launch).  Measured (round 3, bench.py, median of 3 passes of 50 iterations, same box):
// one iteration per graph 172.2 us against 166.3 enqueued at 1 M vertices, 62.3 / 57.3 at 100 K, 121.8 / 117.3 on the
// 16-component SNAP shape; ten iterations per graph 170.1 / 167.9, 58.6 / 57.6, 117.8 / 117.7 -- a graph launch costs
// more than it saves on this runtime (the 1.9 - 2.8 us per replayed boundary of tools/micro/grid_barrier.hip did not
// carry over to kernels with 200-byte argument blocks), so the enqueued loop stays the default.
static bool graph_replay_applies(gh_engine *h) {
    // the fused path, or the inverted-file path (its ~18 launches and memsets per iteration: the device counter is moved on
    // by stats_reduce_kernel there)
    const bool path = gh_ivf_path(h) || (h->fused_scan && !gh_grid_path(h));
    return whole_graph(h) && !h->d_gbuf && !h->g_world && !h->cdist && !h->timing && !h->d_stamps && path &&
           gh_knn_scan_path(h) && h->S > 0 && h->k > 0 && h->K <= 128 && h->LD <= 16 &&
           h->opt_graph && !h->opt_no_presetup;
}
static int graph_iters() { return 10; }   // iterations per captured graph (a graph launch has a cost of its own: one iteration per
                                          // graph was SLOWER than enqueuing, 172.2 against 166.3 us per iteration at 1 M vertices)
// Replays as many whole graphs (graph_iters() iterations each) as fit into `count`; *done = iterations replayed.
static gh_status graph_replay(gh_engine *h, int32_t count, int32_t *done) {
    *done = 0;
    const int G = graph_iters();
    if (count < G) return GH_OK;
    if (!h->graph_exec) {
        // captured from the steady state: the previous launch has done this iteration's set-up (presetup_valid)
        if (!h->presetup_valid || h->presetup_iter != h->iter) { h->err = "graph replay: not in the steady state (enqueuing instead)"; return GH_ERR_RUNTIME; }   // (caller falls back to enqueuing)
        const bool pv = h->presetup_valid, trp = h->tcount_reset_pending;
        const int pm = h->presetup_mode;
        const int32_t *pi = h->presetup_ids;
        const uint64_t it0 = h->iter, pit = h->presetup_iter;
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); h->err = "graph replay: hipStreamBeginCapture failed (enqueuing instead)"; return GH_ERR_RUNTIME; }
        h->graph_capturing = true;
        gh_status st = GH_OK;
        for (int g = 0; g < G && st == GH_OK; ++g) st = run_one_device_sampled(h);   // (iteration numbers: offsets to the device counter)
        h->graph_capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(h->stream, &g);
        // nothing ran: the host-side state goes back to where the captured iterations started
        h->iter = it0; h->presetup_valid = pv; h->presetup_mode = pm; h->presetup_ids = pi; h->presetup_iter = pit;
        h->tcount_reset_pending = trp;
        h->new0_ready = false; h->stats_reduced = false; h->intersect_done = false; h->sample_pending = false;
        if (st != GH_OK || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); if (st == GH_OK) h->err = "graph replay: capture failed (enqueuing instead)"; return GH_ERR_RUNTIME; }
        if (hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGraphDestroy(g); (void)hipGetLastError(); h->graph_exec = nullptr;
            h->err = "graph replay: hipGraphInstantiate failed (enqueuing instead)";
            return GH_ERR_RUNTIME;
        }
        if (h->graph) (void)hipGraphDestroy(h->graph);   // (a previous capture whose executable was dropped)
        h->graph = g;
    }
    GH_HIP(hipMemcpyAsync(h->d_iter, &h->iter, sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));   // (the source is a member of *h: the copy must not outlive this call's view of it)
    const int32_t launches = count / G;
    for (int32_t t = 0; t < launches; ++t) GH_HIP(hipGraphLaunch(h->graph_exec, h->stream));
    *done = launches * G;
    h->iter += (uint64_t)*done;
    h->presetup_valid = true;          // the last replayed normalise launch set the next iteration up
    h->presetup_iter = h->iter;
    h->presetup_mode = h->S >= h->E ? 2 : 1;
    h->presetup_ids = h->d_sampled;
    h->tcount_reset_pending = false;
    return GH_OK;
}

extern "C" gh_status gh_run(gh_handle h, int32_t iters, const int32_t *sample_stream) {
    GH_TRY(check_handle(h));
    if (h->f64) return iters < 0 ? GH_ERR_INVALID : gh_f64_run(h, iters, sample_stream);
    if (iters < 0) { h->err = "negative iteration count"; return GH_ERR_INVALID; }
    if (iters == 0) return GH_OK;
    GH_TRY(check_whole(h, "gh_run"));
    GH_TRY(check_k(h));
    const int32_t *d_ids = nullptr;
    GH_TRY(gh_upload_sample_stream(h, iters, sample_stream, &d_ids));
    const bool use_stream = d_ids != nullptr;
    int32_t t = 0;
    if (!use_stream && iters > graph_iters() && graph_replay_applies(h)) {
        GH_TRY(run_one_device_sampled(h));   // into the steady state (the set-up of iteration 2 rides in this one's normalise launch)
        t = 1;
        int32_t done = 0;
        if (graph_replay(h, iters - 1, &done) == GH_OK) t += done;   // whole graphs; what is left is enqueued below
    }
    for (; t < iters; ++t) {
        GH_TRY(set_sample(h, nullptr, use_stream ? h->d_stream_ids + (size_t)t * h->S : nullptr));
        GH_TRY(step_begin(h, true));
        GH_TRY(step_merge(h, h->d_partial, 1));
        const bool more = t + 1 < iters;
        if (h->S >= h->E) GH_TRY(step_finish(h, 2));
        else if (use_stream && more) GH_TRY(step_finish(h, 0, h->d_stream_ids + (size_t)(t + 1) * h->S));
        else GH_TRY(step_finish(h, use_stream ? -1 : 1));  // after the last id row: nothing to prepare
    }
    h->d_sampled_cur = h->d_sampled;
    return GH_OK;
}

extern "C" gh_status gh_set_cdist_replay(gh_handle h, int32_t all_ties) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_set_cdist_replay"));
    if (!h->cdist) { h->err = "gh_set_cdist_replay: not a GH_DIST_CDIST engine"; return GH_ERR_INVALID; }
    h->cd_all_ties = all_ties != 0;
    return GH_OK;
}

// ---- the reference's own sampler, drawn beside the loop (pt.py:403-413) ---------------------------------------------
extern "C" gh_status gh_torch_randperm_prefix(uint8_t *rng_state, int64_t state_bytes, int64_t n, int64_t S, int32_t iters,
                                              int32_t *ids) {
    if (!rng_state || state_bytes != GH_TORCH_RNG_STATE_BYTES || n < 0 || n >= ((int64_t)1 << 31) || S < 0 || S > n || iters < 0 ||
        (!ids && iters > 0 && S > 0))
        return GH_ERR_INVALID;
    gh_mt19937 mt;
    if (!gh_mt_load(&mt, rng_state)) return GH_ERR_INVALID;
    std::vector<int64_t> scratch((size_t)gh_rp_scratch_words(S));
    for (int32_t t = 0; t < iters; ++t) gh_torch_randperm_prefix_one(&mt, n, S, ids + (size_t)t * (size_t)S, scratch.data());
    gh_mt_store(&mt, rng_state);
    return GH_OK;
}
extern "C" const char *gh_torch_randperm_isa(void) { return gh_mt_isa(); }

// iters iterations whose sample ids are torch.randperm(E)[:S] of the generator state handed in -- what run_layout of the
// reference's CPU backend consumes (one randperm per iteration, pt.py:409) -- drawn by a host thread into a circular host
// buffer while this thread enqueues: before iteration t goes out, the rows drawn so far (at least row t + 1, whose query
// records the normalise launch of iteration t sets up; at most GH_RING_CHUNK at a time) are copied to a pinned slot and
// from there, in one copy on the engine's stream, into a device ring of GH_DEV_RING rows.  The GPU starts after ONE draw;
// when the producer is ahead the uploads are whole slots, when it is the bottleneck they are single rows.  The only host
// synchronisation with the stream is for a pinned slot to come back (GH_RING_SLOTS uploads later).
#define GH_DEV_RING 128    /* rows of the device ring: a row is overwritten GH_DEV_RING - GH_RING_CHUNK - 1 iterations after its own at the earliest */
#define GH_HOST_RING 256   /* rows the producer may be ahead of the uploads */
extern "C" gh_status gh_run_torch_sampled(gh_handle h, int32_t iters, uint8_t *rng_state, int64_t state_bytes) {
    GH_TRY(check_handle(h));
    if (iters < 0) { h->err = "negative iteration count"; return GH_ERR_INVALID; }
    if (!rng_state || state_bytes != GH_TORCH_RNG_STATE_BYTES) { h->err = "rng_state must be the 5056 bytes of torch.get_rng_state()"; return GH_ERR_INVALID; }
    if (iters == 0) return GH_OK;
    if (h->S >= h->E) return gh_run(h, iters, nullptr);   // arange(E): no randomness consumed (pt.py:412)
    gh_mt19937 mt;
    if (!gh_mt_load(&mt, rng_state)) { h->err = "rng_state is not a torch CPU generator state (mt19937, legacy layout)"; return GH_ERR_INVALID; }
    const size_t S = (size_t)h->S;
    if (h->f64) {   // (the float64 engine takes host ids step by step: drawn up front)
        std::vector<int32_t> ids((size_t)iters * S);
        std::vector<int64_t> scratch((size_t)gh_rp_scratch_words(h->S));
        for (int32_t t = 0; t < iters; ++t) gh_torch_randperm_prefix_one(&mt, h->E, h->S, ids.data() + (size_t)t * S, scratch.data());
        GH_TRY(gh_f64_run(h, iters, ids.data()));
        gh_mt_store(&mt, rng_state);
        return GH_OK;
    }
    GH_TRY(check_whole(h, "gh_run_torch_sampled"));
    GH_TRY(check_k(h));
    const size_t ring_words = (size_t)GH_RING_SLOTS * GH_RING_CHUNK * S;
    if (h->ring_cap < ring_words) {
        GH_HIP(hipStreamSynchronize(h->stream));
        if (h->h_ring) { (void)hipHostFree(h->h_ring); h->h_ring = nullptr; h->ring_cap = 0; }
        GH_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->h_ring), ring_words * sizeof(int32_t), hipHostMallocDefault));
        h->ring_cap = ring_words;
        for (hipEvent_t &e : h->ring_ev)
            if (!e) GH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if ((size_t)GH_DEV_RING * S > h->stream_ids_cap) {
        if (h->d_stream_ids) { GH_HIP(hipStreamSynchronize(h->stream)); GH_HIP(hipFree(h->d_stream_ids)); h->d_stream_ids = nullptr; h->stream_ids_cap = 0; }
        GH_TRY(dev_alloc(h, &h->d_stream_ids, (size_t)GH_DEV_RING * S, false));
        h->stream_ids_cap = (size_t)GH_DEV_RING * S;
    }

    std::vector<int32_t> hbuf((size_t)GH_HOST_RING * S);
    std::mutex mu;
    std::condition_variable cv;
    int32_t drawn = 0;      // rows in hbuf (producer -> this thread)
    int32_t taken = 0;      // rows copied out of hbuf (this thread -> producer)
    bool stop = false;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    const clk::time_point t_begin = clk::now();
    double draw_ms = 0.0, slot_wait_ms = 0.0, main_wait_ms = 0.0;
    std::thread producer([&]() {
        std::vector<int64_t> scratch((size_t)gh_rp_scratch_words(h->S));
        for (int32_t t = 0; t < iters; ++t) {
            if (t >= GH_HOST_RING) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || taken > t - GH_HOST_RING; });
                if (stop) return;
            }
            const clk::time_point t0 = clk::now();
            gh_torch_randperm_prefix_one(&mt, h->E, h->S, hbuf.data() + (size_t)(t % GH_HOST_RING) * S, scratch.data());
            draw_ms += ms_since(t0);
            {
                std::lock_guard<std::mutex> lk(mu);
                drawn = t + 1;
                if (stop) return;
            }
            cv.notify_all();
        }
    });
    int32_t uploaded = 0, uploads = 0;
    // rows [uploaded, uploaded + m) -> device ring; m >= 1 once row `need` is drawn
    auto upload_through = [&](int32_t need) -> gh_status {
        while (uploaded <= need) {
            int32_t have;
            {
                const clk::time_point t0 = clk::now();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return drawn > uploaded; });
                have = drawn;
                main_wait_ms += ms_since(t0);
            }
            int32_t m = std::min<int32_t>(have - uploaded, GH_RING_CHUNK);
            m = std::min<int32_t>(m, GH_DEV_RING - uploaded % GH_DEV_RING);     // neither ring wraps inside one copy
            m = std::min<int32_t>(m, GH_HOST_RING - uploaded % GH_HOST_RING);
            const int slot = uploads % GH_RING_SLOTS;
            if (uploads >= GH_RING_SLOTS) {
                const clk::time_point t0 = clk::now();
                GH_HIP(hipEventSynchronize(h->ring_ev[slot]));
                slot_wait_ms += ms_since(t0);
            }
            int32_t *pin = h->h_ring + (size_t)slot * GH_RING_CHUNK * S;
            memcpy(pin, hbuf.data() + (size_t)(uploaded % GH_HOST_RING) * S, sizeof(int32_t) * (size_t)m * S);
            GH_HIP(hipMemcpyAsync(h->d_stream_ids + (size_t)(uploaded % GH_DEV_RING) * S, pin, sizeof(int32_t) * (size_t)m * S, hipMemcpyHostToDevice, h->stream));
            GH_HIP(hipEventRecord(h->ring_ev[slot], h->stream));
            uploaded += m;
            ++uploads;
            {
                std::lock_guard<std::mutex> lk(mu);
                taken = uploaded;
            }
            cv.notify_all();
        }
        return GH_OK;
    };
    auto row = [&](int32_t t) { return h->d_stream_ids + (size_t)(t % GH_DEV_RING) * S; };
    auto loop = [&]() -> gh_status {
        for (int32_t t = 0; t < iters; ++t) {
            GH_TRY(upload_through(std::min(t + 1, iters - 1)));   // this iteration's ids, and the next one's for its normalise launch
            GH_TRY(set_sample(h, nullptr, row(t)));
            GH_TRY(step_begin(h, true));
            GH_TRY(step_merge(h, h->d_partial, 1));
            if (t + 1 < iters) GH_TRY(step_finish(h, 0, row(t + 1)));
            else GH_TRY(step_finish(h, -1));
        }
        return GH_OK;
    };
    const gh_status st = loop();
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = st != GH_OK;
    }
    cv.notify_all();
    producer.join();
    h->d_sampled_cur = h->d_sampled;
    h->sampler_stats[0] = draw_ms; h->sampler_stats[1] = slot_wait_ms; h->sampler_stats[2] = main_wait_ms; h->sampler_stats[3] = ms_since(t_begin);
    if (st != GH_OK) return st;
    gh_mt_store(&mt, rng_state);
    return GH_OK;
}

extern "C" gh_status gh_sampler_stats(gh_handle h, double *out4) {
    GH_TRY(check_handle(h));
    if (!out4) { h->err = "out4 is NULL"; return GH_ERR_INVALID; }
    for (int i = 0; i < 4; ++i) out4[i] = h->sampler_stats[i];
    return GH_OK;
}

extern "C" gh_status gh_sync(gh_handle h) {
    GH_TRY(check_handle(h));
    GH_HIP(hipStreamSynchronize(h->stream));
    resolve_timers(h);
    return check_device_waits(h);
}

// ---- multi-GPU split step ----------------------------------------------------------
extern "C" gh_status gh_step_begin(gh_handle h, const int32_t *sampled) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_step_begin"));
    GH_TRY(check_k(h));
    h->last_step_own_ids = sampled == nullptr;
    GH_TRY(set_sample(h, sampled, nullptr));
    return step_begin(h, false);
}
// Part 1 of a split step with the ids already on the device (a row of an uploaded stream), or nullptr: the
// engine draws them itself, identically on every rank (comm.hip gh_run_partitioned).
gh_status gh_step_begin_device_ids(gh_engine *h, const int32_t *dev_ids) {
    GH_TRY(check_k(h));
    h->last_step_own_ids = dev_ids == nullptr;
    GH_TRY(set_sample(h, nullptr, dev_ids));
    return step_begin(h, false);
}
extern "C" gh_status gh_set_stream(gh_handle h, void *hip_stream, int32_t use_own) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_set_stream"));
    GH_HIP(hipStreamSynchronize(h->stream));
    resolve_timers(h);
    h->stream = use_own ? h->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    return GH_OK;
}
extern "C" int64_t gh_positions_rows_allocated(gh_handle h) { return h ? h->pos_rows : 0; }
extern "C" uint64_t *gh_knn_partial_device(gh_handle h) { return h ? h->d_partial : nullptr; }
extern "C" const uint64_t *gh_knn_merged_device(gh_handle h) { return h ? h->d_keys_cur : nullptr; }
extern "C" int32_t gh_knn_partial_cols(gh_handle h) { return h ? h->K + (h->cd_part ? 2 : 0) : 0; }
extern "C" gh_status gh_step_merge(gh_handle h, const uint64_t *gathered, int32_t world) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_step_merge"));
    if (!gathered || world < 1) { h->err = "bad gathered buffer / world size"; return GH_ERR_INVALID; }
    return step_merge(h, gathered, world);
}
extern "C" double *gh_stats_partial_device(gh_handle h) { return h ? h->d_stats : nullptr; }
extern "C" int32_t gh_stats_rows(gh_handle h) { return h ? 2 + 2 * gh_fix_blocks(h->LD) : 0; }
extern "C" gh_status gh_step_finish(gh_handle h) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_step_finish"));
    return step_finish(h);
}

extern "C" gh_status gh_gather_layout(gh_handle h, int32_t world, int32_t rank, int64_t chunk) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_gather_layout"));
    if (world < 1 || rank < 0 || rank >= world || chunk < 1 || chunk * world < h->n ||
        h->part.row_lo != std::min<int64_t>(h->n, rank * chunk) || h->part.row_hi != std::min<int64_t>(h->n, (rank + 1) * chunk)) {
        h->err = "gather layout does not match the engine's row partition";
        return GH_ERR_INVALID;
    }
    if (h->d_gbuf || h->g_world) { h->err = "rank / gather layout already set"; return GH_ERR_INVALID; }
    const int64_t stats_bytes = (int64_t)sizeof(double) * (2 + 2 * gh_fix_blocks(h->LD)) * h->LD;
    const int64_t slot = (chunk * h->LD * (int64_t)sizeof(float) + stats_bytes + 15) / 16 * 16;
    GH_HIP(hipStreamSynchronize(h->stream));
    GH_TRY(dev_alloc(h, &h->d_gbuf, (size_t)(slot * world), true));
    h->d_new_own = h->d_new;
    h->d_stats_own = h->d_stats;
    h->d_new = reinterpret_cast<float *>(h->d_gbuf + rank * slot);
    h->d_stats = reinterpret_cast<double *>(h->d_gbuf + rank * slot + chunk * h->LD * (int64_t)sizeof(float));
    h->g_slot = slot; h->g_chunk = chunk; h->g_world = world; h->g_rank = rank;
    return GH_OK;
}
extern "C" gh_status gh_rank_layout(gh_handle h, int32_t world, int32_t rank, int64_t chunk) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_rank_layout"));
    if (world < 1 || rank < 0 || rank >= world || chunk < 1 || chunk * world < h->n || chunk * world > h->pos_rows ||
        h->part.row_lo != std::min<int64_t>(h->n, rank * chunk) || h->part.row_hi != std::min<int64_t>(h->n, (rank + 1) * chunk)) {
        h->err = "rank layout does not match the engine's row partition";
        return GH_ERR_INVALID;
    }
    if (h->d_gbuf || h->g_world) { h->err = "rank / gather layout already set"; return GH_ERR_INVALID; }
    GH_TRY(dev_alloc(h, &h->d_stats_comb, (size_t)2 * h->LD, true));
    // fewer components than the row stride (3 of 4, 5..7 of 8, 9..15 of 16): the finished blocks travel unpadded
    if (h->D < h->LD && world > 1) {
        GH_TRY(dev_alloc(h, &h->d_rows_packed, (size_t)world * chunk * h->D, true));
        h->packed_exchange = h->n >= ((int64_t)1 << 21);
    }
    h->g_chunk = chunk; h->g_world = world; h->g_rank = rank;
    return GH_OK;
}
// Form D: form B's finish (every rank normalises all n rows from the gathered un-normalised rows) with the big collective
// moved to the front of the KNN tail -- see include/graphem_hip.h.
extern "C" gh_status gh_overlap_layout(gh_handle h, int32_t world, int32_t rank, int64_t chunk) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_overlap_layout"));
    if (world < 1 || rank < 0 || rank >= world || chunk < 1 || chunk * world < h->n ||
        h->part.row_lo != std::min<int64_t>(h->n, rank * chunk) || h->part.row_hi != std::min<int64_t>(h->n, (rank + 1) * chunk)) {
        h->err = "overlap layout does not match the engine's row partition";
        return GH_ERR_INVALID;
    }
    if (h->d_gbuf || h->g_world) { h->err = "rank / gather layout already set"; return GH_ERR_INVALID; }
    if (h->LD > 16) { h->err = "gh_overlap_layout: up to 16 components (use gh_rank_layout / gh_gather_layout beyond)"; return GH_ERR_INVALID; }
    const size_t R = (size_t)(2 + 2 * gh_fix_blocks(h->LD));
    // a rank's block of the late all-gather: statistics rows, 16 bytes for the patch count, the patch records
    h->patch_cap = std::max<int64_t>(1, std::min<int64_t>(4 * h->S * h->k, chunk));
    h->stats_block = (int64_t)(R * h->LD) + 2 + ((int64_t)h->patch_cap * (1 + h->LD) * 4 + 7) / 8;
    h->stats_block = (h->stats_block + 1) / 2 * 2;   // 16-byte multiples
    GH_HIP(hipStreamSynchronize(h->stream));
    GH_TRY(dev_alloc(h, &h->d_rows_all, (size_t)world * chunk * h->LD, true));
    GH_TRY(dev_alloc(h, &h->d_stats_all, (size_t)world * (size_t)h->stats_block, true));
    if (h->D < h->LD && world > 1) GH_TRY(dev_alloc(h, &h->d_rows_pk, (size_t)world * chunk * h->D, true));
    GH_HIP(hipStreamSynchronize(h->stream));
    h->d_new_own = h->d_new;
    h->d_stats_own = h->d_stats;
    h->d_new = h->d_rows_all + (size_t)rank * chunk * h->LD;
    h->d_stats = h->d_stats_all + (size_t)rank * (size_t)h->stats_block;
    h->g_chunk = chunk; h->g_world = world; h->g_rank = rank;
    h->overlap = true;
    return GH_OK;
}
extern "C" float *gh_rows_all_device(gh_handle h) { return !h || !h->overlap ? nullptr : h->d_rows_pk ? h->d_rows_pk : h->d_rows_all; }
extern "C" int32_t gh_rows_all_row_floats(gh_handle h) { return !h || !h->overlap ? 0 : h->d_rows_pk ? h->D : h->LD; }
extern "C" double *gh_stats_all_device(gh_handle h) { return h && h->overlap ? h->d_stats_all : nullptr; }
extern "C" int64_t gh_stats_all_block_doubles(gh_handle h) { return h && h->overlap ? h->stats_block : 0; }
extern "C" int32_t gh_step_rows_early(gh_handle h) { return h && h->overlap && h->rows_early ? 1 : 0; }
extern "C" gh_status gh_step_pack_rows(gh_handle h, void *hip_stream, int32_t use_engine_stream) {
    GH_TRY(check_handle(h));
    if (!h->overlap) { h->err = "gh_overlap_layout has not been called"; return GH_ERR_INVALID; }
    return gh_launch_pack_rows(h, use_engine_stream ? h->stream : reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" gh_status gh_step_finish_overlap(gh_handle h) {
    GH_TRY(check_handle(h));
    if (!h->overlap) { h->err = "gh_overlap_layout has not been called"; return GH_ERR_INVALID; }
    GH_TRY(gh_launch_patch_rows(h));
    GH_TRY(gh_launch_normalise_gathered(h, h->last_step_own_ids ? (h->S >= h->E ? 2 : 1) : -1));
    h->rows_early = false;
    h->iter += 1;
    return GH_OK;
}

extern "C" gh_status gh_set_packed_rows(gh_handle h, int32_t on) {
    GH_TRY(check_handle(h));
    if (on && !h->d_rows_packed) { h->err = "no packed block exchange for this engine (needs gh_rank_layout with world > 1 and fewer components than the row stride)"; return GH_ERR_INVALID; }
    h->packed_exchange = on != 0;
    return GH_OK;
}
extern "C" float *gh_rows_packed_device(gh_handle h) { return h && h->packed_exchange ? h->d_rows_packed : nullptr; }
extern "C" gh_status gh_step_unpack_rows(gh_handle h) {
    GH_TRY(check_handle(h));
    if (!h->d_stats_comb || h->d_gbuf) { h->err = "gh_rank_layout has not been called"; return GH_ERR_INVALID; }
    if (!h->packed_exchange) { h->err = "the packed block exchange is not in use (gh_set_packed_rows)"; return GH_ERR_INVALID; }
    return gh_launch_unpack_rows(h);
}
extern "C" gh_status gh_step_finish_own(gh_handle h, const double *stats_all, int32_t world) {
    GH_TRY(check_handle(h));
    if (!h->d_stats_comb || h->d_gbuf) { h->err = "gh_rank_layout has not been called"; return GH_ERR_INVALID; }
    if (!stats_all || world != h->g_world) { h->err = "bad statistics buffer / world size"; return GH_ERR_INVALID; }
    GH_TRY(gh_launch_normalise_own(h, stats_all, world));
    h->iter += 1;
    return GH_OK;
}
extern "C" void *gh_gather_buffer_device(gh_handle h) { return h ? h->d_gbuf : nullptr; }
extern "C" int64_t gh_gather_slot_bytes(gh_handle h) { return h ? h->g_slot : 0; }
extern "C" gh_status gh_step_finish_gathered(gh_handle h) {
    GH_TRY(check_handle(h));
    if (!h->d_gbuf) { h->err = "gh_gather_layout has not been called"; return GH_ERR_INVALID; }
    // a rank that drew this step's ids on the device will do so again: prepare them in the same launch
    GH_TRY(gh_launch_normalise_gathered(h, h->last_step_own_ids ? (h->S >= h->E ? 2 : 1) : -1));
    h->iter += 1;
    return GH_OK;
}

extern "C" gh_status gh_radial_topk(gh_handle h, int32_t k, int32_t *ids) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_radial_topk"));
    if (!ids) { h->err = "ids is NULL"; return GH_ERR_INVALID; }
    if (k < 1 || k > 64 || k > h->n) { h->err = "gh_radial_topk: k must be in [1, min(n, 64)]"; return GH_ERR_INVALID; }
    int nparts = (int)((h->n + 2047) / 2048);
    if (nparts > 256) nparts = 256;
    uint64_t *d_part = nullptr;
    int32_t *d_ids = nullptr;
    GH_TRY(dev_alloc(h, &d_part, (size_t)nparts * k, false));
    gh_status st = dev_alloc(h, &d_ids, (size_t)k, false);
    if (st == GH_OK) st = gh_radial_topk_device(h, k, d_part, nparts, d_ids);
    if (st == GH_OK && hipMemcpyAsync(ids, d_ids, sizeof(int32_t) * k, hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
        h->err = "gh_radial_topk: copy failed";
        st = GH_ERR_HIP;
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess && st == GH_OK) { h->err = "gh_radial_topk: sync failed"; st = GH_ERR_HIP; }
    (void)hipFree(d_part);
    (void)hipFree(d_ids);
    return st;
}

// ---- per-phase entry points --------------------------------------------------------

extern "C" gh_status gh_spring_forces(gh_handle h, float *F) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_spring_forces (use gh_spring_forces_f64)"));
    if (!F) { h->err = "F is NULL"; return GH_ERR_INVALID; }
    GH_TRY(gh_launch_spring_only(h, h->d_tmpF));
    return download_padded(h, h->d_tmpF, F);
}

extern "C" gh_status gh_knn_midpoints(gh_handle h, const int32_t *sampled, int32_t *knn) {
    GH_TRY(check_handle(h));
    if (h->f64) return knn ? gh_f64_knn_midpoints(h, sampled, knn) : GH_ERR_INVALID;
    if (!knn) { h->err = "knn is NULL"; return GH_ERR_INVALID; }
    GH_TRY(check_whole(h, "gh_knn_midpoints"));
    GH_TRY(check_k(h));
    if (!sampled && h->S < h->E) { h->err = "sampled is NULL"; return GH_ERR_INVALID; }
    GH_TRY(set_sample(h, sampled, nullptr));
    GH_TRY(step_begin(h, false));  // the same kernels a step runs (spring forces are a by-product)
    const uint64_t *d_keys = h->d_partial;
    if (h->cd_part) {   // a GH_DIST_CDIST engine created with a (whole-graph) partition: its rows are decided at the merge
        h->intersect_done = true;   // (no intersection phase here)
        GH_TRY(gh_knn_merge(h, h->d_partial, 1));
        h->intersect_done = false;
        d_keys = h->d_merged;
    }
    std::vector<uint64_t> keys((size_t)h->S * h->K);
    GH_HIP(hipMemcpyAsync(keys.data(), d_keys, sizeof(uint64_t) * keys.size(), hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    for (int64_t s = 0; s < h->S; ++s)  // column 0 dropped blindly (pt.py:421)
        for (int c = 1; c < h->K; ++c) knn[s * h->k + (c - 1)] = (int32_t)(keys[(size_t)s * h->K + c] & 0xFFFFFFFFu);
    return GH_OK;
}

extern "C" gh_status gh_intersection_forces(gh_handle h, const int32_t *sampled, const int32_t *knn, float *F) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_intersection_forces (use gh_intersection_forces_f64)"));
    if (!knn || !F) { h->err = "NULL argument"; return GH_ERR_INVALID; }
    GH_TRY(check_whole(h, "gh_intersection_forces"));
    if (!sampled && h->S < h->E) { h->err = "sampled is NULL"; return GH_ERR_INVALID; }
    for (int64_t i = 0; i < h->S * h->k; ++i)
        if (knn[i] < 0 || knn[i] >= h->E) { h->err = "neighbour edge id out of range"; return GH_ERR_INVALID; }
    GH_TRY(set_sample(h, sampled, nullptr));
    GH_TRY(gh_ensure_sample(h));
    std::vector<uint64_t> keys((size_t)h->S * h->K, 0);  // the kernel reads ids from key columns 1..k
    for (int64_t s = 0; s < h->S; ++s)
        for (int c = 1; c < h->K; ++c) keys[(size_t)s * h->K + c] = (uint32_t)knn[s * h->k + (c - 1)];
    GH_HIP(hipMemcpyAsync(h->d_merged, keys.data(), sizeof(uint64_t) * keys.size(), hipMemcpyHostToDevice, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    h->d_keys_cur = h->d_merged;
    GH_TRY(gh_launch_intersect(h));
    GH_TRY(gh_launch_inter_to_dense(h, h->d_tmpF));
    GH_TRY(gh_launch_inter_cleanup(h));
    return download_padded(h, h->d_tmpF, F);
}

extern "C" gh_status gh_integrate_normalise(gh_handle h, const float *Fs, const float *Fi, float *out) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_integrate_normalise"));
    if (!Fs || !Fi || !out) { h->err = "NULL argument"; return GH_ERR_INVALID; }
    GH_TRY(check_whole(h, "gh_integrate_normalise"));
    const size_t bytes = sizeof(float) * (size_t)h->n * h->D;
    GH_HIP(hipMemcpyAsync(h->d_io, Fs, bytes, hipMemcpyHostToDevice, h->stream));
    GH_TRY(gh_launch_pad(h, h->d_io, h->d_tmpF));
    GH_HIP(hipStreamSynchronize(h->stream));
    GH_HIP(hipMemcpyAsync(h->d_io, Fi, bytes, hipMemcpyHostToDevice, h->stream));
    GH_TRY(gh_launch_pad(h, h->d_io, h->d_tmpF2));
    GH_HIP(hipStreamSynchronize(h->stream));
    GH_TRY(gh_launch_integrate_given(h, h->d_tmpF, h->d_tmpF2));
    // normalise into scratch so the current positions stay unchanged
    GH_HIP(hipMemcpyAsync(h->d_tmpF, h->d_pos, sizeof(float) * (size_t)h->n * h->LD, hipMemcpyDeviceToDevice, h->stream));
    GH_TRY(gh_launch_normalise(h, false));
    GH_TRY(gh_launch_unpad(h, h->d_pos, h->d_io));
    GH_HIP(hipMemcpyAsync(out, h->d_io, bytes, hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipMemcpyAsync(h->d_pos, h->d_tmpF, sizeof(float) * (size_t)h->n * h->LD, hipMemcpyDeviceToDevice, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}

// ---- instrumentation ---------------------------------------------------------------
extern "C" gh_status gh_timing_enable(gh_handle h, int32_t on) {
    GH_TRY(check_handle(h));
    h->timing = on != 0;
    return GH_OK;
}
extern "C" gh_status gh_timing_reset(gh_handle h) {
    GH_TRY(check_handle(h));
    GH_HIP(hipStreamSynchronize(h->stream));
    resolve_timers(h);
    h->timers.clear();
    return GH_OK;
}
extern "C" int32_t gh_timing_count(gh_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    resolve_timers(h);
    return (int32_t)h->timers.size();
}
extern "C" gh_status gh_timing_get(gh_handle h, int32_t i, const char **name, double *total_ms, int64_t *launches) {
    if (!h) return GH_ERR_INVALID;
    if (i < 0 || i >= (int32_t)h->timers.size()) { h->err = "timer index out of range"; return GH_ERR_INVALID; }
    if (name) *name = h->timers[(size_t)i].name.c_str();
    if (total_ms) *total_ms = h->timers[(size_t)i].total_ms;
    if (launches) *launches = h->timers[(size_t)i].launches;
    return GH_OK;
}

// Diagnostic builds of a run (GRAPHEM_HIP_STAMPS set at gh_create): wall-clock stamps (100 MHz) of the last fused
// launch, 8 per workgroup: start, after the spring phase, after its barrier, scan operands ready, scan done, hits flushed;
// after those n_vblocks records, GH_STAMP_EXTRA records of the last normalise launch's first workgroups (set-up
// workgroups first): start, mean / std known, [tile staged], end, -, -, 1 = set-up / 2 = normalising.
extern "C" gh_status gh_debug_stamps(gh_handle h, unsigned long long *out, int64_t count) {
    GH_TRY(check_handle(h));
    if (!h->d_stamps) { h->err = "GRAPHEM_HIP_STAMPS was not set when the engine was created"; return GH_ERR_INVALID; }
    const int64_t have = ((int64_t)std::max(h->n_vblocks, 1) + GH_STAMP_EXTRA) * 8;
    GH_HIP(hipStreamSynchronize(h->stream));
    GH_HIP(hipMemcpy(out, h->d_stamps, sizeof(unsigned long long) * (size_t)std::min(count, have), hipMemcpyDeviceToHost));
    return GH_OK;
}

extern "C" gh_status gh_knn_last_counts(gh_handle h, int32_t *subset_counts, int32_t *final_counts, int32_t *overflow) {
    GH_TRY(check_handle(h));
    GH_TRY(reject_f64(h, "gh_knn_last_counts"));
    const size_t bytes = sizeof(int32_t) * (size_t)h->S;
    if (subset_counts) GH_HIP(hipMemcpyAsync(subset_counts, h->d_dbg_cnt, bytes, hipMemcpyDeviceToHost, h->stream));
    if (final_counts) GH_HIP(hipMemcpyAsync(final_counts, h->d_dbg_cnt + h->S, bytes, hipMemcpyDeviceToHost, h->stream));
    if (overflow) GH_HIP(hipMemcpyAsync(overflow, h->d_ovf, bytes, hipMemcpyDeviceToHost, h->stream));
    GH_HIP(hipStreamSynchronize(h->stream));
    return GH_OK;
}

extern "C" gh_status gh_knn_cdist_stats(gh_handle h, int32_t *full_pass_rows, int32_t *unresolved_tie_rows) {
    GH_TRY(check_handle(h));
    int32_t rare = 0, stat = 0;
    if (h->cdist && h->d_rare) {
        const int32_t *hdr = h->d_cd_stat + 4 * (h->cd_set ^ 1);   // the counters of the last search
        GH_HIP(hipMemcpyAsync(&rare, hdr, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        GH_HIP(hipMemcpyAsync(&stat, hdr + 1, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        GH_HIP(hipStreamSynchronize(h->stream));
        if (!gh_knn_scan_path(h)) rare = (int32_t)h->S;   // a graph too small for the filtered scan: every row
    }
    if (full_pass_rows) *full_pass_rows = rare;
    if (unresolved_tie_rows) *unresolved_tie_rows = stat;
    return GH_OK;
}

extern "C" gh_status gh_knn_points(int device_id, const float *q, int64_t nq, const float *ref, int64_t nref,
                                   int32_t D, int32_t k, int64_t *out) {
    auto fail = [&](gh_status st, const std::string &msg) { g_create_error = msg; return st; };
    if (!q || !ref || !out || nq < 0 || nref < 0 || D <= 0 || k <= 0) return fail(GH_ERR_INVALID, "bad argument");
    if ((int64_t)k > nref) return fail(GH_ERR_K_TOO_LARGE, "selected index k out of range");
    if (k > GH_SEL_BUF - GH_SEL_CHUNK) return fail(GH_ERR_INVALID, "k too large for the HIP backend (max 2048)");
    if (nref >= ((int64_t)1 << 31)) return fail(GH_ERR_INVALID, "too many reference points");
    if (hipSetDevice(device_id) != hipSuccess) return fail(GH_ERR_RUNTIME, "invalid device ordinal " + std::to_string(device_id));
    float *d_q = nullptr, *d_ref = nullptr;
    uint64_t *d_keys = nullptr;
    std::vector<uint64_t> keys((size_t)nq * k);
    gh_status st = GH_OK;
    std::string err;
    auto cleanup = [&]() { if (d_q) (void)hipFree(d_q); if (d_ref) (void)hipFree(d_ref); if (d_keys) (void)hipFree(d_keys); };
    if (hipMalloc((void **)&d_q, sizeof(float) * (size_t)std::max<int64_t>(nq * D, 1)) != hipSuccess ||
        hipMalloc((void **)&d_ref, sizeof(float) * (size_t)std::max<int64_t>(nref * D, 1)) != hipSuccess ||
        hipMalloc((void **)&d_keys, sizeof(uint64_t) * std::max<size_t>(keys.size(), 1)) != hipSuccess) {
        cleanup();
        return fail(GH_ERR_NOMEM, "hipMalloc failed");
    }
    if (hipMemcpy(d_q, q, sizeof(float) * (size_t)nq * D, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ref, ref, sizeof(float) * (size_t)nref * D, hipMemcpyHostToDevice) != hipSuccess) {
        cleanup();
        return fail(GH_ERR_HIP, "upload failed");
    }
    st = gh_knn_points_device(nullptr, d_q, nq, d_ref, nref, D, k, d_keys, &err);
    if (st == GH_OK && hipMemcpy(keys.data(), d_keys, sizeof(uint64_t) * keys.size(), hipMemcpyDeviceToHost) != hipSuccess) {
        st = GH_ERR_HIP;
        err = "download failed";
    }
    cleanup();
    if (st != GH_OK) return fail(st, err);
    for (size_t i = 0; i < keys.size(); ++i) out[i] = (int64_t)(keys[i] & 0xFFFFFFFFu);
    return GH_OK;
}

extern "C" int32_t gh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" const char *gh_version(void) { return "graphem_hip 0.1 (gfx950)"; }

// Internal engine state shared by the translation units of libgraphem_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/graphem_hip.h"

// Candidate-list capacity per query in the filtered KNN scan, and the LDS sort size.
#define GH_CAND_CAP 16384   /* (8192 until round 3: one query of a 16 M-vertex run reached 8777 candidates and its exhaustive fallback cost 0.5 s) */
#define GH_SEL_BUF 4096
#define GH_SEL_CHUNK 2048
// Candidate counters are padded to one per 128-byte line: adjacent counters serialise their
// returning atomics on one L2 line (~11 ns each, measured: 16 K appends per line cost 180 us).
#define GH_CNT_STRIDE 32
// Spare rows behind the n positions so equal all-gather chunks fit for any world size <= this.
#define GH_POS_PAD_ROWS 1024
// Row pairs of corrections behind the (2, LD) column statistics: one per stats_fix workgroup.
inline int gh_fix_blocks(int LD) { return LD <= 16 ? 2 * LD : 0; }
// Below this many reference edges the per-query block kernel scans everything itself.
#define GH_SCAN_MIN_EDGES 16384

struct gh_timer_slot {
    std::string name;
    double total_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct gh_comm;   // comm.hip: collective backend of the native partitioned loop
struct gh_f64;    // f64.hip: state of a float64 engine (gh_create_f64)
struct gh_ivf;    // ivf.hip: buffers of the inverted-file search (GH_KNN_IVF)

struct gh_engine {
    int device = 0;
    gh_comm *comm = nullptr;
    gh_f64 *f64 = nullptr;    // non-null: a float64 engine -- only this, the sizes, the parameters and the stream are in use
    int64_t n = 0, E = 0;
    int D = 0, LD = 0, k = 0, K = 0;
    int64_t S = 0;
    gh_params prm{};
    gh_partition part{};
    int64_t rows = 0;       // part.row_hi - part.row_lo
    uint64_t iter = 0;      // iterations done (device sampler counter)
    std::string err;

    hipStream_t stream = nullptr;       // stream all work is enqueued on
    hipStream_t own_stream = nullptr;   // created by gh_create
    int64_t pos_rows = 0;               // rows allocated in d_pos (n + GH_POS_PAD_ROWS)

    // graph
    int32_t *d_edges = nullptr;   // (E, 2)
    int32_t *d_rowptr = nullptr;  // (rows + 1) pull lists of own rows, reference summation order
    int32_t *d_adj = nullptr;     // neighbours
    int64_t adj_len = 0;
    int32_t *d_first_edge = nullptr; // (rows + 1) offset of each own row's owned edges: the first owned edge id
                                     // (rule A: smaller endpoint owns) or a prefix count into d_own_eids (rule B)
    int32_t *d_own_eids = nullptr;   // rule B (balanced ownership of partitioned engines): ids of the owned edges
                                     // in (row, list) order; null under rule A
    int64_t own_count = 0;           // edges this rank owns (midpoints in d_mid, searched by its KNN kernels)
    int64_t mid_base = 0;            // d_mid row of an owned edge = d_first_edge offset - mid_base
    bool fused_mid = false;       // own edge range == edges owned by own rows: spring kernel writes midpoints
    int32_t *d_long_rows = nullptr;   // local ids of the own rows with more than GH_LONG_DEG neighbours (common.h)
    int32_t *d_long_ownptr = nullptr; // their owned (hub-hub) edges: offsets ...
    int32_t *d_long_ownadj = nullptr; // ... and neighbours
    int32_t *d_long_eptr = nullptr;   // (nlong + 1) prefix of their degrees
    int32_t *d_long_erow = nullptr;   // (long_entries) index of the long row a list entry belongs to
    uint8_t *d_own_long = nullptr;    // (own_count) owned-edge slots whose owner row is long (common.h gh_long_midpoints)
    float *d_long_terms = nullptr;    // (long_entries * D) force terms of their neighbours, component-major per row
    int nlong = 0;
    int long_deg = 128;               // rows with more neighbours than this are long (common.h gh_long_degree)
    int64_t long_entries = 0;
    int long_max_deg = 0;             // longest long row (forces.hip: one launch for the long rows when moderate)
    float *d_mid = nullptr;       // (own_count, LD) midpoints of the own edges, current iteration
    float *d_Fs = nullptr;        // (rows, LD) spring forces of the own rows
    float *d_gmin = nullptr;      // (S, Gpad) group minima of the threshold subset (setup_core.h), float bits
    int32_t *d_sub_uv = nullptr;  // (thr_M1, 2) endpoints of the subset edges
    int64_t thr_stride = 0, thr_M1 = 0;  // the subset: every thr_stride-th own edge, thr_M1 of them (0: not chosen yet)
    int32_t *d_vblock = nullptr;  // (n_vblocks + 1) vertex ranges of the fused spring+scan workgroups
    int n_vblocks = 0;
    bool opt_no_presetup = false, opt_graph = false;   // GRAPHEM_HIP_NO_PRESETUP / GRAPHEM_HIP_GRAPH (read at gh_create)
    bool fused_scan = false;      // fused spring+scan kernel usable for this graph / partition

    // state
    float *d_pos = nullptr;       // (n, LD)
    float *d_new = nullptr;       // (rows, LD) un-normalised update of own rows
    float *d_tmpF = nullptr;      // (n, LD) scratch for the per-phase entry points
    float *d_tmpF2 = nullptr;
    float *d_io = nullptr;        // (n, D) staging for unpadded host copies

    // intersection accumulators
    double *d_acc = nullptr;      // (n, LD) zero between iterations
    int32_t *d_tflag = nullptr;   // (n) zero between iterations
    int32_t *d_touched = nullptr; // (4 * S * k)
    int32_t *d_tcount = nullptr;  // (1)

    // knn
    int32_t *d_sampled = nullptr; // (S) owned buffer
    int32_t *d_sampled_cur = nullptr; // ids of the current iteration (d_sampled or a row of d_stream_ids)
    int32_t *d_stream_ids = nullptr;  // (iters, S) uploaded sample stream of gh_run
    size_t stream_ids_cap = 0;
    // gh_run_torch_sampled (api.hip): pinned upload slots for the rows a host thread draws (torch.randperm's prefixes), one
    // event per slot (recorded behind the slot's copy: the slot is reused once it has fired)
#define GH_RING_SLOTS 8
#define GH_RING_CHUNK 32   /* rows per slot = per copy, at most */
    int32_t *h_ring = nullptr;
    size_t ring_cap = 0;              // int32 words allocated in h_ring (GH_RING_SLOTS * GH_RING_CHUNK * S)
    hipEvent_t ring_ev[GH_RING_SLOTS] = {};
    double sampler_stats[4] = {0, 0, 0, 0};   // last gh_run_torch_sampled, host ms: producer drawing, caller waiting for a pinned slot, caller waiting for ids, the call
    bool new0_ready = false;      // the fused kernel of this step wrote d_new = pos + Fs and its block sums
    bool intersect_done = false;  // the KNN kernels of this step already ran the intersection phase
    bool stats_reduced = false;   // ... and reduced the fused kernel's workgroup sums into d_stats (knn_select_kernel)
    bool presetup_valid = false;  // the last normalise launch also ran the KNN set-up of iteration presetup_iter
    int presetup_mode = 0;        //   with this sample mode / id pointer (gh_knn_prepare then skips its kernel)
    const int32_t *presetup_ids = nullptr;
    uint64_t presetup_iter = 0;
    bool last_step_own_ids = false;     // gh_step_begin was called without ids (device sampler / arange)
    bool tcount_reset_pending = false;  // this step's threshold kernel must reset d_tcount
    bool sample_pending = false;  // ids of this iteration still to be produced (inside knn_setup_kernel)
    int sample_mode = 0;          // 1 device sampler, 2 arange
    float *d_iscratch = nullptr;  // (S * k, LD) per-pair scratch of the intersection kernel
    float *d_q = nullptr;         // (S, QS) query records: midpoint coordinates + tau (knn.hip gh_qs)
    float *d_qscan = nullptr;     // (S, QS) pre-filter records (-2q, t) written by the threshold kernel
    uint16_t *d_qA = nullptr;     // (S, 16) f16 A-operand rows of the split-f16 MFMA pre-filter (D <= 3), same kernel
    int32_t *d_order = nullptr;       // internal row of every vertex (BFS reordering), or null: identity
    std::vector<int32_t> order_host;  // the same on the host (empty: identity)
    std::vector<int32_t> edges_internal;  // gh_create scratch: the edge list in internal vertex numbers
    unsigned char *d_gbuf = nullptr;  // gather buffer of the one-collective finish (gh_gather_layout), or null
    float *d_new_own = nullptr;       // allocations behind d_new / d_stats while they point into d_gbuf
    double *d_stats_own = nullptr;
    int64_t g_slot = 0, g_chunk = 0;  // slot bytes, rows per rank
    int g_world = 0, g_rank = 0;
    double *d_stats_comb = nullptr;   // form C (gh_rank_layout): the ranks' statistics added in rank order, (2, LD)
    float *d_rows_packed = nullptr;   // form C, D < LD: (world, chunk, D) the finished blocks WITHOUT the pad columns -- what travels
                                      // (12 instead of 16 bytes per row at 3 components); gh_step_unpack_rows expands it into d_pos
    bool packed_exchange = false;     // ... in use (gh_set_packed_rows; default: from 2 M vertices on, where the saved quarter of
                                      // the all-gather outweighs the expansion kernel -- 30 us at 4 M vertices, 14 at 1 M)
    // form D (gh_overlap_layout): the ranks' un-normalised rows new0 = pos + Fs are all-gathered EARLY, beside the KNN tail
    bool overlap = false;
    bool rows_early = false;          // this step: new0 of the own rows is in its block (the rows may travel right after step_begin)
    float *d_rows_all = nullptr;      // (world, chunk, LD): d_new is block g_rank of it
    float *d_rows_pk = nullptr;       // (world, chunk, D): the same without pad columns -- what travels when D < LD -- or null
    double *d_stats_all = nullptr;    // (world, stats_block) doubles: per rank its statistics rows, then its PATCH LIST -- two int32 counters used by alternate iterations (16 bytes
                                      // reserved) and patch_cap records (row as int32 bits, LD floats): the own rows the intersection phase touched, as
                                      // finished by their owner, pos + (Fs + Fi); d_stats is block g_rank of it
    int64_t stats_block = 0;          // doubles per rank in d_stats_all
    int64_t patch_cap = 0;            // records per rank: min(4 S k, chunk)
    int32_t *d_qexact = nullptr;  // [0] = count, [1..] = queries outside the f16 range (scanned exactly)
    uint64_t *d_cand = nullptr;   // (S, GH_CAND_CAP)
    int32_t *d_cnt = nullptr;     // (S * GH_CNT_STRIDE) one counter per 128-byte line
    int32_t *d_ovf = nullptr;     // (S)
    int32_t *d_sel_redo = nullptr;// (S) queries knn_select_wave_kernel left to the workgroup form (zero between launches)
    int32_t *d_tq_count = nullptr, *d_tq_base = nullptr, *d_tq_touched = nullptr;   // (S), (S), (S, 4 k): per-query runs of the touched list (S >= 2048)
    int32_t *d_dbg_cnt = nullptr; // (2, S) candidate-list lengths seen by the last subset / final select
    uint64_t *d_partial = nullptr;// (S, K) this rank's best keys, ascending
    uint64_t *d_merged = nullptr; // (S, K) keys merged over the ranks (world > 1)
    const uint64_t *d_keys_cur = nullptr;  // keys the intersection phase reads: d_partial or d_merged

    // GH_DIST_CDIST (cdist.hip): the reference's cdist + topk values and tie order
    bool cdist = false;
    bool cd_part = false;             // ... on a row partition: d_partial holds (S, K + 2) per-rank records, the rows are decided at the merge (cdist.hip)
    bool cd_all_ties = false;         // gh_set_cdist_replay: the loop lists every tie too (rows column for column), not only those that can change a force
    int Ksel = 0;                     // keys the candidate selection extracts: K, or K + 1 with cdist (boundary ties)
    int32_t *d_rare = nullptr;        // [1..S] = the listed queries: partial_sort's heap is replayed for them
    int32_t *d_cd_rows = nullptr;     // (2, S) per listed slot: prefix length P (ids below it are valued), tail length
    float *d_cd_vbuf = nullptr;       // (cd_R, cd_nchunks * 64) cdist values of those queries against the edges below P
    float *d_cd_cmin = nullptr;       // (cd_R, cd_nchunks) minimum of each chunk of 64 edge ids
    int32_t *d_cd_stat = nullptr;     // two sets (used alternately) of [0] listed rows, [1] rows whose tie order ATen leaves to std::nth_element (not reproduced), [2] the longest prefix
    int cd_R = 0, cd_nchunks = 0, cd_set = 0;   // cd_set: the counter set the NEXT search uses

    // grid KNN (grid.hip; GH_KNN_GRID)
    int grid_G = 0, grid_bits = 0;
    int64_t grid_cells = 0;
    size_t grid_temp_bytes = 0;
    uint32_t *d_grid_u32 = nullptr;   // keys, rows, sorted keys, sorted rows, cell starts, sorted edge ids
    void *d_grid_smid = nullptr;      // (own_count) float4 midpoints in cell order
    void *d_grid_temp = nullptr;      // radix sort scratch

    gh_ivf *ivf = nullptr;            // GH_KNN_IVF (ivf.hip)

    // normalisation
    double *d_blockstats = nullptr; // (nblocks, 2, LD)
    int nblocks_update = 0;
    double *d_stats = nullptr;      // (2 + 2*gh_fix_blocks(LD), LD): sum, sum of squares, then correction row pairs;
                                    // summed elementwise over the ranks by the caller when partitioned

    // timing
    // thresholds inside the fused launch (tau_core.h)
    bool tau_embedded = false;
    unsigned *d_tau_flag = nullptr;      // queries published so far by the current fused launch
    // iterations replayed from a hipGraph (api.hip gh_run): one captured iteration, the iteration number on the device
    uint64_t *d_iter = nullptr;
    bool graph_capturing = false;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    int32_t *d_wait_failed = nullptr;    // a consumer gave up waiting: reported by gh_sync / gh_get_positions

#define GH_STAMP_EXTRA 8192
    unsigned long long *d_stamps = nullptr;   // GRAPHEM_HIP_STAMPS: (n_vblocks, 8) wall-clock stamps of the last fused launch
    bool timing = false;
    std::vector<gh_timer_slot> timers;
};

// RAII-free helper: records start/stop events around a launch when timing is on.
struct gh_scope {
    gh_engine *h;
    int slot = -1;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t stream = nullptr;
    gh_scope(gh_engine *h_, const char *name, hipStream_t on = nullptr /* null: the engine's stream */);
    ~gh_scope();
};

// api.hip / comm.hip
extern "C" float *gh_rows_all_device(gh_handle h);
extern "C" int32_t gh_rows_all_row_floats(gh_handle h);
extern "C" int32_t gh_step_rows_early(gh_handle h);
extern "C" gh_status gh_step_finish_overlap(gh_handle h);
gh_status gh_upload_sample_stream(gh_engine *h, int32_t iters, const int32_t *sample_stream, const int32_t **d_ids);
gh_status gh_step_begin_device_ids(gh_engine *h, const int32_t *dev_ids);
void gh_comm_free(gh_engine *h);
// f64.hip
void gh_set_create_error(const std::string &msg);   // (api.hip) message gh_last_error(NULL) returns
void gh_f64_free(gh_engine *h);
gh_status gh_f64_set_positions_f32(gh_engine *h, const float *pos);
gh_status gh_f64_get_positions_f32(gh_engine *h, float *pos);
gh_status gh_f64_step(gh_engine *h, const int32_t *sampled);
gh_status gh_f64_run(gh_engine *h, int32_t iters, const int32_t *sample_stream);
gh_status gh_f64_knn_midpoints(gh_engine *h, const int32_t *sampled, int32_t *knn);
// knn.hip
gh_status gh_knn_local(gh_engine *h, bool fuse_intersect);  // d_sampled, d_mid -> d_partial (unfused)
bool gh_knn_scan_path(const gh_engine *h);
struct gh_setup_args;
gh_setup_args gh_make_setup_args(gh_engine *h, int mode, int32_t *sampled, uint64_t iter);  // setup_core.h
unsigned gh_setup_blocks(const gh_setup_args &a);
int64_t gh_gmin_floats(const gh_engine *h);   // size of d_gmin
gh_status gh_knn_prepare(gh_engine *h);
gh_status gh_knn_thresholds(gh_engine *h, int64_t groups = 0);   // groups > 0: d_gmin holds (S, groups) minima written by the caller (ivf.hip)
struct gh_tau_args;
gh_tau_args gh_make_tau_args(gh_engine *h);  // tau_core.h
gh_status gh_knn_finish(gh_engine *h, bool have_mid, bool fuse_intersect);
gh_status gh_knn_points_device(hipStream_t stream, const float *d_q, int64_t nq, const float *d_ref, int64_t nref,
                               int D, int K, uint64_t *d_keys, std::string *err);
// cdist.hip
gh_status gh_cdist_alloc(gh_engine *h);
gh_status gh_knn_merge_cdist(gh_engine *h, const uint64_t *gathered, int world);   // row partitions: the ranks' (S, K + 2) records -> d_merged, the reference's rows
gh_status gh_knn_finish_cdist(gh_engine *h, bool all_rows, bool fuse_intersect);   // candidate lists (or nothing) -> d_partial, the reference's rows
// grid.hip
bool gh_grid_path(const gh_engine *h);
gh_status gh_grid_alloc(gh_engine *h);
gh_status gh_grid_search(gh_engine *h);            // d_mid + tau -> candidate lists
// ivf.hip
bool gh_ivf_path(const gh_engine *h);
gh_status gh_ivf_alloc(gh_engine *h);
void gh_ivf_free(gh_engine *h);
gh_status gh_ivf_search(gh_engine *h);             // d_mid + query records -> tau and candidate lists (probed lists only)
// fused.hip
gh_status gh_radial_topk_device(gh_engine *h, int K, uint64_t *d_part, int nparts, int32_t *d_ids);
int gh_fused_mfma_kb(const gh_engine *h);          // operand rows the thresholds must write: 0 = split-f16 MFMA form (D <= 3), -1 = none
bool gh_fused_uses_mfma(const gh_engine *h);       // the fused kernel's pre-filter runs on the matrix pipe
int gh_fused_tile(const gh_engine *h);              // edges per fused workgroup
gh_status gh_launch_spring_scan(gh_engine *h);             // d_Fs + final-level candidates in one kernel
gh_status gh_knn_merge(gh_engine *h, const uint64_t *gathered, int world);  // -> d_keys_cur
// forces.hip
gh_status gh_launch_intersect(gh_engine *h);               // d_sampled, d_keys_cur -> d_acc/d_touched
gh_status gh_launch_inter_cleanup(gh_engine *h);
gh_status gh_launch_spring_mid(gh_engine *h);              // -> d_Fs, d_mid
gh_status gh_launch_mid_only(gh_engine *h);                // -> d_mid
gh_status gh_launch_integrate(gh_engine *h);               // d_Fs, d_acc -> d_new, d_stats
gh_status gh_launch_spring_only(gh_engine *h, float *d_F); // F (n, LD), own rows
gh_status gh_launch_inter_to_dense(gh_engine *h, float *d_F);
gh_status gh_launch_integrate_given(gh_engine *h, const float *d_Fs, const float *d_Fi);
gh_status gh_launch_normalise(gh_engine *h, bool with_cleanup, bool presetup = false, int next_mode = 0,
                              int32_t *next_ids = nullptr);
gh_status gh_launch_normalise_gathered(gh_engine *h, int next_mode = -1);
gh_status gh_launch_normalise_own(gh_engine *h, const double *stats_all, int world);   // form C: own rows from every rank's statistics
gh_status gh_launch_unpack_rows(gh_engine *h);   // form C: the gathered packed blocks of the OTHER ranks -> their rows of d_pos
struct gh_long_args;
gh_long_args gh_make_long_args(const gh_engine *h, bool coop_mid = false);   // common.h; coop_mid: fused kernels
gh_status gh_launch_spring_long(gh_engine *h, float *outF, int64_t f_row0);  // spring forces of the hub rows  // gathered slots of every rank -> all n rows of d_pos
inline int32_t *gh_patch_count(gh_engine *h) { return reinterpret_cast<int32_t *>(h->d_stats + (size_t)(2 + 2 * gh_fix_blocks(h->LD)) * h->LD); }
inline float *gh_patch_records(gh_engine *h) { return reinterpret_cast<float *>(gh_patch_count(h) + 4); }
gh_status gh_launch_new0(gh_engine *h);                            // form D without the fused kernel: d_new = pos + Fs of the own rows
gh_status gh_launch_pack_rows(gh_engine *h, hipStream_t stream);   // form D: own block of new0 -> its packed slot (on the given stream)
gh_status gh_launch_patch_rows(gh_engine *h);                      // form D: touched rows of the gathered array += Fi; accumulators zeroed
gh_status gh_launch_pad(gh_engine *h, const float *d_src_nD, float *d_dst_nLD);
gh_status gh_launch_unpad(gh_engine *h, const float *d_src_nLD, float *d_dst_nD);
gh_status gh_launch_sample(gh_engine *h);                  // device sampler -> d_sampled
gh_status gh_launch_arange(gh_engine *h);
gh_status gh_ensure_sample(gh_engine *h);                  // run a pending stand-alone sampler launch

#define GH_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return GH_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define GH_TRY_ST(x)                                                                        \
    do {                                                                                    \
        gh_status st_ = (x);                                                                \
        if (st_ != GH_OK) return st_;                                                       \
    } while (0)

#define GH_LAUNCH_CHECK()                                                                   \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            h->err = std::string("kernel launch: ") + hipGetErrorString(e_);                \
            return GH_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

/*
 * graphem_hip.h -- C ABI of the MI355X (gfx950) force-directed layout engine.
 *
 * This is the drop-in boundary for ONE path of sashakolpakov/graphem-rapids: the
 * per-iteration loop behind GraphEmbedderPyTorch.run_layout().  The reference
 * has no FFI of its own (it is 100 % Python); each entry point
 * below cites the reference method (graphem_rapids/backends/embedder_pytorch.py,
 * "pt.py") whose work it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; host pointers unless a name says "device";
 *   - every function returns a gh_status; gh_last_error() gives the message;
 *   - all work is enqueued on the handle's HIP streams; only the functions
 *     documented as blocking (copies to host, gh_sync) wait for the GPU;
 *   - a handle is not thread-safe; distinct handles are independent;
 *   - positions cross the boundary as (n, D) row-major float32, edges as
 *     (E, 2) row-major int32 with u < v in the reference's edge order
 *     (pt.py:220-245), edge ids are positions in that list.
 */
#ifndef GRAPHEM_HIP_H
#define GRAPHEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gh_engine *gh_handle;

typedef enum {
    GH_OK = 0,
    GH_ERR_INVALID = 1,     /* bad argument            -> ValueError in the Python mirror   */
    GH_ERR_RUNTIME = 2,     /* runtime failure         -> RuntimeError                      */
    GH_ERR_K_TOO_LARGE = 3, /* n_neighbors + 1 > E     -> RuntimeError (torch.topk, pt.py:583) */
    GH_ERR_HIP = 4,         /* HIP API / kernel error  -> RuntimeError                      */
    GH_ERR_NOMEM = 5        /* allocation failure      -> MemoryError                       */
} gh_status;

/* Constructor arguments of the reference that the loop reads (pt.py:51-67, 130-156). */
typedef struct {
    float L_min;          /* pt.py:57  */
    float k_attr;         /* pt.py:58  */
    float k_inter;        /* pt.py:59  */
    int32_t n_neighbors;  /* pt.py:60, k */
    int32_t sample_size;  /* pt.py:61, already min(sample_size, E) as pt.py:156 */
    uint64_t seed;        /* seed of the on-device sampler (used when no sample ids are passed) */
    int32_t reorder;      /* internal vertex order: GH_REORDER_AUTO / _OFF / _BFS (no reference counterpart) */
    int32_t knn_method;   /* GH_KNN_AUTO / _SCAN / _GRID / _IVF: how the KNN of the sampled midpoints is searched */
    int32_t knn_distance; /* GH_DIST_EXACT / GH_DIST_CDIST: which distance ranks the neighbours (below) */
    int32_t ivf_lists;    /* GH_KNN_IVF: inverted lists (0: about sqrt(own edges) / 2, at most 512 up to 4 components and 1024 above;
                             always a multiple of 64 in 64 ... 2048) */
    int32_t ivf_probes;   /* GH_KNN_IVF: lists a query searches (0: lists / 16 for more than 8 components, / 32 for 5 - 8, / 64 below;
                             < 0: exact mode, every list that can hold a neighbour) */
} gh_params;

/* Distance the KNN ranks on (pt.py:543-593).
 *   GH_DIST_EXACT  squared Euclidean distance in exact-difference form, sum_d (q_d - m_d)^2 as an fma chain in
 *                  coordinate order, ties on the smaller edge id: what the reference's KeOps path computes
 *                  (pt.py:527-534) and the fp64-correct ranking.  The speed mode; 0, so a zeroed struct selects it.
 *   GH_DIST_CDIST  the value and order the reference's PyTorch-CPU backend gets from torch.cdist + torch.topk
 *                  (pt.py:580-583): ATen's matmul form  acc = fma(-2 q_d, m_d, acc) ...; acc += |q|^2; acc += |m|^2;
 *                  sqrt(max(acc, 0))  with |x|^2 = sum of rounded squares, left to right, and equal values in the
 *                  order std::partial_sort leaves them (ATen/native/TopKImpl.h; K * 64 <= E).  The sampled edge is not
 *                  special-cased: column 0 is whatever ranks first (pt.py:417-421).  Neighbour ids then equal the
 *                  reference's row by row at every size.  The candidates of the filtered scan are re-valued with that
 *                  formula; for the rows whose K + 1 smallest values hold a tie (about 1 in 100 at a million vertices)
 *                  partial_sort's heap is replayed over the ids below a prefix bound (about E / stride of them) and the
 *                  id-sorted rest of the candidate list (csrc/cdist.hip): two more launches per iteration, 220 against
 *                  165 us at a million vertices.  Works on the candidates of GH_KNN_SCAN only: knn_method = GH_KNN_AUTO
 *                  takes the scan whatever the sample size, an explicit GH_KNN_GRID / GH_KNN_IVF is refused
 *                  (GH_ERR_INVALID).  On a row partition (gh_partition given) a rank sends, instead of its K best keys, a
 *                  record of K + 2 words per query (gh_knn_partial_cols): its K + 1 best cdist keys and 1 where it could
 *                  prove them its K + 1 best; gh_step_merge decides the queries whose merged K + 1 smallest values are
 *                  pairwise different and replays partial_sort's heap for the others -- every rank holds all positions
 *                  and the whole edge list, so every rank gets the same rows and no further collective is needed; the
 *                  replay covers the ids below a bound taken from the gathered keys (about E / world) and the gathered
 *                  keys behind it, or all edges where a rank could not prove its keys. */
#define GH_DIST_EXACT 0
#define GH_DIST_CDIST 1

/* KNN search (the reference's cdist + topk, pt.py:543-593; its cuVS backend reaches for IVF indexes,
 * embedder_cuvs.py:255-313).  SCAN and GRID return the EXACT k+1 nearest midpoints, identical ids; IVF is approximate.
 *   GH_KNN_SCAN  filtered brute-force scan fused with the spring phase: S * E pre-filter evaluations on the matrix
 *                pipe, hidden under the spring phase's gathers up to a few thousand queries;
 *   GH_KNN_GRID  n_components <= 3: a grid over the midpoints rebuilt every iteration (O(E)), then per query only the
 *                cells its threshold ball touches: sub-quadratic, pays from several thousand queries on.  (With more
 *                components the engine searches with GH_KNN_SCAN instead: a grid over the first three coordinates stays
 *                exact but was measured 50-80x slower than the scan at a million vertices; removed in round 4.)
 *   GH_KNN_IVF   2 <= n_components <= 16, GH_DIST_EXACT (a partitioned engine indexes the edges it owns): an inverted-file index rebuilt every iteration,
 *                the counterpart of the cuVS backend's IVF-Flat (embedder_cuvs.py:255-313, 384-430).  ivf_lists centroids
 *                (midpoints of evenly spaced edges), every midpoint filed under its nearest one (f16 scores on the matrix
 *                pipe), a query searches the ivf_probes lists whose centroids are nearest and gets the exact k+1 nearest
 *                AMONG THEIR MEMBERS (exact-difference distances, ties on the smaller id).  APPROXIMATE: a neighbour filed
 *                under an unprobed list is missed.  Measured on a million vertices / 4 M edges, 4096 queries, defaults
 *                (1024 lists; 64 / 32 probes): recall 0.992 in 16 dimensions (1.7 ms per iteration against 2.6 for SCAN), 1.0000
 *                in 6 (1.0 against 1.5); pays from a few thousand queries on; AUTO never takes the approximate mode.
 *                ivf_probes < 0 selects its EXACT mode: a query probes every list that can hold one of its k+1 nearest -- with
 *                tau >= the (k+1)-th smallest distance, the lists whose centroid lies within sqrt(tau) + min(list radius,
 *                sqrt(tau) + distance to the query's nearest centroid) -- and the rows are those of SCAN, id for id.  A few
 *                lists per query up to 4 components, a few dozen at 6, most of them beyond 8 (then it costs a scan plus the
 *                index).  rr1m, SCAN / exact IVF us per iteration: D = 3 S = 4096 1063 / 591, 16384 3691 / 766 (GRID 1926);
 *                D = 6 S = 4096 1507 / 915, 16384 5059 / 1401; D = 8 S = 16384 5264 / 2135; D = 16 S = 16384 8827 / 6677.
 *   GH_KNN_AUTO  exact methods only: engines with 2-8 components, GH_DIST_EXACT, >= 262144 (own) edges and thousands of
 *                queries (sample_size >= 4096 up to 4 components, >= 8192 for 5-8) take IVF in its exact mode; else GRID when
 *                n_components <= 3 and sample_size >= 12288; else SCAN. */
#define GH_KNN_AUTO 0
#define GH_KNN_SCAN 1
#define GH_KNN_GRID 2
#define GH_KNN_IVF 3

/* Internal vertex order.  The spring phase gathers the position row of every neighbour; with
 * breadth-first vertex numbers a vertex sits next to its BFS siblings and close to its parent and
 * children, so more of those gathers hit the L2; within blocks of 16384 such numbers the rows are put in
 * order of falling degree (the lanes of a wave walk their lists in lock-step).  Purely internal: vertex arrays cross the API in
 * the caller's order, edge ids are unchanged and every summation keeps the reference's order, so
 * results do not depend on it.  AUTO = BFS when the position array outgrows an L2 (n * row bytes >
 * 3 MB) and the partition (if any) uses GH_EDGES_HASHED.  gh_positions_device() exposes the
 * INTERNAL order; gh_vertex_order() returns the internal row of every vertex. */
#define GH_REORDER_AUTO 0
#define GH_REORDER_OFF 1
#define GH_REORDER_BFS 2

/* Row partition for multi-GPU runs (no reference counterpart).
 * A rank integrates vertices [row_lo, row_hi) and searches the edges it OWNS in the
 * KNN phase; it still holds all n positions and all E edges.  Every edge must be owned
 * by exactly one rank of the job:
 *   GH_EDGES_RANGE   the rank owns edges [edge_lo, edge_hi) (the caller cuts the ranges);
 *   GH_EDGES_HASHED  each edge belongs to one of its endpoints, chosen by a fixed hash of
 *                    the edge id, and so to the rank whose rows hold that endpoint: every
 *                    rank owns ~E/world edges whatever the vertex numbering.  The row
 *                    ranges of the ranks must tile [0, n); edge_lo/edge_hi are ignored.
 * Single GPU: row_lo = 0, row_hi = n, edge_lo = 0, edge_hi = E, GH_EDGES_RANGE. */
#define GH_EDGES_RANGE 0
#define GH_EDGES_HASHED 1
typedef struct {
    int64_t row_lo, row_hi;
    int64_t edge_lo, edge_hi;
    int32_t edge_rule;
} gh_partition;

/* ---- lifetime -------------------------------------------------------------- */

/* Replaces the device-side part of GraphEmbedderPyTorch.__init__ (pt.py:150-159):
 * uploads the edge list, builds the pull lists for the spring phase, allocates all
 * state.  part may be NULL (whole graph).  Positions start at zero. */
gh_status gh_create(gh_handle *out, int device_id, int64_t n, int32_t n_components, int64_t n_edges,
                    const int32_t *edges, const gh_params *params, const gh_partition *part);
void gh_destroy(gh_handle h);
/* Message of the last failure on h (h may be NULL for a failed gh_create). */
const char *gh_last_error(gh_handle h);

/* ---- float64 engine (pt.py:56: the reference computes in the dtype it is created with; tests/test_pytorch_backend.py:
 *      169-181) --------------------------------------------------------------------------------------------------
 * gh_create_f64 makes an engine whose every phase runs in double: positions (n, D) doubles, spring forces, the exact
 * KNN ranked on double distances (ties on the smaller id), intersection forces, means and unbiased std in double.
 * Plain kernels, one per phase (csrc/f64.hip): what this mode is for is the reference's fp64 numbers, not speed.
 * On such a handle: gh_set_positions / gh_get_positions convert from / to float32, the *_f64 accessors below move
 * doubles; gh_step, gh_run, gh_sync, gh_knn_midpoints, gh_destroy, gh_last_error work as on a float32 engine; every
 * other entry point (partitions, collectives, float32 per-phase calls, instrumentation of the fused kernels) returns
 * GH_ERR_INVALID.  gh_params.reorder / knn_method / knn_distance are ignored.  Up to 32 components, 255 neighbours. */
gh_status gh_create_f64(gh_handle *out, int device_id, int64_t n, int32_t n_components, int64_t n_edges,
                        const int32_t *edges, const gh_params *params /* its three float constants are NOT used: */,
                        double L_min, double k_attr, double k_inter /* pt.py:57-59 as the doubles Python holds */);
gh_status gh_set_positions_f64(gh_handle h, const double *pos /* (n, D) host */);
gh_status gh_get_positions_f64(gh_handle h, double *pos /* (n, D) host, blocking */);
double *gh_positions_device_f64(gh_handle h);                 /* (n, D) doubles, caller's vertex order, no padding */
gh_status gh_spring_forces_f64(gh_handle h, double *F);      /* _compute_spring_forces (pt.py:595-636) in double */
gh_status gh_intersection_forces_f64(gh_handle h, const int32_t *sampled, const int32_t *knn, double *F);   /* pt.py:638-774 */

/* ---- positions accessors: `positions` property / setter, get_positions
 *      (pt.py:324-335, 835-844) ------------------------------------------------ */
gh_status gh_set_positions(gh_handle h, const float *pos /* (n, D) host */);
gh_status gh_get_positions(gh_handle h, float *pos /* (n, D) host, blocking */);
/* Device view for callers that keep data on the GPU (RCCL all-gather, torch tensors):
 * (n, ld) float32 rows, ld = gh_row_stride(h) >= D floats, columns >= D are zero. */
float *gh_positions_device(gh_handle h);
/* order[v] = row of vertex v in the device position array (identity without reordering). */
gh_status gh_vertex_order(gh_handle h, int32_t *order);
/* Device copy of the positions in the CALLER's vertex order, (n, n_components) floats without
 * padding; valid until the next call on this handle. */
const float *gh_positions_unpadded_device(gh_handle h);
int32_t gh_row_stride(gh_handle h);

/* ---- the loop: update_positions / run_layout (pt.py:776-806, 808-833) ------- */

/* One iteration.  sampled: the S edge ids of this iteration (what
 * torch.randperm(E)[:S] returned, pt.py:409), host pointer; NULL = draw them on the
 * device (ignored when S >= E, where the reference uses arange(E), pt.py:412). */
gh_status gh_step(gh_handle h, const int32_t *sampled);
/* iters iterations without host synchronisation.  sample_stream: (iters, S) host ids
 * or NULL for the device sampler. */
gh_status gh_run(gh_handle h, int32_t iters, const int32_t *sample_stream);
/* The reference's own sampler beside the loop (pt.py:403-413: `torch.randperm(n_edges)[:sample_size]` on the global CPU
 * generator, once per iteration).  ATen's CPU randperm is a forward Fisher-Yates over an mt19937 (z = random() % (n - i);
 * swap(r[i], r[i + z])): entries [:S] are final after S draws, the other n - 1 - S draws only move the generator on.
 *   gh_torch_randperm_prefix  pure host code, no GPU: rng_state = the 5056 bytes of torch.get_rng_state() (in / out);
 *                             writes `iters` rows of S ids = what `iters` calls of torch.randperm(n)[:S] return, and
 *                             leaves in rng_state the state those calls leave (hand it to torch.set_rng_state).
 *                             O(S) swaps per row in a small table + an AVX2 / AVX-512 twist over the skipped words
 *                             (gh_torch_randperm_isa names the one chosen on this host).
 *   gh_run_torch_sampled      gh_run with those ids: a host thread draws them (up to 256 iterations ahead) while the calling
 *                             thread uploads what is drawn and enqueues; rng_state in / out as above (written on success only).
 *                             S >= E consumes nothing, like pt.py:412.  Whole-graph engines (float32 or float64). */
gh_status gh_torch_randperm_prefix(uint8_t *rng_state, int64_t state_bytes, int64_t n, int64_t sample_size, int32_t iters,
                                   int32_t *ids /* (iters, sample_size) host */);
const char *gh_torch_randperm_isa(void);
gh_status gh_run_torch_sampled(gh_handle h, int32_t iters, uint8_t *rng_state, int64_t state_bytes);
/* Host milliseconds of the last gh_run_torch_sampled on h: [0] the producer thread drawing, [1] the calling thread
 * waiting for a pinned upload slot to come back from the GPU, [2] the calling thread waiting for ids, [3] the whole call
 * (enqueue only). */
gh_status gh_sampler_stats(gh_handle h, double *out4);
/* Blocks until everything enqueued on h has finished. */
gh_status gh_sync(gh_handle h);

/* ---- per-phase entry points (tests, profiling).  Each runs on the CURRENT
 *      positions, leaves them unchanged, blocks, and writes host buffers. -------- */

/* _compute_spring_forces (pt.py:595-636): F (n, D). */
gh_status gh_spring_forces(gh_handle h, float *F);
/* _locate_knn_midpoints + _compute_knn_chunked/_compute_knn_torch (pt.py:381-424,
 * 426-483, 543-593): knn (S, k) edge ids, column 0 already dropped (pt.py:421). */
gh_status gh_knn_midpoints(gh_handle h, const int32_t *sampled, int32_t *knn);
/* _compute_intersection_forces + _check_line_intersections (pt.py:638-774): F (n, D). */
gh_status gh_intersection_forces(gh_handle h, const int32_t *sampled, const int32_t *knn, float *F);
/* combine + normalise (pt.py:796-804): out = normalise(pos + (Fs + Fi)); (n, D) each. */
gh_status gh_integrate_normalise(gh_handle h, const float *Fs, const float *Fi, float *out);

/* ---- multi-GPU hooks.  With a gh_partition a step is split so the
 *      caller can run its collectives (RCCL through torch.distributed) between the
 *      parts; all buffers are device pointers owned by the handle. ---------------- */

/* Run all later work of h on the caller's HIP stream (e.g. torch's current stream, so the
 * engine's kernels are stream-ordered with RCCL collectives).  hip_stream may be NULL, which
 * is HIP's default (null) stream -- torch's default stream.  use_own != 0 switches back to the
 * stream gh_create made and ignores hip_stream. */
gh_status gh_set_stream(gh_handle h, void *hip_stream, int32_t use_own);
/* Rows allocated in the device position array: >= n, padded so that world equal chunks of
 * ceil(n / world) rows fit for an in-place all-gather (rows >= n are zero and never read). */
int64_t gh_positions_rows_allocated(gh_handle h);

/* Part 1: spring pull for own rows, KNN scan of own edges.  Afterwards
 * gh_knn_partial_device() holds this rank's S x (k+1) best (dist2, id) keys. */
gh_status gh_step_begin(gh_handle h, const int32_t *sampled);
uint64_t *gh_knn_partial_device(gh_handle h);      /* (S, gh_knn_partial_cols) uint64: k+1 keys, ascending */
/* 64-bit words per query of that record: k + 1; a GH_DIST_CDIST engine on a partition: k + 3 (its k + 2 best cdist keys,
 * then 1 where they are provably its k + 2 best).  gathered of gh_step_merge is (world, S, gh_knn_partial_cols). */
int32_t gh_knn_partial_cols(gh_handle h);
/* After gh_step_merge: the (S, k+1) keys of the global KNN the intersection phase read (column 0 included; the id of a
 * neighbour is the low 32 bits of its key).  Device pointer owned by the handle; NULL before the first merge. */
const uint64_t *gh_knn_merged_device(gh_handle h);
/* Part 2: gathered = (world, S, k+1) keys from all ranks (device pointer; may alias a
 * caller buffer).  Merges them, computes intersection forces, integrates own rows and
 * leaves this rank's column sums in gh_stats_partial_device(). */
gh_status gh_step_merge(gh_handle h, const uint64_t *gathered, int32_t world);
double *gh_stats_partial_device(gh_handle h);      /* (gh_stats_rows, ld) doubles, to be summed elementwise over ranks */
int32_t gh_stats_rows(gh_handle h);
/* Part 3, form A: after the caller all-reduced (SUM) the whole statistics buffer: normalise own rows in place in the
 * full position array; the caller then all-gathers the row blocks (two collectives). */
gh_status gh_step_finish(gh_handle h);

/* Part 3, form B (ONE collective): the rank's un-normalised new rows and its statistics live side by side in
 * one slot of a gather buffer; the caller all-gathers the slots in place and every rank then normalises ALL n
 * rows from the gathered slots, summing the per-rank statistics in rank order (identical on every rank).
 *   gh_gather_layout(h, world, rank, chunk)  once after gh_create: this rank is `rank` of `world`, rank r owns
 *                                             rows [r*chunk, min(n, (r+1)*chunk)); allocates the buffer
 *   gh_gather_buffer_device(h)               world * gh_gather_slot_bytes(h) bytes; slot r belongs to rank r
 *   gh_step_finish_gathered(h)               after the all-gather: d_pos <- normalised rows of every rank */
gh_status gh_gather_layout(gh_handle h, int32_t world, int32_t rank, int64_t chunk);
void *gh_gather_buffer_device(gh_handle h);
int64_t gh_gather_slot_bytes(gh_handle h);
gh_status gh_step_finish_gathered(gh_handle h);

/* Part 3, form C (the default of the drivers; no pass over all n rows): the rank keeps its un-normalised rows to itself,
 * the caller all-gathers only the ranks' STATISTICS (gh_stats_rows * ld doubles each, rank order), every rank
 * normalises ITS rows into its block of the position array -- per-rank sums added in rank order, so all ranks use the
 * same mean / std bits -- and the caller then all-gathers the finished blocks IN PLACE in gh_positions_device():
 * block r = rows [r*chunk, (r+1)*chunk), chunk * ld floats (gh_positions_rows_allocated() >= world * chunk).
 *   gh_rank_layout(h, world, rank, chunk)   once after gh_create (instead of gh_gather_layout)
 *   gh_step_finish_own(h, stats_all, world) stats_all: (world, gh_stats_rows, ld) doubles, device
 * The position array is complete again only after the caller's all-gather.
 * With fewer components than the row stride (3 of 4, 5..7 of 8, 9..15 of 16) and world > 1 the blocks can travel WITHOUT
 * their pad columns (12 instead of 16 bytes per row at 3 components: a quarter off the largest collective of the iteration):
 * gh_step_finish_own also writes the own block into slot `rank` of gh_rows_packed_device(), a (world, chunk, D) float
 * array; the caller all-gathers THAT in place instead of the position blocks and calls gh_step_unpack_rows, which expands
 * the other ranks' blocks into the position array.  The expansion is a kernel over all n rows (30 us at 4 M vertices, 14 at
 * 1 M, measured), so the packed exchange is in use by default from 2 M vertices on; gh_set_packed_rows switches it on or
 * off after gh_rank_layout.  gh_rows_packed_device() is NULL while it is not in use (D == ld, world == 1, switched off):
 * the caller then all-gathers the position blocks in place. */
gh_status gh_rank_layout(gh_handle h, int32_t world, int32_t rank, int64_t chunk);
/* Part 3, form D (round 5; the default of bench.py --gpus N and of distributed.PartitionedLayout): form B's finish -- every
 * rank normalises ALL n rows from the ranks' gathered un-normalised rows, no collective after the normalisation -- with the
 * big collective moved to the FRONT of the KNN tail.  Part 1 leaves new0 = pos + Fs of the own rows in block `rank` of a
 * (world, chunk, ld) array (the fused spring+scan kernel writes it; a rank too small for that kernel makes it with a launch
 * of its own, so every rank sends at the same point: gh_step_rows_early() is 1 after every gh_step_begin); the caller
 * starts the all-gather of those blocks at once, on a second stream / communicator, and runs select -> all-gather of the
 * keys -> gh_step_merge -> all-gather of the statistics beside it.  Only the <= 4 S k rows the intersection phase touches
 * differ from new0 afterwards: gh_step_merge puts the finished value of every OWN touched row -- pos + (Fs + Fi), the single
 * engine's expression -- into the rank's PATCH LIST, which sits behind its statistics and travels with them.
 * gh_step_finish_overlap (enqueued behind BOTH collectives) writes every rank's patch list over those rows of the gathered
 * array and normalises all n rows into the position array (the next iteration's KNN set-up rides in that launch).  Results:
 * what forms B / C give (the single engine's up to the order in which the ranks' statistics are added).  What travels:
 *   gh_rows_all_device()      (world, chunk, gh_rows_all_row_floats()) floats, block r = rank r's rows: WITHOUT pad columns
 *                             when n_components < ld and world > 1 (row_floats = n_components; the own block is put there
 *                             by gh_step_pack_rows on the stream given -- call it on the side stream before the all-gather;
 *                             a no-op otherwise), else the (world, chunk, ld) array itself;
 *   gh_stats_all_device()     (world, gh_stats_all_block_doubles()) doubles, block r = rank r's statistics rows
 *                             (= gh_stats_partial_device(), gh_stats_rows * ld doubles), then 16 bytes holding its two patch
 *                             counters (int32; iteration t uses counter t & 1, the other is zeroed meanwhile) and
 *                             min(4 S k, chunk) records of (row as int32, ld floats).
 * Up to 16 components. */
gh_status gh_overlap_layout(gh_handle h, int32_t world, int32_t rank, int64_t chunk);
float *gh_rows_all_device(gh_handle h);
int32_t gh_rows_all_row_floats(gh_handle h);
double *gh_stats_all_device(gh_handle h);
int64_t gh_stats_all_block_doubles(gh_handle h);
int32_t gh_step_rows_early(gh_handle h);
gh_status gh_step_pack_rows(gh_handle h, void *hip_stream, int32_t use_engine_stream);
gh_status gh_step_finish_overlap(gh_handle h);
gh_status gh_step_finish_own(gh_handle h, const double *stats_all, int32_t world);
gh_status gh_set_packed_rows(gh_handle h, int32_t on);
float *gh_rows_packed_device(gh_handle h);
gh_status gh_step_unpack_rows(gh_handle h);

/* ---- the whole partitioned run as ONE call (no host language in the loop) ----------------------
 * After gh_create(partition) + gh_rank_layout(world, rank, chunk) (form C: all-gathers of the keys, of the statistics
 * and, in place, of the finished position blocks) or gh_gather_layout (form B: keys, slots) + one of the communicator
 * calls, gh_run_partitioned enqueues `iters` iterations on the handle's stream; only the collective backend may block.
 * Every rank must call it with the same iters and the same sample_stream ((iters, S) host ids, or NULL:
 * each rank's engine then draws identical ids from (seed, iteration)).
 *   gh_comm_unique_id        rank 0: 128 bytes to hand to every rank (ncclGetUniqueId)
 *   gh_comm_init_rccl        RCCL communicator on the engine's device; collectives = ncclAllGather on the engine's
 *                            stream over xGMI.  librccl.so is opened here, not at load time.
 *   gh_loopback_group_create / gh_comm_init_loopback
 *                            `world` engines of one process, one host thread each, exchange by device copies:
 *                            the same loop on a single GPU (where RCCL refuses two ranks on one device)
 * After gh_overlap_layout (form D) the communicator call also makes the side stream and, with RCCL, a second communicator
 * (ncclCommSplit) for the early all-gather of the rows; without one the rows go out after the merge (form B's order).
 * Kernel and collective times appear under gh_timing_get as "allgather_keys" / "allgather_slots" / "allgather_rows" (form
 * D: measured on the side stream) / "allgather_rows_exposed" (form D: how long the engine's stream then still waited). */
typedef struct gh_loop_group gh_loop_group;
int32_t gh_comm_available(void);         /* 1 when librccl.so opens and has every entry point used here (no communicator made) */
gh_status gh_comm_unique_id(void *out128);
gh_status gh_comm_init_rccl(gh_handle h, int32_t world, int32_t rank, const void *unique_id128);
gh_loop_group *gh_loopback_group_create(int32_t world);
void gh_loopback_group_destroy(gh_loop_group *group);
gh_status gh_comm_init_loopback(gh_handle h, gh_loop_group *group, int32_t rank);
gh_status gh_comm_destroy(gh_handle h);
gh_status gh_run_partitioned(gh_handle h, int32_t iters, const int32_t *sample_stream);
const char *gh_comm_last_error(void);   /* message of a failed gh_comm_unique_id */

/* ---- instrumentation --------------------------------------------------------- */

/* Names and accumulated GPU milliseconds of the kernels launched since the last
 * reset, measured with HIP events on the launching stream when timing is enabled. */
gh_status gh_timing_enable(gh_handle h, int32_t on);
gh_status gh_timing_reset(gh_handle h);
int32_t gh_timing_count(gh_handle h);
gh_status gh_timing_get(gh_handle h, int32_t i, const char **name, double *total_ms, int64_t *launches);

/* Environment variables the library reads -- diagnostics only, all of them at gh_create, none needed in production:
 *   GRAPHEM_HIP_TAU_SEPARATE=1|0  thresholds always in a launch of their own / always by the first workgroups of the fused
 *                                 launch (default: inside for <= 2048 fused workgroups);
 *   GRAPHEM_HIP_NO_PRESETUP=1     the next iteration's KNN set-up as a launch of its own instead of inside the normalise
 *                                 launch (the per-query flags of gh_knn_last_counts then survive a step);
 *   GRAPHEM_HIP_REORDER=1|2       overrides gh_params.reorder (1 off, 2 breadth-first);
 *   GRAPHEM_HIP_STAMPS=1          allocates the stamp buffer gh_debug_stamps reads;
 *   GRAPHEM_HIP_GRAPH=1           gh_run replays iterations 2.. from a hipGraph of ten iterations (measured slower than
 *                                 enqueuing: 170.1 against 167.9 us per iteration at a million vertices). */

/* Diagnostic runs only (environment GRAPHEM_HIP_STAMPS set at gh_create): 8 wall-clock stamps (100 MHz) per workgroup
 * of the last fused spring+scan launch (tools/stamp_probe.py).  Blocking. */
gh_status gh_debug_stamps(gh_handle h, unsigned long long *out, int64_t count);

/* Diagnostics of the last KNN search: per query, the candidate-list length the last subset
 * level and the final level saw, and whether the exact fallback had to redo the query
 * (any of the three (S,) host pointers may be NULL).  Blocking. */
gh_status gh_knn_last_counts(gh_handle h, int32_t *subset_counts, int32_t *final_counts, int32_t *overflow);
/* GH_DIST_CDIST engines, last KNN search: rows whose partial_sort heap was replayed (a tie among their K + 1 smallest
 * cdist values, or a candidate list that could not be proven complete), and rows NOT reproduced.  Tiny graphs
 * (K * 64 > E), where ATen ranks with std::nth_element + std::sort, get libstdc++'s introselect and introsort replayed on
 * single-engine runs (a row partition's merge replays the heap for them: equal values in (value, id) order, counted)
 * (E <= 8000: tie order reproduced; the second count then only holds rows whose introselect depth limit ran out, which
 * adversarial inputs alone do); with K * 64 > E > 8000 (more than 125 neighbours on a graph of a few thousand edges)
 * equal values come out in (value, id) order and a row with a tie is counted.  Either pointer may be NULL.  Blocking. */
gh_status gh_knn_cdist_stats(gh_handle h, int32_t *full_pass_rows, int32_t *unresolved_tie_rows);
/* GH_DIST_CDIST engines: which ties the LOOP (gh_step / gh_run / gh_run_torch_sampled) replays.  The intersection phase
 * reads a neighbour row as a set of pairs (column 0 dropped, pt.py:417-421; the other k ids paired with the sampled edge,
 * pt.py:668-699), so only a tie between values 0 and 1 (which id is dropped) or between values k and k + 1 (which id is a
 * member) can change a force; a tie strictly inside permutes columns of the same set.  all_ties = 0 (default): the loop
 * replays partial_sort's heap for those two kinds of tie only -- same positions, bit for bit, as with all_ties = 1 (every
 * tie, rows column for column: what gh_knn_midpoints always does, and what row-partitioned engines always do). */
gh_status gh_set_cdist_replay(gh_handle h, int32_t all_ties);
/* GH_KNN_IVF engines: the number of inverted lists and of lists probed per query the engine settled on (0, 0 when the
 * engine searches another way).  Either pointer may be NULL. */
gh_status gh_knn_ivf_config(gh_handle h, int32_t *lists, int32_t *probes);
/* Members of every inverted list after the last search (diagnostic; count must equal the number of lists).  Blocking. */
gh_status gh_knn_ivf_list_sizes(gh_handle h, int32_t *sizes, int32_t count);

/* Plain point-set KNN without a handle: the reference's _compute_knn_chunked /
 * _compute_knn_torch (pt.py:426-483, 543-593).  q (nq, D), ref (nref, D) host float32 row-major;
 * out (nq, k) int64: ids of the k nearest reference rows, ascending distance (exact squared
 * Euclidean distance, ties on the smaller id).  k > nref -> GH_ERR_K_TOO_LARGE like torch.topk.
 * On failure gh_last_error(NULL) has the message.  Blocking. */
gh_status gh_knn_points(int device_id, const float *q, int64_t nq, const float *ref, int64_t nref,
                        int32_t n_components, int32_t k, int64_t *out);

/* ---- caller-side reduction on the device (influence.py:28-37) ----
 * The k vertices farthest from the origin, farthest first: np.argsort(-np.linalg.norm(positions,
 * axis=1))[:k] of the reference's graphem_seed_selection without the (n, D) device-to-host copy.
 * Radial distance = sqrtf of the sum of squares accumulated in coordinate order with separate
 * multiply and add: numpy's own order for rows shorter than 8 (longer rows it sums pairwise, so the
 * last bit -- and with it only the order of near-ties -- may differ); equal distances -> smaller
 * vertex id first (numpy's unstable sort leaves that order open).  ids: k host int32.  1 <= k <= min(n, 64).
 * Blocking. */
gh_status gh_radial_topk(gh_handle h, int32_t k, int32_t *ids);

/* ---- spectral initialisation (_compute_laplacian_embedding, pt.py:337-379) ----
 * y = (2I - L) x for the normalised Laplacian L of a symmetric, unweighted graph in CSR form:
 * y_i = x_i + s_i * sum_j s_j x_j over the neighbours j of i (s = degree^-1/2), y_i = 2 x_i for an
 * isolated vertex.  The operator of the Lanczos iteration in graphem-rapids_amd/spectral.py.
 * All pointers are DEVICE pointers; fp64; asynchronous on hip_stream (NULL = default stream).
 * gh_spectral_last_error() has the message of a failed call. */
gh_status gh_spmv_symnorm(void *hip_stream, int64_t n, const int64_t *indptr, const int32_t *indices,
                          const double *inv_sqrt_deg, const double *x, double *y);
/* Steps k .. m-1 of a Lanczos sweep on B (thick-restart form: the first k columns of the projected matrix are given):
 * matvec, classical Gram-Schmidt twice against the nl locked vectors and the basis so far, coefficients into column j of
 * Td, beta[j] = |w|, hmax[j] = max |coefficient|, next basis vector -- seven launches per step on hip_stream, no host
 * synchronisation.  V (nl + m + 1, n), Td (m, m) row-major, work: n + 2 (nl + m + 1) + ceil(n / 512) (nl + m + 1)
 * doubles, beta / hmax (m); all DEVICE pointers; nl + m + 1 <= 256. */
gh_status gh_trlan_sweep(void *hip_stream, int64_t n, const int64_t *indptr, const int32_t *indices,
                         const double *inv_sqrt_deg, double *V, int32_t nl, int32_t m, int32_t k, double *Td,
                         double *work, double *beta, double *hmax);
const char *gh_spectral_last_error(void);

/* Self-test (no reference counterpart): the spring phase computes sqrt and its D divisions by one distance with leaner
 * instruction sequences than the compiler's general ones; this compares them bit for bit with sqrtf and '/' on `samples`
 * pseudo-random operand sets over and beyond their fast domain.  Both counts must come back 0.  Blocking. */
gh_status gh_selftest_arith(int device_id, uint64_t seed, int64_t samples, int64_t *bad_sqrt, int64_t *bad_div);

/* Device / build facts for the host mirror's get_backend_info(). */
int32_t gh_device_count(void);
const char *gh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHEM_HIP_H */

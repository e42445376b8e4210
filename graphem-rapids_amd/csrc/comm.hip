// The partitioned layout loop without the host language in it:
// gh_run_partitioned enqueues, per iteration, part 1 (spring pull + KNN scan of the own rows / edges), the
// all-gather of the ranks' S x (k+1) keys, part 2 (merge, intersection forces, integrate own rows) and then
//   form C (gh_rank_layout):   all-gather of the ranks' column statistics (a few hundred bytes), normalise the OWN rows,
//                              in-place all-gather of the finished position blocks;
//   form B (gh_gather_layout): in-place all-gather of the ranks' slots (new rows + statistics), normalise all n rows --
// kernels and collectives on ONE stream, no host synchronisation inside the loop.
//
// Collective backends behind one small interface:
//   RCCL      ncclAllGather on the engine's stream.  librccl.so is opened at gh_comm_init_rccl() (dlopen), so the
//             library itself has no link-time dependency on it; the communicator is created from a 128-byte
//             unique id that rank 0 makes (gh_comm_unique_id) and the caller distributes (torch.distributed,
//             MPI, a file: whatever launched the ranks).
//   loopback  several engines of ONE process (one thread per engine) exchange through device-to-device copies
//             at a host rendezvous.  Lets the multi-rank loop run -- and be tested -- on a single GPU, where
//             RCCL refuses two ranks on one device.
#include "common.h"
#include "engine.h"

#include <dlfcn.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h>

// ---- RCCL through dlopen -------------------------------------------------------------------
namespace {

struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr;   // optional (form D's second communicator)
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

rccl_api *rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names) {
            api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.err = std::string("cannot open librccl.so: ") + dlerror(); return; }
        auto sym = [&](const char *s) { void *p = dlsym(api.lib, s); if (!p) api.err = std::string("librccl.so lacks ") + s; return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.CommSplit = reinterpret_cast<decltype(api.CommSplit)>(dlsym(api.lib, "ncclCommSplit"));
    });
    return &api;
}

thread_local std::string g_comm_error;

}  // namespace

// ---- loopback group -----------------------------------------------------------------------
struct gh_loop_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> send;
    bool failed = false;

    // every rank calls with the same sequence of collectives; returns true after all ranks arrived, false once any rank
    // has poisoned the group (a rank that failed will never arrive: the others must not wait for it)
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return !failed;
    }
    void poison() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

struct gh_comm {
    int world = 1, rank = 0;
    ncclComm_t nccl = nullptr;          // RCCL backend
    gh_loop_group *loop = nullptr;      // loopback backend
    uint64_t *d_gathered = nullptr;     // (world, S, K) keys of all ranks
    double *d_stats_all = nullptr;      // (world, stats rows, LD) statistics of all ranks (form C)
    // form D: the all-gather of the new0 blocks runs beside the KNN tail, on a stream and a communicator of its own
    ncclComm_t nccl_b = nullptr;        // ncclCommSplit of `nccl` (same ranks); null: the loopback backend, or no overlap
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_fused = nullptr, ev_rows = nullptr;
};

// side: the collective goes on the second stream / communicator (form D's early all-gather of the rows)
static gh_status comm_all_gather(gh_engine *h, const void *send, void *recv, size_t bytes, const char *what, bool side = false) {
    gh_comm *c = h->comm;
    const hipStream_t stream = side ? c->stream_b : h->stream;
    gh_scope t(h, what, stream);
    if (c->nccl) {
        const ncclResult_t r = rccl()->AllGather(send, recv, bytes, ncclUint8, side ? c->nccl_b : c->nccl, stream);
        if (r != ncclSuccess) { h->err = std::string("ncclAllGather: ") + rccl()->GetErrorString(r); return GH_ERR_RUNTIME; }
        return GH_OK;
    }
    // loopback: what the ranks send must be complete before anybody copies it, and nobody may start
    // overwriting its send buffer (the next iteration) before everybody has copied
    gh_loop_group *g = c->loop;
    auto fail = [&](const char *msg) { g->poison(); h->err = msg; return GH_ERR_RUNTIME; };
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("loopback all-gather: stream synchronisation failed");
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->send[(size_t)c->rank] = send;
    }
    if (!g->barrier()) { h->err = "loopback all-gather: another rank of the group failed"; return GH_ERR_RUNTIME; }
    for (int r = 0; r < c->world; ++r) {
        void *dst = static_cast<unsigned char *>(recv) + (size_t)r * bytes;
        const void *src = g->send[(size_t)r];
        if (dst != src && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail("loopback all-gather: copy failed");
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("loopback all-gather: stream synchronisation failed");
    if (!g->barrier()) { h->err = "loopback all-gather: another rank of the group failed"; return GH_ERR_RUNTIME; }
    return GH_OK;
}

extern "C" const char *gh_comm_last_error(void) { return g_comm_error.c_str(); }

extern "C" int32_t gh_comm_available(void) { return rccl()->err.empty() && rccl()->lib ? 1 : 0; }

extern "C" gh_status gh_comm_unique_id(void *out128) {
    if (!out128) { g_comm_error = "out is NULL"; return GH_ERR_INVALID; }
    rccl_api *api = rccl();
    if (!api->err.empty()) { g_comm_error = api->err; return GH_ERR_RUNTIME; }
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return GH_ERR_RUNTIME; }
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return GH_OK;
}

static gh_status comm_common(gh_engine *h, int world, int rank) {
    if (h->comm) { h->err = "a communicator is already attached"; return GH_ERR_INVALID; }
    if (h->g_world != world || h->g_rank != rank) {
        h->err = "call gh_rank_layout (or gh_gather_layout) with the same world and rank first";
        return GH_ERR_INVALID;
    }
    h->comm = new (std::nothrow) gh_comm();
    if (!h->comm) { h->err = "out of host memory"; return GH_ERR_NOMEM; }
    h->comm->world = world;
    h->comm->rank = rank;
    const size_t stats_doubles = (size_t)world * (2 + 2 * gh_fix_blocks(h->LD)) * h->LD;
    if (hipMalloc(reinterpret_cast<void **>(&h->comm->d_gathered), sizeof(uint64_t) * (size_t)world * h->S * (h->K + (h->cd_part ? 2 : 0)) + 16) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&h->comm->d_stats_all), sizeof(double) * stats_doubles + 16) != hipSuccess) {
        if (h->comm->d_gathered) (void)hipFree(h->comm->d_gathered);
        delete h->comm; h->comm = nullptr;
        h->err = "hipMalloc of the gather buffers failed";
        return GH_ERR_NOMEM;
    }
    if (h->overlap) {   // form D: a second stream for the early all-gather of the rows, ordered against the engine's by two events
        if (hipStreamCreateWithFlags(&h->comm->stream_b, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->comm->ev_fused, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->comm->ev_rows, hipEventDisableTiming) != hipSuccess) {
            gh_comm_free(h);
            h->err = "form D: creating the side stream failed";
            return GH_ERR_HIP;
        }
    }
    return GH_OK;
}

extern "C" gh_status gh_comm_init_rccl(gh_handle h, int32_t world, int32_t rank, const void *unique_id128) {
    if (!h) return GH_ERR_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GH_ERR_HIP; }
    if (!unique_id128 || world < 1 || rank < 0 || rank >= world) { h->err = "bad communicator arguments"; return GH_ERR_INVALID; }
    rccl_api *api = rccl();
    if (!api->err.empty()) { h->err = api->err; return GH_ERR_RUNTIME; }
    GH_TRY_ST(comm_common(h, world, rank));
    ncclUniqueId id;
    memcpy(id.internal, unique_id128, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = api->CommInitRank(&h->comm->nccl, world, id, rank);
    if (r != ncclSuccess) {
        h->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
        (void)hipFree(h->comm->d_gathered);
        (void)hipFree(h->comm->d_stats_all);
        delete h->comm; h->comm = nullptr;
        return GH_ERR_RUNTIME;
    }
    // form D: the early all-gather needs a communicator of its own to be in flight beside the keys' and the statistics'.
    // Without ncclCommSplit (or if it fails -- on every rank alike: it is a collective call) the rows go out on the engine's
    // stream after the merge, form B's order: correct, nothing hidden.
    if (h->overlap && api->CommSplit) {   // (also at world 1: the rehearsal of the N > 1 path on a one-GPU box takes the same route)
        if (api->CommSplit(h->comm->nccl, 0, rank, &h->comm->nccl_b, nullptr) != ncclSuccess) h->comm->nccl_b = nullptr;
    }
    return GH_OK;
}

extern "C" gh_loop_group *gh_loopback_group_create(int32_t world) {
    if (world < 1) return nullptr;
    gh_loop_group *g = new (std::nothrow) gh_loop_group();
    if (!g) return nullptr;
    g->world = world;
    g->send.assign((size_t)world, nullptr);
    return g;
}
extern "C" void gh_loopback_group_destroy(gh_loop_group *g) { delete g; }

extern "C" gh_status gh_comm_init_loopback(gh_handle h, gh_loop_group *group, int32_t rank) {
    if (!h) return GH_ERR_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GH_ERR_HIP; }
    if (!group || rank < 0 || rank >= group->world) { h->err = "bad loopback group / rank"; return GH_ERR_INVALID; }
    GH_TRY_ST(comm_common(h, group->world, rank));
    h->comm->loop = group;
    return GH_OK;
}

void gh_comm_free(gh_engine *h) {
    if (!h->comm) return;
    if (h->comm->nccl_b) (void)rccl()->CommDestroy(h->comm->nccl_b);
    if (h->comm->nccl) (void)rccl()->CommDestroy(h->comm->nccl);
    if (h->comm->stream_b) (void)hipStreamDestroy(h->comm->stream_b);
    if (h->comm->ev_fused) (void)hipEventDestroy(h->comm->ev_fused);
    if (h->comm->ev_rows) (void)hipEventDestroy(h->comm->ev_rows);
    if (h->comm->d_gathered) (void)hipFree(h->comm->d_gathered);
    if (h->comm->d_stats_all) (void)hipFree(h->comm->d_stats_all);
    delete h->comm;
    h->comm = nullptr;
}

extern "C" gh_status gh_comm_destroy(gh_handle h) {
    if (!h) return GH_ERR_INVALID;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    gh_comm_free(h);
    return GH_OK;
}

extern "C" gh_status gh_run_partitioned(gh_handle h, int32_t iters, const int32_t *sample_stream) {
    if (!h) return GH_ERR_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return GH_ERR_HIP; }
    if (!h->comm) { h->err = "no communicator: call gh_comm_init_rccl / gh_comm_init_loopback first"; return GH_ERR_INVALID; }
    if (iters < 0) { h->err = "negative iteration count"; return GH_ERR_INVALID; }
    gh_comm *c = h->comm;
    const size_t key_bytes = sizeof(uint64_t) * (size_t)h->S * (h->K + (h->cd_part ? 2 : 0));   // (a GH_DIST_CDIST partition sends K + 1 keys and a flag)
    const int32_t *d_ids = nullptr;
    GH_TRY_ST(gh_upload_sample_stream(h, iters, sample_stream, &d_ids));   // nullptr: device sampler / arange on every rank
    // a rank that fails leaves the loop: the loopback group must not wait for it (RCCL has its own abort paths)
    auto run = [&]() -> gh_status {
        const size_t stats_bytes = sizeof(double) * (size_t)(2 + 2 * gh_fix_blocks(h->LD)) * h->LD;
        for (int32_t t = 0; t < iters; ++t) {
            GH_TRY_ST(gh_step_begin_device_ids(h, d_ids ? d_ids + (size_t)t * h->S : nullptr));
            // form D: new0 = pos + Fs of the own rows is complete -- its all-gather starts here, on the side stream when there
            // is a second communicator (RCCL) or on the engine's (loopback: the host rendezvous blocks either way), and runs
            // beside select -> keys -> merge + intersection -> statistics
            const bool early = h->overlap && gh_step_rows_early(h);
            const bool side = early && c->stream_b && (c->nccl_b || c->loop);
            if (early && (side || c->loop)) {
                const size_t block = sizeof(float) * (size_t)h->g_chunk * gh_rows_all_row_floats(h);
                unsigned char *rows = reinterpret_cast<unsigned char *>(gh_rows_all_device(h));
                if (side) {
                    if (hipEventRecord(c->ev_fused, h->stream) != hipSuccess || hipStreamWaitEvent(c->stream_b, c->ev_fused, 0) != hipSuccess) { h->err = "form D: event hand-off failed"; return GH_ERR_HIP; }
                }
                GH_TRY_ST(gh_launch_pack_rows(h, side ? c->stream_b : h->stream));
                GH_TRY_ST(comm_all_gather(h, rows + (size_t)c->rank * block, rows, block, "allgather_rows", side));
                if (side && hipEventRecord(c->ev_rows, c->stream_b) != hipSuccess) { h->err = "form D: event record failed"; return GH_ERR_HIP; }
            }
            const bool rows_sent = early && (side || c->loop);
            if (h->S > 0 && h->k > 0) {
                GH_TRY_ST(comm_all_gather(h, h->d_partial, c->d_gathered, key_bytes, "allgather_keys"));
                GH_TRY_ST(gh_step_merge(h, c->d_gathered, c->world));
            } else {
                GH_TRY_ST(gh_step_merge(h, h->d_partial, 1));   // spring forces only: nothing to merge
            }
            if (h->overlap) {   // form D: (the rows went out above, or go now)
                const size_t block = sizeof(float) * (size_t)h->g_chunk * gh_rows_all_row_floats(h);
                unsigned char *rows = reinterpret_cast<unsigned char *>(gh_rows_all_device(h));
                if (!rows_sent) {   // a step without new0 (no fused kernel), or RCCL without a second communicator: form B's order
                    GH_TRY_ST(gh_launch_pack_rows(h, h->stream));
                    GH_TRY_ST(comm_all_gather(h, rows + (size_t)c->rank * block, rows, block, "allgather_rows"));
                }
                GH_TRY_ST(comm_all_gather(h, h->d_stats, h->d_stats_all, sizeof(double) * (size_t)h->stats_block, "allgather_stats"));   // statistics + patch list
                if (early && side) {   // what of the early all-gather is still outstanding is this iteration's EXPOSED collective time
                    gh_scope t(h, "allgather_rows_exposed");
                    if (hipStreamWaitEvent(h->stream, c->ev_rows, 0) != hipSuccess) { h->err = "hipStreamWaitEvent failed"; return GH_ERR_HIP; }
                }
                GH_TRY_ST(gh_step_finish_overlap(h));
            } else if (h->d_gbuf) {   // form B
                GH_TRY_ST(comm_all_gather(h, h->d_gbuf + (size_t)c->rank * h->g_slot, h->d_gbuf, (size_t)h->g_slot, "allgather_slots"));
                GH_TRY_ST(gh_step_finish_gathered(h));
            } else {           // form C
                GH_TRY_ST(comm_all_gather(h, h->d_stats, c->d_stats_all, stats_bytes, "allgather_stats"));
                GH_TRY_ST(gh_step_finish_own(h, c->d_stats_all, c->world));
                if (h->packed_exchange) {   // the blocks travel without their pad columns (12 instead of 16 bytes per row at D = 3)
                    const size_t block = sizeof(float) * (size_t)h->g_chunk * h->D;
                    GH_TRY_ST(comm_all_gather(h, reinterpret_cast<unsigned char *>(h->d_rows_packed) + (size_t)c->rank * block, h->d_rows_packed, block, "allgather_rows"));
                    GH_TRY_ST(gh_launch_unpack_rows(h));
                } else {
                    const size_t block = sizeof(float) * (size_t)h->g_chunk * h->LD;
                    GH_TRY_ST(comm_all_gather(h, reinterpret_cast<unsigned char *>(h->d_pos) + (size_t)c->rank * block, h->d_pos, block, "allgather_rows"));
                }
            }
        }
        return GH_OK;
    };
    const gh_status st = run();
    if (st != GH_OK && c->loop) c->loop->poison();
    return st;
}

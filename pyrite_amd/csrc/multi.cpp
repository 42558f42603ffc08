// multi.cpp -- the multi-GPU entry points of include/pyrite_gpu.h.
//
// The reference has one process and shared memory: simple::render hands independent tiles to worker threads
// (pyrite/src/renderer/simple.rs:36-55, renderer/mod.rs:125-189) and every tile exposes its own pixels. Here a "worker" is a
// GPU: the scene is replicated, rank r of n renders the tiles r, r + n, ... in one launch into a private buffer of ringed tile
// blocks (PYR_FILM_TILE_BLOCKS) and ONE gather -- a group of ncclSend / ncclRecv, RCCL over xGMI -- brings the blocks to rank
// 0, which adds them into the film (assemble kernels, kernels.hip). No data-path collective before that.
//
// librccl is loaded with dlopen when the first communicator is made: single-GPU users of the library never pay for it, and
// a host process that already holds an RCCL (PyTorch ships its own copy) shares that one instead of loading a second.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "api_internal.h"

using namespace pyr;

namespace {

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return api_fail(PYR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL the process already holds first (RTLD_NOLOAD), then the ROCm installation's
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!lib.handle) lib.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : paths)
            if (!lib.handle) lib.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!lib.handle) {
            lib.error = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(lib.handle, name);
            if (!p && lib.error.empty()) lib.error = std::string("librccl lacks ") + name;
            return p;
        };
        lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(sym("ncclGetUniqueId"));
        lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(sym("ncclCommInitRank"));
        lib.CommInitAll = reinterpret_cast<decltype(lib.CommInitAll)>(sym("ncclCommInitAll"));
        lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
        lib.GroupStart = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
        lib.GroupEnd = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
        lib.Send = reinterpret_cast<decltype(lib.Send)>(sym("ncclSend"));
        lib.Recv = reinterpret_cast<decltype(lib.Recv)>(sym("ncclRecv"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &lib;
}

int rccl_ready(Rccl*& out) {
    out = rccl();
    if (!out->error.empty()) return api_fail(PYR_ERR_DEVICE, out->error);
    return PYR_OK;
}

#define NCCL_TRY(lib, expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return api_fail(PYR_ERR_DEVICE, std::string(#expr) + ": " + (lib)->GetErrorString(r_)); \
    } while (0)

struct Grown { // a device buffer that only ever grows
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t n, hipStream_t stream) {
        if (n <= bytes) return PYR_OK;
        if (ptr) {
            HIP_TRY(hipStreamSynchronize(stream)); // an earlier call on this stream may still use the old one
            HIP_TRY(hipFree(ptr));
            ptr = nullptr;
            bytes = 0;
        }
        HIP_TRY(hipMalloc(&ptr, n));
        bytes = n;
        return PYR_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

// The tiles rank `rank` of `num_ranks` renders out of those `params` selects, as parameters of its own call.
PyrRenderParams rank_share(const PyrFilmDesc* film, const PyrRenderParams* params, uint32_t rank, uint32_t num_ranks) {
    const uint32_t ts = params->tile_size;
    const uint64_t total = (uint64_t)((film->width + ts - 1) / ts) * ((film->height + ts - 1) / ts);
    const uint64_t end = params->tile_end ? params->tile_end : total;
    const uint64_t stride = std::max(1u, params->tile_stride);
    PyrRenderParams mine = *params;
    mine.tile_end = (uint32_t)end;
    mine.tile_begin = (uint32_t)std::min<uint64_t>(end, params->tile_begin + rank * stride); // == end: no tile for this rank
    mine.tile_stride = (uint32_t)(stride * num_ranks);
    mine.film_layout = PYR_FILM_TILE_BLOCKS;
    mine.film_row_begin = mine.film_row_count = 0;
    return mine;
}

} // namespace

struct PyrComm {
    int rank = 0, num_ranks = 1, device = 0;
    ncclComm_t comm = nullptr; // nullptr for a communicator of one rank
    bool owns_comm = true;
    Grown window;   // this rank's blocks
    Grown gathered; // rank 0: the other ranks' blocks, one after the other
};

extern "C" {

int pyr_comm_unique_id(uint8_t id_out[PYR_COMM_ID_BYTES]) {
    static_assert(PYR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    Rccl* lib;
    int rc = rccl_ready(lib);
    if (rc != PYR_OK) return rc;
    ncclUniqueId id;
    NCCL_TRY(lib, lib->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, PYR_COMM_ID_BYTES);
    return PYR_OK;
}

int pyr_comm_create(const uint8_t id_in[PYR_COMM_ID_BYTES], int rank, int num_ranks, int device, PyrComm** out_comm) {
    if (!out_comm) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null out pointer");
    *out_comm = nullptr;
    if (num_ranks < 1 || rank < 0 || rank >= num_ranks) return api_fail(PYR_ERR_INVALID_ARGUMENT, "rank out of range");
    if (device < 0 || device >= pyr_device_count()) return api_fail(PYR_ERR_DEVICE, "no such HIP device; pyrite_gpu has no CPU path");
    HIP_TRY(hipSetDevice(device));
    PyrComm* c = new PyrComm();
    c->rank = rank, c->num_ranks = num_ranks, c->device = device;
    if (num_ranks > 1) {
        if (!id_in) {
            delete c;
            return api_fail(PYR_ERR_INVALID_ARGUMENT, "null communicator id");
        }
        Rccl* lib;
        int rc = rccl_ready(lib);
        if (rc != PYR_OK) {
            delete c;
            return rc;
        }
        ncclUniqueId id;
        std::memcpy(id.internal, id_in, PYR_COMM_ID_BYTES);
        ncclResult_t r = lib->CommInitRank(&c->comm, num_ranks, id, rank);
        if (r != ncclSuccess) {
            delete c;
            return api_fail(PYR_ERR_DEVICE, std::string("ncclCommInitRank: ") + lib->GetErrorString(r));
        }
    }
    *out_comm = c;
    return PYR_OK;
}

void pyr_comm_destroy(PyrComm* comm) {
    if (!comm) return;
    (void)hipSetDevice(comm->device);
    comm->window.release();
    comm->gathered.release();
    if (comm->comm && comm->owns_comm) (void)rccl()->CommDestroy(comm->comm);
    delete comm;
}

int pyr_render_simple_sharded(PyrComm* comm, PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params,
                              PyrGrain* film_device_rank0, void* hip_stream) {
    if (!comm || !scene || !camera || !film || !params) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (comm->rank == 0 && !film_device_rank0) return api_fail(PYR_ERR_INVALID_ARGUMENT, "rank 0 needs the film");
    if (scene_device(scene) != comm->device) return api_fail(PYR_ERR_INVALID_ARGUMENT, "the scene lives on another device than the communicator");
    if (params->film_layout != PYR_FILM_ROWS || params->film_row_begin || params->film_row_count)
        return api_fail(PYR_ERR_INVALID_ARGUMENT, "a sharded render adds into the whole-image film on rank 0");
    if (params->tile_size == 0 || film->width == 0 || film->height == 0 || film->bins == 0) return api_fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
    HIP_TRY(hipSetDevice(comm->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    const uint32_t n = (uint32_t)comm->num_ranks, me = (uint32_t)comm->rank;

    std::vector<PyrRenderParams> share(n);
    std::vector<uint64_t> grains(n);
    for (uint32_t r = 0; r < n; ++r) {
        share[r] = rank_share(film, params, r, n);
        grains[r] = share[r].tile_begin < share[r].tile_end ? pyr_film_blocks_grains(film, &share[r]) : 0;
        if (share[r].tile_begin < share[r].tile_end && grains[r] == 0) return PYR_ERR_INVALID_ARGUMENT; // message set by the callee
    }
    int rc;
    if (grains[me]) {
        if ((rc = comm->window.reserve(grains[me] * sizeof(PyrGrain), stream)) != PYR_OK) return rc;
        HIP_TRY(hipMemsetAsync(comm->window.ptr, 0, grains[me] * sizeof(PyrGrain), stream));
        if ((rc = pyr_render_simple_device(scene, camera, film, &share[me], (PyrGrain*)comm->window.ptr, stream)) != PYR_OK) return rc;
    }
    if (n > 1) { // the gather: every rank's blocks to rank 0
        Rccl* lib = rccl();
        uint64_t others = 0;
        for (uint32_t r = 1; r < n; ++r) others += grains[r];
        if (me == 0 && others && (rc = comm->gathered.reserve(others * sizeof(PyrGrain), stream)) != PYR_OK) return rc;
        NCCL_TRY(lib, lib->GroupStart());
        if (me == 0) {
            uint64_t offset = 0;
            for (uint32_t r = 1; r < n; ++r) {
                if (grains[r]) NCCL_TRY(lib, lib->Recv((PyrGrain*)comm->gathered.ptr + offset, grains[r] * 2, ncclFloat, (int)r, comm->comm, stream));
                offset += grains[r];
            }
        } else if (grains[me]) {
            NCCL_TRY(lib, lib->Send(comm->window.ptr, grains[me] * 2, ncclFloat, 0, comm->comm, stream));
        }
        NCCL_TRY(lib, lib->GroupEnd());
    }
    if (me == 0) {
        uint64_t offset = 0;
        for (uint32_t r = 0; r < n; ++r) {
            if (!grains[r]) continue;
            const PyrGrain* blocks = r == 0 ? (const PyrGrain*)comm->window.ptr : (const PyrGrain*)comm->gathered.ptr + offset;
            if ((rc = pyr_film_blocks_assemble_device(film, &share[r], blocks, film_device_rank0, comm->device, stream)) != PYR_OK) return rc;
            if (r) offset += grains[r];
        }
    }
    return PYR_OK;
}

int pyr_render_simple_multi(PyrScene* const* scenes, uint32_t num_devices, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params,
                            PyrGrain* film_inout, PyrProgressFn on_status, void* user) {
    if (!scenes || num_devices == 0 || !camera || !film || !params || !film_inout) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (params->film_layout != PYR_FILM_ROWS || params->film_row_begin || params->film_row_count)
        return api_fail(PYR_ERR_INVALID_ARGUMENT, "a multi-device render adds into a whole-image film");
    if (params->tile_size == 0 || film->width == 0 || film->height == 0 || film->bins == 0) return api_fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
    std::vector<int> devices(num_devices);
    bool distinct = true;
    for (uint32_t i = 0; i < num_devices; ++i) {
        if (!scenes[i]) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null scene");
        devices[i] = scene_device(scenes[i]);
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
    }
    const char* message = "Rendering"; // simple.rs:30
    if (on_status) on_status(user, 0, message);

    // communicators: RCCL (ncclCommInitAll, kept for the process: making them takes seconds) when every rank has a GPU of
    // its own; otherwise the blocks travel by hipMemcpyPeerAsync below
    std::vector<PyrComm> comms(num_devices);
    for (uint32_t i = 0; i < num_devices; ++i) {
        comms[i].rank = (int)i, comms[i].num_ranks = (int)num_devices, comms[i].device = devices[i], comms[i].owns_comm = false;
    }
    const bool use_rccl = distinct && num_devices > 1;
    if (use_rccl) {
        static std::mutex cache_mutex;
        static std::map<std::vector<int>, std::vector<ncclComm_t>> cache;
        Rccl* lib;
        int rc = rccl_ready(lib);
        if (rc != PYR_OK) return rc;
        std::lock_guard<std::mutex> lock(cache_mutex);
        auto it = cache.find(devices);
        if (it == cache.end()) {
            std::vector<ncclComm_t> made(num_devices);
            NCCL_TRY(lib, lib->CommInitAll(made.data(), (int)num_devices, devices.data()));
            it = cache.emplace(devices, made).first;
        }
        for (uint32_t i = 0; i < num_devices; ++i) comms[i].comm = it->second[i];
    }

    const size_t film_bytes = (size_t)film->width * film->height * film->bins * sizeof(PyrGrain);
    PyrGrain* film_dev = nullptr;
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMalloc((void**)&film_dev, film_bytes));
    struct FilmGuard {
        PyrGrain* p;
        int device;
        ~FilmGuard() {
            (void)hipSetDevice(device);
            (void)hipFree(p);
        }
    } film_guard{film_dev, devices[0]};
    HIP_TRY(hipMemcpy(film_dev, film_inout, film_bytes, hipMemcpyHostToDevice));

    std::vector<int> status(num_devices, PYR_OK);
    std::vector<std::string> messages(num_devices);
    std::vector<hipStream_t> streams(num_devices, nullptr);
    int result = PYR_OK;
    std::string result_message;
    auto note = [&](uint32_t i) {
        if (status[i] != PYR_OK && result == PYR_OK) result = status[i], result_message = messages[i];
    };
    if (use_rccl || num_devices == 1) {
        // one host thread per device: each issues its render, its side of the gather and (rank 0) the assembly, then waits
        auto work = [&](uint32_t i) {
            if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) {
                status[i] = PYR_ERR_DEVICE, messages[i] = "hipStreamCreate failed";
                return;
            }
            status[i] = pyr_render_simple_sharded(&comms[i], scenes[i], camera, film, params, i == 0 ? film_dev : nullptr, streams[i]);
            if (status[i] != PYR_OK) messages[i] = pyr_last_error();
            if (hipStreamSynchronize(streams[i]) != hipSuccess && status[i] == PYR_OK) status[i] = PYR_ERR_DEVICE, messages[i] = "hipStreamSynchronize failed";
        };
        std::vector<std::thread> pool;
        for (uint32_t i = 1; i < num_devices; ++i) pool.emplace_back(work, i);
        work(0);
        for (auto& t : pool) t.join();
        for (uint32_t i = 0; i < num_devices; ++i) note(i);
    } else {
        // test rig (several logical ranks on one GPU): the ranks render one after the other, each as a communicator of its own
        // size-1 world over its share, and the blocks are copied and assembled here
        HIP_TRY(hipSetDevice(devices[0]));
        HIP_TRY(hipStreamCreateWithFlags(&streams[0], hipStreamNonBlocking));
        Grown staged;
        for (uint32_t i = 0; i < num_devices && result == PYR_OK; ++i) {
            const PyrRenderParams mine = rank_share(film, params, i, num_devices);
            if (mine.tile_begin >= mine.tile_end) continue;
            const uint64_t grains = pyr_film_blocks_grains(film, &mine);
            int rc = PYR_OK;
            if (hipSetDevice(devices[i]) != hipSuccess || (i > 0 && hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess))
                rc = api_fail(PYR_ERR_DEVICE, "hipStreamCreate failed");
            hipStream_t s = streams[i];
            if (rc == PYR_OK) rc = comms[i].window.reserve(grains * sizeof(PyrGrain), s);
            if (rc == PYR_OK && hipMemsetAsync(comms[i].window.ptr, 0, grains * sizeof(PyrGrain), s) != hipSuccess) rc = api_fail(PYR_ERR_DEVICE, "hipMemsetAsync failed");
            if (rc == PYR_OK) rc = pyr_render_simple_device(scenes[i], camera, film, &mine, (PyrGrain*)comms[i].window.ptr, s);
            if (rc == PYR_OK && hipStreamSynchronize(s) != hipSuccess) rc = api_fail(PYR_ERR_DEVICE, "hipStreamSynchronize failed");
            if (rc == PYR_OK) {
                (void)hipSetDevice(devices[0]);
                rc = staged.reserve(grains * sizeof(PyrGrain), streams[0]);
                if (rc == PYR_OK && hipMemcpyPeerAsync(staged.ptr, devices[0], comms[i].window.ptr, devices[i], grains * sizeof(PyrGrain), streams[0]) != hipSuccess)
                    rc = api_fail(PYR_ERR_DEVICE, "hipMemcpyPeerAsync failed");
                if (rc == PYR_OK) rc = pyr_film_blocks_assemble_device(film, &mine, (const PyrGrain*)staged.ptr, film_dev, devices[0], streams[0]);
                if (rc == PYR_OK && hipStreamSynchronize(streams[0]) != hipSuccess) rc = api_fail(PYR_ERR_DEVICE, "hipStreamSynchronize failed");
            }
            if (rc != PYR_OK) result = rc, result_message = pyr_last_error();
        }
        (void)hipSetDevice(devices[0]);
        staged.release();
    }
    for (uint32_t i = 0; i < num_devices; ++i) {
        (void)hipSetDevice(devices[i]);
        if (streams[i]) (void)hipStreamDestroy(streams[i]);
        comms[i].window.release();
        comms[i].gathered.release();
    }
    if (result != PYR_OK) return api_fail(result, result_message);
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMemcpy(film_inout, film_dev, film_bytes, hipMemcpyDeviceToHost));
    if (on_status) on_status(user, 100, message);
    return PYR_OK;
}

} // extern "C"

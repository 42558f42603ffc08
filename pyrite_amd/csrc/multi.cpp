// multi.cpp -- the multi-GPU entry points of include/pyrite_gpu.h.
//
// The reference has one process and shared memory: simple::render hands independent tiles to worker threads
// (pyrite/src/renderer/simple.rs:36-55, renderer/mod.rs:125-189) and every tile exposes its own pixels. Here a "worker" is a
// GPU: the scene is replicated, rank r of n renders the tiles r, r + n, ... in one launch into a private buffer of ringed tile
// blocks (PYR_FILM_TILE_BLOCKS) and ONE gather -- a group of ncclSend / ncclRecv, RCCL over xGMI -- brings the blocks to rank
// 0, which adds them into the film (assemble kernels, kernels.hip). No data-path collective before that.
//
// A failed rank must not leave its peers blocked in the gather (the reference collects its workers' results on one thread,
// renderer/mod.rs:181-183, and a worker that dies takes the whole render down with it -- the equivalent here is an error on
// every rank). The protocol of pyr_render_simple_sharded:
//   1. everything that can fail on the host before the gather (arguments, buffer growth) happens first and yields a status;
//   2. the ranks AGREE on it: one ncclAllReduce(max) of the status word, read back on the host. If any rank failed, every
//      rank returns an error and nobody posts a send or a receive;
//   3. the render is enqueued. What can still go wrong now -- the launch itself, or the kernels flagging their film invalid
//      (spectral tape overflow) -- travels WITH the data: every block buffer ends in one trailer grain that carries the
//      sender's status, so the message sizes never depend on an outcome and every rank enters the group;
//   4. the group always reaches ncclGroupEnd; an error inside it aborts the communicator (ncclCommAbort), which is then dead;
//   5. rank 0 assembles and copies the trailers to the host; pyr_comm_status() reports them once the stream has been waited for.
//
// librccl is loaded with dlopen when the first communicator is made: single-GPU users of the library never pay for it, and
// a host process that already holds an RCCL (PyTorch ships its own copy) shares that one instead of loading a second.
// PYRITE_RCCL_LIB names another library to load instead (a newer RCCL; the tests' in-process stand-in, tests/fake_rccl).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#if !defined(PYR_NO_RCCL_HEADER) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// The handful of NCCL declarations this file needs, for builds on machines without the RCCL headers (the library itself is
// only ever dlopen'ed). Values are those of nccl.h 2.x, which RCCL follows.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
    char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;
}
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool substitute = false; // PYRITE_RCCL_LIB is set: the library named there, not the system's RCCL
    std::string error;
};

Rccl* rccl() {
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* named = std::getenv("PYRITE_RCCL_LIB");
        if (named && *named) {
            lib.substitute = true;
            lib.handle = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
        } else {
            // an RCCL the process already holds first (RTLD_NOLOAD), then the ROCm installation's
            const char* names[] = {"librccl.so", "librccl.so.1"};
            for (const char* n : names)
                if (!lib.handle) lib.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : paths)
                if (!lib.handle) lib.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!lib.handle) {
            const char* why = dlerror(); // one call: dlerror() clears the message it returns
            lib.error = std::string("librccl could not be loaded") + (named && *named ? std::string(" from ") + named : std::string()) + ": " + (why ? why : "not found");
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
        lib.CommAbort = reinterpret_cast<decltype(lib.CommAbort)>(sym("ncclCommAbort"));
        lib.GroupStart = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
        lib.GroupEnd = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
        lib.Send = reinterpret_cast<decltype(lib.Send)>(sym("ncclSend"));
        lib.Recv = reinterpret_cast<decltype(lib.Recv)>(sym("ncclRecv"));
        lib.AllReduce = reinterpret_cast<decltype(lib.AllReduce)>(sym("ncclAllReduce"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &lib;
}

int rccl_ready(Rccl*& out) {
    out = rccl();
    if (!out->error.empty()) return api_fail(PYR_ERR_DEVICE, out->error);
    return PYR_OK;
}

std::string nccl_message(Rccl* lib, const char* what, ncclResult_t r) { return std::string(what) + ": " + lib->GetErrorString(r); }

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

// Test switches (never set outside the test suite): the rank named by PYRITE_TEST_FAIL_RANK pretends its buffers could not be
// grown (a failure BEFORE the agreement); the one named by PYRITE_TEST_FAIL_RENDER_RANK pretends its launch failed (a failure
// AFTER it, which travels in the trailer).
bool test_switch_names(const char* variable, int rank) {
    const char* v = std::getenv(variable);
    return v && *v && std::atoi(v) == rank;
}

constexpr uint32_t kTrailerLaunchFailed = 0x7F000000u; // a trailer word no kernel writes: the sender's launch failed on the host

} // namespace

struct PyrComm {
    int rank = 0, num_ranks = 1, device = 0;
    ncclComm_t comm = nullptr; // nullptr for a communicator of one rank (unless PYRITE_FORCE_RCCL made a real one)
    bool owns_comm = true;
    std::atomic<bool> dead{false}; // aborted after an error inside a collective: every later call fails. Atomic: a sibling rank's host thread may abort it (pyr_render_simple_multi)
    Grown window;      // this rank's blocks + one trailer grain
    Grown gathered;    // rank 0: the senders' blocks (each with its trailer), one after the other
    Grown agree;       // two words: this rank's status, the agreed one
    uint32_t* host_words = nullptr; // pinned: [0] the agreed status, [1] this rank's, [2 + r] trailer word of rank r (rank 0: everybody's; others: their own)
    bool pending = false;           // trailers of the last call have not been looked at yet

    int host_reserve() {
        if (host_words) return PYR_OK;
        HIP_TRY(hipHostMalloc((void**)&host_words, sizeof(uint32_t) * (size_t)(2 + num_ranks), hipHostMallocDefault));
        std::memset(host_words, 0, sizeof(uint32_t) * (size_t)(2 + num_ranks));
        return PYR_OK;
    }
    void abort_comm() { // exactly once per communicator, whichever thread gets here first
        if (!dead.exchange(true) && comm) {
            Rccl* lib = rccl();
            if (lib->CommAbort) (void)lib->CommAbort(comm);
        }
    }
    void release() {
        window.release();
        gathered.release();
        agree.release();
        if (host_words) (void)hipHostFree(host_words);
        host_words = nullptr;
    }
};

extern "C" {

int pyr_comm_unique_id(uint8_t id_out[PYR_COMM_ID_BYTES]) {
    static_assert(PYR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    Rccl* lib;
    int rc = rccl_ready(lib);
    if (rc != PYR_OK) return rc;
    ncclUniqueId id;
    ncclResult_t r = lib->GetUniqueId(&id);
    if (r != ncclSuccess) return api_fail(PYR_ERR_DEVICE, nccl_message(lib, "ncclGetUniqueId", r));
    std::memcpy(id_out, id.internal, PYR_COMM_ID_BYTES);
    return PYR_OK;
}

int pyr_comm_create(const uint8_t id_in[PYR_COMM_ID_BYTES], int rank, int num_ranks, int device, PyrComm** out_comm) {
    if (!out_comm) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null out pointer");
    *out_comm = nullptr;
    if (num_ranks < 1 || rank < 0 || rank >= num_ranks) return api_fail(PYR_ERR_INVALID_ARGUMENT, "rank out of range");
    if (device < 0 || device >= pyr_device_count()) return api_fail(PYR_ERR_DEVICE, "no such HIP device; pyrite_gpu has no CPU path");
    HIP_TRY(hipSetDevice(device));
    // A world of one needs no communicator -- unless PYRITE_FORCE_RCCL=1 asks for a real one-rank RCCL communicator: the
    // rank's blocks then travel through a grouped self ncclSend / ncclRecv (and the agreement through ncclAllReduce), i.e. the
    // code a multi-rank gather runs, on one GPU.
    const char* force = std::getenv("PYRITE_FORCE_RCCL");
    const bool real = num_ranks > 1 || (force && std::string(force) == "1");
    PyrComm* c = new PyrComm();
    c->rank = rank, c->num_ranks = num_ranks, c->device = device;
    if (real) {
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
            return api_fail(PYR_ERR_DEVICE, nccl_message(lib, "ncclCommInitRank", r));
        }
    }
    *out_comm = c;
    return PYR_OK;
}

void pyr_comm_destroy(PyrComm* comm) {
    if (!comm) return;
    (void)hipSetDevice(comm->device);
    comm->release();
    if (comm->comm && comm->owns_comm && !comm->dead) (void)rccl()->CommDestroy(comm->comm);
    delete comm;
}

int pyr_comm_uses_rccl(const PyrComm* comm) { return comm && comm->comm != nullptr ? 1 : 0; }

int pyr_comm_status(PyrComm* comm) {
    if (!comm) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (comm->dead) return api_fail(PYR_ERR_DEVICE, "the communicator was aborted after an error inside a collective");
    if (!comm->pending || !comm->host_words) return PYR_OK;
    comm->pending = false;
    for (int r = 0; r < comm->num_ranks; ++r) {
        const uint32_t word = comm->host_words[2 + r];
        if (word == 0) continue;
        const std::string who = "rank " + std::to_string(r);
        if (word == kTrailerLaunchFailed) return api_fail(PYR_ERR_DEVICE, who + " could not launch its render: the gathered film is invalid");
        if (word == 2) return api_fail(PYR_ERR_DEVICE, who + ": a wave gave up waiting on its workgroup's LDS queues: the gathered film is invalid");
        return api_fail(PYR_ERR_DEVICE, who + ": a path appended more records than the spectral tape's bound allows: the gathered film is invalid");
    }
    return PYR_OK;
}

int pyr_render_simple_sharded(PyrComm* comm, PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params,
                              PyrGrain* film_device_rank0, void* hip_stream) {
    if (!comm) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (comm->dead) return api_fail(PYR_ERR_DEVICE, "the communicator was aborted after an error inside a collective");
    HIP_TRY(hipSetDevice(comm->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    const uint32_t n = (uint32_t)comm->num_ranks, me = (uint32_t)comm->rank;
    const bool travels = comm->comm != nullptr; // blocks go through RCCL (several ranks, or the forced one-rank communicator)
    Rccl* lib = travels ? rccl() : nullptr;

    // ---- 1. what can fail on the host before the gather; `local` is this rank's verdict, with pyr_last_error() set by whoever failed
    int local = PYR_OK;
    std::vector<PyrRenderParams> share(n);
    std::vector<uint64_t> grains(n, 0);
    auto prepare = [&]() -> int {
        // the few bytes the agreement itself needs come first: a rank that fails further down can still tell its peers
        int rc = comm->host_reserve();
        if (rc != PYR_OK) return rc;
        if (travels && (rc = comm->agree.reserve(2 * sizeof(uint32_t), stream)) != PYR_OK) return rc;
        if (!scene || !camera || !film || !params) return api_fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
        if (me == 0 && !film_device_rank0) return api_fail(PYR_ERR_INVALID_ARGUMENT, "rank 0 needs the film");
        if (scene_device(scene) != comm->device) return api_fail(PYR_ERR_INVALID_ARGUMENT, "the scene lives on another device than the communicator");
        if (params->film_layout != PYR_FILM_ROWS || params->film_row_begin || params->film_row_count)
            return api_fail(PYR_ERR_INVALID_ARGUMENT, "a sharded render adds into the whole-image film on rank 0");
        if (params->tile_size == 0 || film->width == 0 || film->height == 0 || film->bins == 0) return api_fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
        for (uint32_t r = 0; r < n; ++r) {
            share[r] = rank_share(film, params, r, n);
            grains[r] = share[r].tile_begin < share[r].tile_end ? pyr_film_blocks_grains(film, &share[r]) : 0;
            if (share[r].tile_begin < share[r].tile_end && grains[r] == 0) return PYR_ERR_INVALID_ARGUMENT; // message set by the callee
        }
        if (test_switch_names("PYRITE_TEST_FAIL_RANK", (int)me)) return api_fail(PYR_ERR_DEVICE, "test switch: this rank's buffers could not be grown");
        if (grains[me] && (rc = comm->window.reserve((grains[me] + 1) * sizeof(PyrGrain), stream)) != PYR_OK) return rc;
        if (travels && me == 0) {
            uint64_t incoming = 0;
            for (uint32_t r = (n == 1 ? 0u : 1u); r < n; ++r) incoming += grains[r] ? grains[r] + 1 : 0;
            if (incoming && (rc = comm->gathered.reserve(incoming * sizeof(PyrGrain), stream)) != PYR_OK) return rc;
        }
        return PYR_OK;
    };
    local = prepare();
    const std::string local_message = local != PYR_OK ? pyr_last_error() : "";

    // ---- 2. the agreement: nobody posts a send or a receive unless every rank got this far
    if (travels) {
        if (!comm->agree.ptr || !comm->host_words) { // not even the status word could be set up: peers cannot be told
            comm->abort_comm();
            return api_fail(local != PYR_OK ? local : PYR_ERR_DEVICE, local_message.empty() ? "no memory for the status exchange" : local_message);
        }
        uint32_t* words = (uint32_t*)comm->agree.ptr;
        comm->host_words[0] = 1u;
        comm->host_words[1] = local != PYR_OK ? 1u : 0u;
        bool ok = hipMemcpyAsync(words, comm->host_words + 1, sizeof(uint32_t), hipMemcpyHostToDevice, stream) == hipSuccess;
        ncclResult_t r = ok ? lib->AllReduce(words, words + 1, 1, ncclInt32, ncclMax, comm->comm, stream) : ncclSuccess;
        ok = ok && r == ncclSuccess;
        ok = ok && hipMemcpyAsync(comm->host_words, words + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(stream) == hipSuccess;
        if (!ok) {
            comm->abort_comm();
            return api_fail(PYR_ERR_DEVICE, r != ncclSuccess ? nccl_message(lib, "ncclAllReduce (status agreement)", r) : std::string("the status agreement failed on the device"));
        }
        if (comm->host_words[0] != 0u) {
            if (local != PYR_OK) return api_fail(local, local_message);
            return api_fail(PYR_ERR_DEVICE, "another rank failed before the gather; nothing was rendered");
        }
    } else if (local != PYR_OK) {
        return api_fail(local, local_message);
    }

    // ---- 3. the render; its outcome goes into the trailer grain behind the blocks
    int rc;
    if (grains[me]) {
        PyrGrain* blocks = (PyrGrain*)comm->window.ptr;
        uint32_t* trailer = (uint32_t*)(blocks + grains[me]);
        bool launched = hipMemsetAsync(blocks, 0, (grains[me] + 1) * sizeof(PyrGrain), stream) == hipSuccess;
        if (launched && test_switch_names("PYRITE_TEST_FAIL_RENDER_RANK", (int)me)) launched = false;
        if (launched) launched = pyr_render_simple_device(scene, camera, film, &share[me], blocks, stream) == PYR_OK;
        if (!launched) {
            (void)hipMemsetD32Async((hipDeviceptr_t)trailer, (int)kTrailerLaunchFailed, 1, stream);
        } else if (uint32_t* word = scene_overflow_word(scene)) {
            // the kernels' verdict on their own film (0 = fine), taken and cleared in stream order
            (void)hipMemcpyAsync(trailer, word, sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
            (void)hipMemsetAsync(word, 0, sizeof(uint32_t), stream);
        }
        if (me != 0) (void)hipMemcpyAsync(comm->host_words + 2 + me, trailer, sizeof(uint32_t), hipMemcpyDeviceToHost, stream); // rank 0 reads everybody's below
    } else {
        comm->host_words[2 + me] = 0u;
    }

    // ---- 4. the gather: every rank's blocks (+ trailer) to rank 0. The group is always closed; an error inside it kills the communicator
    if (travels) {
        ncclResult_t first = lib->GroupStart();
        const char* where = "ncclGroupStart";
        if (first == ncclSuccess) {
            auto note = [&](ncclResult_t r, const char* what) {
                if (r != ncclSuccess && first == ncclSuccess) first = r, where = what;
            };
            // One message per 256 MiB: RCCL 2.26.6 delivers about half of a single ncclSend / ncclRecv pair above 1 GiB (seen with
            // the one-rank communicator at 1920 x 1080: 1.2 GB of blocks, rows 544-1079 never arrived, no error reported;
            // tools/native_one_rank.py). Sends and receives between the same two ranks are matched in the order they are posted.
            constexpr uint64_t kMessageFloats = 1ull << 26;
            auto in_messages = [&](uint64_t floats, auto&& post) {
                for (uint64_t done = 0; done < floats; done += kMessageFloats) post(done, std::min(kMessageFloats, floats - done));
            };
            if (me == 0) {
                uint64_t offset = 0;
                for (uint32_t r = (n == 1 ? 0u : 1u); r < n; ++r) {
                    if (!grains[r]) continue;
                    float* into = reinterpret_cast<float*>((PyrGrain*)comm->gathered.ptr + offset);
                    in_messages((grains[r] + 1) * 2, [&](uint64_t at, uint64_t count) { note(lib->Recv(into + at, count, ncclFloat, (int)r, comm->comm, stream), "ncclRecv"); });
                    offset += grains[r] + 1;
                }
            }
            if ((me != 0 || n == 1) && grains[me]) {
                const float* from = reinterpret_cast<const float*>(comm->window.ptr);
                in_messages((grains[me] + 1) * 2, [&](uint64_t at, uint64_t count) { note(lib->Send(from + at, count, ncclFloat, 0, comm->comm, stream), "ncclSend"); });
            }
            note(lib->GroupEnd(), "ncclGroupEnd");
        }
        if (first != ncclSuccess) {
            comm->abort_comm();
            return api_fail(PYR_ERR_DEVICE, nccl_message(lib, where, first));
        }
    }

    // ---- 5. rank 0 adds everybody's blocks into the film and brings the trailers to the host
    if (me == 0) {
        uint64_t offset = 0;
        for (uint32_t r = 0; r < n; ++r) {
            if (!grains[r]) {
                comm->host_words[2 + r] = 0u;
                continue;
            }
            const bool sent = travels && (r != 0 || n == 1);
            const PyrGrain* blocks = sent ? (const PyrGrain*)comm->gathered.ptr + offset : (const PyrGrain*)comm->window.ptr;
            if ((rc = pyr_film_blocks_assemble_device(film, &share[r], blocks, film_device_rank0, comm->device, stream)) != PYR_OK) return rc;
            HIP_TRY(hipMemcpyAsync(comm->host_words + 2 + r, blocks + grains[r], sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            if (sent) offset += grains[r] + 1;
        }
    }
    comm->pending = true;
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
    // its own; otherwise the blocks travel by hipMemcpyPeerAsync below. A substitute library (PYRITE_RCCL_LIB: the tests'
    // in-process stand-in) is asked about repeated devices instead of being bypassed.
    std::vector<PyrComm> comms(num_devices);
    for (uint32_t i = 0; i < num_devices; ++i) {
        comms[i].rank = (int)i, comms[i].num_ranks = (int)num_devices, comms[i].device = devices[i], comms[i].owns_comm = false;
    }
    bool use_rccl = distinct && num_devices > 1;
    if (!use_rccl && num_devices > 1) {
        const char* named = std::getenv("PYRITE_RCCL_LIB");
        use_rccl = named && *named;
    }
    static std::mutex cache_mutex;
    static std::map<std::vector<int>, std::vector<ncclComm_t>> cache;
    if (use_rccl) {
        Rccl* lib;
        int rc = rccl_ready(lib);
        if (rc != PYR_OK) return rc;
        std::lock_guard<std::mutex> lock(cache_mutex);
        auto it = cache.find(devices);
        if (it == cache.end()) {
            std::vector<ncclComm_t> made(num_devices);
            ncclResult_t r = lib->CommInitAll(made.data(), (int)num_devices, devices.data());
            if (r != ncclSuccess) return api_fail(PYR_ERR_DEVICE, nccl_message(lib, "ncclCommInitAll", r));
            it = cache.emplace(devices, made).first;
        }
        for (uint32_t i = 0; i < num_devices; ++i) comms[i].comm = it->second[i];
    }

    const size_t film_bytes = (size_t)film->width * film->height * film->bins * sizeof(PyrGrain);
    PyrGrain* film_dev = nullptr;
    std::vector<hipStream_t> streams(num_devices, nullptr);
    struct Cleanup { // everything this call made, whichever way it ends
        std::vector<PyrComm>& comms;
        std::vector<hipStream_t>& streams;
        std::vector<int>& devices;
        PyrGrain*& film_dev;
        ~Cleanup() {
            for (size_t i = 0; i < comms.size(); ++i) {
                (void)hipSetDevice(devices[i]);
                if (streams[i]) (void)hipStreamDestroy(streams[i]);
                comms[i].release();
            }
            (void)hipSetDevice(devices[0]);
            if (film_dev) (void)hipFree(film_dev);
        }
    } cleanup{comms, streams, devices, film_dev};
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMalloc((void**)&film_dev, film_bytes));
    HIP_TRY(hipMemcpy(film_dev, film_inout, film_bytes, hipMemcpyHostToDevice));
    // the streams before any thread starts: a rank that cannot even get a stream must not leave its peers in a collective
    const bool threaded = use_rccl || num_devices == 1;
    for (uint32_t i = 0; i < (threaded ? num_devices : 1u); ++i) {
        HIP_TRY(hipSetDevice(devices[i]));
        HIP_TRY(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
    }

    std::vector<int> status(num_devices, PYR_OK);
    std::vector<std::string> messages(num_devices);
    int result = PYR_OK;
    std::string result_message;
    if (threaded) {
        // one host thread per device: each issues its render, its side of the gather and (rank 0) the assembly, then waits
        auto work = [&](uint32_t i) {
            status[i] = pyr_render_simple_sharded(&comms[i], scenes[i], camera, film, params, i == 0 ? film_dev : nullptr, streams[i]);
            if (status[i] != PYR_OK) messages[i] = pyr_last_error();
            // A rank whose collective failed has aborted its own communicator; its siblings may already sit in an untimed
            // hipStreamSynchronize on a gather that can no longer complete. Abort theirs from here, now -- ncclCommAbort is what
            // unblocks them -- instead of after a join() that waits for those very threads.
            if (use_rccl && comms[i].dead.load())
                for (uint32_t j = 0; j < num_devices; ++j) comms[j].abort_comm();
            (void)hipSetDevice(devices[i]);
            if (hipStreamSynchronize(streams[i]) != hipSuccess && status[i] == PYR_OK) status[i] = PYR_ERR_DEVICE, messages[i] = "hipStreamSynchronize failed";
            if (status[i] == PYR_OK && (status[i] = pyr_comm_status(&comms[i])) != PYR_OK) messages[i] = pyr_last_error();
        };
        std::vector<std::thread> pool;
        for (uint32_t i = 1; i < num_devices; ++i) pool.emplace_back(work, i);
        work(0);
        for (auto& t : pool) t.join();
        // the message of the rank that failed on its own, not of one that only heard about it
        for (uint32_t i = 0; i < num_devices; ++i)
            if (status[i] != PYR_OK && (result == PYR_OK || result_message.find("another rank failed") != std::string::npos)) result = status[i], result_message = messages[i];
        bool any_dead = false;
        for (uint32_t i = 0; i < num_devices; ++i) any_dead = any_dead || comms[i].dead;
        if (use_rccl && any_dead) { // the cached communicators are of no use any more: the next call makes new ones
            std::lock_guard<std::mutex> lock(cache_mutex);
            Rccl* lib = rccl();
            (void)lib;
            for (uint32_t i = 0; i < num_devices; ++i) comms[i].abort_comm();
            cache.erase(devices);
        }
    } else {
        // test rig (several logical ranks on one GPU): the ranks render one after the other, each as a communicator of its own
        // size-1 world over its share, and the blocks are copied and assembled here
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
            if (rc == PYR_OK) rc = scene_check_overflow(scenes[i]); // the kernels' verdict on the film they just wrote
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
    if (result != PYR_OK) return api_fail(result, result_message);
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMemcpy(film_inout, film_dev, film_bytes, hipMemcpyDeviceToHost));
    if (on_status) on_status(user, 100, message);
    return PYR_OK;
}

} // extern "C"

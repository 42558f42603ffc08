// fake_rccl.cpp -- TEST INFRASTRUCTURE: an in-process stand-in for the ten librccl entry points pyrite_amd/csrc/multi.cpp binds.
//
// Real RCCL refuses two ranks on one device, and the build's GPU box has one GPU, so the multi-rank flow of
// pyr_render_simple_multi / pyr_render_simple_sharded -- status agreement, grouped send / receive with trailers, abort on
// error -- could never execute there. Here a "rank" is a thread of the one process and all ranks may share a device: a send
// is a record in a mailbox, the matching receive turns it into a device-to-device copy on the receiver's stream (ordered
// after the sender's stream by an event, and the sender's stream after the copy by another), an all-reduce is a rendezvous of
// the ranks' host threads. Nothing here is RCCL's algorithm; what it keeps of RCCL is the CONTRACT multi.cpp relies on:
//   * a receive completes only when the peer posts the matching send (otherwise it blocks -- here: fails after a timeout,
//     FAKE_RCCL_TIMEOUT_MS, so that a protocol error fails a test instead of hanging the GPU box);
//   * send and receive counts must match (a mismatch is reported as ncclInvalidArgument);
//   * operations between ncclGroupStart and ncclGroupEnd are issued together, so a rank may send to itself.
// Loaded through PYRITE_RCCL_LIB by tests/fake_rccl/run_cases.py only; the product never sees it.
// Fault injection: FAKE_RCCL_DROP_SEND_FROM=<rank> makes that rank's sends vanish (its peer's receive then times out).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}
struct ncclComm;
typedef struct ncclComm* ncclComm_t;

namespace {

struct Message {
    const void* src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr; // recorded on the sender's stream behind everything the buffer depends on
    hipEvent_t done = nullptr;  // recorded on the receiver's stream behind the copy
    bool consumed = false, failed = false;
};

struct World {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Message>>> box; // (from, to) -> posted sends
    bool aborted = false;
    // all-reduce rendezvous
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<int32_t> accum, result;
    std::atomic<int> refs{0};
};

struct Op {
    bool send;
    void* buffer;
    size_t bytes;
    int peer;
    ncclComm* comm;
    hipStream_t stream;
};

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

std::mutex g_registry_mutex;
std::map<std::string, World*> g_registry; // unique id -> world (ncclCommInitRank)
std::atomic<uint64_t> g_next_id{1};

int timeout_ms() {
    const char* v = std::getenv("FAKE_RCCL_TIMEOUT_MS");
    return v && *v ? std::atoi(v) : 20000;
}
size_t type_bytes(ncclDataType_t t) { return t == 0 || t == 1 ? 1 : (t == 6 || t == 9 ? 2 : (t == 4 || t == 5 || t == 8 ? 8 : 4)); }

} // namespace

struct ncclComm {
    World* world;
    int rank;
};

namespace {

ncclResult_t run_ops(std::vector<Op>& ops) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms());
    const char* drop = std::getenv("FAKE_RCCL_DROP_SEND_FROM");
    std::vector<std::pair<World*, std::shared_ptr<Message>>> posted;
    ncclResult_t verdict = ncclSuccess;
    for (Op& op : ops) { // 1. post the sends
        if (!op.send) continue;
        if (drop && *drop && std::atoi(drop) == op.comm->rank) continue; // fault injection: the message is lost
        auto msg = std::make_shared<Message>();
        msg->src = op.buffer, msg->bytes = op.bytes;
        if (hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(msg->ready, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        World* w = op.comm->world;
        {
            std::lock_guard<std::mutex> lock(w->m);
            w->box[{op.comm->rank, op.peer}].push_back(msg);
        }
        w->cv.notify_all();
        posted.emplace_back(w, msg);
    }
    for (Op& op : ops) { // 2. the receives: wait for the matching send, then copy on this rank's stream
        if (op.send) continue;
        World* w = op.comm->world;
        std::shared_ptr<Message> msg;
        {
            std::unique_lock<std::mutex> lock(w->m);
            auto& queue = w->box[{op.peer, op.comm->rank}];
            if (!w->cv.wait_until(lock, deadline, [&] { return w->aborted || !queue.empty(); })) {
                verdict = ncclSystemError; // the peer never sent: real RCCL would block for ever
                break;
            }
            if (w->aborted) {
                verdict = ncclInternalError;
                break;
            }
            msg = queue.front();
            queue.pop_front();
        }
        bool ok = msg->bytes == op.bytes;
        if (!ok) verdict = ncclInvalidArgument;
        ok = ok && hipStreamWaitEvent(op.stream, msg->ready, 0) == hipSuccess;
        ok = ok && hipMemcpyAsync(op.buffer, msg->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&msg->done, hipEventDisableTiming) == hipSuccess && hipEventRecord(msg->done, op.stream) == hipSuccess;
        {
            std::lock_guard<std::mutex> lock(w->m);
            msg->consumed = true, msg->failed = !ok;
        }
        w->cv.notify_all();
        if (!ok) {
            if (verdict == ncclSuccess) verdict = ncclUnhandledCudaError;
            break;
        }
    }
    size_t k = 0;
    for (Op& op : ops) { // 3. a send is complete when its receiver has taken it: the buffer may be reused behind the copy
        if (!op.send) continue;
        if (drop && *drop && std::atoi(drop) == op.comm->rank) continue;
        World* w = posted[k].first;
        std::shared_ptr<Message> msg = posted[k].second;
        ++k;
        std::unique_lock<std::mutex> lock(w->m);
        if (!w->cv.wait_until(lock, deadline, [&] { return w->aborted || msg->consumed; })) {
            if (verdict == ncclSuccess) verdict = ncclSystemError;
            continue;
        }
        if (!msg->consumed || msg->failed) {
            if (verdict == ncclSuccess) verdict = ncclInternalError;
            continue;
        }
        lock.unlock();
        if (hipStreamWaitEvent(op.stream, msg->done, 0) != hipSuccess && verdict == ncclSuccess) verdict = ncclUnhandledCudaError;
    }
    return verdict;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id->internal, 0, sizeof(id->internal));
    const uint64_t v = g_next_id.fetch_add(1);
    std::memcpy(id->internal, "fake-rccl", 9);
    std::memcpy(id->internal + 16, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof(id.internal));
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    World*& w = g_registry[key];
    if (!w) {
        w = new World();
        w->n = nranks;
    }
    if (w->n != nranks) return ncclInvalidArgument;
    w->refs++;
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* /*devlist*/) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World* w = new World();
    w->n = ndev;
    w->refs = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{w, i};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    if (--comm->world->refs == 0) {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (auto it = g_registry.begin(); it != g_registry.end(); ++it)
            if (it->second == comm->world) {
                g_registry.erase(it);
                break;
            }
        delete comm->world;
    }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    {
        std::lock_guard<std::mutex> lock(comm->world->m);
        comm->world->aborted = true;
    }
    comm->world->cv.notify_all();
    return ncclSuccess; // the world is leaked on purpose: peers may still be inside a call that looks at it
}

ncclResult_t ncclGroupStart() {
    g_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_group_depth == 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

static ncclResult_t enqueue(bool send, void* buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->world->n) return ncclInvalidArgument;
    if (comm->world->aborted) return ncclInternalError;
    g_ops.push_back(Op{send, buffer, count * type_bytes(type), peer, comm, stream});
    if (g_group_depth == 0) {
        std::vector<Op> ops;
        ops.swap(g_ops);
        return run_ops(ops);
    }
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(true, const_cast<void*>(buffer), count, type, peer, comm, stream);
}
ncclResult_t ncclRecv(void* buffer, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(false, buffer, count, type, peer, comm, stream);
}

// int32 max only (what the status agreement uses); a rendezvous of the ranks' host threads
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!comm || type != 2 || op != 2 || count == 0) return ncclInvalidArgument;
    World* w = comm->world;
    std::vector<int32_t> mine(count);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(mine.data(), sendbuff, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<int32_t> out;
    {
        std::unique_lock<std::mutex> lock(w->m);
        if (w->aborted) return ncclInternalError;
        if (w->arrived == 0) w->accum = mine;
        else
            for (size_t i = 0; i < count && i < w->accum.size(); ++i) w->accum[i] = std::max(w->accum[i], mine[i]);
        const uint64_t my_generation = w->generation;
        if (++w->arrived == w->n) {
            w->result = w->accum;
            w->arrived = 0;
            w->generation++;
            w->cv.notify_all();
        } else {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms());
            if (!w->cv.wait_until(lock, deadline, [&] { return w->aborted || w->generation != my_generation; })) {
                w->arrived--; // withdraw
                return ncclSystemError;
            }
            if (w->aborted) return ncclInternalError;
        }
        out = w->result;
    }
    if (hipMemcpy(recvbuff, out.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: a HIP call failed";
    case ncclSystemError: return "fake rccl: timed out waiting for the peer (real RCCL would block for ever)";
    case ncclInternalError: return "fake rccl: the communicator was aborted";
    case ncclInvalidArgument: return "fake rccl: invalid argument (send / receive sizes differ?)";
    default: return "fake rccl: invalid usage";
    }
}

} // extern "C"

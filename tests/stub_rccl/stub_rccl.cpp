/*
 * stub_rccl.cpp — TEST RIG, not product code.  A stand-in for the eight RCCL entry points raytracer_amd/csrc/mirt_node.hip
 * looks up with dlsym, built as tests/stub_rccl/librccl.so.1.  It runs every "rank" of a single-process communicator with
 * plain HIP copies ordered by events, and does not mind several ranks sharing one GPU, so the N-rank logic of mirt_node
 * (band partition, exchange slots and offsets, K4 scatter, statistics) can be exercised on a one-GPU box
 * (tests/test_gpu_node.py::test_node_n_ranks_on_one_gpu_through_the_stub).  It says nothing about xGMI or about RCCL's own
 * behaviour; it implements the documented semantics of grouped ncclSend / ncclRecv and of ncclAllGather and nothing else.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

struct ncclComm {
    int rank, nranks, dev;
};

namespace {
struct P2p { int from, to; void* buf; size_t bytes; hipStream_t stream; int dev; bool done; };
struct Gather { int rank, nranks; const void* send; void* recv; size_t bytes; hipStream_t stream; int dev; };
std::mutex g_lock;
int g_depth = 0;
std::vector<P2p> g_sends, g_recvs;
std::vector<Gather> g_gathers;

size_t elem(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
#define H(expr) do { if ((expr) != hipSuccess) return ncclUnhandledCudaError; } while (0)

/* dst stream waits for what src stream has queued so far */
ncclResult_t after(hipStream_t dst, int dst_dev, hipStream_t src, int src_dev) {
    hipEvent_t ev;
    H(hipSetDevice(src_dev));
    H(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    H(hipEventRecord(ev, src));
    H(hipSetDevice(dst_dev));
    H(hipStreamWaitEvent(dst, ev, 0));
    H(hipEventDestroy(ev)); /* released once it has completed */
    return ncclSuccess;
}

ncclResult_t flush() {
    ncclResult_t rc = ncclSuccess;
    for (P2p& r : g_recvs) {
        P2p* s = nullptr;
        for (P2p& c : g_sends)
            if (!c.done && c.from == r.from && c.to == r.to) { s = &c; break; }
        if (!s || s->bytes != r.bytes) { rc = ncclInvalidUsage; break; }
        s->done = true;
        if ((rc = after(r.stream, r.dev, s->stream, s->dev)) != ncclSuccess) break;
        if (hipMemcpyAsync(r.buf, s->buf, r.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        if ((rc = after(s->stream, s->dev, r.stream, r.dev)) != ncclSuccess) break; /* the send is over when the data has moved */
    }
    if (rc == ncclSuccess)
        for (P2p& c : g_sends)
            if (!c.done) rc = ncclInvalidUsage; /* a send nobody receives would hang the real library */
    if (rc == ncclSuccess && !g_gathers.empty()) {
        const int n = g_gathers[0].nranks;
        if ((int)g_gathers.size() != n) rc = ncclInvalidUsage;
        for (const Gather& r : g_gathers) {
            if (rc != ncclSuccess) break;
            for (const Gather& s : g_gathers) {
                if (s.bytes != r.bytes) { rc = ncclInvalidUsage; break; }
                if ((rc = after(r.stream, r.dev, s.stream, s.dev)) != ncclSuccess) break;
                if (hipMemcpyAsync((unsigned char*)r.recv + (size_t)s.rank * r.bytes, s.send, r.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess) {
                    rc = ncclUnhandledCudaError;
                    break;
                }
            }
        }
        for (const Gather& s : g_gathers) /* nobody's send buffer is free before every reader is through */
            for (const Gather& r : g_gathers)
                if (rc == ncclSuccess && r.rank != s.rank) rc = after(s.stream, s.dev, r.stream, r.dev);
    }
    g_sends.clear();
    g_recvs.clear();
    g_gathers.clear();
    return rc;
}
} /* namespace */

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{i, ndev, devlist ? devlist[i] : i};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "stub: invalid usage" : r == ncclUnhandledCudaError ? "stub: HIP error" : "stub: error"; }
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> g(g_lock); ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> g(g_lock);
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? flush() : ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    std::lock_guard<std::mutex> g(g_lock);
    if (!c || peer < 0 || peer >= c->nranks || !elem(t)) return ncclInvalidArgument;
    g_sends.push_back({c->rank, peer, (void*)buf, count * elem(t), st, c->dev, false});
    return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    std::lock_guard<std::mutex> g(g_lock);
    if (!c || peer < 0 || peer >= c->nranks || !elem(t)) return ncclInvalidArgument;
    g_recvs.push_back({peer, c->rank, buf, count * elem(t), st, c->dev, false});
    return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    std::lock_guard<std::mutex> g(g_lock);
    if (!c || !elem(t)) return ncclInvalidArgument;
    g_gathers.push_back({c->rank, c->nranks, send, recv, count * elem(t), st, c->dev});
    return g_depth ? ncclSuccess : flush();
}

} /* extern "C" */

/*
 * mirt_node.hip — one host thread drives N GPUs of one node (SURVEY.md §8b "one host thread drives N devices", §8e).
 *
 * The seam is still Camera::render(&self, &Scene) (reference src/camera/camera.rs:232): a drop-in render call that wants the
 * whole node creates a mirt_node instead of a mirt_scene and calls mirt_node_render.  Inside:
 *   - one scene replica, one stream and one band buffer per device (the scene is ~100 KB - 10 MB: replicated, not sharded);
 *   - the frame is cut into interleaved strips of mirt_opts.strip_rows rows, strip s -> device s mod N; every device renders
 *     its band compactly with the very kernels of mirt_render_device (MIRT_FLAG_COMPACT_ROWS);
 *   - ONE exchange step over xGMI with RCCL (ncclCommInitAll, single process): either every peer sends its band to the root
 *     (grouped ncclSend / ncclRecv: each peer uses its own direct link to the root) or a ring ncclAllGather, selectable per
 *     call and both timed, as SURVEY.md §5/§8e asks;
 *   - K4 (mirt_deinterleave_device) scatters the bands into frame order on the root.
 * The image is the one-GPU image bit for bit (counter RNG keyed on the global pixel; the work decomposition follows the
 * whole frame), whatever N and whichever exchange.
 *
 * RCCL is looked up at run time (dlopen of librccl.so.1: the copy the process already holds - PyTorch ships one - or the
 * ROCm one), so libmirt.so has no link-time dependency on it and single-GPU users never load it.  This file uses only the
 * public C ABI of include/mirt.h plus the HIP runtime.
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> /* types and prototypes only; nothing is linked */

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mirt.h"

void mirt_internal_set_error(const char* msg); /* mirt_api.hip: thread-local text behind mirt_last_error() */

namespace {

int nfail(int code, const std::string& msg) { /* the text goes where mirt_last_error() finds it */
    mirt_internal_set_error(msg.c_str());
    return code;
}

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    std::string error;
};
Rccl* rccl() { /* loaded once per process, on the first node that asks for it */
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    });
    return &r;
}

struct Dev {
    int id = 0;
    mirt_scene* scene = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;            /* band rendered (+ exchanged) on this device's stream */
    void* band[2] = {nullptr, nullptr};   /* compact band: [0] linear reals, [1] bytes */
    size_t band_bytes[2] = {0, 0};
    void* all[2] = {nullptr, nullptr};    /* N bands (root always; every device for the all-gather) */
    size_t all_bytes[2] = {0, 0};
    ncclComm_t comm = nullptr;
};

} /* namespace */

struct mirt_node {
    std::vector<Dev> dev;
    bool comms = false;
    void* frame[2] = {nullptr, nullptr}; /* root: full frames behind the host-buffer entry point */
    size_t frame_bytes[2] = {0, 0};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; /* root stream: before / after the exchange, after K4 */
    std::mutex lock;
};

#define NODE_HIP(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return nfail(MIRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)
#define NODE_NCCL(expr)                                                                                        \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return nfail(MIRT_ERR_HIP, std::string(#expr) + ": " + R->GetErrorString(r_));  \
    } while (0)

static int grow(void** buf, size_t* have, size_t need) {
    if (need <= *have) return MIRT_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    if (hipMalloc(buf, need) != hipSuccess) return nfail(MIRT_ERR_NOMEM, "hipMalloc failed (" + std::to_string(need) + " bytes of band / frame buffer)");
    *have = need;
    return MIRT_OK;
}

extern "C" {

int mirt_node_create(const mirt_scene_desc* desc, const int* device_ids, uint32_t n_devices, uint32_t flags, mirt_node** out) {
    if (!desc || !out || !device_ids) return nfail(MIRT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (flags & ~(uint32_t)MIRT_NODE_SHARED_DEVICES) return nfail(MIRT_ERR_INVALID, "unknown mirt_node_create flag");
    if (n_devices == 0 || n_devices > MIRT_NODE_MAX_DEVICES) return nfail(MIRT_ERR_INVALID, "n_devices must be 1.." + std::to_string(MIRT_NODE_MAX_DEVICES));
    const int ndev = mirt_device_count();
    for (uint32_t i = 0; i < n_devices; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev) return nfail(MIRT_ERR_NO_DEVICE, "device id " + std::to_string(device_ids[i]) + " out of range");
        for (uint32_t j = 0; j < i; ++j)
            if (device_ids[j] == device_ids[i] && !(flags & MIRT_NODE_SHARED_DEVICES))
                return nfail(MIRT_ERR_INVALID, "device ids must be distinct (RCCL needs one rank per GPU)");
    }
    mirt_node* n = new (std::nothrow) mirt_node();
    if (!n) return nfail(MIRT_ERR_NOMEM, "out of host memory");
    n->dev.resize(n_devices);
    for (uint32_t i = 0; i < n_devices; ++i) {
        Dev& d = n->dev[i];
        d.id = device_ids[i];
        int rc = mirt_scene_create(desc, d.id, &d.scene); /* the scene is replicated */
        hipError_t e = hipSuccess;
        if (!rc) {
            e = hipSetDevice(d.id);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&d.done, hipEventDisableTiming);
        }
        if (rc || e != hipSuccess) {
            std::string msg = rc ? std::string(mirt_last_error()) : std::string(hipGetErrorString(e));
            mirt_node_destroy(n);
            return nfail(rc ? rc : MIRT_ERR_HIP, "device " + std::to_string(device_ids[i]) + ": " + msg);
        }
    }
    hipError_t e = hipSetDevice(n->dev[0].id);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&n->ev[k]);
    if (e != hipSuccess) { mirt_node_destroy(n); return nfail(MIRT_ERR_HIP, hipGetErrorString(e)); }
    *out = n;
    return MIRT_OK;
}

void mirt_node_destroy(mirt_node* n) {
    if (!n) return;
    Rccl* R = n->comms ? rccl() : nullptr;
    for (Dev& d : n->dev) {
        (void)hipSetDevice(d.id);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm && R && R->CommDestroy) (void)R->CommDestroy(d.comm);
        for (int k = 0; k < 2; ++k) {
            if (d.band[k]) (void)hipFree(d.band[k]);
            if (d.all[k]) (void)hipFree(d.all[k]);
        }
        if (d.done) (void)hipEventDestroy(d.done);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.scene) mirt_scene_destroy(d.scene);
    }
    if (!n->dev.empty()) (void)hipSetDevice(n->dev[0].id);
    for (int k = 0; k < 2; ++k) if (n->frame[k]) (void)hipFree(n->frame[k]);
    for (int k = 0; k < 4; ++k) if (n->ev[k]) (void)hipEventDestroy(n->ev[k]);
    delete n;
}

uint32_t mirt_node_device_count(const mirt_node* n) { return n ? (uint32_t)n->dev.size() : 0u; }

} /* extern "C" */

/* The body of mirt_node_render_device; the caller holds n->lock (mirt_node_render keeps it from growing its frames until it has
 * copied them back, so that no second thread frees a frame in between). */
static int node_render_locked(mirt_node* n, const mirt_camera* cam, const mirt_opts* opts, uint32_t gather, void* d_rgb_linear, uint8_t* d_rgb8,
                              mirt_node_stats* stats) {
    if (gather != MIRT_GATHER_TO_ROOT && gather != MIRT_GATHER_ALLGATHER) return nfail(MIRT_ERR_INVALID, "unknown gather kind");
    if (opts->flags & (MIRT_FLAG_COMPACT_ROWS | MIRT_FLAG_DEFER_STATS)) return nfail(MIRT_ERR_INVALID, "the node sets MIRT_FLAG_COMPACT_ROWS / MIRT_FLAG_DEFER_STATS itself");
    if (opts->part_count > 1 || opts->part_index != 0) return nfail(MIRT_ERR_INVALID, "the node partitions the frame itself (part_index / part_count must be 0 / 0 or 1)");
    if (opts->precision != MIRT_F32 && opts->precision != MIRT_F64) return nfail(MIRT_ERR_INVALID, "mirt_opts.precision must be 32 or 64");
    const auto wall0 = std::chrono::steady_clock::now();
    const uint32_t N = (uint32_t)n->dev.size(), W = cam->img_width, H = cam->img_height, S = opts->strip_rows ? opts->strip_rows : 8u;
    if (W == 0 || H == 0) return nfail(MIRT_ERR_INVALID, "empty image");
    const size_t rsz = opts->precision == MIRT_F64 ? 8 : 4;
    const bool want[2] = {d_rgb_linear != nullptr, d_rgb8 != nullptr};
    const size_t px_bytes[2] = {3 * rsz, 3};
    const uint32_t rows0 = mirt_part_rows(H, S, 0, N); /* the largest band: every band buffer and exchange slot has this size */
    Rccl* R = rccl();
    if (!R->error.empty()) return nfail(MIRT_ERR_HIP, R->error);
    if (!n->comms) { /* first render: one communicator per device, single process */
        std::vector<int> ids(N);
        std::vector<ncclComm_t> comms(N);
        for (uint32_t i = 0; i < N; ++i) ids[i] = n->dev[i].id;
        NODE_NCCL(R->CommInitAll(comms.data(), (int)N, ids.data()));
        for (uint32_t i = 0; i < N; ++i) n->dev[i].comm = comms[i];
        n->comms = true;
    }
    /* buffers */
    for (uint32_t i = 0; i < N; ++i) {
        Dev& d = n->dev[i];
        NODE_HIP(hipSetDevice(d.id));
        for (int k = 0; k < 2; ++k) {
            if (!want[k]) continue;
            int rc = grow(&d.band[k], &d.band_bytes[k], (size_t)rows0 * W * px_bytes[k]);
            if (!rc && (i == 0 || gather == MIRT_GATHER_ALLGATHER)) rc = grow(&d.all[k], &d.all_bytes[k], (size_t)N * rows0 * W * px_bytes[k]);
            if (rc) return rc;
        }
    }
    /* every device renders its band */
    mirt_opts o = *opts;
    o.strip_rows = S;
    o.part_count = N;
    o.flags |= MIRT_FLAG_COMPACT_ROWS | MIRT_FLAG_DEFER_STATS;
    /* From the first launch on there is ONE way out on failure: every replica that took a launch (MIRT_FLAG_DEFER_STATS) has its
     * statistics collected and its stream drained first - a handle left `pending` refuses every later render, so one transient
     * HIP / RCCL error would otherwise make the node unusable for good. */
    uint32_t launched = 0;
    auto bail = [&](int code, const std::string& msg) {
        for (uint32_t j = 0; j < launched; ++j) {
            (void)mirt_render_wait(n->dev[j].scene, nullptr);
            if (hipSetDevice(n->dev[j].id) == hipSuccess) (void)hipStreamSynchronize(n->dev[j].stream);
        }
        return nfail(code, msg);
    };
#define NODE_HIP_L(expr)                                                                                       \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return bail(MIRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)
    for (uint32_t i = 0; i < N; ++i) {
        Dev& d = n->dev[i];
        o.part_index = i;
        int rc = mirt_render_device(d.scene, cam, &o, want[0] ? d.band[0] : nullptr, want[1] ? (uint8_t*)d.band[1] : nullptr, d.stream, nullptr);
        if (rc) return bail(rc, "device " + std::to_string(d.id) + ": " + std::string(mirt_last_error()));
        launched = i + 1;
    }
    /* the one exchange step, on each device's own stream behind its band */
    Dev& root = n->dev[0];
    NODE_HIP_L(hipSetDevice(root.id));
    NODE_HIP_L(hipEventRecord(n->ev[0], root.stream));
    for (int k = 0; k < 2; ++k) {
        if (!want[k]) continue;
        const size_t count = (size_t)rows0 * W * px_bytes[k]; /* bytes: the bands travel as ncclUint8, whatever they hold */
        ncclResult_t gr = R->GroupStart();
        if (gr != ncclSuccess) return bail(MIRT_ERR_HIP, std::string("ncclGroupStart: ") + R->GetErrorString(gr));
        for (uint32_t i = 0; i < N && gr == ncclSuccess; ++i) {
            Dev& d = n->dev[i];
            if (hipSetDevice(d.id) != hipSuccess) { gr = ncclUnhandledCudaError; break; }
            if (gather == MIRT_GATHER_ALLGATHER) {
                gr = R->AllGather(d.band[k], d.all[k], count, ncclUint8, d.comm, d.stream);
            } else {
                gr = R->Send(d.band[k], count, ncclUint8, 0, d.comm, d.stream); /* the root sends to itself as well: one code path */
                if (i == 0)
                    for (uint32_t r = 0; r < N && gr == ncclSuccess; ++r)
                        gr = R->Recv((unsigned char*)root.all[k] + (size_t)r * count, count, ncclUint8, (int)r, root.comm, root.stream);
            }
        }
        ncclResult_t ge = R->GroupEnd(); /* always: a group left open would swallow the next call's collectives */
        if (gr != ncclSuccess || ge != ncclSuccess) return bail(MIRT_ERR_HIP, std::string("RCCL exchange: ") + R->GetErrorString(gr != ncclSuccess ? gr : ge));
    }
    NODE_HIP_L(hipSetDevice(root.id));
    NODE_HIP_L(hipEventRecord(n->ev[1], root.stream));
    /* K4 on the root */
    int rc = MIRT_OK;
    if (want[0]) rc = mirt_deinterleave_device(root.all[0], (size_t)rows0 * W * px_bytes[0], N, S, W, H, (uint32_t)px_bytes[0], d_rgb_linear, root.stream);
    if (!rc && want[1]) rc = mirt_deinterleave_device(root.all[1], (size_t)rows0 * W * px_bytes[1], N, S, W, H, (uint32_t)px_bytes[1], d_rgb8, root.stream);
    if (rc) return bail(rc, std::string(mirt_last_error()));
    NODE_HIP_L(hipEventRecord(n->ev[2], root.stream));
#undef NODE_HIP_L
    /* wait: peers first (their sends), then the root; collect the per-device statistics.  Every replica is collected whatever
     * the others return (nothing stays pending); a replica that failed contributes zeros. */
    mirt_node_stats local;
    memset(&local, 0, sizeof(local));
    local.n_devices = N;
    local.gather = gather;
    int first_rc = MIRT_OK;
    std::string first_msg;
    for (uint32_t i = 0; i < N; ++i) {
        Dev& d = n->dev[i];
        mirt_stats st;
        memset(&st, 0, sizeof(st));
        int rc2 = mirt_render_wait(d.scene, &st);
        if (rc2) memset(&st, 0, sizeof(st));
        {
            hipError_t e = hipSetDevice(d.id);
            if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
            if (e != hipSuccess && !rc2) rc2 = nfail(MIRT_ERR_HIP, hipGetErrorString(e));
        }
        if (rc2 && !first_rc) { first_rc = rc2; first_msg = mirt_last_error(); }
        local.segments += st.segments;
        local.nan_pixels += st.nan_pixels;
        local.band_kernel_ms[i] = st.kernel_ms;
        local.band_segments[i] = st.segments;
    }
    if (first_rc) return nfail(first_rc, first_msg);
    float ms = 0.f;
    NODE_HIP(hipSetDevice(root.id));
    NODE_HIP(hipEventElapsedTime(&ms, n->ev[0], n->ev[1]));
    local.gather_ms = ms; /* on the root's stream: includes waiting for the slowest peer's band */
    NODE_HIP(hipEventElapsedTime(&ms, n->ev[1], n->ev[2]));
    local.scatter_ms = ms;
    local.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = local;
    return MIRT_OK;
}

extern "C" {

/* d_rgb_linear / d_rgb8: device pointers on the ROOT device (device_ids[0]), full frames; either may be NULL. */
int mirt_node_render_device(mirt_node* n, const mirt_camera* cam, const mirt_opts* opts, uint32_t gather, void* d_rgb_linear, uint8_t* d_rgb8,
                            mirt_node_stats* stats) {
    if (!n || !cam || !opts) return nfail(MIRT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(n->lock);
    return node_render_locked(n, cam, opts, gather, d_rgb_linear, d_rgb8, stats);
}

/* host buffers: what a drop-in Camera::render calls */
int mirt_node_render(mirt_node* n, const mirt_camera* cam, const mirt_opts* opts, uint32_t gather, void* rgb_linear, uint8_t* rgb8, mirt_node_stats* stats) {
    if (!n || !cam || !opts) return nfail(MIRT_ERR_INVALID, "null argument");
    if (!rgb_linear && !rgb8) return nfail(MIRT_ERR_INVALID, "no output requested");
    const size_t npx = (size_t)cam->img_width * cam->img_height, rsz = opts->precision == MIRT_F64 ? 8 : 4;
    const size_t bytes[2] = {npx * 3 * rsz, npx * 3};
    void* d[2] = {nullptr, nullptr};
    /* the lock is held from growing the node's frames to copying them back: a second thread on the same node waits here
     * instead of freeing or overwriting a frame this call still reads */
    std::lock_guard<std::mutex> guard(n->lock);
    if (hipSetDevice(n->dev[0].id) != hipSuccess) return nfail(MIRT_ERR_HIP, "hipSetDevice failed");
    if (rgb_linear) { int rc = grow(&n->frame[0], &n->frame_bytes[0], bytes[0]); if (rc) return rc; d[0] = n->frame[0]; }
    if (rgb8) { int rc = grow(&n->frame[1], &n->frame_bytes[1], bytes[1]); if (rc) return rc; d[1] = n->frame[1]; }
    mirt_node_stats local;
    int rc = node_render_locked(n, cam, opts, gather, d[0], (uint8_t*)d[1], &local);
    if (rc) return rc;
    if (hipSetDevice(n->dev[0].id) != hipSuccess) return nfail(MIRT_ERR_HIP, "hipSetDevice failed");
    if (rgb_linear && hipMemcpy(rgb_linear, d[0], bytes[0], hipMemcpyDeviceToHost) != hipSuccess) return nfail(MIRT_ERR_HIP, "copy back failed");
    if (rgb8 && hipMemcpy(rgb8, d[1], bytes[1], hipMemcpyDeviceToHost) != hipSuccess) return nfail(MIRT_ERR_HIP, "copy back failed");
    if (stats) *stats = local;
    if (rgb8 && local.nan_pixels) return nfail(MIRT_ERR_NAN, "a NaN reached the quantiser (the reference panics: rgb.rs:33)");
    return MIRT_OK;
}

} /* extern "C" */

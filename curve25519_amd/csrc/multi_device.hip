// curve25519_amd/csrc/multi_device.hip -- the C-ABI multi-GPU entry points (include/curve25519_amd.h, "Multi-GPU").
//
// SURVEY.md 8(e) / BASELINE.json north_star: the batch shards embarrassingly over the GPUs of one node -- element i never
// looks at element j -- and the only exchange step is ONE gather of every GPU's result rows to the root over xGMI
// (RCCL ncclGather, /opt/rocm/include/rccl/rccl.h:745).  The C counterpart of the one-process-per-GPU
// torch.distributed layer in curve25519_amd/sharded.py (which bench.py uses), in one process:
//   1. contiguous shards: device d owns elements [n*d/D, n*(d+1)/D);
//   2. ONE WORKER THREAD PER DEVICE, alive as long as the handle: it binds to its device and runs its shard through the
//      same pinned, pieced pipeline as the single-GPU *_batch entry points (host_pipeline.hpp: stage-in threads, upload /
//      kernel / download streams), so every device's uploads run over its own PCIe link at the same time and no copy
//      ever leaves from or lands in pageable memory.  Gathered results stay on the device (Arr::dev);
//   3. one grouped ncclGather per output array to devices[0] (rows padded to the largest shard, every rank sends the
//      same count), then the root's worker streams the gathered slab to the caller through the same pipeline.
// RCCL is loaded with dlopen on first use, so single-GPU users of the library do not pay for (or need) it; without its
// header the few declarations used here are spelled out below, so the library builds on a machine that lacks RCCL.
#include "capi_common.hpp"
#include "host_pipeline.hpp"

#include "../../include/curve25519_amd.h"

#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else                                       // the subset of rccl.h this file calls (rccl.h:36-60, :232, :745, :880-890)
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root,
                        ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <functional>
#include <memory>

using c25519_host::Arr;
using c25519_host::bad_arg;
using c25519_host::last_error;
using c25519_host::run_batch;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int load_rccl(Rccl& r)
{
    // a copy already mapped into the process (e.g. torch's) wins; otherwise the ROCm installation's
    const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (const char* n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) {
        last_error() = std::string("RCCL not found (librccl.so): ") + dlerror();
        return (int)hipErrorSharedObjectInitFailed;
    }
#define SYM(f) if (!(r.f = (decltype(r.f))dlsym(r.handle, "nccl" #f))) { last_error() = "RCCL lacks nccl" #f; return (int)hipErrorSharedObjectSymbolNotFound; }
    SYM(CommInitAll) SYM(CommDestroy) SYM(Gather) SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
    return 0;
}

constexpr int MAX_ARR = 5;

// One per device: a thread bound to that device for the life of the handle (its thread-local ThreadState -- streams,
// pinned and device staging, work scratch -- is created on first use and reused by every call), fed one job at a time.
struct Worker {
    int device = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, stop = false;
    int rc = 0;
    std::string err;

    void start(int dev)
    {
        device = dev;
        th = std::thread([this] {
            (void)hipSetDevice(device);
            for (;;) {
                std::function<int()> j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return has_job || stop; });
                    if (stop) break;
                    j = std::move(job);
                    has_job = false;
                }
                const int r = j();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    rc = r;
                    err = r ? last_error() : std::string();      // the error text is thread-local: carry it to the caller
                    done = true;
                }
                cv.notify_all();
            }
            c25519_amd_thread_release();                         // staging zeroed and freed on the worker's own device
        });
    }
    void submit(std::function<int()> j)
    {
        { std::lock_guard<std::mutex> lk(mu); job = std::move(j); has_job = true; done = false; }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        if (rc) last_error() = err;
        return rc;
    }
    void shutdown()
    {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        th.join();
    }
};

}  // namespace

struct c25519_amd_multi {
    Rccl rccl;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;          // the gather's stream on each device
    std::vector<ncclComm_t> comm;
    std::vector<std::unique_ptr<Worker>> worker;
    std::vector<void*> buf[MAX_ARR];          // per device: the shard's rows of gathered output array a (grow-only)
    std::vector<size_t> cap[MAX_ARR];
    void* gathered[MAX_ARR] = {};             // on devices[0]: D x (largest shard) rows of output array a
    size_t gcap[MAX_ARR] = {};
};

namespace {

#define NCCL_TRY(m, expr)                                                                 \
    do {                                                                                  \
        ncclResult_t r_ = (expr);                                                         \
        if (r_ != ncclSuccess) {                                                          \
            last_error() = std::string(#expr ": ") + (m)->rccl.GetErrorString(r_);        \
            return (int)hipErrorUnknown;                                                  \
        }                                                                                 \
    } while (0)

// grow-only device buffer on the current device; the old one held results (shared secrets, signatures): zeroed first
int reserve(void*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap) return 0;
    if (p) {
        C25519_TRY(hipMemset(p, 0, cap));
        C25519_TRY(hipFree(p));
        p = nullptr; cap = 0;
    }
    const size_t want = bytes < 4096 ? 4096 : bytes;
    C25519_TRY(hipMalloc(&p, want));
    C25519_TRY(hipMemset(p, 0, want));               // pad rows of an uneven last shard travel through the gather: defined bytes
    cap = want;
    return 0;
}

struct MArr {
    const void* in;      // host source (nullptr: output only)
    void* out;           // host destination (nullptr: input only)
    size_t elem;         // bytes per element
    bool gather;         // output travels through the RCCL gather to the root (else it is read back from its own device)
};

// shard; per device (worker thread): pipeline(upload, launch(d, device pointers, count, stream)), results resident;
// gather; the root's worker downloads the slab
template <typename Launch>
int run_multi(c25519_amd_multi* m, size_t n, const MArr* arr, int na, Launch launch)
{
    const int D = (int)m->dev.size();
    // C25519_AMD_MULTI_FORCE_GATHER=1: a one-device handle takes the gather path too (how the tests run the N > 1 code --
    // resident results, grouped ncclGather, slab download -- on a one-GPU box)
    const bool gathers = D > 1 || getenv("C25519_AMD_MULTI_FORCE_GATHER") != nullptr;
    int prev = 0;
    C25519_TRY(hipGetDevice(&prev));
    std::vector<size_t> lo(D + 1);
    size_t rows = 0;                                                  // largest shard
    for (int d = 0; d <= D; d++) lo[d] = n * (size_t)d / (size_t)D;
    for (int d = 0; d < D; d++) rows = lo[d + 1] - lo[d] > rows ? lo[d + 1] - lo[d] : rows;
    auto body = [&]() -> int {
        // 1. every device at once: its worker pipelines the shard's pieces (pinned staging, upload / kernel / download
        //    streams) and leaves the gathered outputs in buf[a][d]
        for (int d = 0; d < D; d++) {
            const size_t cnt = lo[d + 1] - lo[d], off = lo[d];
            m->worker[d]->submit([=, &launch]() -> int {
                Arr pa[MAX_ARR];
                for (int a = 0; a < na; a++) {
                    pa[a] = Arr{ arr[a].in ? (const char*)arr[a].in + off * arr[a].elem : nullptr,
                                 arr[a].out && !arr[a].gather ? (char*)arr[a].out + off * arr[a].elem : nullptr, arr[a].elem };
                    if (arr[a].out && arr[a].gather && gathers) {
                        C25519_RC(reserve(m->buf[a][d], m->cap[a][d], arr[a].elem * rows));
                        pa[a].dev = m->buf[a][d];
                    } else if (arr[a].out && arr[a].gather) {
                        // a handle of ONE device: the gather would hand the device its own rows back, so they leave
                        // through the worker's pipeline like any output (piece by piece, under the next piece's kernels)
                        pa[a].out = (char*)arr[a].out + off * arr[a].elem;
                    }
                }
                if (!cnt) return 0;
                auto piece = [&](void** ptr, size_t c, size_t, hipStream_t st) -> int { return launch(d, ptr, c, st); };
                switch (na) {
                    case 2: return run_batch(cnt, { pa[0], pa[1] }, piece);
                    case 3: return run_batch(cnt, { pa[0], pa[1], pa[2] }, piece);
                    case 4: return run_batch(cnt, { pa[0], pa[1], pa[2], pa[3] }, piece);
                    default: return bad_arg("internal: unsupported array count");
                }
            });
        }
        int rc = 0;
        for (int d = 0; d < D; d++) { const int r = m->worker[d]->wait(); if (r && !rc) rc = r; }
        if (rc || !gathers) return rc;
        // 2. the one exchange step: every device's rows of each gathered output -> devices[0]
        for (int a = 0; a < na; a++) {
            if (!arr[a].out || !arr[a].gather || !rows) continue;
            C25519_TRY(hipSetDevice(m->dev[0]));
            C25519_RC(reserve(m->gathered[a], m->gcap[a], arr[a].elem * rows * D));
            NCCL_TRY(m, m->rccl.GroupStart());
            for (int d = 0; d < D; d++) {
                C25519_TRY(hipSetDevice(m->dev[d]));
                NCCL_TRY(m, m->rccl.Gather(m->buf[a][d], d == 0 ? m->gathered[a] : nullptr, arr[a].elem * rows, ncclUint8, 0,
                                            m->comm[d], m->stream[d]));
            }
            NCCL_TRY(m, m->rccl.GroupEnd());
        }
        for (int d = 0; d < D; d++) {
            C25519_TRY(hipSetDevice(m->dev[d]));
            C25519_TRY(hipStreamSynchronize(m->stream[d]));
        }
        // 3. the root's worker streams the gathered slab to the caller: shard d's rows sit at d * rows
        m->worker[0]->submit([=]() -> int {
            for (int a = 0; a < na; a++) {
                if (!arr[a].out || !arr[a].gather) continue;
                for (int d = 0; d < D; d++) {
                    const size_t cnt = lo[d + 1] - lo[d];
                    if (!cnt) continue;
                    Arr g{ nullptr, (char*)arr[a].out + lo[d] * arr[a].elem, arr[a].elem };
                    g.dev = (char*)m->gathered[a] + arr[a].elem * rows * d;
                    C25519_RC(run_batch(cnt, { g }, [](void**, size_t, size_t, hipStream_t) -> int { return 0; }));
                }
            }
            return 0;
        });
        return m->worker[0]->wait();
    };
    const int rc = body();
    (void)hipSetDevice(prev);
    return rc;
}

}  // namespace

extern "C" {

int c25519_amd_multi_create(c25519_amd_multi** out, const int* devices, int n_dev)
{
    if (!out || !devices || n_dev < 1) return bad_arg("c25519_amd_multi_create: bad arguments");
    int have = 0;
    C25519_TRY(hipGetDeviceCount(&have));
    for (int d = 0; d < n_dev; d++)
        if (devices[d] < 0 || devices[d] >= have) return bad_arg("c25519_amd_multi_create: no such device");
    c25519_amd_multi* m = new c25519_amd_multi();
    int prev = 0;
    (void)hipGetDevice(&prev);
    auto init = [&]() -> int {
        C25519_RC(load_rccl(m->rccl));
        m->dev.assign(devices, devices + n_dev);
        m->stream.assign(n_dev, nullptr);
        m->comm.assign(n_dev, nullptr);
        for (int a = 0; a < MAX_ARR; a++) { m->buf[a].assign(n_dev, nullptr); m->cap[a].assign(n_dev, 0); }
        for (int d = 0; d < n_dev; d++) {
            C25519_TRY(hipSetDevice(devices[d]));
            C25519_TRY(hipStreamCreateWithFlags(&m->stream[d], hipStreamNonBlocking));
        }
        NCCL_TRY(m, m->rccl.CommInitAll(m->comm.data(), n_dev, devices));
        for (int d = 0; d < n_dev; d++) {
            m->worker.emplace_back(new Worker());
            m->worker.back()->start(devices[d]);
        }
        return 0;
    };
    int rc = 0;
    try { rc = init(); } catch (const std::system_error&) { rc = bad_arg("c25519_amd_multi_create: cannot start a worker thread"); }
    (void)hipSetDevice(prev);
    if (rc) { c25519_amd_multi_destroy(m); return rc; }
    *out = m;
    return 0;
}

void c25519_amd_multi_destroy(c25519_amd_multi* m)
{
    if (!m) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto& w : m->worker) w->shutdown();
    for (size_t d = 0; d < m->dev.size(); d++) {
        (void)hipSetDevice(m->dev[d]);
        if (m->stream[d]) (void)hipStreamSynchronize(m->stream[d]);
        if (m->comm[d] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comm[d]);
        for (int a = 0; a < MAX_ARR; a++)
            if (m->buf[a][d]) { (void)hipMemset(m->buf[a][d], 0, m->cap[a][d]); (void)hipFree(m->buf[a][d]); }
        if (d == 0)
            for (int a = 0; a < MAX_ARR; a++)                       // X25519 shared secrets passed through here
                if (m->gathered[a]) { (void)hipMemset(m->gathered[a], 0, m->gcap[a]); (void)hipFree(m->gathered[a]); }
        if (m->stream[d]) (void)hipStreamDestroy(m->stream[d]);
    }
    (void)hipSetDevice(prev);
    delete m;
}

int c25519_amd_multi_device_count(const c25519_amd_multi* m) { return m ? (int)m->dev.size() : 0; }

int curve25519_dh_CreateSharedKey_multi(c25519_amd_multi* m, unsigned char* shared, const unsigned char* pk,
                                        unsigned char* sk, size_t n)
{
    if (!m || !shared || !pk || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    const MArr arr[3] = { { pk, nullptr, 32, false }, { sk, sk, 32, false }, { nullptr, shared, 32, true } };
    return run_multi(m, n, arr, 3, [&](int, void** d, size_t c, hipStream_t st) -> int {
        return curve25519_dh_CreateSharedKey_dev(d[2], d[0], d[1], c, st);
    });
}

int ed25519_SignMessage_multi(c25519_amd_multi* m, unsigned char* sig, const unsigned char* priv, const unsigned char* msg,
                              size_t msg_size, size_t n)
{
    if (!m || !sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    const MArr arr[3] = { { priv, nullptr, 64, false }, { msg, nullptr, msg_size, false }, { nullptr, sig, 64, true } };
    return run_multi(m, n, arr, 3, [&](int, void** d, size_t c, hipStream_t st) -> int {
        return ed25519_SignMessage_dev(d[2], d[0], d[1], msg_size, c, st);
    });
}

int ed25519_VerifySignature_multi(c25519_amd_multi* m, int* verdict, const unsigned char* sig, const unsigned char* pk,
                                  const unsigned char* msg, size_t msg_size, size_t n)
{
    if (!m || !verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    const MArr arr[4] = { { sig, nullptr, 64, false }, { pk, nullptr, 32, false }, { msg, nullptr, msg_size, false },
                          { nullptr, verdict, sizeof(int), true } };
    return run_multi(m, n, arr, 4, [&](int, void** d, size_t c, hipStream_t st) -> int {
        return ed25519_VerifySignature_dev(d[3], d[0], d[1], d[2], msg_size, c, st);
    });
}

}  // extern "C"

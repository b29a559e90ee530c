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
//   3. the one exchange step: a grouped ncclGather of every device's result rows to devices[0] (rows padded to the
//      largest shard, every rank sends the same count), and a second thread on the root streams the gathered rows to the
//      caller (the copies themselves: the process-wide copy threads of host_pipeline.hpp).  All of it PIECE BY PIECE: every device cuts its shard at the same rows, a piece is gathered as soon
//      as every device has enqueued its kernels (the gather streams wait for the pieces' events on the device; no worker
//      ever stops for it) and handed to the caller while the devices compute the next pieces, so only the last
//      piece's gather and download are not hidden (round 3 ran the three phases strictly one after the other).
//   c25519_amd_multi_set_gather(handle, 0) leaves the gather out: every device hands its own rows to the caller over
//   its own PCIe link (no root-link bound for host destinations); the gather mode is what north_star names and the default.
//   Virtual devices: a device list that names one device several times (or C25519_AMD_MULTI_VIRTUAL=V with a one-device
//   list) runs V workers, shards, pipelines and gather streams on that ONE device -- everything of the D > 1 path except
//   RCCL, which refuses a duplicate device: the gather of a piece is then D device-to-device copies on the gather streams.
//   It is how the D > 1 code (cut rows, pad rows, rank-major blocks, drain / copier hand-over, thread budget) runs on a
//   one-GPU box; it is not a way to go faster.
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

#include <deque>
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
constexpr int MAX_DEVICES = 64;             // entries of a handle's device list (the copier hands one block per device to the copy threads)

// One per device (plus one more on the root for the gathered rows): a thread bound to that device for the life of the
// handle (its thread-local ThreadState -- streams, pinned and device staging, work scratch -- is created on first use and
// reused by every call), fed jobs in FIFO order.  wait() returns once every job submitted so far has run, with the first
// error among them (later jobs of a failed sequence are skipped).
struct Worker {
    int device = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<int(bool)>> jobs;     // int job(bool skipped): skipped = a job before it in this sequence failed
    size_t submitted = 0, finished = 0;
    bool stop = false;
    int rc = 0;
    std::string err;

    void start(int dev)
    {
        device = dev;
        th = std::thread([this] {
            c25519_host::mark_thread_inside_a_call();     // a worker only ever runs jobs of a call that holds the gate (capi_common.hpp)
            (void)hipSetDevice(device);
            for (;;) {
                std::function<int(bool)> j;
                bool skip;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !jobs.empty() || stop; });
                    if (jobs.empty()) break;
                    j = std::move(jobs.front());
                    jobs.pop_front();
                    skip = rc != 0;
                }
                const int r = j(skip);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (r && !rc) { rc = r; err = last_error(); }     // the error text is thread-local: carry it to the caller
                    finished++;
                }
                cv.notify_all();
            }
            c25519_amd_thread_release();                         // staging zeroed and freed on the worker's own device
        });
    }
    // an ordinary job is skipped behind a failed one ...
    void submit(std::function<int()> j)
    {
        submit_always([j = std::move(j)](bool skipped) -> int { return skipped ? 0 : j(); });
    }
    // ... a job that hands something back (a pinned slot) runs either way and is told whether to do its work
    void submit_always(std::function<int(bool)> j)
    {
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(j)); submitted++; }
        cv.notify_all();
    }
    // every job submitted so far has run (or was skipped behind a failed one); returns and clears the first error
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return finished == submitted; });
        const int r = rc;
        if (r) last_error() = err;
        rc = 0;
        return r;
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
    // the root's hand-over of gathered pieces to the caller, beside worker[0]'s compute: `drain` enqueues a piece's
    // device-to-host copy on drain_stream (into a pinned slot, or straight into a page-locked destination), `copier`
    // waits for the slot and copies it out to the caller's pageable rows (the process-wide copy threads, host_pipeline.hpp)
    std::unique_ptr<Worker> drain, copier;
    hipStream_t drain_stream = nullptr;
    static constexpr int SLOTS = 4;
    struct Slot { void* pinned = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; } slot[SLOTS];
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    std::vector<hipEvent_t> gathered_ev;      // on devices[0]: piece c has arrived in `gathered`
    bool gather = true;                       // c25519_amd_multi_set_gather
    bool virtual_devices = false;             // the list names a device more than once: no RCCL, device-to-device copies
    std::vector<hipEvent_t> copy_ev;          // virtual devices: "worker d's rows of this piece are in `gathered`"
    int copy_threads = 1;
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
        C25519_RC(c25519_host::zero_device_now(p, cap));
        C25519_TRY(hipFree(p));
        p = nullptr; cap = 0;
    }
    const size_t want = bytes < 4096 ? 4096 : bytes;
    C25519_TRY(hipMalloc(&p, want));
    C25519_RC(c25519_host::zero_device_now(p, want));   // pad rows of an uneven last shard travel through the gather: defined bytes -- and
                                                        // the fill must have RUN before the call's kernels write rows into the buffer
    cap = want;
    return 0;
}

struct MArr {
    const void* in;      // host source (nullptr: output only)
    void* out;           // host destination (nullptr: input only)
    size_t elem;         // bytes per element
    bool gather;         // output travels through the RCCL gather to the root (else it is read back from its own device)
};

// shard; per device (worker thread): ONE pipeline over the shard (upload, launch(d, device pointers, count, stream)),
// gathered outputs resident; the calling thread gathers piece c as soon as every device has enqueued it and hands it to
// the root's drain thread, which streams it to the caller while the devices compute the following pieces
template <typename Launch>
int run_multi(c25519_amd_multi* m, size_t n, const MArr* arr, int na, Launch launch)
{
    C25519_API_CALL();
    const int D = (int)m->dev.size();
    // C25519_AMD_MULTI_FORCE_GATHER=1: a one-device handle takes the gather path too (how the tests run the N > 1 code --
    // resident results, piece-wise grouped ncclGather, hand-over -- on a one-GPU box)
    const bool gathers = m->gather && (D > 1 || c25519_host::tunable_or(c25519_host::T_MULTI_FORCE_GATHER, 0) != 0);
    int prev = 0;
    C25519_TRY(hipGetDevice(&prev));
    std::vector<size_t> lo(D + 1);
    size_t rows = 0, row_bytes = 0;                                   // largest shard; bytes per element over all arrays
    for (int d = 0; d <= D; d++) lo[d] = n * (size_t)d / (size_t)D;
    for (int d = 0; d < D; d++) rows = lo[d + 1] - lo[d] > rows ? lo[d + 1] - lo[d] : rows;
    for (int a = 0; a < na; a++) row_bytes += arr[a].elem;
    // every device cuts its shard at the same rows: piece c = local rows [c * chunk, min((c + 1) * chunk, shard))
    const size_t chunk = c25519_host::piece_rows(rows, row_bytes);
    const size_t P = (rows + chunk - 1) / chunk;

    std::mutex mu;
    std::condition_variable cv;
    std::vector<size_t> enq(D, 0);                                    // pieces device d has enqueued so far
    std::vector<std::vector<hipEvent_t>> piece_ev(D, std::vector<hipEvent_t>(P, nullptr));
    bool any_failed = false;

    // everything of the gather's set-up that can fail, BEFORE a worker gets a job: the jobs capture this frame's locals by
    // reference, so from the first submit on no path may leave run_multi without waiting for the workers
    bool pinned_out[MAX_ARR] = {};
    auto gather_setup = [&]() -> int {
        C25519_TRY(hipSetDevice(m->dev[0]));
        while (m->gathered_ev.size() < P) {
            hipEvent_t e = nullptr;
            C25519_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            m->gathered_ev.push_back(e);
        }
        for (int a = 0; a < na; a++)
            if (arr[a].out && arr[a].gather) {
                C25519_RC(reserve(m->gathered[a], m->gcap[a], arr[a].elem * rows * D));
                pinned_out[a] = c25519_host::host_pinned(arr[a].out, n * arr[a].elem);
            }
        return 0;
    };
    if (gathers) {
        const int rc = gather_setup();
        (void)hipSetDevice(prev);
        if (rc) return rc;
    }

    auto body = [&]() -> int {
        // 1. every device at once: its worker pipelines the shard (pinned staging, upload / kernel / download streams)
        //    and leaves the gathered outputs in buf[a][d]
        for (int d = 0; d < D; d++) {
            const size_t cnt = lo[d + 1] - lo[d], off = lo[d];
            m->worker[d]->submit([&, d, cnt, off]() -> int {
                auto run = [&]() -> int {
                    Arr pa[MAX_ARR];
                    for (int a = 0; a < na; a++) {
                        pa[a] = Arr{ arr[a].in ? (const char*)arr[a].in + off * arr[a].elem : nullptr,
                                     arr[a].out && !arr[a].gather ? (char*)arr[a].out + off * arr[a].elem : nullptr, arr[a].elem };
                        if (arr[a].out && arr[a].gather && gathers) {
                            C25519_RC(reserve(m->buf[a][d], m->cap[a][d], arr[a].elem * rows));
                            pa[a].dev = m->buf[a][d];
                            // a short shard's pad rows travel through the gather and the pinned slots: zero, not what an
                            // earlier call left there (nothing reads buf now: every call ends with the gather streams idle)
                            if (cnt < rows) C25519_RC(c25519_host::zero_device_now((char*)m->buf[a][d] + cnt * arr[a].elem, (rows - cnt) * arr[a].elem));
                        } else if (arr[a].out && arr[a].gather) {
                            // no gather (one device, or switched off): the rows leave through the worker's pipeline like any
                            // output (piece by piece, under the next piece's kernels, over this device's own link)
                            pa[a].out = (char*)arr[a].out + off * arr[a].elem;
                        }
                    }
                    if (!cnt) return 0;
                    c25519_host::concurrent_pipelines() = D;              // D pipelines share this process' copy threads
                    c25519_host::PieceHook hook;
                    hook.chunk = chunk;
                    if (gathers)
                        hook.enqueued = [&, d](size_t c, size_t, size_t, hipEvent_t ev) {
                            { std::lock_guard<std::mutex> lk(mu); piece_ev[d][c] = ev; enq[d] = c + 1; }
                            cv.notify_all();
                        };
                    auto piece = [&](void** ptr, size_t c, size_t, hipStream_t st) -> int { return launch(d, ptr, c, st); };
                    switch (na) {
                        case 2: return run_batch(cnt, { pa[0], pa[1] }, piece, &hook);
                        case 3: return run_batch(cnt, { pa[0], pa[1], pa[2] }, piece, &hook);
                        case 4: return run_batch(cnt, { pa[0], pa[1], pa[2], pa[3] }, piece, &hook);
                        default: return bad_arg("internal: unsupported array count");
                    }
                };
                const int rc = run();
                // whatever happened, the calling thread must not wait for pieces that will not come (a short shard has
                // one piece fewer than the largest; a failed pipeline stops early)
                { std::lock_guard<std::mutex> lk(mu); enq[d] = P; any_failed = any_failed || rc != 0; }
                cv.notify_all();
                return rc;
            });
        }
        int rc = 0;
        for (size_t c = 0; gathers && c < P && !rc; c++) {
            {   // every device has enqueued piece c: its event says when the rows are there
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { for (int d = 0; d < D; d++) if (enq[d] <= c) return false; return true; });
                if (any_failed) break;                                 // the workers' wait() below reports it
            }
            const size_t r0 = c * chunk, cnt = rows - r0 < chunk ? rows - r0 : chunk;
            // 2. the one exchange step, for this piece: every device's rows of each gathered output -> devices[0].
            //    Block d of piece c lands at gathered + elem * (D * r0 + d * cnt); a short shard sends pad rows.
            auto gather_piece = [&]() -> int {
                for (int d = 0; d < D; d++)
                    if (piece_ev[d][c]) {
                        C25519_TRY(hipSetDevice(m->dev[d]));
                        C25519_TRY(hipStreamWaitEvent(m->stream[d], piece_ev[d][c], 0));
                    }
                for (int a = 0; a < na; a++) {
                    if (!arr[a].out || !arr[a].gather) continue;
                    char* dst = (char*)m->gathered[a] + arr[a].elem * (size_t)D * r0;
                    if (m->virtual_devices) {                          // one physical device: RCCL would refuse the duplicates
                        for (int d = 0; d < D; d++)
                            C25519_TRY(hipMemcpyAsync(dst + arr[a].elem * cnt * d, (char*)m->buf[a][d] + r0 * arr[a].elem,
                                                      arr[a].elem * cnt, hipMemcpyDeviceToDevice, m->stream[d]));
                        continue;
                    }
                    NCCL_TRY(m, m->rccl.GroupStart());
                    for (int d = 0; d < D; d++) {
                        C25519_TRY(hipSetDevice(m->dev[d]));
                        NCCL_TRY(m, m->rccl.Gather((char*)m->buf[a][d] + r0 * arr[a].elem, d == 0 ? dst : nullptr,
                                                    arr[a].elem * cnt, ncclUint8, 0, m->comm[d], m->stream[d]));
                    }
                    NCCL_TRY(m, m->rccl.GroupEnd());
                }
                C25519_TRY(hipSetDevice(m->dev[0]));
                if (m->virtual_devices)                                // the root's stream is behind every worker's copies
                    for (int d = 1; d < D; d++) {
                        C25519_TRY(hipEventRecord(m->copy_ev[d], m->stream[d]));
                        C25519_TRY(hipStreamWaitEvent(m->stream[0], m->copy_ev[d], 0));
                    }
                C25519_TRY(hipEventRecord(m->gathered_ev[c], m->stream[0]));
                return 0;
            };
            rc = gather_piece();
            if (rc) break;
            // 3. the root hands the piece to the caller while the devices compute the following ones
            hipEvent_t ev = m->gathered_ev[c];
            m->drain->submit([&, r0, cnt, ev]() -> int {
                C25519_TRY(hipStreamWaitEvent(m->drain_stream, ev, 0));
                for (int a = 0; a < na; a++) {
                    if (!arr[a].out || !arr[a].gather) continue;
                    const size_t elem = arr[a].elem;
                    const char* src = (const char*)m->gathered[a] + elem * (size_t)D * r0;
                    auto rows_of = [&](int d) -> size_t {            // real (not pad) rows of block d of this piece
                        const size_t cnt_d = lo[d + 1] - lo[d];
                        return r0 >= cnt_d ? 0 : (r0 + cnt < cnt_d ? cnt : cnt_d - r0);
                    };
                    if (pinned_out[a]) {                               // page-locked destination: DMA straight into it
                        for (int d = 0; d < D; d++)
                            if (rows_of(d))
                                C25519_TRY(hipMemcpyAsync((char*)arr[a].out + (lo[d] + r0) * elem, src + elem * cnt * d, rows_of(d) * elem,
                                                          hipMemcpyDeviceToHost, m->drain_stream));
                        continue;
                    }
                    int k = -1;
                    {   // a free pinned slot (the copier hands them back)
                        std::unique_lock<std::mutex> lk(m->slot_mu);
                        m->slot_cv.wait(lk, [&] { for (int i = 0; i < c25519_amd_multi::SLOTS; i++) if (!m->slot[i].busy) { k = i; return true; } return false; });
                        m->slot[k].busy = true;
                    }
                    c25519_amd_multi::Slot& sl = m->slot[k];
                    auto fill = [&]() -> int {
                        const size_t bytes = elem * cnt * D;
                        if (sl.cap < bytes) {
                            if (sl.pinned) { memset(sl.pinned, 0, sl.cap); C25519_TRY(hipHostFree(sl.pinned)); sl.pinned = nullptr; sl.cap = 0; }
                            C25519_TRY(hipHostMalloc(&sl.pinned, bytes, hipHostMallocDefault));
                            sl.cap = bytes;
                        }
                        if (!sl.ev) C25519_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
                        C25519_TRY(hipMemcpyAsync(sl.pinned, src, bytes, hipMemcpyDeviceToHost, m->drain_stream));
                        C25519_TRY(hipEventRecord(sl.ev, m->drain_stream));
                        return 0;
                    };
                    const int r = fill();
                    if (r) { { std::lock_guard<std::mutex> lk(m->slot_mu); sl.busy = false; } m->slot_cv.notify_all(); return r; }
                    char* out = (char*)arr[a].out;
                    // (submit_always: behind a failed copy the remaining ones are skipped, but every one of them still hands
                    // its slot back -- or the drain thread would wait for a free slot for ever, in this call and the next)
                    m->copier->submit_always([&, k, out, elem, r0, cnt](bool skipped) -> int {
                        c25519_amd_multi::Slot& s2 = m->slot[k];
                        const hipError_t e = hipEventSynchronize(s2.ev);   // also when skipped: the DMA into the slot must have ended
                        if (e == hipSuccess && !skipped) {
                            // D blocks of real (not pad) rows, all of them to the process-wide copy threads at once
                            c25519_host::SharedCopyPool::Range rg[MAX_DEVICES];
                            int cntr = 0;
                            for (int d = 0; d < D; d++) {
                                const size_t cnt_d = lo[d + 1] - lo[d];
                                const size_t real = r0 >= cnt_d ? 0 : (r0 + cnt < cnt_d ? cnt : cnt_d - r0);
                                if (real) rg[cntr++] = { out + (lo[d] + r0) * elem, (const char*)s2.pinned + cnt * d * elem, real * elem };
                            }
                            c25519_host::SharedCopyPool::instance().copy(rg, cntr);
                        }
                        { std::lock_guard<std::mutex> lk(m->slot_mu); s2.busy = false; }
                        m->slot_cv.notify_all();
                        if (e != hipSuccess) (void)hipGetLastError();
                        return e == hipSuccess || skipped ? 0 : c25519_host::fail(e, "hipEventSynchronize(gathered piece)", __FILE__, __LINE__);
                    });
                }
                return 0;
            });
        }
        for (int d = 0; d < D; d++) { const int r = m->worker[d]->wait(); if (r && !rc) rc = r; }
        if (gathers) {
            int r = m->drain->wait();
            if (r && !rc) rc = r;
            r = m->copier->wait();
            if (r && !rc) rc = r;
            (void)hipSetDevice(m->dev[0]);
            hipError_t e = hipStreamSynchronize(m->drain_stream);     // direct copies into a page-locked destination
            if (e != hipSuccess && !rc) rc = c25519_host::fail(e, "hipStreamSynchronize(drain stream)", __FILE__, __LINE__);
            for (int d = 0; d < D; d++) {                              // the senders' side of the gathers: buf is free again
                (void)hipSetDevice(m->dev[d]);
                e = hipStreamSynchronize(m->stream[d]);
                if (e != hipSuccess && !rc) rc = c25519_host::fail(e, "hipStreamSynchronize(gather stream)", __FILE__, __LINE__);
            }
        }
        return rc;
    };
    const int rc = body();
    (void)hipSetDevice(prev);
    return rc;
}

}  // namespace

extern "C" {

int c25519_amd_multi_create(c25519_amd_multi** out, const int* devices, int n_dev)
{
    C25519_API_CALL();
    if (!out || !devices || n_dev < 1 || n_dev > MAX_DEVICES) return bad_arg("c25519_amd_multi_create: bad arguments (1..64 devices)");
    int have = 0;
    C25519_TRY(hipGetDeviceCount(&have));
    for (int d = 0; d < n_dev; d++)
        if (devices[d] < 0 || devices[d] >= have) return bad_arg("c25519_amd_multi_create: no such device");
    // virtual devices: a list that names a device twice, or C25519_AMD_MULTI_VIRTUAL=V with a one-device list
    std::vector<int> list(devices, devices + n_dev);
    const long v = c25519_host::tunable_or(c25519_host::T_MULTI_VIRTUAL, 0);
    if (n_dev == 1 && v > 1) list.assign((size_t)(v > MAX_DEVICES ? MAX_DEVICES : v), devices[0]);
    n_dev = (int)list.size();
    bool dup = false, distinct = false;
    for (int d = 0; d < n_dev; d++)
        for (int e = 0; e < d; e++) { dup = dup || list[d] == list[e]; distinct = distinct || list[d] != list[e]; }
    // ONE device named D times is D virtual devices on it (device-to-device copies in RCCL's place); a list that repeats some
    // devices and not others ([0, 0, 1]) is neither that nor a set of real devices RCCL would accept
    if (dup && distinct) return bad_arg("c25519_amd_multi_create: a device list either names distinct devices or ONE device several times");
    c25519_amd_multi* m = new c25519_amd_multi();
    int prev = 0;
    (void)hipGetDevice(&prev);
    auto init = [&]() -> int {
        m->virtual_devices = dup;
        m->dev = list;
        m->stream.assign(n_dev, nullptr);
        m->comm.assign(n_dev, nullptr);
        m->copy_ev.assign(n_dev, nullptr);
        for (int a = 0; a < MAX_ARR; a++) { m->buf[a].assign(n_dev, nullptr); m->cap[a].assign(n_dev, 0); }
        for (int d = 0; d < n_dev; d++) {
            C25519_TRY(hipSetDevice(list[d]));
            C25519_TRY(hipStreamCreateWithFlags(&m->stream[d], hipStreamNonBlocking));
            if (dup) C25519_TRY(hipEventCreateWithFlags(&m->copy_ev[d], hipEventDisableTiming));
        }
        if (!dup) {                                       // (RCCL refuses a communicator with a duplicate device)
            C25519_RC(load_rccl(m->rccl));
            NCCL_TRY(m, m->rccl.CommInitAll(m->comm.data(), n_dev, list.data()));
        }
        for (int d = 0; d < n_dev; d++) {
            m->worker.emplace_back(new Worker());
            m->worker.back()->start(list[d]);
        }
        C25519_TRY(hipSetDevice(list[0]));
        C25519_TRY(hipStreamCreateWithFlags(&m->drain_stream, hipStreamNonBlocking));
        m->drain.reset(new Worker());
        m->drain->start(list[0]);
        m->copier.reset(new Worker());
        m->copier->start(list[0]);
        m->copy_threads = c25519_host::SharedCopyPool::instance().size();   // the process-wide copy threads (host_pipeline.hpp)
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
    if (!c25519_host::runtime_alive().load()) return;         // exit() has begun: nothing to give back to a runtime that is going away
    C25519_API_CALL_OR((void)0);
    if (!m) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto& w : m->worker) w->shutdown();
    if (m->drain) m->drain->shutdown();
    if (m->copier) m->copier->shutdown();
    if (!m->dev.empty()) {
        (void)hipSetDevice(m->dev[0]);
        for (hipEvent_t e : m->gathered_ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : m->copy_ev) if (e) (void)hipEventDestroy(e);
        for (auto& sl : m->slot) {                                    // results passed through the pinned slots
            if (sl.pinned) { memset(sl.pinned, 0, sl.cap); (void)hipHostFree(sl.pinned); }
            if (sl.ev) (void)hipEventDestroy(sl.ev);
        }
        if (m->drain_stream) { (void)hipStreamSynchronize(m->drain_stream); (void)hipStreamDestroy(m->drain_stream); }
    }
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

// threads of this handle that copy memory on the host while a call runs: every device pipeline's staging helpers plus the
// root's copy-out threads (the D workers, the drain and the copier thread only enqueue and wait)
int c25519_amd_multi_helper_threads(const c25519_amd_multi* m)
{
    if (!m) return 0;
    // a one-device handle's pipeline parks stagers and drainers of its own; several devices share the process-wide copy
    // threads (their pipelines' one helper each, like the workers, the drain and the copier thread, only enqueues and waits)
    const int D = (int)m->dev.size();
    return D == 1 ? c25519_host::pipeline_helpers(1).total() : m->copy_threads;
}

int c25519_amd_multi_set_gather(c25519_amd_multi* m, int on)
{
    if (!m) return bad_arg("null handle");
    m->gather = on != 0;
    return 0;
}

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

// curve25519_amd/csrc/host_pipeline.hpp -- the host-pointer pipeline behind the *_batch entry points (engine.hip) and the
// per-device workers of the *_multi entry points (multi_device.hip): pageable caller arrays <-> pinned staging <-> device,
// in pieces, on the calling thread's streams (ThreadState, capi_common.hpp).  No arithmetic here: `launch` enqueues kernels.
#pragma once
#include "capi_common.hpp"

#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <initializer_list>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace c25519_host {

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- helper threads of the pipeline: created once per calling thread, parked between calls --------------------------
// (round 3 created and joined up to six std::threads inside every *_batch call.)  run() hands task(i) to helper i for
// i < k and returns at once; wait() blocks until all of them have come back.  The tasks reference the caller's locals, so
// every run() is paired with a wait() before those go out of scope.  A pool that could not start its threads reports
// ok() == false and the pipeline runs its pieces one after the other instead.
class HelperPool {
public:
    explicit HelperPool(int n)
    {
        try {
            for (int i = 0; i < n; i++) th_.emplace_back([this, i] { loop(i); });
        } catch (const std::system_error&) {
            shutdown();
            failed_ = true;
        }
    }
    ~HelperPool() { shutdown(); }
    bool ok() const { return !failed_; }
    int size() const { return (int)th_.size(); }
    template <typename Task>
    void run(int k, Task& task)
    {
        std::lock_guard<std::mutex> lk(mu_);
        task_ = [&task](int i) { task(i); };
        active_ = k;
        pending_ = k;
        gen_++;
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        task_ = nullptr;
    }

private:
    void loop(int i)
    {
        unsigned long seen = 0;
        for (;;) {
            std::function<void(int)> t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && i < active_); });
                if (stop_) return;
                seen = gen_;
                t = task_;
            }
            t(i);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
        th_.clear();
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::function<void(int)> task_;
    int active_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false, failed_ = false;
};

// ---- one pool of copy threads for the whole process, used when SEVERAL pipelines run beside each other (the multi-GPU
// layer: one pipeline per device) -- so that whichever phase the call is in, staging in on eight devices or handing the
// gathered rows to the caller on the root, every copy thread the CPU budget allows is copying, and the thread count does
// not grow with the number of devices.  copy() cuts its ranges into 512 KiB tasks, queues them and blocks until they are
// done; the threads that call it (a pipeline's helper, the root's copier) only wait.  Size: the CPUs this process may use
// (usable_cpus() / C25519_AMD_HELPER_THREADS) minus four for the threads that enqueue and wait, at least 2, at most 12.
class SharedCopyPool {
public:
    struct Range { void* dst; const void* src; size_t bytes; };
    // never destroyed: exit() runs static destructors while *_multi calls of other threads may still be copying through the pool
    // (the API gate's exit handler waits for them AFTER this object's destructor would have run); the threads are detached
    // and end with the process
    static SharedCopyPool& instance()
    {
        static SharedCopyPool* const pool = new SharedCopyPool();
        return *pool;
    }
    int size() const { return (int)th_.size(); }
    void copy(const Range* r, int count)
    {
        constexpr size_t CHUNK = (size_t)512 << 10;
        Batch b;
        size_t tasks = 0;
        for (int i = 0; i < count; i++) tasks += (r[i].bytes + CHUNK - 1) / CHUNK;
        if (!tasks) return;
        if (th_.empty() || tasks == 1) {                  // nobody to share with (or nothing to share)
            for (int i = 0; i < count; i++) if (r[i].bytes) memcpy(r[i].dst, r[i].src, r[i].bytes);
            return;
        }
        b.pending = tasks;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < count; i++)
                for (size_t o = 0; o < r[i].bytes; o += CHUNK)
                    q_.push_back(Task{ (char*)r[i].dst + o, (const char*)r[i].src + o, r[i].bytes - o < CHUNK ? r[i].bytes - o : CHUNK, &b });
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(b.mu);
        b.cv.wait(lk, [&] { return b.pending == 0; });
    }
    void copy(void* dst, const void* src, size_t bytes) { const Range r{ dst, src, bytes }; copy(&r, 1); }

private:
    struct Batch { std::mutex mu; std::condition_variable cv; size_t pending = 0; };
    struct Task { char* dst; const char* src; size_t bytes; Batch* batch; };
    SharedCopyPool()
    {
        long n = tunable_or(T_HELPER_THREADS, usable_cpus()) - 4;
        n = n < 2 ? 2 : n > 12 ? 12 : n;
        try {
            for (long i = 0; i < n; i++) { th_.emplace_back([this] { loop(); }); th_.back().detach(); }
        } catch (const std::system_error&) {              // fewer threads than asked for: still a pool; none: copy() copies inline
        }
    }
    ~SharedCopyPool() = delete;
    void loop()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                t = q_.front();
                q_.pop_front();
            }
            memcpy(t.dst, t.src, t.bytes);
            bool last;
            { std::lock_guard<std::mutex> lk(t.batch->mu); last = --t.batch->pending == 0; if (last) t.batch->cv.notify_all(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    bool stop_ = false;
};

// ---- host-pointer pipeline used by the *_batch entry points -------------------------------------------------------
// A call is cut into pieces; piece c uses buffer set c % SETS (pinned host + device staging), and these roles work on
// different pieces at the same time:
//     stage-in  : helper threads (pipeline_helpers(): 4, 2 or 1 by the CPUs this process may use and the pipelines that run
//                 beside this one; C25519_AMD_STAGERS) copy the caller's pageable arrays into a set's pinned buffers
//     submit    : the calling thread enqueues the piece: upload on the upload stream, the *_dev kernels on one of two
//                 kernel streams, download on the download stream, chained by events (pinned memory: hipMemcpyAsync is
//                 a real DMA).  With copies and kernels on the same stream, piece c+4's upload queued behind piece c's
//                 kernels and the device idled between rounds of four.
//     stage-out : helper threads (2, or 1; C25519_AMD_DRAINERS) wait for the set's event and copies the results out
//                 (a pipeline that is granted ONE helper has it do both: stage the first SETS pieces, then drain piece
//                 c - SETS and stage piece c in turn)
// so both CPU copies and both PCIe directions ride under the kernels of the neighbouring pieces.  Pieces are n/8 for
// big batches: two of them (2^18 lanes) fill every kernel's occupancy, and a piece cannot finish faster than one
// ladder's latency (~0.7-1.2 ms), so fewer, larger pieces in flight beat many small ones.  Eight buffer sets let a 2^20
// call stage every piece in without waiting for an earlier one to leave; four streams because the runtime drives four
// hardware queues (timelines and the rejected shapes: profiles/r02_hostapi_trace.txt, rates: profiles/r02_hostapi.txt).
struct Arr {
    const void* in;      // caller's source (nullptr: output only)
    void* out;           // caller's destination (nullptr: input only); in and out may both be set (IN/OUT array)
    size_t elem;         // bytes per element
    void* dev = nullptr; // a device-resident array of the current device instead of staging: a piece works on
                         // dev + lo * elem in place (no upload; downloaded to `out` if that is set).  The multi-GPU
                         // entry points keep a shard's results on its device this way, for the RCCL gather.
};

constexpr int MAX_STAGERS = 8, MAX_DRAINERS = 4;
inline int env_count(const char* name, int dflt, int max)
{
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= max ? v : dflt;
}

// How many helper threads a pipelined call may park, from the CPUs this process may use (usable_cpus(): affinity and cgroup
// quota, not the host's thread count; C25519_AMD_HELPER_THREADS overrides) and the number of pipelines running beside each
// other (the multi-GPU layer runs one per device and says so through concurrent_pipelines()): 4 + 2 with cores to spare
// (sign moves 160 B per 1.8 ns of kernel time: one copier per direction cannot keep up), 2 + 1, 1 + 1, or ONE helper that
// stages and drains in turn.  SEVERAL pipelines beside each other always take the last shape, and that one helper only
// orchestrates: its copies are cut into tasks for the process-wide SharedCopyPool above -- eight devices' pipelines on a
// 16-CPU container run 12 copy threads in all, not 48, and all twelve work on whichever device has something to copy.
inline int& concurrent_pipelines() { thread_local int n = 1; return n; }
struct HelperPlan { int stagers, drainers; bool combined; int total() const { return combined ? 1 : stagers + drainers; } };
inline HelperPlan pipeline_helpers(int concurrent, int reserved = 0)
{
    if (concurrent > 1) return HelperPlan{ 0, 0, true };   // one orchestrating helper per pipeline; the copies go to SharedCopyPool
    const long budget = tunable_or(T_HELPER_THREADS, usable_cpus());
    const long share = (budget - reserved) / (concurrent < 1 ? 1 : concurrent);
    HelperPlan p = share >= 16 ? HelperPlan{ 4, 2, false }
                 : share >= 3 ? HelperPlan{ 2, 1, false } : share >= 2 ? HelperPlan{ 1, 1, false } : HelperPlan{ 0, 0, true };
    if (!p.combined) {
        p.stagers = env_count("C25519_AMD_STAGERS", p.stagers, MAX_STAGERS);
        p.drainers = env_count("C25519_AMD_DRAINERS", p.drainers, MAX_DRAINERS);
    }
    return p;
}

// is [p, p + bytes) page-locked host memory the device can DMA from (hipHostMalloc / hipHostRegister /
// c25519_amd_host_register)?  Such arrays skip the staging copies: the H2D / D2H copies run on the caller's memory.
inline bool host_pinned(const void* p, size_t bytes)
{
    if (!p || !bytes) return false;
    for (const char* q : { (const char*)p, (const char*)p + bytes - 1 }) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }   // pageable
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// The kernels pick a workgroup shape from the element count they are given (narrow workgroups when the batch would not
// fill the chip, engine.hip).  The pieces of a pipelined call are small only because they are pieces -- their neighbours
// fill the chip -- so while a call is in the pipeline its TOTAL count decides the shape.
inline size_t& batch_shape_hint() { thread_local size_t total = 0; return total; }
struct ShapeHint {
    size_t saved;
    explicit ShapeHint(size_t n) : saved(batch_shape_hint()) { batch_shape_hint() = n; }
    ~ShapeHint() { batch_shape_hint() = saved; }
};

// the calling thread's parked helpers (created on the first pipelined call, joined when the thread exits or calls
// c25519_amd_thread_release())
inline std::unique_ptr<HelperPool>& helper_pool_slot() { thread_local std::unique_ptr<HelperPool> p; return p; }
inline HelperPool& helper_pool(int n)
{
    auto& p = helper_pool_slot();
    if (!p || (p->ok() && p->size() < n)) p.reset(new HelperPool(n));
    return *p;
}

// rows per piece of a pipelined call of n rows of `row` bytes: n/8 (C25519_AMD_BATCH_PIECES) in multiples of 256 for big
// batches but not below 2^16 rows -- two pieces' kernels are in flight at a time, and 2^15 lanes and fewer leave most of
// the chip idle (a 2^17-row shard of an eight-GPU call is two pieces of 2^16, not eight of 2^14) --, everything at once
// for small ones, at most 256 MiB of staging per buffer set
inline size_t piece_rows(size_t n, size_t row)
{
    static const size_t pieces = [] { const char* e = getenv("C25519_AMD_BATCH_PIECES"); int v = e ? atoi(e) : 0; return (size_t)(v >= 2 && v <= 64 ? v : 8); }();
    static const size_t floor_rows = [] { const char* e = getenv("C25519_AMD_PIECE_MIN_ROWS"); long v = e ? atol(e) : 0; return (size_t)(v >= 256 ? v : 1 << 16); }();
    size_t chunk = n;
    if (n >= ((size_t)1 << 17)) {
        chunk = round_up((n + pieces - 1) / pieces, 256);
        if (chunk < floor_rows) chunk = floor_rows;
    }
    const size_t cap = round_up(((size_t)256 << 20) / (row ? row : 1) + 1, 256);
    return chunk > cap ? cap : chunk;
}

// A call of a few elements -- the reference's own prototypes are calls of ONE -- does not copy at all: the kernels read their
// operands straight out of the pinned staging buffers (page-locked host memory is mapped into the device's address space) and
// write their results into them.  Four copy commands of ~4 us each on the device, and their submission, cost a single call
// more than the few hundred bytes cost over PCIe.  The *_dev entry points, which refuse host pointers from callers, accept them
// while this flag is up (engine.hip: check_dev_args).
constexpr size_t ZERO_COPY_MAX_ROWS = 64;
constexpr size_t ZERO_COPY_MAX_BYTES = 16384;             // ... of all arrays together: a message of KiB .. MiB is hashed byte by byte
                                                          // (sha512.cuh), twice when signing -- over PCIe that is a DMA upload's job
inline bool& zero_copy_call() { thread_local bool on = false; return on; }

// What the multi-GPU layer hangs on a device's pipeline: the piece size (so that every device cuts its shard at the same
// rows) and a call per piece once its kernels are enqueued -- `computed` is the event behind them on this device (null:
// the piece has already completed), which another stream can wait for while this pipeline keeps going.
struct PieceHook {
    size_t chunk = 0;                                      // 0: piece_rows(n, row)
    std::function<void(size_t c, size_t lo, size_t cnt, hipEvent_t computed)> enqueued;
};

template <typename Launch>
int run_batch(size_t n, std::initializer_list<Arr> arrays, Launch launch, const PieceHook* hook = nullptr)
{
    C25519_API_CALL();
    ThreadState& t = tls();
    C25519_RC(t.ensure());
    const ShapeHint shape_hint(n);
    const Arr* arr = arrays.begin();
    const int na = (int)arrays.size();
    if (na > ThreadState::SLOTS) return bad_arg("internal: too many arrays");
    size_t row = 0;
    for (int a = 0; a < na; a++) row += arr[a].elem;
    const size_t chunk = hook && hook->chunk ? hook->chunk : piece_rows(n, row);
    const size_t nchunks = (n + chunk - 1) / chunk;
    const int sets = nchunks < (size_t)ThreadState::SETS ? (int)nchunks : ThreadState::SETS;
    bool direct[ThreadState::SLOTS] = {};                  // the caller's array is pinned: no staging copy either way
    for (int a = 0; a < na && n >= 4096; a++) {            // (not worth two attribute queries per array on a tiny call)
        direct[a] = (!arr[a].in || host_pinned(arr[a].in, n * arr[a].elem)) && (!arr[a].out || host_pinned(arr[a].out, n * arr[a].elem));
        if (arr[a].in && arr[a].out && arr[a].in != arr[a].out) direct[a] = false;
    }
    bool resident_result = false;                          // a device-resident array nobody downloads: synchronise at the end
    bool any_dev = false;
    for (int a = 0; a < na; a++)
        if (arr[a].dev) {
            if (arr[a].in) return bad_arg("internal: a device-resident array has no host source");
            resident_result = resident_result || !arr[a].out;
            any_dev = true;
        }
    static const bool zero_copy_on = [] { const char* e = getenv("C25519_AMD_ZERO_COPY"); return !(e && atoi(e) == 0); }();
    const bool zero_copy = zero_copy_on && n <= ZERO_COPY_MAX_ROWS && row * n <= ZERO_COPY_MAX_BYTES && nchunks == 1 && !any_dev;
    static const bool done_word_on = [] { const char* e = getenv("C25519_AMD_DONE_WORD"); return !(e && atoi(e) == 0); }();
    if (zero_copy && n == 1 && done_word_on && !t.done_word) {
        if (hipHostMalloc((void**)&t.done_word, 64, hipHostMallocDefault) == hipSuccess) *t.done_word = 0;
        else { (void)hipGetLastError(); t.done_word = nullptr; }      // (no word: the call waits for the stream's event, as it used to)
    }
    for (int l = 0; l < sets; l++)
        for (int a = 0; a < na; a++) {
            if (!arr[a].dev && !zero_copy) C25519_RC(t.reserve_dev(l, a, arr[a].elem * chunk));
            if (!direct[a]) C25519_RC(t.reserve_host(l, a, arr[a].elem * chunk));
        }
    auto span = [&](size_t c, size_t& lo, size_t& cnt) { lo = c * chunk; cnt = (n - lo < chunk) ? n - lo : chunk; };
    const bool shared_copies = concurrent_pipelines() > 1;
    auto copy_bytes = [&](void* dst, const void* src, size_t bytes) {
        if (shared_copies) SharedCopyPool::instance().copy(dst, src, bytes);
        else memcpy(dst, src, bytes);
    };
    auto stage_in = [&](size_t c, int part, int parts) {   // rows [part, part+1) / parts of piece c
        size_t lo, cnt;
        span(c, lo, cnt);
        const size_t r0 = cnt * part / parts, r1 = cnt * (part + 1) / parts;
        const int l = (int)(c % sets);
        for (int a = 0; a < na; a++)
            if (arr[a].in && !direct[a] && (r1 - r0) * arr[a].elem)
                copy_bytes((char*)t.hbuf[l][a] + r0 * arr[a].elem, (const char*)arr[a].in + (lo + r0) * arr[a].elem, (r1 - r0) * arr[a].elem);
    };
    // one piece: upload on the upload stream, kernels on one of the two kernel streams, download on the download
    // stream, chained by events -- so the upload of a later piece never queues behind an earlier piece's kernels, and
    // two pieces' kernels (2^18 lanes: full occupancy for every pass) are in flight while others move over PCIe
    auto submit = [&](size_t c, bool one_stream) -> int {
        size_t lo, cnt;
        span(c, lo, cnt);
        const int l = (int)(c % sets);
        hipStream_t kern = t.stream[c & 1];
        hipStream_t up = one_stream ? kern : t.stream[2], down = one_stream ? kern : t.stream[3];
        void* dptr[ThreadState::SLOTS] = {};
        if (zero_copy) {                                   // (one piece, one stream: `sequential` below)
            for (int a = 0; a < na; a++) dptr[a] = t.hbuf[l][a];
            zero_copy_call() = true;
            t.done_taken = false;
            t.done_offered = done_word_on && n == 1 && t.done_word != nullptr;     // a call of one: its last kernel may signal completion itself
            const int rc = launch(dptr, cnt, lo, kern);
            t.done_offered = false;
            zero_copy_call() = false;
            C25519_RC(rc);
            C25519_TRY(hipEventRecord(t.done[l], kern));
            return 0;
        }
        for (int a = 0; a < na; a++) {
            dptr[a] = arr[a].dev ? (void*)((char*)arr[a].dev + lo * arr[a].elem) : t.dbuf[l][a];
            if (arr[a].in && cnt * arr[a].elem)
                C25519_TRY(hipMemcpyAsync(dptr[a], direct[a] ? (const char*)arr[a].in + lo * arr[a].elem : (const char*)t.hbuf[l][a],
                                          cnt * arr[a].elem, hipMemcpyHostToDevice, up));
        }
        if (!one_stream) {
            C25519_TRY(hipEventRecord(t.uploaded[l], up));
            C25519_TRY(hipStreamWaitEvent(kern, t.uploaded[l], 0));
        }
        C25519_RC(launch(dptr, cnt, lo, kern));
        if (!one_stream) {
            C25519_TRY(hipEventRecord(t.computed[l], kern));
            C25519_TRY(hipStreamWaitEvent(down, t.computed[l], 0));
            if (hook && hook->enqueued) hook->enqueued(c, lo, cnt, t.computed[l]);
        }
        for (int a = 0; a < na; a++)
            if (arr[a].out && cnt * arr[a].elem)
                C25519_TRY(hipMemcpyAsync(direct[a] ? (char*)arr[a].out + lo * arr[a].elem : (char*)t.hbuf[l][a], dptr[a],
                                          cnt * arr[a].elem, hipMemcpyDeviceToHost, down));
        C25519_TRY(hipEventRecord(t.done[l], down));
        return 0;
    };
    auto drain = [&](size_t c) -> int {
        size_t lo, cnt;
        span(c, lo, cnt);
        const int l = (int)(c % sets);
        if (zero_copy && t.done_taken) {
            // the call's last kernel stores done_seq behind its results (engine.hip: take_done_word / signal_done): spin on the
            // word; the event is the way out if the stream gets past the kernel without it (a failed launch, a fault)
            t.done_taken = false;
            for (unsigned long spins = 1;; spins++) {
                if (__atomic_load_n(t.done_word, __ATOMIC_ACQUIRE) == t.done_seq) break;
                __builtin_ia32_pause();                    // (a spin-wait hint: the sibling hyperthread keeps its issue slots)
                if ((spins & 0x3fff) == 0 && hipEventQuery(t.done[l]) != hipErrorNotReady) {
                    C25519_TRY(hipEventSynchronize(t.done[l]));
                    break;
                }
            }
        } else
            C25519_TRY(hipEventSynchronize(t.done[l]));
        for (int a = 0; a < na; a++)
            if (arr[a].out && !direct[a] && cnt * arr[a].elem) copy_bytes((char*)arr[a].out + lo * arr[a].elem, t.hbuf[l][a], cnt * arr[a].elem);
        return 0;
    };

    // every stream of the thread idle again: after an error (copies may still be writing the caller's pinned arrays or the
    // staging buffers), and when a result stays on the device with no download to wait for
    auto quiesce = [&]() {
        for (int l = 0; l < ThreadState::LANES; l++)
            if (t.stream[l]) (void)hipStreamSynchronize(t.stream[l]);
        (void)hipGetLastError();
    };
    auto sequential = [&]() -> int {                      // no helper threads: one piece after the other
        for (size_t c = 0; c < nchunks; c++) {
            stage_in(c, 0, 1);
            int rc = submit(c, true);
            if (!rc) rc = drain(c);
            if (rc) { quiesce(); return rc; }
            if (hook && hook->enqueued) {                   // (drain waited for the piece's last copy: it has completed)
                size_t lo, cnt;
                span(c, lo, cnt);
                hook->enqueued(c, lo, cnt, nullptr);
            }
        }
        if (resident_result) quiesce();
        return 0;
    };
    if (nchunks == 1) return sequential();

    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> staged(nchunks, 0), drained(nchunks, 0);
    size_t submitted = 0;
    int failed = 0;                                       // first error of any role; everybody stops
    std::string failed_text;                              // ... and its text: last_error() is thread-local, helpers have their own
    // helper threads: pipeline_helpers() above; parked in the calling thread's pool between calls
    const HelperPlan plan = pipeline_helpers(concurrent_pipelines());
    const int STAGERS = plan.combined ? 1 : plan.stagers, DRAINERS = plan.combined ? 0 : plan.drainers;
    HelperPool& pool = helper_pool(plan.total());
    if (!pool.ok()) return sequential();                  // the process cannot have more threads: do without them
    auto drain_piece = [&](size_t c) -> bool {            // false: somebody failed, stop
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed || submitted > c; });
            if (failed) return false;
        }
        const int rc = drain(c);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rc && !failed) { failed = rc; failed_text = last_error(); }
            drained[c] = 1;
        }
        cv.notify_all();
        return rc == 0;
    };
    auto helper = [&](int idx) {
        if (plan.combined) {                              // one helper: piece c's buffer set is free once piece c - sets has left
            for (size_t c = 0; c < nchunks + (size_t)sets; c++) {
                if (c >= (size_t)sets && !drain_piece(c - sets)) return;
                if (c < nchunks) {
                    stage_in(c, 0, 1);
                    { std::lock_guard<std::mutex> lk(mu); staged[c]++; }
                    cv.notify_all();
                }
            }
        } else if (idx < STAGERS) {
            const int sidx = idx;
            for (size_t c = 0; c < nchunks; c++) {        // every stager copies its share of every piece: pieces
                                                          // become ready in order, each in 1/STAGERS of the time
                {   // the previous piece in this buffer set must have left its pinned buffers
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return failed || c < (size_t)sets || drained[c - sets]; });
                    if (failed) return;
                }
                stage_in(c, sidx, STAGERS);
                { std::lock_guard<std::mutex> lk(mu); staged[c]++; }
                cv.notify_all();
            }
        } else {
            const int didx = idx - STAGERS;
            for (size_t c = didx; c < nchunks; c += DRAINERS)
                if (!drain_piece(c)) return;
        }
    };
    pool.run(plan.total(), helper);
    for (size_t c = 0; c < nchunks; c++) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed || staged[c] == STAGERS; });
            if (failed) break;
        }
        const int rc = submit(c, false);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rc && !failed) { failed = rc; failed_text = last_error(); }
            submitted = c + 1;
        }
        cv.notify_all();
        if (rc) break;
    }
    pool.wait();
    if (failed || resident_result) quiesce();
    if (failed) last_error() = failed_text;               // whichever thread saw it first: the caller gets rc AND text
    return failed;
}

}  // namespace c25519_host

// curve25519_amd/csrc/capi_common.hpp -- host-side plumbing shared by the C-ABI entry points: per-thread error
// text, and the per-thread device resources (streams, pinned + device staging, work scratch) with their release.
// No torch types, no CPU arithmetic: everything that computes is a HIP kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <pthread.h>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <sched.h>

namespace c25519_host {

inline std::string& last_error()
{
    static thread_local std::string e;
    return e;
}

inline int fail(hipError_t err, const char* what, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(err), file, line);
    last_error() = buf;
    (void)hipGetLastError();       // reported through rc + text: do not leave it for the next launch check to trip over
    return (int)err ? (int)err : -1;
}

#define C25519_TRY(expr)                                                             \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) return c25519_host::fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)

#define C25519_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Fill / upload device memory and KNOW it has happened before the next line runs.  hipMemset of device memory may return
// before the fill has run, and the null stream it runs on does not order with this library's streams (all created
// hipStreamNonBlocking): a fill that is still queued when a kernel of the call writes results into the same buffer wipes them
// afterwards (seen on a busy device: tools/stress_host_api.py, zeros where a shard's signatures belonged).  Likewise for a
// small upload the next kernels read.  Both helpers come back only when the null stream has drained.
inline int zero_device_now(void* p, size_t bytes)
{
    C25519_TRY(hipMemsetAsync(p, 0, bytes, nullptr));
    C25519_TRY(hipStreamSynchronize(nullptr));
    return 0;
}
inline int upload_now(void* dst, const void* src, size_t bytes)
{
    C25519_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    C25519_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

inline int bad_arg(const char* msg)
{
    last_error() = msg;
    return (int)hipErrorInvalidValue;
}

// The single-call reference API has no error channel (void functions).  A device failure there is
// fatal by design: say why and abort instead of returning garbage or falling back to a CPU path.
[[noreturn]] inline void die(const char* fn, int rc)
{
    fprintf(stderr, "curve25519_amd: %s failed (rc=%d): %s\n"
                    "curve25519_amd: this library has no CPU fallback; a gfx950 device is required.\n",
            fn, rc, last_error().c_str());
    abort();
}

// ---- tuning / A-B knobs (include/curve25519_amd.h: c25519_amd_tunable_set / _get) ------------------------------------
// One table of process-wide atomics.  Every knob is initialised ONCE from the environment variable C25519_AMD_<NAME> --
// when the first of them is read -- so no call path runs getenv() afterwards (which is not safe against a setenv() in
// another thread), and tests / bench.py / the A-B tools change a knob at run time through the C ABI instead of through the
// environment.  T_UNSET = the library's built-in choice.
enum Tunable {
    T_COOP_MAX,                 // largest batch that runs one operation per WAVE (0: never; unset: per operation, engine.hip)
    T_XF_SPLIT,                 // X25519 as ladder + shared inversion in two launches (1) or one fused launch (0)
    T_INV_K,                    // elements per inverting lane of k_batch_invert, 1..16
    T_VERIFY_REFERENCE_ORDER,   // 1: every verification through the reference-order kernels (BASELINE.json configs[3] as worded)
    T_MULTI_FORCE_GATHER,       // 1: a one-device *_multi handle takes the gather path too
    T_MULTI_VIRTUAL,            // c25519_amd_multi_create: a one-device list becomes this many virtual devices on it
    T_BASE_COMB,                // fixed-base walks: 0 = the 8-table signed comb in LDS, 1 = the wide comb read from L2 (default)
    T_HELPER_THREADS,           // cap on the staging helper threads of one process (unset: the CPUs this process may use)
    T_VERIFY_LAT_CAP_BITS,      // TEST knob: verification's lattice walk takes short vectors up to this many bits (100..157; anything else: the production cap, 158)
    T_ONE_KEY_WIDE,             // ed25519_Verify_Check: smallest batch that builds a wide comb for its key (0: never; default 2^16)
    T_LADDER2_MAX,              // curve25519_dh_CreateSharedKey: largest call that runs the ladder on TWO waves per element (0: never)
    T_QUAD_MIN,                 // calls of MORE than QUAD_MIN and at most QUAD_MAX elements run FOUR LANES per element (quad25519.cuh):
    T_QUAD_MAX,                 //   QUAD_MAX = 0: never; unset: per operation (engine.hip)
    T_COUNT
};
constexpr long T_UNSET = -1;
inline const char* const* tunable_names()
{
    static const char* const names[T_COUNT] = { "COOP_MAX", "XF_SPLIT", "INV_K", "VERIFY_REFERENCE_ORDER", "MULTI_FORCE_GATHER",
                                                "MULTI_VIRTUAL", "BASE_COMB", "HELPER_THREADS", "VERIFY_LAT_CAP_BITS", "ONE_KEY_WIDE", "LADDER2_MAX",
                                                "QUAD_MIN", "QUAD_MAX" };
    return names;
}
inline std::atomic<long>* tunable_table()
{
    static std::atomic<long> v[T_COUNT];
    static const bool once = [] {
        for (int i = 0; i < T_COUNT; i++) {
            const std::string name = std::string("C25519_AMD_") + tunable_names()[i];
            const char* e = getenv(name.c_str());
            v[i].store(e && *e ? atol(e) : T_UNSET, std::memory_order_relaxed);
        }
        return true;
    }();
    (void)once;
    return v;
}
inline long tunable(Tunable t) { return tunable_table()[t].load(std::memory_order_relaxed); }
inline long tunable_or(Tunable t, long dflt) { const long v = tunable(t); return v == T_UNSET ? dflt : v; }

// CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota (a container on a 256-thread host
// is typically allowed 8-16) -- NOT std::thread::hardware_concurrency(), which reports the host.
inline int usable_cpus()
{
    static const int n = [] {
        int cpus = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) cpus = CPU_COUNT(&set);
        auto quota_of = [](const char* path, bool v2) -> double {
            FILE* f = fopen(path, "r");
            if (!f) return 0.0;
            char a[64] = {}, b[64] = {};
            double q = 0.0;
            if (v2) {                                              // cgroup v2: "max 100000" or "<quota> <period>"
                if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0 && atof(b) > 0) q = atof(a) / atof(b);
            } else if (fscanf(f, "%63s", a) == 1 && atof(a) > 0) { // cgroup v1: cpu.cfs_quota_us (-1: none) / cpu.cfs_period_us
                FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
                if (g) { if (fscanf(g, "%63s", b) == 1 && atof(b) > 0) q = atof(a) / atof(b); fclose(g); }
            }
            fclose(f);
            return q;
        };
        double q = quota_of("/sys/fs/cgroup/cpu.max", true);
        if (q <= 0.0) q = quota_of("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
        if (q > 0.0 && (int)(q + 0.5) < cpus) cpus = (int)(q + 0.5);
        return cpus < 1 ? 1 : cpus;
    }();
    return n;
}

// Set to false by an atexit handler registered after the first HIP call (so it runs before the HIP runtime's own
// teardown): thread-exit destructors that fire later must not call into a runtime that is already gone.
inline std::atomic<bool>& runtime_alive()
{
    static std::atomic<bool> alive{ true };
    return alive;
}
// ... and calls that are IN FLIGHT when exit() begins (another thread is in the middle of a batch while main returns) must be
// allowed to finish before the HIP runtime's own teardown runs -- a kernel launch into a runtime that is being torn down is a
// segmentation fault inside libamdhip64 (tests/c/exit_midcall.c) --, and calls that BEGIN after that point must not reach the
// runtime at all.  Every C entry point that touches HIP holds an ApiCall for its duration (C25519_API_CALL()): the outermost one
// of a thread counts itself in; the atexit handler below closes the gate and waits (up to 10 s) for the count to drain; a call
// that arrives at a closed gate parks its thread for the rest of the process' life (which is being ended by exit()).
struct ApiGate {
    static std::atomic<long>& inflight() { static std::atomic<long> n{ 0 }; return n; }
    static int& depth() { thread_local int d = 0; return d; }     // > 0: inside a call already (a *_batch call running its *_dev form)
    static std::atomic<unsigned long>& exiting_thread() { static std::atomic<unsigned long> t{ 0 }; return t; }
};
class ApiCall {
public:
    ApiCall()
    {
        if (ApiGate::depth()++ > 0) return;
        last_error().clear();                             // c25519_amd_last_error(): "" unless THIS call fails
        ApiGate::inflight().fetch_add(1);
        if (!runtime_alive().load()) {
            ApiGate::inflight().fetch_sub(1);
            // the thread that RUNS exit() (a static destructor or an atexit handler of the caller's that reaches the library) must
            // go on: its call is refused; any other thread has nothing left to do in this process
            if (ApiGate::exiting_thread().load() == (unsigned long)pthread_self()) { refused_ = true; return; }
            for (;;) pause();
        }
    }
    ~ApiCall() { if (--ApiGate::depth() == 0 && !refused_) ApiGate::inflight().fetch_sub(1); }
    bool refused() const { return refused_; }
    ApiCall(const ApiCall&) = delete;
    ApiCall& operator=(const ApiCall&) = delete;

private:
    bool refused_ = false;
};
inline int refused_call()
{
    last_error() = "the process is exiting: the HIP runtime is being torn down";
    return (int)hipErrorDeinitialized;
}
#define C25519_API_CALL_OR(ret) c25519_host::ApiCall api_call_guard_; if (api_call_guard_.refused()) return ret
#define C25519_API_CALL() C25519_API_CALL_OR(c25519_host::refused_call())
// a thread that only ever works INSIDE somebody's call (a multi-GPU worker): its nested entry points pass the gate
inline void mark_thread_inside_a_call() { ApiGate::depth() = 1; }

inline void arm_exit_guard()
{
    static std::atomic<bool> armed{ false };
    if (!armed.exchange(true))
        atexit([] {
            ApiGate::exiting_thread().store((unsigned long)pthread_self());
            runtime_alive().store(false);
            for (int ms = 0; ms < 10000 && ApiGate::inflight().load() > 0; ms++) usleep(1000);
        });
}

// Per-host-thread device resources.  The staging side (streams, pinned + device buffers of the *_batch pipeline)
// belongs to ONE device (`device`): when the thread calls a *_batch function with another current device, the old
// device's staging is released first, on that device.  Work scratch of the *_dev functions is kept per device.
// Everything is released by c25519_amd_thread_release(), or when the thread exits while the runtime is still alive.
struct ThreadState {
    // the host-pointer (*_batch) pipeline: pieces rotate over SETS buffer sets (so staging a piece in never waits for
    // an earlier piece to be copied out); LANES streams -- two for kernels, one per copy direction (the runtime drives
    // four hardware queues by default: more streams than that only queue up behind each other, measured)
    static constexpr int LANES = 4;
    static constexpr int SETS = 8;
    static constexpr int SLOTS = 5;        // arrays per set (inputs, outputs, in/out)
    int device = -1;
    hipStream_t stream[LANES] = {};        // [0], [1]: kernels of even / odd pieces; [2]: uploads; [3]: downloads
    hipEvent_t done[SETS] = {};            // end of a set's last download
    hipEvent_t uploaded[SETS] = {};        // end of a set's last upload
    hipEvent_t computed[SETS] = {};        // end of a set's last kernels
    void* dbuf[SETS][SLOTS] = {};          // device staging
    size_t dcap[SETS][SLOTS] = {};
    void* vctx = nullptr;                  // the 2080-byte context of this thread's last ed25519_Verify_Check_batch (nothing else writes it)
    unsigned char vctx_host[2080] = {};    // ... and the bytes that were uploaded into it
    bool vctx_valid = false;
    void* bctx = nullptr;                  // the same for the 192-byte blinding context of this thread's last blinded *_batch call
    unsigned char bctx_host[192] = {};
    bool bctx_valid = false;
    void* hbuf[SETS][SLOTS] = {};          // pinned host staging (hipHostMalloc)
    size_t hcap[SETS][SLOTS] = {};
    // The completion word of a call of ONE element (host_pipeline.hpp: zero-copy calls): pinned host memory the call's last
    // kernel stores done_seq into behind its results, and the calling thread spins on -- the runtime's own completion path
    // (event record + hipEventSynchronize) costs 4.6 us more (tools/scratch/launch_latency.hip, profiles/r06_launch_latency.txt).
    // done_offered: run_batch is inside `launch` and would wait for the word; done_taken: a kernel of this call has it.
    unsigned* done_word = nullptr;
    unsigned done_seq = 0;
    bool done_offered = false, done_taken = false;
    // work scratch of the *_dev entry points: grow-only slabs, CALLER_SLABS PER DEVICE (a thread may drive several GPUs, see
    // multi_device.hip).  Consecutive calls on one stream reuse one slab in stream order; calls on up to CALLER_SLABS
    // different streams get a slab each and may overlap on the device (a caller that splits a mixed batch over streams
    // fills one operation's last, partly empty wave of workgroups with the next operation's); a further stream takes the
    // least recently used slab and first waits for its previous use.
    static constexpr int MAX_DEV = 64;
    static constexpr int CALLER_SLABS = 4;
    struct WorkSlab {
        void* ptr = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        hipStream_t last = nullptr;
        bool used = false;
        unsigned long stamp = 0;           // when it was last handed out (LRU)
        unsigned* report = nullptr;        // a device word that outlives a call's scratch (a counter the call reports afterwards)
    };
    WorkSlab work[MAX_DEV][CALLER_SLABS];
    unsigned long work_clock = 0;
    unsigned long generation = 0;          // bumped whenever streams / slabs / report words are destroyed: a handle to one of
                                           // them remembered across calls (engine.hip: LastVerify) is stale afterwards
    WorkSlab keep[MAX_DEV];                // per device: a buffer whose CONTENT outlives the call (the comb of the last one-key
                                           // verification batch's key, engine.hip); same stream-order rules as a work slab
    WorkSlab lane_work[LANES];             // ... and one per pipeline lane, so that the pieces of a *_batch call do not
                                           // wait for each other's kernels (they belong to the staging device)

    WorkSlab* slab_for(hipStream_t s, int dev)
    {
        if (s && dev == device)
            for (int l = 0; l < LANES; l++)
                if (s == stream[l]) return &lane_work[l];
        WorkSlab* row = work[dev];
        for (int k = 0; k < CALLER_SLABS; k++)
            if (row[k].used && row[k].last == s) return &row[k];
        WorkSlab* pick = &row[0];
        for (int k = 0; k < CALLER_SLABS; k++) {
            if (!row[k].used) return &row[k];
            if (row[k].stamp < pick->stamp) pick = &row[k];
        }
        return pick;
    }
    void free_slab(WorkSlab& w)            // on the slab's device, after a synchronize
    {
        generation++;
        if (w.ptr) { (void)hipMemset(w.ptr, 0, w.cap); (void)hipFree(w.ptr); }
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.report) (void)hipFree(w.report);
        w = WorkSlab();
    }

    // bind to the current device (creating streams/events on first use)
    int ensure()
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        arm_exit_guard();
        if (device >= 0 && dev != device) {
            release_staging();
            C25519_TRY(hipSetDevice(dev));
        }
        if (device < 0) {
            for (int l = 0; l < LANES; l++) C25519_TRY(hipStreamCreateWithFlags(&stream[l], hipStreamNonBlocking));
            for (int l = 0; l < SETS; l++) {
                C25519_TRY(hipEventCreateWithFlags(&done[l], hipEventDisableTiming));
                C25519_TRY(hipEventCreateWithFlags(&uploaded[l], hipEventDisableTiming));
                C25519_TRY(hipEventCreateWithFlags(&computed[l], hipEventDisableTiming));
            }
            device = dev;
        }
        return 0;
    }
    int reserve_dev(int lane /* = buffer set */, int slot, size_t bytes)
    {
        if (bytes <= dcap[lane][slot]) return 0;
        if (dbuf[lane][slot]) {                           // held staged secrets: zeroed before it goes back to the allocator
            C25519_RC(zero_device_now(dbuf[lane][slot], dcap[lane][slot]));
            C25519_TRY(hipFree(dbuf[lane][slot]));
            dbuf[lane][slot] = nullptr; dcap[lane][slot] = 0;
        }
        const size_t want = bytes < 4096 ? 4096 : bytes;
        C25519_TRY(hipMalloc(&dbuf[lane][slot], want));
        dcap[lane][slot] = want;
        return 0;
    }
    int reserve_host(int lane, int slot, size_t bytes)
    {
        if (bytes <= hcap[lane][slot]) return 0;
        if (hbuf[lane][slot]) {
            memset(hbuf[lane][slot], 0, hcap[lane][slot]);
            C25519_TRY(hipHostFree(hbuf[lane][slot]));
            hbuf[lane][slot] = nullptr; hcap[lane][slot] = 0;
        }
        const size_t want = bytes < 4096 ? 4096 : bytes;
        C25519_TRY(hipHostMalloc(&hbuf[lane][slot], want, hipHostMallocDefault));
        hcap[lane][slot] = want;
        return 0;
    }
    int acquire_work(void** out, size_t bytes, hipStream_t s)
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEV) return bad_arg("device ordinal out of range");
        arm_exit_guard();
        WorkSlab& w = *slab_for(s, dev);
        if (w.ptr && bytes > w.cap) {
            C25519_TRY(hipDeviceSynchronize());
            C25519_RC(zero_device_now(w.ptr, w.cap));
            C25519_TRY(hipFree(w.ptr));
            w.ptr = nullptr; w.cap = 0; w.used = false;
        }
        if (!w.ptr) {
            const size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes;
            C25519_TRY(hipMalloc(&w.ptr, want));
            w.cap = want;
        }
        if (!w.done) C25519_TRY(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        if (w.used && s != w.last) C25519_TRY(hipStreamWaitEvent(s, w.done, 0));
        w.last = s; w.used = true; w.stamp = ++work_clock;       // the slab is this stream's until release_work
        *out = w.ptr;
        return 0;
    }
    // the persistent buffer of the current device, at least `bytes` (zero-filled when it is (re)allocated: *fresh says so);
    // a call on another stream than the last one first waits for that one's use (release_keep)
    int acquire_keep(void** out, size_t bytes, hipStream_t s, bool* fresh)
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEV) return bad_arg("device ordinal out of range");
        arm_exit_guard();
        WorkSlab& w = keep[dev];
        *fresh = false;
        if (w.ptr && bytes > w.cap) {
            C25519_TRY(hipDeviceSynchronize());
            C25519_TRY(hipFree(w.ptr));
            w.ptr = nullptr; w.cap = 0; w.used = false;
        }
        if (!w.ptr) {
            C25519_TRY(hipMalloc(&w.ptr, bytes));
            C25519_RC(zero_device_now(w.ptr, bytes));
            w.cap = bytes;
            *fresh = true;
        }
        if (!w.done) C25519_TRY(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        if (w.used && s != w.last) C25519_TRY(hipStreamWaitEvent(s, w.done, 0));
        w.last = s; w.used = true;
        *out = w.ptr;
        return 0;
    }
    int release_keep(hipStream_t s)
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        C25519_TRY(hipEventRecord(keep[dev].done, s));
        return 0;
    }
    // has a call of this thread on the current device left a kept buffer behind (a key's comb and the context it belongs to)?
    bool has_keep() const
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
        return keep[dev].ptr != nullptr;
    }
    // the report word of the slab acquire_work(…, s) handed out (call between acquire_work and release_work)
    int report_word_for(unsigned** out, hipStream_t s)
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= MAX_DEV) return bad_arg("device ordinal out of range");
        WorkSlab& w = *slab_for(s, dev);
        if (!w.report) {
            C25519_TRY(hipMalloc(&w.report, 256));
            C25519_RC(zero_device_now(w.report, 256));
        }
        *out = w.report;
        return 0;
    }
    int release_work(hipStream_t s)
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        WorkSlab& w = *slab_for(s, dev);
        C25519_TRY(hipEventRecord(w.done, s));
        return 0;
    }
    // free the staging side on its device (the buffers held staged secrets: they are zeroed first)
    void release_staging()
    {
        if (device < 0) return;
        generation++;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        for (int l = 0; l < SETS; l++) {
            for (int i = 0; i < SLOTS; i++) {
                if (dbuf[l][i]) { (void)hipMemset(dbuf[l][i], 0, dcap[l][i]); (void)hipFree(dbuf[l][i]); }
                if (hbuf[l][i]) { memset(hbuf[l][i], 0, hcap[l][i]); (void)hipHostFree(hbuf[l][i]); }
                dbuf[l][i] = hbuf[l][i] = nullptr; dcap[l][i] = hcap[l][i] = 0;
            }
            for (hipEvent_t* e : { &done[l], &uploaded[l], &computed[l] }) {
                if (*e) (void)hipEventDestroy(*e);
                *e = nullptr;
            }
        }
        if (done_word) { (void)hipHostFree(done_word); done_word = nullptr; }
        done_offered = done_taken = false;
        if (vctx) { (void)hipMemset(vctx, 0, 2080); (void)hipFree(vctx); vctx = nullptr; }
        vctx_valid = false;
        if (bctx) { (void)hipMemset(bctx, 0, 192); (void)hipFree(bctx); bctx = nullptr; }
        memset(bctx_host, 0, sizeof bctx_host);
        bctx_valid = false;
        for (int l = 0; l < LANES; l++) {
            free_slab(lane_work[l]);
            if (stream[l]) (void)hipStreamDestroy(stream[l]);
            stream[l] = nullptr;
        }
        (void)hipGetLastError();
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        device = -1;
    }
    // ... and everything: staging plus every device's work scratch
    void release()
    {
        release_staging();
        int cur = -1;
        (void)hipGetDevice(&cur);
        for (int d = 0; d < MAX_DEV; d++) {
            bool any = false;
            for (const WorkSlab& w : work[d]) any = any || w.ptr || w.done || w.report;
            any = any || keep[d].ptr || keep[d].done;
            if (!any) continue;
            (void)hipSetDevice(d);
            (void)hipDeviceSynchronize();
            for (WorkSlab& w : work[d]) free_slab(w);
            free_slab(keep[d]);
        }
        (void)hipGetLastError();
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    ~ThreadState()
    {
        if (runtime_alive().load()) release();     // else: process teardown, the HIP runtime may already be gone
    }
};

inline ThreadState& tls()
{
    static thread_local ThreadState s;
    return s;
}

// A work slab between acquire_work and release_work.  Leaving the scope without release() -- an error return between the
// two, with kernels possibly queued on the slab already -- still records the slab's `done` event on the stream, so that a
// stream that takes the slab over later waits for everything that was enqueued, not for a stale event.
class WorkLease {
public:
    WorkLease() = default;
    WorkLease(const WorkLease&) = delete;
    WorkLease& operator=(const WorkLease&) = delete;
    ~WorkLease() { if (live_) { (void)tls().release_work(stream_); (void)hipGetLastError(); } }
    int acquire(void** out, size_t bytes, hipStream_t s)
    {
        C25519_RC(tls().acquire_work(out, bytes, s));
        stream_ = s; live_ = true;
        return 0;
    }
    int release()
    {
        live_ = false;
        return tls().release_work(stream_);
    }

private:
    hipStream_t stream_ = nullptr;
    bool live_ = false;
};

// ... and of the kept buffer (ThreadState::acquire_keep): whatever way the call leaves, the buffer's `done` event is recorded
// behind what was enqueued on it, so that a later call on another stream waits for this call's kernels before it rewrites the
// key's comb
class KeepLease {
public:
    KeepLease() = default;
    KeepLease(const KeepLease&) = delete;
    KeepLease& operator=(const KeepLease&) = delete;
    ~KeepLease() { if (live_) { (void)tls().release_keep(stream_); (void)hipGetLastError(); } }
    int acquire(void** out, size_t bytes, hipStream_t s, bool* fresh)
    {
        C25519_RC(tls().acquire_keep(out, bytes, s, fresh));
        stream_ = s; live_ = true;
        return 0;
    }
    int release()
    {
        if (!live_) return 0;
        live_ = false;
        return tls().release_keep(stream_);
    }

private:
    hipStream_t stream_ = nullptr;
    bool live_ = false;
};

}  // namespace c25519_host

// curve25519_amd/csrc/multi_device.hip -- the C-ABI multi-GPU entry points (include/curve25519_amd.h, "Multi-GPU").
//
// SURVEY.md 8(e) / BASELINE.json north_star: the batch shards embarrassingly over the GPUs of one node -- element i never
// looks at element j -- and the only exchange step is ONE gather of every GPU's result rows to the root over xGMI
// (RCCL ncclGather, /opt/rocm/include/rccl/rccl.h:745).  Here ONE host thread drives all devices (the C counterpart of
// the one-process-per-GPU torch.distributed layer in curve25519_amd/sharded.py, which bench.py uses):
//   1. contiguous shards: device d owns elements [n*d/D, n*(d+1)/D);
//   2. per device, on its own stream: upload the shard, run the same *_dev kernels as the single-GPU path;
//   3. one grouped ncclGather per output array to devices[0] (rows padded to the largest shard, every rank sends the
//      same count), then the host reads the gathered slab from the root device.
// RCCL is loaded with dlopen on first use, so single-GPU users of the library do not pay for (or need) it.
#include "capi_common.hpp"

#include "../../include/curve25519_amd.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

using c25519_host::bad_arg;
using c25519_host::last_error;

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

}  // namespace

struct c25519_amd_multi {
    Rccl rccl;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    std::vector<void*> buf[MAX_ARR];          // per device staging of array a (shard-sized, grow-only)
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

int reserve(void*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap) return 0;
    if (p) { C25519_TRY(hipFree(p)); p = nullptr; cap = 0; }
    C25519_TRY(hipMalloc(&p, bytes < 4096 ? 4096 : bytes));
    cap = bytes < 4096 ? 4096 : bytes;
    return 0;
}

struct MArr {
    const void* in;      // host source (nullptr: output only)
    void* out;           // host destination (nullptr: input only)
    size_t elem;         // bytes per element
    bool gather;         // output travels through the RCCL gather to the root (else read back from its own device)
};

// shard, upload, launch(d, device pointers, count, stream), gather, download
template <typename Launch>
int run_multi(c25519_amd_multi* m, size_t n, const MArr* arr, int na, Launch launch)
{
    const int D = (int)m->dev.size();
    int prev = 0;
    C25519_TRY(hipGetDevice(&prev));
    std::vector<size_t> lo(D + 1);
    size_t rows = 0;                                                  // largest shard
    for (int d = 0; d <= D; d++) lo[d] = n * (size_t)d / (size_t)D;
    for (int d = 0; d < D; d++) rows = lo[d + 1] - lo[d] > rows ? lo[d + 1] - lo[d] : rows;
    auto body = [&]() -> int {
        for (int d = 0; d < D; d++) {
            C25519_TRY(hipSetDevice(m->dev[d]));
            const size_t cnt = lo[d + 1] - lo[d];
            void* ptr[MAX_ARR] = {};
            for (int a = 0; a < na; a++) {
                C25519_RC(reserve(m->buf[a][d], m->cap[a][d], arr[a].elem * rows));
                ptr[a] = m->buf[a][d];
                if (arr[a].in && cnt * arr[a].elem)
                    C25519_TRY(hipMemcpyAsync(ptr[a], (const char*)arr[a].in + lo[d] * arr[a].elem, cnt * arr[a].elem,
                                              hipMemcpyHostToDevice, m->stream[d]));
            }
            if (cnt) C25519_RC(launch(d, ptr, cnt, m->stream[d]));
        }
        // the one exchange step: every device's rows of each gathered output -> devices[0]
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
        // results: gathered arrays from the root's slab, IN/OUT arrays (the clamped sk) from their own device
        for (int a = 0; a < na; a++) {
            if (!arr[a].out) continue;
            for (int d = 0; d < D; d++) {
                const size_t cnt = lo[d + 1] - lo[d];
                if (!cnt) continue;
                const bool g = arr[a].gather;
                C25519_TRY(hipSetDevice(m->dev[g ? 0 : d]));
                const char* src = g ? (const char*)m->gathered[a] + arr[a].elem * rows * d : (const char*)m->buf[a][d];
                C25519_TRY(hipMemcpyAsync((char*)arr[a].out + lo[d] * arr[a].elem, src, cnt * arr[a].elem,
                                          hipMemcpyDeviceToHost, m->stream[g ? 0 : d]));
            }
        }
        for (int d = 0; d < D; d++) {
            C25519_TRY(hipSetDevice(m->dev[d]));
            C25519_TRY(hipStreamSynchronize(m->stream[d]));
        }
        return 0;
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
        return 0;
    };
    const int rc = init();
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
    for (size_t d = 0; d < m->dev.size(); d++) {
        (void)hipSetDevice(m->dev[d]);
        if (m->stream[d]) (void)hipStreamSynchronize(m->stream[d]);
        if (m->comm[d] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comm[d]);
        for (int a = 0; a < MAX_ARR; a++)
            if (m->buf[a][d]) { (void)hipMemset(m->buf[a][d], 0, m->cap[a][d]); (void)hipFree(m->buf[a][d]); }
        if (d == 0)
            for (int a = 0; a < MAX_ARR; a++)
                if (m->gathered[a]) (void)hipFree(m->gathered[a]);
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

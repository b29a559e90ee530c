// curve25519_amd/csrc/capi_common.hpp -- host-side plumbing shared by the C-ABI entry points:
// per-thread error text, per-thread stream + growable device staging buffers.  No torch types, no
// CPU arithmetic: everything that computes is a HIP kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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
    return (int)err ? (int)err : -1;
}

#define C25519_TRY(expr)                                                             \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) return c25519_host::fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)

// Growable device buffers owned by one host thread.  Slots 0-3 belong to pipeline lane 0, 4-7 to lane 1.
struct Staging {
    static constexpr int SLOTS = 8;
    void* ptr[SLOTS] = {};
    size_t cap[SLOTS] = {};
    hipStream_t stream = nullptr;      // lane 0
    hipStream_t stream2 = nullptr;     // lane 1 of the chunked host pipeline
    int device = -1;

    int ensure_stream()
    {
        int dev = 0;
        C25519_TRY(hipGetDevice(&dev));
        if (stream && dev != device) release();
        if (!stream) {
            C25519_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            C25519_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
            device = dev;
        }
        return 0;
    }
    int reserve(int slot, size_t bytes)
    {
        if (bytes <= cap[slot]) return 0;
        if (ptr[slot]) { C25519_TRY(hipFree(ptr[slot])); ptr[slot] = nullptr; cap[slot] = 0; }
        size_t want = bytes < 4096 ? 4096 : bytes;
        C25519_TRY(hipMalloc(&ptr[slot], want));
        cap[slot] = want;
        return 0;
    }
    void release()
    {
        for (int i = 0; i < SLOTS; i++) { if (ptr[i]) (void)hipFree(ptr[i]); ptr[i] = nullptr; cap[i] = 0; }
        if (stream) (void)hipStreamDestroy(stream);
        if (stream2) (void)hipStreamDestroy(stream2);
        stream = nullptr;
        stream2 = nullptr;
        device = -1;
    }
    ~Staging() { /* process teardown: the HIP runtime may already be gone, do not call into it */ }
};

inline Staging& staging()
{
    static thread_local Staging s;
    return s;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

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

}  // namespace c25519_host

// tests/c/exit_midcall.c -- the process calls exit() while another thread is in the middle of *_batch calls (its buffers stay
// valid: they are never freed).  Static destructors and atexit handlers run beside the live thread: must not crash or hang
// (capi_common.hpp: ApiCall -- the library lets the call in flight finish before the HIP runtime's teardown and parks the thread
// at its next call).  argv[1]: 0 = *_batch calls, 1 = *_multi calls on three virtual devices; argv[2]: microseconds between the worker's first completed call and exit().
// Used by tests/test_gpu_parity.py::test_exit_with_a_call_in_flight and tools/scratch/exit_paths.py.
#include <execinfo.h>
#include <signal.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "curve25519_amd.h"
static size_t n = ((size_t)1 << 18) + 5;
static unsigned char *pk, *sk, *out;
static int mode;
static volatile int warmed;                 // the worker's first call has completed (runtime initialised, tables generated)
static void* work(void* arg)
{
    (void)arg;
    c25519_amd_multi* h = NULL;
    int dev[3] = { 0, 0, 0 };
    if (mode == 1 && c25519_amd_multi_create(&h, dev, 3)) { fprintf(stderr, "create: %s\n", c25519_amd_last_error()); _exit(3); }
    for (;;) {
        int rc = mode == 1 ? curve25519_dh_CreateSharedKey_multi(h, out, pk, sk, n) : curve25519_dh_CreateSharedKey_batch(out, pk, sk, n);
        if (rc) { fprintf(stderr, "call failed during exit: %s\n", c25519_amd_last_error()); pause(); }
        warmed = 1;
    }
    return NULL;
}
static void on_fault(int sig)
{
    void* frames[48];
    const int k = backtrace(frames, 48);
    fprintf(stderr, "signal %d in thread %lu\n", sig, (unsigned long)pthread_self());
    backtrace_symbols_fd(frames, k, 2);
    _exit(70);
}
int main(int argc, char** argv)
{
    signal(SIGSEGV, on_fault); signal(SIGABRT, on_fault); signal(SIGBUS, on_fault);
    mode = argc > 1 ? atoi(argv[1]) : 0;
    pk = malloc(32 * n); sk = malloc(32 * n); out = malloc(32 * n);
    for (size_t i = 0; i < 32 * n; i++) { pk[i] = (unsigned char)(i * 131 + 7); sk[i] = (unsigned char)(i * 7); }
    pthread_t th;
    pthread_create(&th, NULL, work, NULL);
    while (!warmed) usleep(1000);               // exit() during the runtime's own start-up is not what this probes
    usleep(argc > 2 ? atoi(argv[2]) : 7000);    // ... but every phase of a later call: staging, kernels in flight, hand-over
    exit(0);
}

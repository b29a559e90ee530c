// tools/scratch/churn_threads.c -- generations of short-lived host threads calling the *_batch entry points with buffers the main
// thread owns: does the library's per-thread state (streams, pinned and device staging, helper threads) go away with the thread?
// Prints host RSS and free device memory per 10 generations.   gcc -O1 -I include churn_threads.c -L curve25519_amd -lcurve25519_amd -lpthread
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "curve25519_amd.h"

#define T 6
static size_t n;
static unsigned char *pk, *sk[T], *out[T], *priv, *msg, *sig[T];
static int release_odd;

static void* work(void* arg)
{
    const long t = (long)arg;
    if (curve25519_dh_CreateSharedKey_batch(out[t], pk, sk[t], n)) { fprintf(stderr, "x25519: %s\n", c25519_amd_last_error()); exit(2); }
    if (ed25519_SignMessage_batch(sig[t], priv, msg, 32, n / 2)) { fprintf(stderr, "sign: %s\n", c25519_amd_last_error()); exit(2); }
    if (curve25519_dh_CreateSharedKey_batch(out[t], pk, sk[t], 1)) exit(2);
    if (release_odd && (t & 1)) c25519_amd_thread_release();
    return NULL;
}

static double rss_mb(void)
{
    long pages = 0, dummy = 0;
    FILE* f = fopen("/proc/self/statm", "r");
    if (!f || fscanf(f, "%ld %ld", &dummy, &pages) != 2) pages = 0;
    if (f) fclose(f);
    return pages * (double)sysconf(_SC_PAGESIZE) / 1048576.0;
}

int main(int argc, char** argv)
{
    const int gens = argc > 1 ? atoi(argv[1]) : 60;
    n = argc > 2 ? (size_t)atol(argv[2]) : ((size_t)1 << 19) + 777;
    release_odd = argc > 3 ? atoi(argv[3]) : 1;
    pk = malloc(32 * n); priv = malloc(64 * n); msg = malloc(32 * n);
    for (size_t i = 0; i < 32 * n; i++) { pk[i] = (unsigned char)(i * 131 + 7); msg[i] = (unsigned char)(i * 29 + 3); }
    for (size_t i = 0; i < 64 * n; i++) priv[i] = (unsigned char)(i * 17 + 1);
    for (int t = 0; t < T; t++) {
        sk[t] = malloc(32 * n); out[t] = malloc(32 * n); sig[t] = malloc(64 * n);
        for (size_t i = 0; i < 32 * n; i++) sk[t][i] = (unsigned char)(i * 7 + t);
    }
    for (int g = 0; g < gens; g++) {
        pthread_t th[T];
        for (long t = 0; t < T; t++) pthread_create(&th[t], NULL, work, (void*)t);
        for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
        if (g % 10 == 9 || g == 0) printf("generation %3d: host RSS %.0f MiB\n", g + 1, rss_mb());
    }
    return 0;
}

/*
 * oracle/orc_batch.c -- batch drivers over the single-call restatement (TEST INFRASTRUCTURE ONLY).
 * Contiguous fixed-stride arrays, one contiguous slice per thread: this is the timed CPU baseline
 * layout named in SURVEY.md 8(d) and the checker for full-size GPU batches.
 */
#include "orc25519.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>

enum { OP_SHARED, OP_PUBLIC, OP_PUBLIC_FAST, OP_KEYPAIR, OP_SIGN, OP_VERIFY };

typedef struct {
    int op;
    size_t lo, hi, msg_size;
    uint8_t *out, *sk_rw;
    const uint8_t *in_a, *in_b, *msg;
    int32_t *ok;
} job_t;

static void *run(void *arg)
{
    job_t *j = (job_t *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        switch (j->op) {
        case OP_SHARED:      orc_x25519_shared(j->out + 32 * i, j->in_a + 32 * i, j->sk_rw + 32 * i); break;
        case OP_PUBLIC:      orc_x25519_public(j->out + 32 * i, j->sk_rw + 32 * i); break;
        case OP_PUBLIC_FAST: orc_x25519_public_fast(j->out + 32 * i, j->sk_rw + 32 * i); break;
        case OP_KEYPAIR:     orc_ed25519_keypair(j->out + 32 * i, j->sk_rw + 64 * i, j->in_a + 32 * i); break;
        case OP_SIGN:        orc_ed25519_sign(j->out + 64 * i, j->in_a + 64 * i, j->msg + j->msg_size * i, j->msg_size); break;
        case OP_VERIFY:      j->ok[i] = orc_ed25519_verify(j->in_a + 64 * i, j->in_b + 32 * i, j->msg + j->msg_size * i, j->msg_size); break;
        }
    }
    return 0;
}

static void fan_out(job_t proto, size_t n, int nthreads)
{
    (void)orc_base_folding8();                 /* build constants before any thread starts */
    if (nthreads <= 1 || n < 2) { proto.lo = 0; proto.hi = n; run(&proto); return; }
    if ((size_t)nthreads > n) nthreads = (int)n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = proto;
        jobs[t].lo = n * (size_t)t / (size_t)nthreads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
        pthread_create(&th[t], 0, run, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    free(jobs); free(th);
}

void orc_x25519_shared_batch(uint8_t *shared, const uint8_t *pk, uint8_t *sk, size_t n, int nthreads)
{
    job_t j = {0}; j.op = OP_SHARED; j.out = shared; j.in_a = pk; j.sk_rw = sk;
    fan_out(j, n, nthreads);
}

void orc_x25519_public_batch(uint8_t *pk, uint8_t *sk, size_t n, int fast, int nthreads)
{
    job_t j = {0}; j.op = fast ? OP_PUBLIC_FAST : OP_PUBLIC; j.out = pk; j.sk_rw = sk;
    fan_out(j, n, nthreads);
}

void orc_ed25519_keypair_batch(uint8_t *pub, uint8_t *priv, const uint8_t *sk, size_t n, int nthreads)
{
    job_t j = {0}; j.op = OP_KEYPAIR; j.out = pub; j.sk_rw = priv; j.in_a = sk;
    fan_out(j, n, nthreads);
}

void orc_ed25519_sign_batch(uint8_t *sig, const uint8_t *priv, const uint8_t *msg, size_t msg_size,
                            size_t n, int nthreads)
{
    job_t j = {0}; j.op = OP_SIGN; j.out = sig; j.in_a = priv; j.msg = msg; j.msg_size = msg_size;
    fan_out(j, n, nthreads);
}

void orc_ed25519_verify_batch(int32_t *ok, const uint8_t *sig, const uint8_t *pk, const uint8_t *msg,
                              size_t msg_size, size_t n, int nthreads)
{
    job_t j = {0}; j.op = OP_VERIFY; j.ok = ok; j.in_a = sig; j.in_b = pk; j.msg = msg; j.msg_size = msg_size;
    fan_out(j, n, nthreads);
}

/* ---- timed CPU baseline over the REAL reference library -------------------------------------------------
 * Threads loop over curve25519_dh_CreateSharedKey / ed25519_* of a dlopen'ed reference build
 * (oracle/_ref/libcurve25519_ref.so), one contiguous slice per thread -- exactly how a caller of the
 * reference would fan a batch out over cores.  Returns 0, or -1 when the library or a symbol is missing. */
typedef void (*ref_shared_fn)(unsigned char *, const unsigned char *, unsigned char *);
typedef void (*ref_sign_fn)(unsigned char *, const unsigned char *, const void *, const unsigned char *, size_t);
typedef int (*ref_verify_fn)(const unsigned char *, const unsigned char *, const unsigned char *, size_t);

typedef struct {
    int op; size_t lo, hi, msg_size;
    ref_shared_fn shared; ref_sign_fn sign; ref_verify_fn verify;
    uint8_t *out, *sk; const uint8_t *a, *b, *msg; int32_t *ok;
} refjob_t;

static void *ref_run(void *arg)
{
    refjob_t *j = (refjob_t *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        if (j->op == 0) j->shared(j->out + 32 * i, j->a + 32 * i, j->sk + 32 * i);
        else if (j->op == 1) j->sign(j->out + 64 * i, j->a + 64 * i, 0, j->msg + j->msg_size * i, j->msg_size);
        else j->ok[i] = j->verify(j->a + 64 * i, j->b + 32 * i, j->msg + j->msg_size * i, j->msg_size);
    }
    return 0;
}

static int ref_fan(refjob_t proto, size_t n, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    refjob_t *jobs = (refjob_t *)malloc(sizeof(refjob_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = proto;
        jobs[t].lo = n * (size_t)t / (size_t)nthreads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
        pthread_create(&th[t], 0, ref_run, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    free(jobs); free(th);
    return 0;
}

int orc_ref_x25519_shared_batch(const char *so_path, uint8_t *shared, const uint8_t *pk, uint8_t *sk, size_t n, int nthreads)
{
    void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    refjob_t j = {0};
    j.op = 0; j.shared = (ref_shared_fn)dlsym(h, "curve25519_dh_CreateSharedKey");
    j.out = shared; j.a = pk; j.sk = sk;
    return j.shared ? ref_fan(j, n, nthreads) : -1;
}

int orc_ref_ed25519_sign_batch(const char *so_path, uint8_t *sig, const uint8_t *priv, const uint8_t *msg, size_t msg_size,
                               size_t n, int nthreads)
{
    void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    refjob_t j = {0};
    j.op = 1; j.sign = (ref_sign_fn)dlsym(h, "ed25519_SignMessage");
    j.out = sig; j.a = priv; j.msg = msg; j.msg_size = msg_size;
    return j.sign ? ref_fan(j, n, nthreads) : -1;
}

int orc_ref_ed25519_verify_batch(const char *so_path, int32_t *ok, const uint8_t *sig, const uint8_t *pk, const uint8_t *msg,
                                 size_t msg_size, size_t n, int nthreads)
{
    void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    refjob_t j = {0};
    j.op = 2; j.verify = (ref_verify_fn)dlsym(h, "ed25519_VerifySignature");
    j.ok = ok; j.a = sig; j.b = pk; j.msg = msg; j.msg_size = msg_size;
    return j.verify ? ref_fan(j, n, nthreads) : -1;
}

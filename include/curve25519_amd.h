/*
 * include/curve25519_amd.h -- batched entry points of the MI355X engine (C ABI).
 *
 * The reference has no batch API: its callers loop over the single-call functions of
 * include/curve25519_dh.h and include/ed25519_signature.h (e.g. test/curve25519_test.c:144-318).
 * Each function below is the N-element form of exactly one reference call and produces, element by
 * element, the bytes that call would produce.  Layouts are contiguous fixed-stride arrays:
 *
 *     sk, pk, shared : n x 32 bytes          priv, sig : n x 64 bytes
 *     msg            : n x msg_size bytes    verdict   : n x int (1 valid / 0 invalid)
 *
 * Two flavours:
 *   *_batch : host pointers.  Synchronous: uploads, runs the kernels, downloads.  Callable from
 *             several host threads at once (each thread owns its stream and staging buffers).
 *   *_dev   : device pointers (hipMalloc'ed memory of the current device, 16-byte aligned) and a
 *             hipStream_t passed as void* (NULL = default stream).  Asynchronous: returns after
 *             enqueueing.  This is what bench.py times with inputs resident in HBM.
 *
 * Return value: 0 on success, otherwise a HIP error code (c25519_amd_last_error() gives the text).
 * There is no CPU fallback: without a usable gfx950 device every entry point fails.
 */
#ifndef CURVE25519_AMD_H
#define CURVE25519_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library / device status ------------------------------------------------------------------- */
const char *c25519_amd_version(void);
const char *c25519_amd_last_error(void);               /* per-thread: the text of the last call of this thread, "" if it succeeded */
int  c25519_amd_device_count(void);                    /* usable HIP devices (0 when none) */
int  c25519_amd_set_device(int device);                /* device used by this host thread */
/* Tuning / A-B knobs.  Each knob is read ONCE from the environment variable C25519_AMD_<name> when the library is first
 * used (no getenv() on any call path afterwards) and can be changed at run time here; value < 0 restores the built-in
 * choice.  Names: COOP_MAX (largest batch that runs one operation per wave; 0 = never), XF_SPLIT (X25519 as two launches 1 /
 * one fused launch 0), INV_K (elements per inverting lane, 1..16), VERIFY_REFERENCE_ORDER (1: every verification in the
 * reference's 4-fold + 8-fold order), MULTI_FORCE_GATHER (1: a one-device *_multi handle gathers too), MULTI_VIRTUAL (V: a
 * one-device list given to c25519_amd_multi_create becomes V virtual devices on it), BASE_COMB (fixed-base walks: 0 = the 8 x 32 comb staged in LDS,
 * 1 = the wide 13 x 20 comb read through L2, the default), HELPER_THREADS (cap on the staging helper threads; default: the CPUs this process may use),
 * VERIFY_LAT_CAP_BITS (test knob, 100..157: verification's lattice walk refuses longer short vectors, which then take the
 * reference-order kernel), ONE_KEY_WIDE (ed25519_Verify_Check_*: the smallest batch that BUILDS a wide comb for its one key,
 * default 65536; a call of more than 1024 pairs whose context is the one the calling thread's last such batch built a comb for
 * walks that comb whatever its size; 0 = never), LADDER2_MAX (curve25519_dh_CreateSharedKey_*: the largest call that runs the
 * ladder on two waves per element, default 512; 0 = never), QUAD_MIN / QUAD_MAX (calls of more than QUAD_MIN and at most QUAD_MAX
 * elements run FOUR LANES per element -- X25519: the whole operation, defaults 3584 / 32768; ed25519_VerifySignature_*: scalars
 * and point tables side by side, the walk on quads, defaults 1024 / 32768; key pairs, signatures, CalculatePublicKey_fast and the
 * one-key ed25519_Verify_Check_* with a comb: the whole operation in one launch, defaults 1024 / 16384; QUAD_MAX = 0: never).
 * _get returns -1 for "built-in choice", -2 for an unknown name.
 * Environment only (read once): C25519_AMD_DONE_WORD=0 -- a host-pointer call of ONE element waits for the stream's event instead of
 * the completion word its last kernel stores behind the results (5 us later; same bytes); C25519_AMD_ZERO_COPY=0 -- calls of a
 * few elements are staged through device buffers like any batch. */
int  c25519_amd_tunable_set(const char *name, long value);
long c25519_amd_tunable_get(const char *name);
int  c25519_amd_usable_cpus(void);                     /* CPUs this process may use (affinity mask cut to the cgroup quota) */
/* *_dev calls a thread issues on different streams may overlap on the device: each stream gets its own work scratch (up
 * to four per device; a further stream reuses the least recently used one after waiting for it).  Splitting a mixed batch
 * over streams is worth ~13 % (one operation's last round of workgroups fills up with the next operation's).
 * Each host thread that calls into the library owns four streams, eight sets of pinned + device staging buffers and work
 * scratch slabs, all on the device that was current at its first call (they follow the thread to another device on
 * the next call, released on the old one first).  They are freed when the thread exits; a long-lived thread can
 * give them back earlier with this call.  Staging buffers are zeroed before they are freed.
 * Process exit: exit() may be called while other threads are inside library calls -- the library's atexit handler lets the
 * calls in flight finish (it waits up to 10 s) before the HIP runtime is torn down, a thread that calls again afterwards is
 * parked for the rest of the process' life, and a call made by the exiting thread itself (from a static destructor or an
 * atexit handler of the caller's) returns hipErrorDeinitialized (release / destroy calls: return at once). */
void c25519_amd_thread_release(void);

/* Optional: page-lock a host array the caller is going to pass to *_batch functions again and again (hipHostRegister).
 * A *_batch call recognises page-locked arguments (registered here, or allocated with hipHostMalloc) and lets the DMA
 * engines read / write them directly instead of staging them through its own pinned buffers: no CPU copy at all.
 * Registration costs about a millisecond per 100 MB: worth it for buffers that are reused.  Unregister before free().
 * Page locking works on whole pages, so p must be 4 KiB-aligned and bytes a multiple of 4 KiB (posix_memalign / mmap a
 * buffer of its own: a heap object that shares a page with a neighbour would lock and unlock the neighbour with it);
 * anything else is refused. */
int c25519_amd_host_register(void *p, size_t bytes);
int c25519_amd_host_unregister(void *p);

/* X25519 ------------------------------------------------------------------------------------ */
/* n x curve25519_dh_CreateSharedKey (reference include/curve25519_dh.h:45); sk is clamped in place */
int curve25519_dh_CreateSharedKey_batch(unsigned char *shared, const unsigned char *pk,
                                        unsigned char *sk, size_t n);
int curve25519_dh_CreateSharedKey_dev(void *shared, const void *pk, void *sk, size_t n, void *stream);

/* n x curve25519_dh_CalculatePublicKey (reference :34): ladder on the base point u = 9 */
int curve25519_dh_CalculatePublicKey_batch(unsigned char *pk, unsigned char *sk, size_t n);
int curve25519_dh_CalculatePublicKey_dev(void *pk, void *sk, size_t n, void *stream);

/* n x curve25519_dh_CalculatePublicKey_fast (reference :40): Edwards 8-fold walk + birational map */
int curve25519_dh_CalculatePublicKey_fast_batch(unsigned char *pk, unsigned char *sk, size_t n);
int curve25519_dh_CalculatePublicKey_fast_dev(void *pk, void *sk, size_t n, void *stream);

/* Ed25519 ----------------------------------------------------------------------------------- */
/* n x ed25519_CreateKeyPair (reference include/ed25519_signature.h:40), blinding = NULL */
int ed25519_CreateKeyPair_batch(unsigned char *pub, unsigned char *priv, const unsigned char *sk, size_t n);
int ed25519_CreateKeyPair_dev(void *pub, void *priv, const void *sk, size_t n, void *stream);

/* n x ed25519_SignMessage (reference :47), blinding = NULL, all messages msg_size bytes long */
int ed25519_SignMessage_batch(unsigned char *sig, const unsigned char *priv, const unsigned char *msg,
                              size_t msg_size, size_t n);
int ed25519_SignMessage_dev(void *sig, const void *priv, const void *msg, size_t msg_size, size_t n,
                            void *stream);

/* Blinding (reference include/ed25519_signature.h:54-64, source/ed25519_sign.c:246-331).  A context is 192 bytes --
 * the reference's EDP_BLINDING_CTX size: scalar bl = L - t, 32 random bytes zr, and t*B as four canonical 32-byte
 * elements.  ed25519_Blinding_Init (declared in ed25519_signature.h) derives it ON THE DEVICE from the caller's seed;
 * the functions below take ONE context for the whole batch and compute (k + bl)*B + BP from a Z-randomised starting
 * point, so the walk and its secret-indexed table lookups see a different scalar under every context.  Outputs are
 * byte-identical to the unblinded calls (as in the reference). */
int ed25519_Blinding_Init_dev(void *ctx /* 192 bytes */, const void *seed, size_t seed_len, void *stream);
int ed25519_CreateKeyPair_blinded_batch(unsigned char *pub, unsigned char *priv, const void *blinding,
                                        const unsigned char *sk, size_t n);
int ed25519_CreateKeyPair_blinded_dev(void *pub, void *priv, const void *blinding, const void *sk, size_t n,
                                      void *stream);
int ed25519_SignMessage_blinded_batch(unsigned char *sig, const unsigned char *priv, const void *blinding,
                                      const unsigned char *msg, size_t msg_size, size_t n);
int ed25519_SignMessage_blinded_dev(void *sig, const void *priv, const void *blinding, const void *msg,
                                    size_t msg_size, size_t n, void *stream);

/* the same with messages of different lengths: message i is msgs[offsets[i] .. offsets[i+1]),
 * offsets has n+1 entries (host memory for _batch, device memory for _dev) */
int ed25519_SignMessage_ragged_batch(unsigned char *sig, const unsigned char *priv, const unsigned char *msgs,
                                     const uint64_t *offsets, size_t n);
int ed25519_SignMessage_ragged_dev(void *sig, const void *priv, const void *msgs, const uint64_t *offsets,
                                   size_t n, void *stream);

/* n x ed25519_VerifySignature (reference :67): full Init + Check per element, distinct keys.
 * Cost depends on the INPUT, which an untrusted sender controls: a key that does not decompress onto the curve sends its
 * element through the reference's own operation order in a kernel behind the walk.  Measured at n = 2^20
 * (profiles/r06_verify_worst_case.txt): all keys on the curve 1.00 x (9.5 ms), ONE off-curve key in the batch 1.15 x (one
 * lane's latency of the reference-order path, about 1.4 ms), one per 256 elements 1.15 x, every second key 1.53 x (the
 * worst case), every key 1.46 x.  Verdicts are the reference's in every case; a caller that must bound latency can pre-screen keys, or keep
 * batches from different senders apart. */
int ed25519_VerifySignature_batch(int *verdict, const unsigned char *sig, const unsigned char *pk,
                                  const unsigned char *msg, size_t msg_size, size_t n);
int ed25519_VerifySignature_dev(void *verdict, const void *sig, const void *pk, const void *msg,
                                size_t msg_size, size_t n, void *stream);

int ed25519_VerifySignature_ragged_batch(int *verdict, const unsigned char *sig, const unsigned char *pk,
                                         const unsigned char *msgs, const uint64_t *offsets, size_t n);
int ed25519_VerifySignature_ragged_dev(void *verdict, const void *sig, const void *pk, const void *msgs,
                                       const uint64_t *offsets, size_t n, void *stream);

/* ed25519_VerifySignature_* decide every element whose key decompresses onto the curve with an exact
 * lattice-shortened walk (csrc/verify_fast.cuh, ~134 doublings instead of 255) and run the reference's own operation
 * order only for the others (set C25519_AMD_VERIFY_REFERENCE_ORDER=1 to force it for everything).  This reports how many
 * elements of the calling thread's last verification on the current device took the reference-order kernel; -1 when
 * there is nothing to report (a *_batch call that was cut into pieces reports one of its pieces).  Synchronises with
 * that call's stream. */
long c25519_amd_verify_last_slow_elements(void);
/* did the calling thread's last ed25519_Verify_Check_* call on this device walk the two wide combs (1) or did the reference-order
 * kernel decide it (0)?  -1: no such call.  Synchronises with that call's stream. */
long c25519_amd_verify_check_last_wide(void);

/* bytes of device scratch ed25519_VerifySignature_dev needs for n elements (per-lane 4-fold tables);
 * the library allocates and caches it per host thread (about 2.8 KB per element). */
size_t ed25519_VerifySignature_scratch_bytes(size_t n);

/* Two-phase verification (reference include/ed25519_signature.h:77-93): one key, many signatures.
 * A context is 2080 bytes -- the reference's EDP_SIGV_CTX size: the 32-byte key followed by the 16-row
 * 4-fold table of 2^(64i)*(-A) subset sums, four canonical 32-byte field elements per row.
 *   Verify_Init_batch / _dev : n keys -> n contexts (n x 2080 bytes)
 *   Verify_Check_batch / _dev: ONE context, n (signature, message) pairs -> n verdicts; this is the
 *                              amortised path, the per-key table is staged in LDS.  From 2^16 pairs per call (tunable
 *                              ONE_KEY_WIDE) a context that is byte for byte Verify_Init's, for a key on the curve, is
 *                              verified over two wide fixed-base combs instead -- the base point's and one generated for the
 *                              key (0.6 ms; kept, with the context it belongs to, in 2 MiB of device memory per calling
 *                              thread until the thread exits or calls c25519_amd_thread_release(), so the next call with
 *                              the same context skips it): the same verdicts at 2.6 x the rate; any other context keeps
 *                              the first kernel, which reads the context's rows as they are, as the reference does. */
int ed25519_Verify_Init_batch(void *ctx, const unsigned char *pk, size_t n);
int ed25519_Verify_Init_dev(void *ctx, const void *pk, size_t n, void *stream);
int ed25519_Verify_Check_batch(int *verdict, const void *ctx, const unsigned char *sig,
                               const unsigned char *msg, size_t msg_size, size_t n);
int ed25519_Verify_Check_dev(void *verdict, const void *ctx, const void *sig, const void *msg,
                             size_t msg_size, size_t n, void *stream);

/* Multi-GPU (BASELINE.json north_star: "batches shard embarrassingly across the 8 GPUs of one node
 * with a single RCCL gather over xGMI") ------------------------------------------------------------------------------
 * A handle owns ONE WORKER THREAD PER DEVICE.  A call cuts the batch into contiguous shards (device d owns elements
 * [n*d/n_dev, n*(d+1)/n_dev)); every worker runs its shard through the same pinned, pieced pipeline as the single-GPU
 * *_batch functions (all devices upload over their own PCIe links at the same time, nothing is copied from or to
 * pageable memory), results stay resident on the device and are gathered to devices[0] with a grouped ncclGather (RCCL,
 * /opt/rocm/include/rccl/rccl.h:745; loaded with dlopen on first use) while a second thread on the root streams the
 * gathered rows to the caller -- piece by piece: every device cuts its shard at the same rows, a piece is gathered as soon
 * as every device has computed it and handed over while the devices compute the next ones, so only the last piece's
 * gather and download are exposed.  A handle of one device skips the gather.  With host destinations every result row of the gather mode crosses the ROOT's
 * PCIe link; c25519_amd_multi_set_gather(m, 0) leaves the gather out and lets every device hand its own rows to the
 * caller over its own link (same results; the gather is what BASELINE.json's north_star names, and the default).
 * Host pointers, synchronous, same byte layouts and results as the *_batch functions.  The handle also owns one stream and
 * one RCCL communicator per device and grow-only result buffers (zeroed before they are freed); it is not thread-safe
 * (one call at a time per handle). */
typedef struct c25519_amd_multi c25519_amd_multi;
int  c25519_amd_multi_create(c25519_amd_multi **m, const int *devices, int n_dev);   /* 1 <= n_dev <= 64 */
void c25519_amd_multi_destroy(c25519_amd_multi *m);
int  c25519_amd_multi_device_count(const c25519_amd_multi *m);
int  c25519_amd_multi_helper_threads(const c25519_amd_multi *m);  /* host threads that copy memory while a call runs */
int  c25519_amd_multi_set_gather(c25519_amd_multi *m, int on);   /* 1 (default): gather to devices[0]; 0: per-device downloads */
int curve25519_dh_CreateSharedKey_multi(c25519_amd_multi *m, unsigned char *shared, const unsigned char *pk,
                                        unsigned char *sk, size_t n);
int ed25519_SignMessage_multi(c25519_amd_multi *m, unsigned char *sig, const unsigned char *priv,
                              const unsigned char *msg, size_t msg_size, size_t n);
int ed25519_VerifySignature_multi(c25519_amd_multi *m, int *verdict, const unsigned char *sig, const unsigned char *pk,
                                  const unsigned char *msg, size_t msg_size, size_t n);

/* introspection used by tests and bench ------------------------------------------------------ */
/* copies the device-generated 256 x 96-byte 8-fold base table (canonical Y+X, Y-X, 2dT rows --
 * the content of reference source/base_folding8.h) to `out` */
int c25519_amd_base_table(unsigned char *out /* 24576 bytes */);

/* test hook: the encoded point T = s*B + h*(-A) that ed25519_Verify_Check compares with enc(R)
 * (reference source/ed25519_verify.c:309-310) instead of the verdict; device pointers, out is n x 32 bytes */
int c25519_amd_verify_point_dev(void *out, const void *sig, const void *pk, const void *msg, size_t msg_size,
                                size_t n, void *stream);

/* device field arithmetic on n pairs of 32-byte little-endian values taken mod p = 2^255-19 (host pointers):
 * out[i] = canonical(op(a[i], b[i])), op 0 mul, 1 square, 2 add, 3 sub, 4 inverse, 5 a^((p-5)/8),
 * 6 canonicalise, 7 (a-b)*(a+b), 8 a^2-b, 9 2a^2+(a+b)-b, 10 a+121665b, 11 9a, 12 inverse as a^(p-2) (the reference's
 * ecp_Inverse chain), 13 inverse by constant-time division steps (what op 4 and every kernel run; 0 -> 0 either way).
 * The unit-test hook for the L0 layer (the reference's ECP_SELF_TEST checks, test/curve25519_selftest.c:640-741). */
int c25519_amd_fe_selftest(unsigned char *out, const unsigned char *a, const unsigned char *b, size_t n, int op);

/* device scalar arithmetic mod L (reference source/curve25519_order.c, unit checks test/curve25519_selftest.c:624-714):
 * a is n x 64 bytes (512-bit little-endian), b n x 32 bytes, out n x 32 bytes.
 *   op 0 canonical(a mod L)   1 raw a mod L            2 canonical(a[0..31] mod L)   3 raw a[0..31]*b
 *   op 4 raw a[0..31]+b       5 raw a[32..35]*2^256 + a[0..31]   (eco_ReduceHiWord)  6 canonical(a[0..31]*b + a[32..63])
 * "raw" = 256 bits congruent to the exact value mod L, not necessarily below L. */
int c25519_amd_sc_selftest(unsigned char *out, const unsigned char *a, const unsigned char *b, size_t n, int op);

/* fold recodings of n 32-byte scalars (reference ecp_8Folds / ecp_4Folds, source/curve25519_utils.c:144 / :125):
 * out is n x 128 bytes: the 32 8-fold columns as the fixed-base walk indexes them, the same 32 as the reference-order
 * verification walk consumes them, and the 64 4-fold columns. */
int c25519_amd_fold_selftest(unsigned char *out, const unsigned char *k, size_t n);

#ifdef __cplusplus
}
#endif
#endif

/* tests/c/readtsc.c -- TEST INFRASTRUCTURE.  The cycle counter the reference's test/openssl_test.c declares
 * (`uint64_t readTSC();`, :11) and takes from its asm library or from test/curve25519_test.c:36-50; needed to link that
 * harness against the HIP drop-in library (oracle/Makefile, target ref-openssl). */
#include <stdint.h>

uint64_t readTSC(void)
{
#if defined(__x86_64__) || defined(__i386__)
    uint32_t lo, hi;
    __asm__ __volatile__("rdtsc" : "=a"(lo), "=d"(hi));
    return ((uint64_t)hi << 32) | lo;
#else
    static uint64_t t;
    return ++t;
#endif
}

// tools/ubench/mad_peak.hip -- issue rates from WHOLE-ASM loops (one asm statement holds the loop, so the compiler can
// neither pad instruction boundaries with s_nop nor move the loop top), 8 independent chains in fixed registers:
//   * how much of a short loop's v_mad_u64_u32 rate is the instruction and how much is the loop: 16 / 64 / 128 / 256
//     MADs per trip, loop top on a 64-byte line or pushed off it
//   * the long-trip (128 per trip) rate of every instruction class the field arithmetic uses, and of one product
//     column as the kernels issue it (10 dependent MADs + mask + 64-bit shift)
// The long-trip accumulating-MAD figure is the denominator of bench.py's roofline.valu (JSON: argv[1]).
// Build: hipcc --offload-arch=gfx950 -O3 mad_peak.hip -o mad_peak
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// eight instructions, one per chain; chain c lives in v[40+2c : 41+2c], factors in v38, v39
#define I8(op, tail) \
    op " v[40:41]" tail(40) "\n\t" op " v[42:43]" tail(42) "\n\t" op " v[44:45]" tail(44) "\n\t" op " v[46:47]" tail(46) "\n\t" \
    op " v[48:49]" tail(48) "\n\t" op " v[50:51]" tail(50) "\n\t" op " v[52:53]" tail(52) "\n\t" op " v[54:55]" tail(54) "\n\t"
#define S8(op, tail) \
    op " v40" tail(40) "\n\t" op " v42" tail(42) "\n\t" op " v44" tail(44) "\n\t" op " v46" tail(46) "\n\t" \
    op " v48" tail(48) "\n\t" op " v50" tail(50) "\n\t" op " v52" tail(52) "\n\t" op " v54" tail(54) "\n\t"
#define STR(x) #x
#define T_MADACC(r) ", vcc, v38, v39, v[" STR(r) ":" "%=" "]"
// (register pair text cannot be computed in the preprocessor: spell the eight out)
#define MAD8 \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v38, v39, v[42:43]\n\t" \
    "v_mad_u64_u32 v[44:45], vcc, v38, v39, v[44:45]\n\tv_mad_u64_u32 v[46:47], vcc, v38, v39, v[46:47]\n\t" \
    "v_mad_u64_u32 v[48:49], vcc, v38, v39, v[48:49]\n\tv_mad_u64_u32 v[50:51], vcc, v38, v39, v[50:51]\n\t" \
    "v_mad_u64_u32 v[52:53], vcc, v38, v39, v[52:53]\n\tv_mad_u64_u32 v[54:55], vcc, v38, v39, v[54:55]\n\t"
#define MADZ8 \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, 0\n\tv_mad_u64_u32 v[42:43], vcc, v38, v39, 0\n\t" \
    "v_mad_u64_u32 v[44:45], vcc, v38, v39, 0\n\tv_mad_u64_u32 v[46:47], vcc, v38, v39, 0\n\t" \
    "v_mad_u64_u32 v[48:49], vcc, v38, v39, 0\n\tv_mad_u64_u32 v[50:51], vcc, v38, v39, 0\n\t" \
    "v_mad_u64_u32 v[52:53], vcc, v38, v39, 0\n\tv_mad_u64_u32 v[54:55], vcc, v38, v39, 0\n\t"
// one dependent chain of eight: what a product column is
#define MADDEP8 \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t"
#define MUL8 \
    "v_mul_lo_u32 v40, v40, v38\n\tv_mul_lo_u32 v42, v42, v38\n\tv_mul_lo_u32 v44, v44, v38\n\tv_mul_lo_u32 v46, v46, v38\n\t" \
    "v_mul_lo_u32 v48, v48, v38\n\tv_mul_lo_u32 v50, v50, v38\n\tv_mul_lo_u32 v52, v52, v38\n\tv_mul_lo_u32 v54, v54, v38\n\t"
#define ADD8 \
    "v_add_u32 v40, v40, v38\n\tv_add_u32 v42, v42, v38\n\tv_add_u32 v44, v44, v38\n\tv_add_u32 v46, v46, v38\n\t" \
    "v_add_u32 v48, v48, v38\n\tv_add_u32 v50, v50, v38\n\tv_add_u32 v52, v52, v38\n\tv_add_u32 v54, v54, v38\n\t"
#define ADDS8 /* VOP3 encoding: an SGPR second operand */ \
    "v_add_u32 v40, v40, s21\n\tv_add_u32 v42, v42, s21\n\tv_add_u32 v44, v44, s21\n\tv_add_u32 v46, v46, s21\n\t" \
    "v_add_u32 v48, v48, s21\n\tv_add_u32 v50, v50, s21\n\tv_add_u32 v52, v52, s21\n\tv_add_u32 v54, v54, s21\n\t"
#define AND8 \
    "v_and_b32 v40, v40, v38\n\tv_and_b32 v42, v42, v38\n\tv_and_b32 v44, v44, v38\n\tv_and_b32 v46, v46, v38\n\t" \
    "v_and_b32 v48, v48, v38\n\tv_and_b32 v50, v50, v38\n\tv_and_b32 v52, v52, v38\n\tv_and_b32 v54, v54, v38\n\t"
#define ANDL8 /* 32-bit literal operand: an 8-byte VOP2 */ \
    "v_and_b32 v40, 0x3ffffff, v40\n\tv_and_b32 v42, 0x3ffffff, v42\n\tv_and_b32 v44, 0x3ffffff, v44\n\tv_and_b32 v46, 0x3ffffff, v46\n\t" \
    "v_and_b32 v48, 0x3ffffff, v48\n\tv_and_b32 v50, 0x3ffffff, v50\n\tv_and_b32 v52, 0x3ffffff, v52\n\tv_and_b32 v54, 0x3ffffff, v54\n\t"
#define SHR64_8 \
    "v_lshrrev_b64 v[40:41], 1, v[40:41]\n\tv_lshrrev_b64 v[42:43], 1, v[42:43]\n\tv_lshrrev_b64 v[44:45], 1, v[44:45]\n\tv_lshrrev_b64 v[46:47], 1, v[46:47]\n\t" \
    "v_lshrrev_b64 v[48:49], 1, v[48:49]\n\tv_lshrrev_b64 v[50:51], 1, v[50:51]\n\tv_lshrrev_b64 v[52:53], 1, v[52:53]\n\tv_lshrrev_b64 v[54:55], 1, v[54:55]\n\t"
#define CNDE32_8 \
    "v_cndmask_b32 v40, v40, v38, vcc\n\tv_cndmask_b32 v42, v42, v38, vcc\n\tv_cndmask_b32 v44, v44, v38, vcc\n\tv_cndmask_b32 v46, v46, v38, vcc\n\t" \
    "v_cndmask_b32 v48, v48, v38, vcc\n\tv_cndmask_b32 v50, v50, v38, vcc\n\tv_cndmask_b32 v52, v52, v38, vcc\n\tv_cndmask_b32 v54, v54, v38, vcc\n\t"
// the shipped multiplication's instruction stream: ten product columns one after the other, each 10 dependent MADs, the
// limb mask and the 64-bit carry shift that feeds the next column (120 instructions); two of them per trip
#define COL(l) \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\t" \
    "v_and_b32 " l ", 0x3ffffff, v40\n\tv_lshrrev_b64 v[40:41], 26, v[40:41]\n\t"
#define FIELD_MUL COL("v58") COL("v59") COL("v60") COL("v61") COL("v62") COL("v63") COL("v64") COL("v65") COL("v66") COL("v67")
// two independent products column by column, their MAD chains interleaved (what a dual-product primitive would issue):
// the same 240 instructions per trip as two FIELD_MULs, but no MAD waits for the one right before it
#define COL2(l, m) \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_and_b32 " l ", 0x3ffffff, v40\n\tv_lshrrev_b64 v[40:41], 26, v[40:41]\n\t" \
    "v_and_b32 " m ", 0x3ffffff, v42\n\tv_lshrrev_b64 v[42:43], 26, v[42:43]\n\t"
#define FIELD_MUL2 COL2("v58", "v68") COL2("v59", "v69") COL2("v60", "v70") COL2("v61", "v71") COL2("v62", "v72") \
                   COL2("v63", "v73") COL2("v64", "v58") COL2("v65", "v59") COL2("v66", "v60") COL2("v67", "v61")
// ONE product, columns in pairs: the even column continues from the carry, the odd one starts from zero beside it and
// takes the even column's carry in one 64-bit add afterwards (125 instructions per product, two per trip)
#define COLPAIR(l, m) \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, 0\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v39, v38, v[42:43]\n\t" \
    "v_and_b32 " l ", 0x3ffffff, v40\n\tv_lshrrev_b64 v[40:41], 26, v[40:41]\n\t" \
    "v_lshl_add_u64 v[42:43], v[40:41], 0, v[42:43]\n\t" \
    "v_and_b32 " m ", 0x1ffffff, v42\n\tv_lshrrev_b64 v[40:41], 25, v[42:43]\n\t"
#define FIELD_MULC COLPAIR("v58", "v59") COLPAIR("v60", "v61") COLPAIR("v62", "v63") COLPAIR("v64", "v65") COLPAIR("v66", "v67")
// one product with a full-rate instruction of the same product (a mask) between every two dependent MADs: does an
// independent VALU instruction in between hide the dependent-issue penalty?  (10 x [10 x (MAD, and), shift]: 210)
#define COLI(l) \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_and_b32 v68, 0x3ffffff, v58\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\tv_and_b32 v69, 0x3ffffff, v59\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_and_b32 v70, 0x3ffffff, v60\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\tv_and_b32 v71, 0x3ffffff, v61\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_and_b32 v72, 0x3ffffff, v62\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\tv_and_b32 v73, 0x3ffffff, v63\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_and_b32 v68, 0x3ffffff, v64\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\tv_and_b32 v69, 0x3ffffff, v65\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v38, v39, v[40:41]\n\tv_and_b32 v70, 0x3ffffff, v66\n\t" \
    "v_mad_u64_u32 v[40:41], vcc, v39, v38, v[40:41]\n\tv_and_b32 " l ", 0x3ffffff, v40\n\t" \
    "v_lshrrev_b64 v[40:41], 26, v[40:41]\n\t"
#define FIELD_MULI COLI("v58") COLI("v59") COLI("v60") COLI("v61") COLI("v62") COLI("v63") COLI("v64") COLI("v65") COLI("v66") COLI("v67")
// full-rate instructions between MADs: the same 8 MADs + 8 masks per group, in runs of 1 / 2 / 4 / 8 (round 4: what does a
// VOP2 instruction cost when it is NOT part of a run of them?)
#define M_(c) "v_mad_u64_u32 v[" #c ":" c1_##c "], vcc, v38, v39, v[" #c ":" c1_##c "]\n\t"
#define c1_40 "41"
#define c1_42 "43"
#define c1_44 "45"
#define c1_46 "47"
#define c1_48 "49"
#define c1_50 "51"
#define c1_52 "53"
#define c1_54 "55"
#define A_(r) "v_and_b32 v" #r ", 0x3ffffff, v" #r "\n\t"
#define RUN1 M_(40) A_(58) M_(42) A_(59) M_(44) A_(60) M_(46) A_(61) M_(48) A_(62) M_(50) A_(63) M_(52) A_(64) M_(54) A_(65)
#define RUN2 M_(40) M_(42) A_(58) A_(59) M_(44) M_(46) A_(60) A_(61) M_(48) M_(50) A_(62) A_(63) M_(52) M_(54) A_(64) A_(65)
#define RUN4 M_(40) M_(42) M_(44) M_(46) A_(58) A_(59) A_(60) A_(61) M_(48) M_(50) M_(52) M_(54) A_(62) A_(63) A_(64) A_(65)
#define RUN8 M_(40) M_(42) M_(44) M_(46) M_(48) M_(50) M_(52) M_(54) A_(58) A_(59) A_(60) A_(61) A_(62) A_(63) A_(64) A_(65)
// the same with the half-rate 64-bit shift in the place of the MAD, and a mask pair after every third MAD (what a column is)
#define S_(c) "v_lshrrev_b64 v[" #c ":" c1_##c "], 1, v[" #c ":" c1_##c "]\n\t"
#define RUN1S S_(40) A_(58) S_(42) A_(59) S_(44) A_(60) S_(46) A_(61) S_(48) A_(62) S_(50) A_(63) S_(52) A_(64) S_(54) A_(65)
#define ADDV_(r) "v_add_u32 v" #r ", v" #r ", v38\n\t"
#define RUN1ADD M_(40) ADDV_(58) M_(42) ADDV_(59) M_(44) ADDV_(60) M_(46) ADDV_(61) M_(48) ADDV_(62) M_(50) ADDV_(63) M_(52) ADDV_(64) M_(54) ADDV_(65)
#define X2(B) B B
#define X8(B) X2(X2(X2(B)))
#define X16(B) X2(X8(B))
#define X32(B) X2(X16(B))

#define PAD0 ""
#define PAD2 "s_nop 0\n\ts_nop 0\n\t"
#define PAD6 PAD2 PAD2 PAD2
#define PAD10 PAD6 PAD2 PAD2
#define PAD14 PAD10 PAD2 PAD2

#define LOOP(PADS, B) LOOPI("", PADS, B)
#define LOOPI(INIT, PADS, B) \
    asm volatile( \
        "v_mov_b32 v38, %1\n\tv_mov_b32 v39, %2\n\tv_mov_b32 v56, 1.0\n\tv_mov_b32 v57, 1.0\n\t" \
        "v_mov_b32 v40, %1\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, %1\n\tv_mov_b32 v45, 0\n\t" \
        "v_mov_b32 v46, %2\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, %1\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, %2\n\tv_mov_b32 v51, 0\n\t" \
        "v_mov_b32 v52, %1\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, %2\n\tv_mov_b32 v55, 0\n\t" \
        "v_mov_b32 v58, %1\n\tv_mov_b32 v59, 0\n\tv_mov_b32 v60, %2\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, %1\n\tv_mov_b32 v63, 0\n\t" \
        "v_mov_b32 v64, %2\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, %1\n\tv_mov_b32 v67, 0\n\tv_mov_b32 v68, %2\n\tv_mov_b32 v69, 0\n\t" \
        "v_mov_b32 v70, %1\n\tv_mov_b32 v71, 0\n\tv_mov_b32 v72, %2\n\tv_mov_b32 v73, 0\n\t" \
        INIT "s_mov_b32 s20, %3\n\ts_mov_b32 s21, 7\n\t" \
        "s_branch 2f\n\t.p2align 6\n\t2:\n\t" PADS \
        "1:\n\t" B \
        "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t" \
        "v_xor_b32 %0, v40, v42\n\tv_xor_b32 %0, %0, v44\n\tv_xor_b32 %0, %0, v46\n\tv_xor_b32 %0, %0, v48\n\t" \
        "v_xor_b32 %0, %0, v50\n\tv_xor_b32 %0, %0, v52\n\tv_xor_b32 %0, %0, v54" \
        : "=v"(r) : "v"(a), "v"(b), "s"(trips) \
        : "vcc", "scc", "s20", "s21", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", \
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
          "v68", "v69", "v70", "v71", "v72", "v73")

// every wave also leaves its own duration in shader cycles (s_memtime: one tick per shader cycle): the longest one of a
// launch in which all waves are resident from the start is the launch's SIMD time, so cycles per instruction and the
// clock the chip sustained under that stream come out of the same run as the lane-op/s figure (round 4)
__device__ __forceinline__ unsigned long long now_cycles()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

template <int ID>
__global__ void __launch_bounds__(256) k_loop(unsigned* out, unsigned seed, int trips, unsigned long long* cyc)
{
    const unsigned long long t0 = now_cycles();
    unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, r = 0;
    if constexpr (ID == 0) LOOP(PAD0, X2(MAD8));
    if constexpr (ID == 1) LOOP(PAD2, X2(MAD8));
    if constexpr (ID == 2) LOOP(PAD6, X2(MAD8));
    if constexpr (ID == 3) LOOP(PAD10, X2(MAD8));
    if constexpr (ID == 4) LOOP(PAD14, X2(MAD8));
    if constexpr (ID == 5) LOOP(PAD0, X8(MAD8));
    if constexpr (ID == 6) LOOP(PAD0, X16(MAD8));
    if constexpr (ID == 7) LOOP(PAD0, X32(MAD8));
    if constexpr (ID == 8) LOOP(PAD0, X16(MADZ8));
    if constexpr (ID == 9) LOOP(PAD0, X16(MADDEP8));
    if constexpr (ID == 10) LOOP(PAD0, X16(MUL8));
    if constexpr (ID == 11) LOOP(PAD0, X16(SHR64_8));
    if constexpr (ID == 12) LOOP(PAD0, X16(ADD8));
    if constexpr (ID == 13) LOOP(PAD0, X16(ADDS8));
    if constexpr (ID == 14) LOOP(PAD0, X16(AND8));
    if constexpr (ID == 15) LOOP(PAD0, X16(ANDL8));
    if constexpr (ID == 16) LOOP(PAD0, X16(CNDE32_8));
    if constexpr (ID == 18) LOOP(PAD0, X2(FIELD_MUL));
    if constexpr (ID == 19) LOOP(PAD0, FIELD_MUL2);
    if constexpr (ID == 20) LOOP(PAD0, X2(FIELD_MULC));
    if constexpr (ID == 21) LOOP(PAD0, FIELD_MULI);
    if constexpr (ID == 22) LOOP(PAD0, X8(RUN1));
    if constexpr (ID == 23) LOOP(PAD0, X8(RUN2));
    if constexpr (ID == 24) LOOP(PAD0, X8(RUN4));
    if constexpr (ID == 25) LOOP(PAD0, X8(RUN8));
    if constexpr (ID == 26) LOOP(PAD0, X8(RUN1S));
    if constexpr (ID == 27) LOOP(PAD0, X8(RUN1ADD));
    // two streams on one SIMD: the workgroups alternate between a pure MAD loop and a pure VOP2 loop (a workgroup is four
    // waves, one per SIMD, so every SIMD holds both kinds); ID 28 at equal wave priority, ID 29 with the VOP2 waves at
    // priority 0 and the MAD waves at 1 -- can the two classes run side by side?
    if constexpr (ID == 28 || ID == 29) {
        if (blockIdx.x & 1) {
            if (ID == 29) __builtin_amdgcn_s_setprio(1);
            LOOP(PAD0, X16(MAD8));
        } else {
            if (ID == 29) __builtin_amdgcn_s_setprio(0);
            LOOP(PAD0, X16(AND8));
        }
    }
    // does a wave-priority instruction between MADs change what a MAD costs?  (the ladder with priority dips around its
    // VOP2 runs costs 4.0 cycles per 4-cycle-class instruction where the plain one costs 4.26: profiles/r04_cycle_probe.txt)
#define DIP "s_setprio 0\n\ts_setprio 1\n\t"
#define ONE "s_setprio 1\n\t"
#define NOP "s_nop 0\n\t"
    if constexpr (ID == 30) LOOP(PAD0, X16(MAD8 DIP));
    if constexpr (ID == 31) LOOP(PAD0, X8(MAD8 MAD8 DIP));
    if constexpr (ID == 32) LOOP(PAD0, X16(MAD8 ONE));
    if constexpr (ID == 33) LOOP(PAD0, X16(MAD8 NOP));
    if constexpr (ID == 34) LOOP(PAD0, X16(MADDEP8 DIP));
    if (r == 0x12345678u) out[0] = r;
    const unsigned long long t1 = now_cycles();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

struct Row { int id; int per_trip; double scale; const char* key; const char* name; };
static const Row rows[] = {
    { 0, 16, 1, "mad_acc_16_aligned", "v_mad_u64_u32 acc,  16 per trip, loop top on a 64 B line" },
    { 1, 16, 1, "mad_acc_16_off8", "v_mad_u64_u32 acc,  16 per trip, loop top + 8 B" },
    { 2, 16, 1, "mad_acc_16_off24", "v_mad_u64_u32 acc,  16 per trip, loop top + 24 B" },
    { 3, 16, 1, "mad_acc_16_off40", "v_mad_u64_u32 acc,  16 per trip, loop top + 40 B" },
    { 4, 16, 1, "mad_acc_16_off56", "v_mad_u64_u32 acc,  16 per trip, loop top + 56 B" },
    { 5, 64, 1, "mad_acc_64", "v_mad_u64_u32 acc,  64 per trip" },
    { 6, 128, 1, "v_mad_u64_u32", "v_mad_u64_u32 acc, 128 per trip   <- roofline.valu peak" },
    { 7, 256, 1, "mad_acc_256", "v_mad_u64_u32 acc, 256 per trip" },
    { 8, 128, 1, "mad_zero_addend", "v_mad_u64_u32 zero addend, 128 per trip" },
    { 9, 128, 1, "mad_one_chain", "v_mad_u64_u32 acc, ONE dependent chain, 128 per trip" },
    { 10, 128, 1, "v_mul_lo_u32", "v_mul_lo_u32, 128 per trip" },
    { 11, 128, 1, "v_lshrrev_b64", "v_lshrrev_b64, 128 per trip" },
    { 12, 128, 1, "v_add_u32", "v_add_u32 (VOP2), 128 per trip" },
    { 13, 128, 1, "v_add_u32_sgpr", "v_add_u32 with an SGPR operand (VOP3), 128 per trip" },
    { 14, 128, 1, "v_and_b32", "v_and_b32 (VOP2), 128 per trip" },
    { 15, 128, 1, "v_and_b32_literal", "v_and_b32 with a 32-bit literal (8-byte VOP2), 128 per trip" },
    { 16, 128, 1, "v_cndmask_b32_vcc_run", "v_cndmask_b32 (VOP2, vcc), back to back, 128 per trip" },
    { 18, 240, 1, "field_mul_stream", "the shipped field multiplication's stream (10 x [10 dependent MADs, mask, shift]), 240 per trip" },
    { 19, 240, 1, "field_mul_two_products", "two products interleaved MAD by MAD (same 240 instructions per trip)" },
    { 20, 250, 1, "field_mul_column_pairs", "one product, columns in pairs + one 64-bit add per pair (2 x 125 per trip)" },
    { 21, 210, 1, "field_mul_mad_and", "one product, a full-rate mask between dependent MADs (210 per trip, 100 MADs)" },
    { 22, 128, 1, "mad_and_runs_of_1", "MAD, mask, MAD, mask ... (64 + 64 per trip, independent)" },
    { 23, 128, 1, "mad_and_runs_of_2", "2 MADs, 2 masks, ... (64 + 64 per trip)" },
    { 24, 128, 1, "mad_and_runs_of_4", "4 MADs, 4 masks, ... (64 + 64 per trip)" },
    { 25, 128, 1, "mad_and_runs_of_8", "8 MADs, 8 masks, ... (64 + 64 per trip)" },
    { 26, 128, 1, "shr64_and_runs_of_1", "v_lshrrev_b64, mask, v_lshrrev_b64, mask ... (64 + 64 per trip)" },
    { 27, 128, 1, "mad_add_runs_of_1", "MAD, v_add_u32 (VOP2, two VGPRs), MAD, v_add_u32 ... (64 + 64 per trip)" },
    { 30, 128, 1, "mad_dip_every_8", "v_mad_u64_u32, s_setprio 0 ; s_setprio 1 behind every 8 (128 MADs per trip)" },
    { 31, 128, 1, "mad_dip_every_16", "v_mad_u64_u32, s_setprio 0 ; s_setprio 1 behind every 16" },
    { 32, 128, 1, "mad_setprio1_every_8", "v_mad_u64_u32, s_setprio 1 behind every 8 (no change of level)" },
    { 33, 128, 1, "mad_nop_every_8", "v_mad_u64_u32, s_nop 0 behind every 8" },
    { 34, 128, 1, "mad_one_chain_dip_every_8", "v_mad_u64_u32, ONE dependent chain, the dip behind every 8" },
    { 28, 128, 1, "mad_waves_and_vop2_waves", "half the waves pure MAD, half pure v_and_b32, equal priority" },
    { 29, 128, 1, "mad_waves_and_vop2_waves_prio", "half the waves pure MAD (priority 1), half pure v_and_b32 (priority 0)" },
};
// the field streams again at lower occupancy: what the dependent-MAD penalty costs with 4 / 2 / 1 waves per SIMD
static const int occupancy_rows[] = { 6, 9, 14, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34 };

static unsigned long long* g_cyc;          // [MAX_WAVES] device, g_cyc_host its mirror
static unsigned long long* g_cyc_host;
constexpr int MAX_WAVES = 256 * 8 * 4 * 2;
template <int ID> static void launch(int blocks, unsigned* d, int trips, hipStream_t s) { k_loop<ID><<<blocks, 256, 0, s>>>(d, 1, trips, g_cyc); }
// the longest wave of the last launch, in shader cycles
static double longest_wave(int blocks)
{
    CHECK(hipMemcpy(g_cyc_host, g_cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
    unsigned long long m = 0;
    for (int i = 0; i < blocks * 4; i++) if (g_cyc_host[i] > m) m = g_cyc_host[i];
    return (double)m;
}
static void dispatch(int id, int blocks, unsigned* d, int trips, hipStream_t s)
{
    switch (id) {
#define C(k) case k: launch<k>(blocks, d, trips, s); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(33) C(34)
#undef C
    }
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* d; CHECK(hipMalloc(&d, 64));
    CHECK(hipMalloc(&g_cyc, sizeof(unsigned long long) * MAX_WAVES));
    g_cyc_host = (unsigned long long*)malloc(sizeof(unsigned long long) * MAX_WAVES);
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (argc > 1 && !strcmp(argv[1], "--quick")) {
        // bench.py's same-box, same-run roof: ONE configuration -- the accumulating v_mad_u64_u32 stream, 128 per trip, 8 waves per
        // SIMD launched: roofline.valu's peak row -- behind ~60 ms of the same launches (the chip's clock ramp); one JSON line on stdout
        const Row* r = nullptr;
        for (const Row& q : rows) if (!strcmp(q.key, "v_mad_u64_u32")) r = &q;
        if (!r) { fprintf(stderr, "mad_peak --quick: no such row\n"); return 1; }
        const int trips = 65536 / r->per_trip, blocks = cus * 8;
        float best = 1e30f, total = 0;
        for (int rep = 0; rep < 400 && (rep < 8 || total < 60.0f); rep++) {          // the ramp: untimed in effect (best-of below)
            CHECK(hipEventRecord(e0, s));
            dispatch(r->id, blocks, d, trips, s);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
        float sum = 0;
        const int reps = 10;
        for (int rep = 0; rep < reps; rep++) {
            CHECK(hipEventRecord(e0, s));
            dispatch(r->id, blocks, d, trips, s);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            sum += ms;
            if (ms < best) best = ms;
        }
        const double per = r->scale * (double)blocks * 4 * (double)trips * r->per_trip * 64;
        printf("{\"v_mad_u64_u32\": %.4e, \"v_mad_u64_u32_mean_of_%d\": %.4e, \"ms_best\": %.4f, \"device_cus\": %d, \"waves_per_simd\": 8, "
               "\"unit\": \"lane-op/s\", \"row\": \"%s\"}\n", per / (best * 1e-3), reps, per / (sum / reps * 1e-3), best, cus, r->name);
        return 0;
    }
    FILE* jf = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (jf) fprintf(jf, "{\"device_cus\": %d, \"waves_per_simd\": 8, \"unit\": \"lane-op/s\", \"rates\": {\n", cus);
    // argv[2]: SIMD cycles per wave-instruction of every row (s_memtime inside the kernels), the issue model's class costs
    FILE* cj = argc > 2 ? fopen(argv[2], "w") : nullptr;
    bool cfirst = true;
    if (cj) fprintf(cj, "{\"unit\": \"SIMD cycles per wave-instruction\", \"cycles\": {\n");
    printf("device %s, %d CUs; 8 waves per SIMD launched (6 resident at a time: 74 VGPRs), ~65536 instructions per wave, best of 5\n", prop.name, cus);
    bool first = true;
    for (const Row& r : rows) {
        const int trips = 65536 / r.per_trip;
        const int blocks = cus * 8;
        float best = 1e30f;
        double cycles = 0;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0, s));
            dispatch(r.id, blocks, d, trips, s);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) { best = ms; cycles = longest_wave(blocks); }
        }
        const double rate = r.scale * (double)blocks * 4 * (double)trips * r.per_trip * 64 / (best * 1e-3);
        // (no cycle figure for these rows: the loops hold 74 VGPRs, so six of a SIMD's eight waves are resident and the other
        // two queue behind them -- the longest wave is not the launch.  The by-occupancy blocks below launch what fits.)
        (void)cycles;
        printf("%-78s %8.3f ms  %7.2f T lane-op/s\n", r.name, best, rate / 1e12);
        if (jf) { fprintf(jf, "%s  \"%s\": %.4e", first ? "" : ",\n", r.key, rate); first = false; }
    }
    if (jf) fprintf(jf, "\n},\n\"by_occupancy\": {\n");
    first = true;
    for (int waves : { 6, 4, 2, 1 }) {
        printf("-- %d wave(s) per SIMD\n", waves);
        for (int id : occupancy_rows) {
            const Row* r = nullptr;
            for (const Row& q : rows) if (q.id == id) r = &q;
            const int trips = 65536 / r->per_trip;
            const int blocks = cus * waves;
            float best = 1e30f;
            double cycles = 0;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipEventRecord(e0, s));
                dispatch(r->id, blocks, d, trips, s);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) { best = ms; cycles = longest_wave(blocks); }
            }
            const double rate = (double)blocks * 4 * (double)trips * r->per_trip * 64 / (best * 1e-3);
            const double cpi = cycles / ((double)waves * trips * r->per_trip);
            printf("%-78s %8.3f ms  %7.2f T lane-op/s  %5.2f cycles/instr  %5.3f GHz\n", r->name, best, rate / 1e12, cpi, cycles / (best * 1e6));
            if (jf) { fprintf(jf, "%s  \"%s@%d\": %.4e", first ? "" : ",\n", r->key, waves, rate); first = false; }
            if (cj) { fprintf(cj, "%s  \"%s@%d\": %.4f", cfirst ? "" : ",\n", r->key, waves, cpi); cfirst = false; }
        }
    }
    if (jf) { fprintf(jf, "\n}}\n"); fclose(jf); }
    if (cj) { fprintf(cj, "\n}}\n"); fclose(cj); }
    return 0;
}

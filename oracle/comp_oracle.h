/* oracle/comp_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned).
 *
 * Restates codec2's COMP type and complex primitives
 * [UPSTREAM-RECALLED: codec2 src/comp.h, src/comp_prim.h; pulled in un-pinned by
 *  /root/reference/build_codec2.sh:3-5, not present under /root/reference].
 * Evaluation order of every expression below is the upstream one, in float32,
 * with no fused multiply-add (the oracle is built with -ffp-contract=off, which
 * is what an x86-64 baseline build of codec2 produces).
 */
#ifndef PIRIP_ORACLE_COMP_H
#define PIRIP_ORACLE_COMP_H
#include <math.h>

typedef struct { float real; float imag; } COMP;

static inline COMP comp0(void) { COMP a = {0.0f, 0.0f}; return a; }
static inline COMP comp_exp_j(float phi) { COMP r; r.real = cosf(phi); r.imag = sinf(phi); return r; }
static inline COMP cconj(COMP a) { COMP r; r.real = a.real; r.imag = -a.imag; return r; }
static inline COMP cadd(COMP a, COMP b) { COMP r; r.real = a.real + b.real; r.imag = a.imag + b.imag; return r; }
static inline COMP fcmult(float a, COMP b) { COMP r; r.real = a * b.real; r.imag = a * b.imag; return r; }
static inline COMP cmult(COMP a, COMP b) {
    COMP r;
    r.real = a.real * b.real - a.imag * b.imag;
    r.imag = a.real * b.imag + a.imag * b.real;
    return r;
}
static inline float cabsolute(COMP a) { return sqrtf((a.real * a.real) + (a.imag * a.imag)); }
static inline COMP comp_normalize(COMP a) {
    COMP b; float av = cabsolute(a);
    b.real = a.real / av; b.imag = a.imag / av;
    return b;
}
#endif

/* detmath.h -- TEST INFRASTRUCTURE (oracle).  Not part of the product path.
 *
 * Deterministic float32 elementary functions for the CPU oracle.
 *
 * Why: the reference's device code (src/trace.metal) calls Metal's sin/cos/acos/atan/exp,
 * which are only defined to a few ulp (MSL fast-math default) and cannot be executed
 * here at all.  A bidirectional path tracer is chaotic in those last bits,
 * so the oracle PINS one concrete definition: classic single-precision
 * argument-reduction + minimax-polynomial algorithms (the published Cephes `sinf/cosf/
 * asinf/atanf/expf` schemes, S. Moshier), written with +,-,*,/ and sqrt only, in a fixed
 * evaluation order.  Every operation is an IEEE-754 binary32 operation (build with
 * -ffp-contract=off, no fast-math), so the HIP kernels -- which carry their own statement
 * of the same algorithms in clive2_amd/csrc/detmath.hpp -- reproduce the oracle bit for
 * bit, and GPU<->oracle parity can be checked exactly instead of statistically.
 * Accuracy (<= 2 ulp on the ranges the tracer uses) is pinned in tests/test_detmath.py.
 */
#ifndef ORACLE_DETMATH_H
#define ORACLE_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define DM_FOPI   1.27323954473516f      /* 4/pi */
#define DM_DP1    0.78515625f            /* pi/4 split in three parts */
#define DM_DP2    2.4187564849853515625e-4f
#define DM_DP3    3.77489497744594108e-8f
#define DM_PIO2   1.5707963267948966192f
#define DM_PIO4   0.7853981633974483096f
#define DM_PI     3.14159265358979323846f

static inline float dm_sin_poly(float x, float z) {
    float y = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x;
    return y + x;
}
static inline float dm_cos_poly(float z) {
    float y = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z
               + 4.166664568298827E-002f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

/* |x| must stay below ~8192 (the tracer only passes [0, 2*pi] and [0, pi/2]); NaN -> NaN. */
static inline float det_sinf(float xx) {
    float x = xx;
    int neg = 0;
    if (x < 0.0f) { neg = 1; x = -x; }
    if (!(x <= 8192.0f)) return x - x;              /* inf/NaN/huge -> NaN or 0: defined, unused */
    int j = (int)(DM_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { neg = !neg; j -= 4; }
    x = ((x - y * DM_DP1) - y * DM_DP2) - y * DM_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? dm_cos_poly(z) : dm_sin_poly(x, z);
    return neg ? -r : r;
}

static inline float det_cosf(float xx) {
    float x = xx;
    int neg = 0;
    if (x < 0.0f) x = -x;
    if (!(x <= 8192.0f)) return x - x;
    int j = (int)(DM_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { j -= 4; neg = !neg; }
    if (j > 1) neg = !neg;
    x = ((x - y * DM_DP1) - y * DM_DP2) - y * DM_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? dm_sin_poly(x, z) : dm_cos_poly(z);
    return neg ? -r : r;
}

static inline float det_asinf(float xx) {
    float x = xx, a, z;
    int neg = 0, flag = 0;
    if (x < 0.0f) { neg = 1; a = -x; } else a = x;
    if (!(a <= 1.0f)) return a - a + (a - a) / (a - a);   /* domain error / NaN -> NaN */
    if (a < 1.0e-4f) { z = a; return neg ? -z : z; }
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = 1; }
    else { x = a; z = x * x; }
    z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z
          + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * x + x;
    if (flag) { z = z + z; z = DM_PIO2 - z; }
    return neg ? -z : z;
}

static inline float det_acosf(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return (x - x) / (x - x);  /* NaN */
    if (x > 0.5f) return 2.0f * det_asinf(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return DM_PI - 2.0f * det_asinf(sqrtf(0.5f * (1.0f + x)));
    return DM_PIO2 - det_asinf(x);
}

static inline float det_atanf(float xx) {
    float x = xx, y;
    int neg = 0;
    if (x < 0.0f) { neg = 1; x = -x; }
    if (x > 2.414213562373095f) { y = DM_PIO2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = DM_PIO4; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z
              - 3.33329491539E-1f) * z * x + x);
    return neg ? -y : y;
}

/* exp for x <= ~88; returns exactly 0 below -87 (no denormal tail), NaN -> NaN. */
static inline float det_expf(float xx) {
    float x = xx;
    if (x != x) return x;
    if (x > 88.0f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    float fz = floorf(1.44269504088896341f * x + 0.5f);
    x = x - fz * 0.693359375f;
    x = x - fz * -2.12194440e-4f;
    int n = (int)fz;
    float z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x
           + 4.1665795894E-2f) * x + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    uint32_t bits = (uint32_t)(n + 127) << 23;      /* 2^n, n in [-126, 127] by the clamps above */
    float scale;
    memcpy(&scale, &bits, 4);
    return z * scale;
}

#endif

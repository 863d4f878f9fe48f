/* TEST INFRASTRUCTURE (CPU oracle).
 * cos(angle) / sin(angle) at src/ORBextractor.cc:113 take a FLOAT argument under `using namespace std;` (:67), so they
 * resolve to std::cos(float) / std::sin(float) = the C library's cosf / sinf, NOT (float)cos((double)angle).
 * glibc >= 2.28 sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c; the ARM
 * optimized-routines algorithm): argument in double, one reduction by n * pi/2 (n from a scaled float->int
 * conversion), a degree-7 / degree-8 polynomial in double, one rounding to float.  Restated for |x| < 120
 * (the descriptor's angle is in [0, 2 pi)); evaluated operation by operation (no FMA contraction).
 * Pinned: identical to this image's libm (glibc 2.35, x86-64, FMA-capable host) for EVERY float in [0, 6.2832]
 * (1 086 918 650 values; tests/test_oracle_sincosf.py runs the exhaustive comparison on a stride and the full sweep on
 * demand), which differs from the double-rounded value for 446 486 (cos) / 1 020 963 (sin) of them. */
#ifndef ORB_ORACLE_SINCOSF_H
#define ORB_ORACLE_SINCOSF_H
#include <stdint.h>
#include <string.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } rs_sincos_t;
static const rs_sincos_t rs_tab[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t rs_abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
static inline float rs_sinf_poly(double x, double x2, const rs_sincos_t *p, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2, s1 = p->s2 + x2 * p->s3, x7 = x3 * x2, s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    } else {
        double x4 = x2 * x2, c2 = p->c3 + x2 * p->c4, c1 = p->c0 + x2 * p->c1, x6 = x4 * x2, c = c1 + x4 * p->c2;
        return (float)(c + x6 * c2);
    }
}
static inline double rs_reduce_fast(double x, const rs_sincos_t *p, int *np) {
    double r = x * p->hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * p->hpi;
}
static inline float rs_sinf(float y) {
    double x = y;
    const rs_sincos_t *p = &rs_tab[0];
    if (rs_abstop12(y) < rs_abstop12(0x1.921FB6p-1f)) {
        double s = x * x;
        if (rs_abstop12(y) < rs_abstop12(0x1p-12f)) return y;
        return rs_sinf_poly(x, s, p, 0);
    }
    int n;
    x = rs_reduce_fast(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &rs_tab[1];
    return rs_sinf_poly(x * s, x * x, p, n);
}
static inline float rs_cosf(float y) {
    double x = y;
    const rs_sincos_t *p = &rs_tab[0];
    if (rs_abstop12(y) < rs_abstop12(0x1.921FB6p-1f)) {
        if (rs_abstop12(y) < rs_abstop12(0x1p-12f)) return 1.0f;
        return rs_sinf_poly(x, x * x, p, 1);
    }
    int n;
    x = rs_reduce_fast(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &rs_tab[1];
    return rs_sinf_poly(x * s, x * x, p, n ^ 1);
}
#endif

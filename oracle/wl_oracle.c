/* wl_oracle.c -- CPU oracle for the WaterLily `sim_step! -> mom_step!` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (waterlily_amd/) never does.
 *
 * A plain-C restatement (C11 + OpenMP) of the reference's `Array` CPU path:
 *   src/Flow.jl, src/Poisson.jl, src/MultiLevelPoisson.jl, src/util.jl (BC!, exitBC!, perBC!),
 *   src/Metrics.jl:84-100 (pressure_force).
 * The reference is Julia and no Julia runtime exists in the build container or on the GPU box, so the
 * reference itself cannot be executed; this oracle is pinned by restating every known-answer and
 * analytic test of test/maintests.jl that touches the path (tests/test_oracle_pins.py).
 *
 * Pinning status: PINNED by the reference's own analytic tests (no golden files exist upstream).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct wlo_grid {
    int D;       /* 2 or 3 */
    int n[3];    /* extents INCLUDING one ghost layer per side; n[2]=1 when D==2 */
    long s[3];   /* element strides: 1, n0, n0*n1 */
    long ncell;  /* n0*n1*n2 */
} wlo_grid;

#define WLO_MAXLEV 16

/* middle eigenvalue of a symmetric 3x3 matrix (closed form): lambda2 = eigvals(Hermitian(S^2+Omega^2))[2] */
static double wlo_sym3_mid_eig(double a00, double a01, double a02, double a11, double a12, double a22) {
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) / 3.0;
    if (p1 == 0.0) {
        double x = a00, y = a11, z = a22, t;
        if (x > y) { t = x; x = y; y = t; }
        if (y > z) { t = y; y = z; z = t; }
        if (x > y) { t = x; x = y; y = t; }
        return y;
    }
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    const double p = sqrt(p2 / 6.0);
    const double c00 = b00 / p, c11 = b11 / p, c22 = b22 / p, c01 = a01 / p, c02 = a02 / p, c12 = a12 / p;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    const double e1 = q + 2.0 * p * cos(phi), e3 = q + 2.0 * p * cos(phi + 2.0943951023931953);
    return 3.0 * q - e1 - e3;
}

#define T float
#define SUF(x) x##_f32
#define WLO_EPS FLT_EPSILON
#include "wlo_impl.h"
#undef T
#undef SUF
#undef WLO_EPS

#define T double
#define SUF(x) x##_f64
#define WLO_EPS DBL_EPSILON
#include "wlo_impl.h"
#undef T
#undef SUF
#undef WLO_EPS

int wlo_abi_version(void) { return 1; }

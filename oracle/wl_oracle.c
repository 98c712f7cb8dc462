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

/* 0 (default): reductions exactly where the reference takes them -- LinearAlgebra.dot and maximum over the
 * WHOLE arrays, ghosts included (src/Poisson.jl:126-146, src/Flow.jl:174).  1: over inside() only, which is
 * what the HIP path computes.  The two differ only through ghost entries that hold stale scratch (sigma is
 * also conv_diff!'s Phi) multiplied by periodic copies; tests use the switch to show that this is the ONLY
 * difference in periodic runs (deliberate deviations). */
static int wlo_interior_reductions = 0;
void wlo_set_interior_reductions(int on) { wlo_interior_reductions = on; }

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

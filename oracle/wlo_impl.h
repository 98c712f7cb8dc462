/* wlo_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * Included twice by wl_oracle.c with  T = float / double  and  SUF(x) = x##_f32 / x##_f64.
 * Every function restates one `@loop` site (or one host function) of the reference
 * WaterLily `Array` path; the reference file:line it follows is cited on each function
 * (paths relative to /root/reference).  Loop structure is kept un-fused, one parallel loop
 * per reference `@loop`, so that the same code doubles as the multithreaded CPU baseline.
 *
 * Conventions:
 *   - arrays are dense column-major with ONE ghost layer per side, extents n[d] include ghosts;
 *     0-based index here = reference 1-based index - 1;
 *   - vector fields are SoA: component c starts at c*ncell; mu1[I,i,j] is component i + D*j;
 *   - Float64 literals in the Julia source promote Float32 intermediates to Float64; this is
 *     mimicked (compute in double, round on store) wherever the reference does it;
 *   - reductions (sum / dot / maximum) accumulate in double and round to T once.  The reference
 *     uses BLAS/pairwise accumulation in T whose order is unspecified (and @fastmath allows
 *     reassociation), so no summation order is "the" reference; double accumulation differs from
 *     any T-order by O(eps_T) relative.
 */

/* ---------------------------------------------------------------- small helpers */

/* Flow.jl:25-34 -- custom 3-argument median */
static inline T SUF(median)(T a, T b, T c) {
    if (a > b) {
        if (b >= c) return b;
        if (a > c) return c;
    } else {
        if (b <= c) return b;
        if (a < c) return c;
    }
    return a;
}
/* Flow.jl:4 -- QUICK with median limiter, all in T (integer literals do not promote) */
static inline T SUF(quick)(T u, T c, T d) {
    T a1 = (((T)5 * c + (T)2 * d) - u) / (T)6;
    T a2 = SUF(median)((T)10 * c - (T)9 * u, c, d);
    return SUF(median)(a1, c, a2);
}
/* Flow.jl:5 -- vanLeer (not used by default; kept for the known-answer tests) */
static inline T SUF(vanleer)(T u, T c, T d) {
    T mn = u < d ? u : d, mx = u > d ? u : d;
    return (c <= mn || c >= mx) ? c : c + (d - c) * (c - u) / (d - u);
}
/* Flow.jl:3 -- phi = (f[I]+f[I-d])*0.5 : T add, then *0.5 in Float64 */
static inline double SUF(phi)(const T *f, long I, long s) {
    return (double)(T)(f[I] + f[I - s]) * 0.5;
}
/* Flow.jl:6 */
static inline double SUF(phiu)(const T *f, long I, long s, double u) {
    return u > 0 ? u * (double)SUF(quick)(f[I - 2 * s], f[I - s], f[I])
                 : u * (double)SUF(quick)(f[I + s], f[I], f[I - s]);
}
/* Flow.jl:7 -- periodic variant: the far-upwind point is given explicitly */
static inline double SUF(phiuP)(const T *f, long Ip, long I, long s, double u) {
    return u > 0 ? u * (double)SUF(quick)(f[Ip], f[I - s], f[I])
                 : u * (double)SUF(quick)(f[I + s], f[I], f[I - s]);
}
/* Flow.jl:8 */
static inline double SUF(phiuL)(const T *f, long I, long s, double u) {
    return u > 0 ? u * SUF(phi)(f, I, s) : u * (double)SUF(quick)(f[I + s], f[I], f[I - s]);
}
/* Flow.jl:9 */
static inline double SUF(phiuR)(const T *f, long I, long s, double u) {
    return u < 0 ? u * SUF(phi)(f, I, s) : u * (double)SUF(quick)(f[I - 2 * s], f[I - s], f[I]);
}

typedef struct { int lo[3], hi[3]; } SUF(rng);

/* util.jl:47 inside(a): 2:N-1 (1-based) in every dimension */
static inline SUF(rng) SUF(inside)(const wlo_grid *g) {
    SUF(rng) r;
    for (int d = 0; d < 3; ++d) {
        if (d < g->D) { r.lo[d] = 1; r.hi[d] = g->n[d] - 2; } else { r.lo[d] = 0; r.hi[d] = 0; }
    }
    return r;
}
static inline SUF(rng) SUF(whole)(const wlo_grid *g) {
    SUF(rng) r;
    for (int d = 0; d < 3; ++d) { r.lo[d] = 0; r.hi[d] = g->n[d] - 1; }
    return r;
}
/* util.jl:180-182 slice(dims,i,j,low): index i (1-based) in direction j, low:dims[k] elsewhere */
static inline SUF(rng) SUF(slice)(const wlo_grid *g, int i1, int j, int low1) {
    SUF(rng) r;
    for (int d = 0; d < 3; ++d) {
        if (d >= g->D) { r.lo[d] = 0; r.hi[d] = 0; }
        else if (d == j) { r.lo[d] = i1 - 1; r.hi[d] = i1 - 1; }
        else { r.lo[d] = low1 - 1; r.hi[d] = g->n[d] - 1; }
    }
    return r;
}
/* util.jl:55-57 inside_u(dims,j): 3:N-1 in j, 2:N (top ghost included) elsewhere */
static inline SUF(rng) SUF(inside_u)(const wlo_grid *g, int j) {
    SUF(rng) r;
    for (int d = 0; d < 3; ++d) {
        if (d >= g->D) { r.lo[d] = 0; r.hi[d] = 0; }
        else if (d == j) { r.lo[d] = 2; r.hi[d] = g->n[d] - 2; }
        else { r.lo[d] = 1; r.hi[d] = g->n[d] - 1; }
    }
    return r;
}
static inline long SUF(rcount)(const SUF(rng) *r) {
    long c = 1;
    for (int d = 0; d < 3; ++d) c *= (long)(r->hi[d] - r->lo[d] + 1);
    return c;
}

#define WLO_PAR _Pragma("omp parallel for collapse(2) schedule(static) if (wlo__big)")
#define WLO_LOOP(R, ...)                                                               \
    do {                                                                               \
        const SUF(rng) R__ = (R);                                                      \
        const int wlo__big = SUF(rcount)(&R__) > 16384;                                \
        (void)wlo__big;                                                                \
        WLO_PAR                                                                        \
        for (int k = R__.lo[2]; k <= R__.hi[2]; ++k)                                   \
            for (int j = R__.lo[1]; j <= R__.hi[1]; ++j)                               \
                for (int i = R__.lo[0]; i <= R__.hi[0]; ++i) {                         \
                    const long I = (long)i + g->s[1] * (long)j + g->s[2] * (long)k;    \
                    (void)I;                                                           \
                    __VA_ARGS__                                                        \
                }                                                                      \
    } while (0)

/* ---------------------------------------------------------------- reductions */

static double SUF(sum_all)(const T *a, long n) {
    double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (n > 16384)
    for (long q = 0; q < n; ++q) s += (double)a[q];
    return s;
}
/* LinearAlgebra.dot over the WHOLE arrays incl. ghosts (Poisson.jl:126,131,137,146) */
T SUF(wlo_dot)(const T *a, const T *b, long n) {
    double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (n > 16384)
    for (long q = 0; q < n; ++q) s += (double)a[q] * (double)b[q];
    return (T)s;
}
/* dot used by pcg!/L2: LinearAlgebra.dot over the whole arrays, ghost entries included */
static T SUF(dot_g)(const T *a, const T *b, const wlo_grid *g) { return SUF(wlo_dot)(a, b, g->ncell); }
static T SUF(max_all)(const T *a, long n) {
    T m = a[0];
#pragma omp parallel for reduction(max : m) schedule(static) if (n > 16384)
    for (long q = 0; q < n; ++q) m = a[q] > m ? a[q] : m;
    return m;
}
/* util.jl:68  L2(a) = sum(abs2, a[inside]) */
double SUF(wlo_L2_inside)(const T *a, const wlo_grid *g) {
    double s = 0;
    SUF(rng) R = SUF(inside)(g);
    for (int k = R.lo[2]; k <= R.hi[2]; ++k)
        for (int j = R.lo[1]; j <= R.hi[1]; ++j)
            for (int i = R.lo[0]; i <= R.hi[0]; ++i) {
                double v = (double)a[i + g->s[1] * j + g->s[2] * k];
                s += v * v;
            }
    return s;
}

/* ---------------------------------------------------------------- boundary conditions */

/* util.jl:192-210  BC!(a,A,saveexit,perdir) on a vector field */
void SUF(wlo_bc_vec)(T *a, const wlo_grid *g, const double *A, int saveexit, int permask) {
    const int D = g->D;
    for (int c = 0; c < D; ++c)
        for (int j = 0; j < D; ++j) {
            T *ac = a + (long)c * g->ncell;
            const long sj = g->s[j];
            const int N = g->n[j];
            if ((permask >> j) & 1) {
                /* util.jl:196-197 */
                WLO_LOOP(SUF(slice)(g, 1, j, 1), { ac[I] = ac[I + (long)(N - 2) * sj]; });
                WLO_LOOP(SUF(slice)(g, N, j, 1), { ac[I] = ac[I - (long)(N - 2) * sj]; });
            } else if (c == j) {
                /* util.jl:200-203 Dirichlet on planes 1,2 and N (N skipped for saveexit && i==1) */
                const T Ac = (T)A[c];
                for (int s = 1; s <= 2; ++s) WLO_LOOP(SUF(slice)(g, s, j, 1), { ac[I] = Ac; });
                if (!saveexit || c > 0) WLO_LOOP(SUF(slice)(g, N, j, 1), { ac[I] = Ac; });
            } else {
                /* util.jl:205-206 zero Neumann on tangential components */
                WLO_LOOP(SUF(slice)(g, 1, j, 1), { ac[I] = ac[I + sj]; });
                WLO_LOOP(SUF(slice)(g, N, j, 1), { ac[I] = ac[I - sj]; });
            }
        }
}

/* util.jl:227-231  perBC!(a,perdir) on a scalar field */
void SUF(wlo_bc_per)(T *a, const wlo_grid *g, int permask) {
    for (int j = 0; j < g->D; ++j)
        if ((permask >> j) & 1) {
            const long sj = g->s[j];
            const int N = g->n[j];
            WLO_LOOP(SUF(slice)(g, 1, j, 1), { a[I] = a[I + (long)(N - 2) * sj]; });
            WLO_LOOP(SUF(slice)(g, N, j, 1), { a[I] = a[I - (long)(N - 2) * sj]; });
        }
}

/* util.jl:216-222  exitBC!(u,u0,U,dt): 1-D convective exit + mass-flux correction */
void SUF(wlo_exit_bc)(T *u, const T *u0, const wlo_grid *g, const double *U, double dt_) {
    SUF(rng) R;
    for (int d = 0; d < 3; ++d) {
        if (d >= g->D) { R.lo[d] = 0; R.hi[d] = 0; }
        else if (d == 0) { R.lo[d] = g->n[0] - 1; R.hi[d] = g->n[0] - 1; }
        else { R.lo[d] = 1; R.hi[d] = g->n[d] - 2; }
    }
    const T U1 = (T)U[0], dt = (T)dt_;
    const T Udt = U1 * dt;
    WLO_LOOP(R, { u[I] = u0[I] - Udt * (u0[I] - u0[I - 1]); });
    double s = 0;
    for (int k = R.lo[2]; k <= R.hi[2]; ++k)
        for (int j = R.lo[1]; j <= R.hi[1]; ++j)
            s += (double)u[(long)R.lo[0] + g->s[1] * j + g->s[2] * k];
    const T corr = (T)s / (T)SUF(rcount)(&R) - U1;
    WLO_LOOP(R, { u[I] -= corr; });
}

/* ---------------------------------------------------------------- Flow.jl */

/* Flow.jl:36-60  conv_diff!(r,u,Phi;nu,perdir) -- faithful scatter form with the Phi scratch */
void SUF(wlo_conv_diff)(T *r, const T *u, T *Phi, const wlo_grid *g, double nu_, int permask) {
    const int D = g->D;
    const long nc = g->ncell;
    const T nu = (T)nu_;
    for (long q = 0; q < nc * D; ++q) r[q] = 0; /* Flow.jl:37 */
    for (int c = 0; c < D; ++c)
        for (int j = 0; j < D; ++j) {
            const T *ui = u + (long)c * nc;
            const T *uj = u + (long)j * nc;
            T *ri = r + (long)c * nc;
            const long sj = g->s[j], si = g->s[c];
            const int Nj = g->n[j];
            const int per = (permask >> j) & 1;
            /* lower boundary, plane I_j = 2 (1-based)  Flow.jl:54 / :58-59 */
            if (!per) {
                WLO_LOOP(SUF(slice)(g, 2, j, 2), {
                    const double uf = SUF(phi)(uj, I, si);
                    const T nud = nu * (T)(ui[I] - ui[I - sj]);
                    const double F = SUF(phiuL)(ui, I, sj, uf) - (double)nud;
                    ri[I] = (T)((double)ri[I] + F);
                });
            } else {
                WLO_LOOP(SUF(slice)(g, 2, j, 2), {
                    const double uf = SUF(phi)(uj, I, si);
                    const T nud = nu * (T)(ui[I] - ui[I - sj]);
                    const long Ip = I + (long)(Nj - 4) * sj; /* j-index N_j-2 (1-based) */
                    Phi[I] = (T)(SUF(phiuP)(ui, Ip, I, sj, uf) - (double)nud);
                    ri[I] += Phi[I];
                });
            }
            /* interior faces  Flow.jl:45-46 */
            WLO_LOOP(SUF(inside_u)(g, j), {
                const double uf = SUF(phi)(uj, I, si);
                const T nud = nu * (T)(ui[I] - ui[I - sj]);
                Phi[I] = (T)(SUF(phiu)(ui, I, sj, uf) - (double)nud);
                ri[I] += Phi[I];
            });
            /* Flow.jl:47 (separate loop: scatter to the cell below) */
            WLO_LOOP(SUF(inside_u)(g, j), { ri[I - sj] -= Phi[I]; });
            /* upper boundary, plane I_j = N_j  Flow.jl:55 / :60 */
            if (!per) {
                WLO_LOOP(SUF(slice)(g, Nj, j, 2), {
                    const double uf = SUF(phi)(uj, I, si);
                    const T nud = nu * (T)(ui[I] - ui[I - sj]);
                    const double F = -SUF(phiuR)(ui, I, sj, uf) + (double)nud;
                    ri[I - sj] = (T)((double)ri[I - sj] + F);
                });
            } else {
                WLO_LOOP(SUF(slice)(g, Nj, j, 2), { ri[I - sj] -= Phi[I - (long)(Nj - 2) * sj]; });
            }
        }
}

/* Flow.jl:68-70  accelerate!: r[..,i] .+= g_i over every element of component i.
 * The host evaluates g(i,t) (+ dU/dt) and passes the D numbers. */
void SUF(wlo_accelerate)(T *r, const wlo_grid *g, const double *acc) {
    for (int c = 0; c < g->D; ++c) {
        T *rc = r + (long)c * g->ncell;
        const double a = acc[c];
        for (long q = 0; q < g->ncell; ++q) rc[q] = (T)((double)rc[q] + a);
    }
}

/* Flow.jl:131-135  BDIM!  (mu_ddn: Flow.jl:18-24, `0.5s` promotes to Float64) */
void SUF(wlo_bdim)(T *u, const T *u0, T *f, const T *V, const T *mu0, const T *mu1,
                   const wlo_grid *g, double dt_) {
    const int D = g->D;
    const long nc = g->ncell;
    const T dt = (T)dt_;
#pragma omp parallel for schedule(static) if (nc > 16384)
    for (long q = 0; q < nc * D; ++q) f[q] = (u0[q] + dt * f[q]) - V[q]; /* Flow.jl:133 */
    for (int c = 0; c < D; ++c) {
        T *uc = u + (long)c * nc;
        const T *fc = f + (long)c * nc, *Vc = V + (long)c * nc, *m0 = mu0 + (long)c * nc;
        WLO_LOOP(SUF(inside)(g), {
            T s = 0;
            for (int j = 0; j < D; ++j) {
                const T *m1 = mu1 + (long)(c + D * j) * nc;
                s += m1[I] * (fc[I + g->s[j]] - fc[I - g->s[j]]);
            }
            const double tmp = (0.5 * (double)s + (double)Vc[I]) + (double)(T)(m0[I] * fc[I]);
            uc[I] = (T)((double)uc[I] + tmp);
        });
    }
}

/* Flow.jl:170  scale_u!: u *= scale over inside_u(size(p)) */
void SUF(wlo_scale_u)(T *u, const wlo_grid *g, double scale) {
    for (int c = 0; c < g->D; ++c) {
        T *uc = u + (long)c * g->ncell;
        WLO_LOOP(SUF(inside)(g), { uc[I] = (T)((double)uc[I] * scale); });
    }
}

/* Flow.jl:11-17  div(I,u) summed in T, forward differences (Flow.jl:2) */
void SUF(wlo_div)(T *z, const T *u, const wlo_grid *g) {
    const int D = g->D;
    const long nc = g->ncell;
    WLO_LOOP(SUF(inside)(g), {
        T s = 0;
        for (int d = 0; d < D; ++d) s += u[I + g->s[d] + d * nc] - u[I + d * nc];
        z[I] = s;
    });
}

/* Flow.jl:172-182  CFL: sigma = flux_out (Float64 via max(0.,.)), dt = min(10, 1/(max(sigma)+5nu)) */
double SUF(wlo_cfl)(T *sigma, const T *u, const wlo_grid *g, double nu_) {
    const int D = g->D;
    const long nc = g->ncell;
    WLO_LOOP(SUF(inside)(g), {
        double s = 0;
        for (int d = 0; d < D; ++d) {
            const double a = (double)u[I + g->s[d] + d * nc], b = -(double)u[I + d * nc];
            s += (a > 0 ? a : 0.0) + (b > 0 ? b : 0.0);
        }
        sigma[I] = (T)s;
    });
    T m = SUF(max_all)(sigma, nc); /* maximum over the WHOLE array (ghosts keep stale Phi) */
    const T d = (T)1 / (m + (T)5 * (T)nu_);
    return (double)(d < (T)10 ? d : (T)10);
}

/* ---------------------------------------------------------------- Poisson.jl */

typedef struct SUF(wlo_poisson) {
    wlo_grid g;
    T *L, *D, *iD, *x, *eps, *r, *z;
    int permask;
    int owns_xLz;
} SUF(wlo_poisson);

/* Poisson.jl:42-54 set_diag! */
void SUF(wlo_set_diag)(T *Dg, T *iD, const T *L, const wlo_grid *g) {
    const int D = g->D;
    const long nc = g->ncell;
    const T eps2 = (T)2 * WLO_EPS;
    WLO_LOOP(SUF(inside)(g), {
        T s = 0;
        for (int d = 0; d < D; ++d) s -= (L[I + d * nc] + L[I + g->s[d] + d * nc]);
        Dg[I] = s;
    });
    WLO_LOOP(SUF(inside)(g), { iD[I] = (Dg[I] * Dg[I] < eps2) ? (T)0 : (T)1 / Dg[I]; });
}

/* Poisson.jl:69-75 mult(I,L,D,x) */
static inline T SUF(mult1)(const T *L, const T *Dg, const T *x, const wlo_grid *g, long I) {
    T s = x[I] * Dg[I];
    for (int d = 0; d < g->D; ++d) {
        const long sd = g->s[d];
        const T *Ld = L + (long)d * g->ncell;
        s += x[I - sd] * Ld[I] + x[I + sd] * Ld[I + sd];
    }
    return s;
}

/* Poisson.jl:62-68 mult!(p,x): z = A x, zero in the ghosts */
void SUF(wlo_mult)(SUF(wlo_poisson) *p, T *x) {
    const wlo_grid *g = &p->g;
    SUF(wlo_bc_per)(x, g, p->permask);
    for (long q = 0; q < g->ncell; ++q) p->z[q] = 0;
    WLO_LOOP(SUF(inside)(g), { p->z[I] = SUF(mult1)(p->L, p->D, x, g, I); });
}

/* Poisson.jl:91-97 residual! */
void SUF(wlo_residual)(SUF(wlo_poisson) *p) {
    const wlo_grid *g = &p->g;
    SUF(wlo_bc_per)(p->x, g, p->permask);
    WLO_LOOP(SUF(inside)(g), {
        p->r[I] = (p->iD[I] == 0) ? (T)0 : p->z[I] - SUF(mult1)(p->L, p->D, p->x, g, I);
    });
    SUF(rng) R = SUF(inside)(g);
    const T s = (T)SUF(sum_all)(p->r, g->ncell) / (T)SUF(rcount)(&R);
    if ((s < 0 ? -s : s) <= (T)2 * WLO_EPS) return;
    WLO_LOOP(SUF(inside)(g), { p->r[I] = p->r[I] - s; });
}

/* Poisson.jl:99-103 increment! */
void SUF(wlo_increment)(SUF(wlo_poisson) *p) {
    const wlo_grid *g = &p->g;
    SUF(wlo_bc_per)(p->eps, g, p->permask);
    WLO_LOOP(SUF(inside)(g), {
        p->r[I] = p->r[I] - SUF(mult1)(p->L, p->D, p->eps, g, I);
        p->x[I] = p->x[I] + p->eps[I];
    });
}

/* Poisson.jl:110-113 Jacobi! */
void SUF(wlo_jacobi)(SUF(wlo_poisson) *p, int it) {
    const wlo_grid *g = &p->g;
    for (int n = 0; n < it; ++n) {
        WLO_LOOP(SUF(inside)(g), { p->eps[I] = p->r[I] * p->iD[I]; });
        SUF(wlo_increment)(p);
    }
}

/* Poisson.jl:123-143 pcg!  -- returns the number of (x,r) updates performed (diagnostic only) */
int SUF(wlo_pcg)(SUF(wlo_poisson) *p, int it) {
    const wlo_grid *g = &p->g;
    T *x = p->x, *r = p->r, *e = p->eps, *z = p->z;
    int nupd = 0;
    WLO_LOOP(SUF(inside)(g), { z[I] = e[I] = r[I] * p->iD[I]; });
    T rho = SUF(dot_g)(r, z, g);
    if ((rho < 0 ? -rho : rho) < (T)10 * WLO_EPS) return nupd;
    for (int i = 1; i <= it; ++i) {
        SUF(wlo_bc_per)(e, g, p->permask);
        WLO_LOOP(SUF(inside)(g), { z[I] = SUF(mult1)(p->L, p->D, e, g, I); });
        const T alpha = rho / SUF(dot_g)(z, e, g);
        const double aa = (double)(alpha < 0 ? -alpha : alpha);
        if (aa < 1e-2 || aa > 1e2) return nupd; /* NaN compares false and falls through, as in Julia */
        WLO_LOOP(SUF(inside)(g), {
            x[I] += alpha * e[I];
            r[I] -= alpha * z[I];
        });
        ++nupd;
        if (i == it) return nupd;
        WLO_LOOP(SUF(inside)(g), { z[I] = r[I] * p->iD[I]; });
        const T rho2 = SUF(dot_g)(r, z, g);
        if ((rho2 < 0 ? -rho2 : rho2) < (T)10 * WLO_EPS) return nupd;
        const T beta = rho2 / rho;
        WLO_LOOP(SUF(inside)(g), { e[I] = beta * e[I] + z[I]; });
        rho = rho2;
    }
    return nupd;
}

/* Poisson.jl:146 */
T SUF(wlo_L2)(const SUF(wlo_poisson) *p) { return SUF(dot_g)(p->r, p->r, &p->g); }

/* Poisson.jl:162-172 solver!(::Poisson) */
int SUF(wlo_solver)(SUF(wlo_poisson) *p, double tol, int itmx) {
    SUF(wlo_residual)(p);
    T r2 = SUF(wlo_L2)(p);
    int n = 0;
    while (n < itmx) {
        SUF(wlo_pcg)(p, 6);
        r2 = SUF(wlo_L2)(p);
        ++n;
        if ((double)r2 < tol) break;
    }
    SUF(wlo_bc_per)(p->x, &p->g, p->permask);
    return n;
}

/* ---------------------------------------------------------------- MultiLevelPoisson.jl */

/* MultiLevelPoisson.jl:10-16,26-32 restrictL!: coarse face = 0.5*sum of the 2^(D-1) fine faces */
void SUF(wlo_restrictL)(T *a, const wlo_grid *ga, const T *b, const wlo_grid *gb, int permask) {
    const int D = ga->D;
    for (int c = 0; c < D; ++c) {
        T *ac = a + (long)c * ga->ncell;
        const T *bc = b + (long)c * gb->ncell;
        const wlo_grid *g = ga;
        WLO_LOOP(SUF(inside)(ga), {
            const int cc[3] = {i, j, k};
            int lo[3], hi[3];
            for (int d = 0; d < 3; ++d) {
                if (d >= D) { lo[d] = hi[d] = 0; }
                else { lo[d] = 2 * cc[d] - 1; hi[d] = (d == c) ? lo[d] : 2 * cc[d]; }
            }
            T s = 0;
            for (int kk = lo[2]; kk <= hi[2]; ++kk)
                for (int jj = lo[1]; jj <= hi[1]; ++jj)
                    for (int ii = lo[0]; ii <= hi[0]; ++ii)
                        s += bc[(long)ii + gb->s[1] * jj + gb->s[2] * kk];
            ac[I] = (T)(0.5 * (double)s);
        });
    }
    const double zero[3] = {0, 0, 0};
    SUF(wlo_bc_vec)(a, ga, zero, 0, permask);
}

/* MultiLevelPoisson.jl:3-9,33 restrict!: coarse = SUM of the 2^D children */
void SUF(wlo_restrict)(T *a, const wlo_grid *ga, const T *b, const wlo_grid *gb) {
    const int D = ga->D;
    const wlo_grid *g = ga;
    WLO_LOOP(SUF(inside)(ga), {
        const int cc[3] = {i, j, k};
        int lo[3], hi[3];
        for (int d = 0; d < 3; ++d) {
            if (d >= D) { lo[d] = hi[d] = 0; }
            else { lo[d] = 2 * cc[d] - 1; hi[d] = 2 * cc[d]; }
        }
        T s = 0;
        for (int kk = lo[2]; kk <= hi[2]; ++kk)
            for (int jj = lo[1]; jj <= hi[1]; ++jj)
                for (int ii = lo[0]; ii <= hi[0]; ++ii)
                    s += b[(long)ii + gb->s[1] * jj + gb->s[2] * kk];
        a[I] = s;
    });
}

/* MultiLevelPoisson.jl:2,34 prolongate!: fine[I] = coarse[down(I)] */
void SUF(wlo_prolongate)(T *a, const wlo_grid *ga, const T *b, const wlo_grid *gb) {
    const wlo_grid *g = ga;
    WLO_LOOP(SUF(inside)(ga), {
        const long J = (long)((i + 1) / 2) + gb->s[1] * (long)(ga->D > 1 ? (j + 1) / 2 : 0) +
                       gb->s[2] * (long)(ga->D > 2 ? (k + 1) / 2 : 0);
        a[I] = b[J];
    });
}

typedef struct SUF(wlo_mg) {
    int nlevels;
    SUF(wlo_poisson) lev[WLO_MAXLEV];
    int permask;
} SUF(wlo_mg);

static T *SUF(zalloc)(long n) { return (T *)calloc((size_t)n, sizeof(T)); }

static void SUF(poisson_init)(SUF(wlo_poisson) *p, const wlo_grid *g, T *x, T *L, T *z, int permask, int owns) {
    p->g = *g;
    p->x = x; p->L = L; p->z = z;
    p->permask = permask;
    p->owns_xLz = owns;
    /* Poisson.jl:33-35 */
    p->r = SUF(zalloc)(g->ncell);
    p->eps = SUF(zalloc)(g->ncell);
    p->D = SUF(zalloc)(g->ncell);
    p->iD = SUF(zalloc)(g->ncell);
    SUF(wlo_set_diag)(p->D, p->iD, p->L, g);
}

/* MultiLevelPoisson.jl:36-37 */
static int SUF(divisible)(const wlo_grid *g) {
    for (int d = 0; d < g->D; ++d)
        if (!(g->n[d] % 2 == 0 && g->n[d] > 4)) return 0;
    return 1;
}

/* MultiLevelPoisson.jl:44-60 (+ restrictML :18-25).  Returns NULL when fewer than 3 levels
 * ("MultiLevelPoisson requires size=a2^n, where n>2"). */
SUF(wlo_mg) *SUF(wlo_mg_create)(T *x, T *L, T *z, const wlo_grid *g, int permask, int maxlevels) {
    SUF(wlo_mg) *ml = (SUF(wlo_mg) *)calloc(1, sizeof(SUF(wlo_mg)));
    ml->permask = permask;
    SUF(poisson_init)(&ml->lev[0], g, x, L, z, permask, 0);
    ml->nlevels = 1;
    while (SUF(divisible)(&ml->lev[ml->nlevels - 1].g) && ml->nlevels <= maxlevels && ml->nlevels < WLO_MAXLEV) {
        const SUF(wlo_poisson) *b = &ml->lev[ml->nlevels - 1];
        wlo_grid ga;
        ga.D = b->g.D;
        for (int d = 0; d < 3; ++d) ga.n[d] = d < ga.D ? 1 + b->g.n[d] / 2 : 1;
        ga.s[0] = 1; ga.s[1] = ga.n[0]; ga.s[2] = (long)ga.n[0] * ga.n[1];
        ga.ncell = (long)ga.n[0] * ga.n[1] * ga.n[2];
        T *aL = SUF(zalloc)(ga.ncell * ga.D);
        T *ax = SUF(zalloc)(ga.ncell);
        T *az = SUF(zalloc)(ga.ncell);
        SUF(wlo_restrictL)(aL, &ga, b->L, &b->g, permask);
        SUF(poisson_init)(&ml->lev[ml->nlevels], &ga, ax, aL, az, permask, 1);
        ml->nlevels++;
    }
    if (ml->nlevels <= 2) { /* caller reports the reference's assertion text */
        ml->nlevels = -ml->nlevels;
    }
    return ml;
}

void SUF(wlo_mg_destroy)(SUF(wlo_mg) *ml) {
    int nl = ml->nlevels < 0 ? -ml->nlevels : ml->nlevels;
    for (int l = 0; l < nl; ++l) {
        SUF(wlo_poisson) *p = &ml->lev[l];
        free(p->r); free(p->eps); free(p->D); free(p->iD);
        if (p->owns_xLz) { free(p->x); free(p->L); free(p->z); }
    }
    free(ml);
}

/* MultiLevelPoisson.jl:62-68 update! */
void SUF(wlo_mg_update)(SUF(wlo_mg) *ml) {
    SUF(wlo_set_diag)(ml->lev[0].D, ml->lev[0].iD, ml->lev[0].L, &ml->lev[0].g);
    for (int l = 1; l < ml->nlevels; ++l) {
        SUF(wlo_restrictL)(ml->lev[l].L, &ml->lev[l].g, ml->lev[l - 1].L, &ml->lev[l - 1].g, ml->lev[l - 1].permask);
        SUF(wlo_set_diag)(ml->lev[l].D, ml->lev[l].iD, ml->lev[l].L, &ml->lev[l].g);
    }
}

/* MultiLevelPoisson.jl:70-82 Vcycle! (l is 0-based here) */
void SUF(wlo_vcycle)(SUF(wlo_mg) *ml, int l) {
    SUF(wlo_poisson) *fine = &ml->lev[l], *coarse = &ml->lev[l + 1];
    SUF(wlo_jacobi)(fine, 1);
    SUF(wlo_restrict)(coarse->r, &coarse->g, fine->r, &fine->g);
    for (long q = 0; q < coarse->g.ncell; ++q) coarse->x[q] = 0;
    if (l + 2 < ml->nlevels) SUF(wlo_vcycle)(ml, l + 1);
    SUF(wlo_pcg)(coarse, 6);
    SUF(wlo_prolongate)(fine->eps, &fine->g, coarse->x, &coarse->g);
    SUF(wlo_increment)(fine);
}

/* MultiLevelPoisson.jl:87-99 solver!(::MultiLevelPoisson) */
int SUF(wlo_mg_solver)(SUF(wlo_mg) *ml, double tol, int itmx) {
    SUF(wlo_poisson) *p = &ml->lev[0];
    SUF(wlo_residual)(p);
    T r2 = SUF(wlo_L2)(p);
    int n = 0;
    while (n < itmx) {
        SUF(wlo_vcycle)(ml, 0);
        SUF(wlo_pcg)(p, 6);
        r2 = SUF(wlo_L2)(p);
        ++n;
        if ((double)r2 < tol) break;
    }
    SUF(wlo_bc_per)(p->x, &p->g, p->permask);
    return n;
}

int SUF(wlo_mg_nlevels)(const SUF(wlo_mg) *ml) { return ml->nlevels; }
/* which: 0 L, 1 D, 2 iD, 3 x, 4 eps, 5 r, 6 z */
T *SUF(wlo_mg_array)(SUF(wlo_mg) *ml, int l, int which, int *n3) {
    SUF(wlo_poisson) *p = &ml->lev[l];
    for (int d = 0; d < 3; ++d) n3[d] = p->g.n[d];
    switch (which) {
        case 0: return p->L;
        case 1: return p->D;
        case 2: return p->iD;
        case 3: return p->x;
        case 4: return p->eps;
        case 5: return p->r;
        default: return p->z;
    }
}
SUF(wlo_poisson) *SUF(wlo_mg_level)(SUF(wlo_mg) *ml, int l) { return &ml->lev[l]; }

/* stand-alone single-level Poisson (Poisson.jl:31-37), used by the reference's Poisson tests */
SUF(wlo_poisson) *SUF(wlo_poisson_create)(T *x, T *L, T *z, const wlo_grid *g, int permask) {
    SUF(wlo_poisson) *p = (SUF(wlo_poisson) *)calloc(1, sizeof(SUF(wlo_poisson)));
    SUF(poisson_init)(p, g, x, L, z, permask, 0);
    return p;
}
void SUF(wlo_poisson_destroy)(SUF(wlo_poisson) *p) {
    free(p->r); free(p->eps); free(p->D); free(p->iD);
    free(p);
}
T *SUF(wlo_poisson_array)(SUF(wlo_poisson) *p, int which) {
    switch (which) {
        case 0: return p->L;
        case 1: return p->D;
        case 2: return p->iD;
        case 3: return p->x;
        case 4: return p->eps;
        case 5: return p->r;
        default: return p->z;
    }
}

/* ---------------------------------------------------------------- Flow.jl: project!, mom_step! */

typedef struct SUF(wlo_flow) {
    wlo_grid g;
    T *u, *u0, *f, *p, *sigma, *V, *mu0, *mu1;
    double nu;
    int exitBC, permask;
} SUF(wlo_flow);

/* Flow.jl:137-145 project!(a,b,w).  dt = w*dt: T for w==1 (Int), Float64 for w==0.5 */
int SUF(wlo_project)(SUF(wlo_flow) *a, SUF(wlo_mg) *b, double dt_, double w, double tol, int itmx) {
    const wlo_grid *g = &a->g;
    const long nc = g->ncell;
    SUF(wlo_poisson) *p = &b->lev[0];
    const T dtT = (T)dt_;
    const double dtd = w * (double)dtT;
    SUF(wlo_div)(p->z, a->u, g);
    if (w == 1.0) { for (long q = 0; q < nc; ++q) p->x[q] = p->x[q] * dtT; }
    else { for (long q = 0; q < nc; ++q) p->x[q] = (T)((double)p->x[q] * dtd); }
    const int n = SUF(wlo_mg_solver)(b, tol, itmx);
    for (int c = 0; c < g->D; ++c) {
        T *uc = a->u + (long)c * nc;
        const T *Lc = p->L + (long)c * nc;
        const long sc = g->s[c];
        WLO_LOOP(SUF(inside)(g), { uc[I] -= Lc[I] * (p->x[I] - p->x[I - sc]); });
    }
    if (w == 1.0) { for (long q = 0; q < nc; ++q) p->x[q] = p->x[q] / dtT; }
    else { for (long q = 0; q < nc; ++q) p->x[q] = (T)((double)p->x[q] / dtd); }
    return n;
}

/* Flow.jl:153-169 mom_step!.  U = BCTuple at t+dt, gp/gc = accelerations (g + dU/dt) at t and t+dt
 * (NULL when `accelerate!` is a no-op, Flow.jl:73).  Returns the next dt from CFL. */
double SUF(wlo_mom_step)(SUF(wlo_flow) *a, SUF(wlo_mg) *b, double dt, const double *U,
                         const double *gp, const double *gc, int *n2) {
    const wlo_grid *g = &a->g;
    const long nv = g->ncell * g->D;
    memcpy(a->u0, a->u, (size_t)nv * sizeof(T));
    SUF(wlo_scale_u)(a->u, g, 0.0);
    /* predictor */
    SUF(wlo_conv_diff)(a->f, a->u0, a->sigma, g, a->nu, a->permask);
    if (gp) SUF(wlo_accelerate)(a->f, g, gp);
    SUF(wlo_bdim)(a->u, a->u0, a->f, a->V, a->mu0, a->mu1, g, dt);
    SUF(wlo_bc_vec)(a->u, g, U, a->exitBC, a->permask);
    if (a->exitBC) SUF(wlo_exit_bc)(a->u, a->u0, g, U, dt);
    n2[0] = SUF(wlo_project)(a, b, dt, 1.0, 1e-4, 32);
    SUF(wlo_bc_vec)(a->u, g, U, a->exitBC, a->permask);
    /* corrector */
    SUF(wlo_conv_diff)(a->f, a->u, a->sigma, g, a->nu, a->permask);
    if (gc) SUF(wlo_accelerate)(a->f, g, gc);
    SUF(wlo_bdim)(a->u, a->u0, a->f, a->V, a->mu0, a->mu1, g, dt);
    SUF(wlo_scale_u)(a->u, g, 0.5);
    SUF(wlo_bc_vec)(a->u, g, U, a->exitBC, a->permask);
    n2[1] = SUF(wlo_project)(a, b, dt, 0.5, 1e-4, 32);
    SUF(wlo_bc_vec)(a->u, g, U, a->exitBC, a->permask);
    return SUF(wlo_cfl)(a->sigma, a->u, g, a->nu);
}

SUF(wlo_flow) *SUF(wlo_flow_create)(const wlo_grid *g, T *u, T *u0, T *f, T *p, T *sigma, T *V, T *mu0,
                                   T *mu1, double nu, int exitBC, int permask) {
    SUF(wlo_flow) *a = (SUF(wlo_flow) *)calloc(1, sizeof(SUF(wlo_flow)));
    a->g = *g;
    a->u = u; a->u0 = u0; a->f = f; a->p = p; a->sigma = sigma; a->V = V; a->mu0 = mu0; a->mu1 = mu1;
    a->nu = nu; a->exitBC = exitBC; a->permask = permask;
    return a;
}
void SUF(wlo_flow_destroy)(SUF(wlo_flow) *a) { free(a); }

/* ---------------------------------------------------------------- Metrics.jl */

/* Metrics.jl:94-100 pressure_force: df[I,:] = p[I]*nds (Float64 product rounded into the T array df),
 * then summed with Float64 accumulation.  `nds` (Metrics.jl:84-87) is evaluated host-side and handed
 * over as a compact band: only cells with d^2 <= 1 have a non-zero nds.  df is zeroed first (:97). */
void SUF(wlo_pforce)(const T *p, T *df, const wlo_grid *g, const long *idx, const double *nds, long nband,
                     double *out) {
    const int D = g->D;
    const long nc = g->ncell;
    for (long q = 0; q < nc * D; ++q) df[q] = 0;
    for (long b = 0; b < nband; ++b)
        for (int c = 0; c < D; ++c) df[idx[b] + c * nc] = (T)((double)p[idx[b]] * nds[b * D + c]);
    for (int c = 0; c < D; ++c) {
        double s = 0;
        for (long b = 0; b < nband; ++b) s += (double)df[idx[b] + c * nc];
        out[c] = s;
    }
}

/* Metrics.jl:28-31  d u_i / d x_j at the cell centre, in T */
static inline T SUF(dudx)(const T *u, const wlo_grid *g, long I, int i, int j) {
    const T *ui = u + (long)i * g->ncell;
    if (i == j) return ui[I + g->s[i]] - ui[I];
    return (ui[I + g->s[j]] + ui[I + g->s[j] + g->s[i]] - ui[I - g->s[j]] - ui[I - g->s[j] + g->s[i]]) / (T)4;
}
/* Metrics.jl:109-113 viscous_force: df[I,:] = -nu*grad2u(I,u)*nds (Float32 matrix times Float64 vector), stored in the
 * T array df, summed in Float64.  nds comes as the same compact band as for wlo_pforce. */
void SUF(wlo_vforce)(const T *u, T *df, const wlo_grid *g, const long *idx, const double *nds, long nband, double nu_,
                     double *out) {
    const int D = g->D;
    const long nc = g->ncell;
    const T nu = (T)nu_;
    for (long q = 0; q < nc * D; ++q) df[q] = 0;
    for (long b = 0; b < nband; ++b)
        for (int i = 0; i < D; ++i) {
            double s = 0;
            for (int j = 0; j < D; ++j) {
                const T m = -nu * (T)(SUF(dudx)(u, g, idx[b], i, j) + SUF(dudx)(u, g, idx[b], j, i));
                s += (double)m * nds[b * D + j];
            }
            df[idx[b] + i * nc] = (T)s;
        }
    for (int c = 0; c < D; ++c) {
        double s = 0;
        for (long b = 0; b < nband; ++b) s += (double)df[idx[b] + c * nc];
        out[c] = s;
    }
}
/* Metrics.jl:130-134 pressure_moment: df[I,:] = p[I]*cross(loc(0,I)-x0, nds) */
void SUF(wlo_pmoment)(const T *p, T *df, const wlo_grid *g, const long *idx, const double *nds, long nband,
                      const double *x0, double *out) {
    const int D = g->D;
    const long nc = g->ncell;
    for (long q = 0; q < nc * D; ++q) df[q] = 0;
    for (long b = 0; b < nband; ++b) {
        const long I = idx[b];
        const long k = D > 2 ? I / g->s[2] : 0, rem = D > 2 ? I - k * g->s[2] : I;
        const long j = rem / g->s[1], i = rem - j * g->s[1];
        const double rx = (double)i - 0.5 - x0[0], ry = (double)j - 0.5 - x0[1], rz = D > 2 ? (double)k - 0.5 - x0[2] : 0.0;
        const double pv = (double)p[I];
        if (D == 3) {
            const double nx = nds[b * 3], ny = nds[b * 3 + 1], nz = nds[b * 3 + 2];
            df[I] = (T)(pv * (ry * nz - rz * ny));
            df[I + nc] = (T)(pv * (rz * nx - rx * nz));
            df[I + 2 * nc] = (T)(pv * (rx * ny - ry * nx));
        } else {
            const T m = (T)(pv * (rx * nds[b * 2 + 1] - ry * nds[b * 2]));
            df[I] = m; df[I + nc] = m;
        }
    }
    for (int c = 0; c < D; ++c) {
        double s = 0;
        for (long b = 0; b < nband; ++b) s += (double)df[idx[b] + c * nc];
        out[c] = s;
    }
}

/* Metrics.jl:14-77 field metrics over inside(out): kind 0 ke(I,u,U), 1 curl(i,I,u), 2 omega_mag, 3 omega_theta(I,z,c,u),
 * 4 lambda2 (see include/wlhip.h wl_metric for the parameter conventions). */
void SUF(wlo_metric)(int kind, T *out, const T *u, const wlo_grid *g, int ipar, const double *par, const double *par2) {
    const int D = g->D;
    const long nc = g->ncell;
    const long *S = g->s;
    WLO_LOOP(SUF(inside)(g), {
        T res = 0;
        if (kind == 0) {
            double s = 0;
            for (int c = 0; c < D; ++c) {
                const double v = (double)(T)(u[I + c * nc] + u[I + S[c] + c * nc]) - 2.0 * (par ? par[c] : 0.0);
                s += v * v;
            }
            res = (T)(0.125 * s);
        } else if (kind == 1) {
            const int a = (ipar + 1) % 3, b = (ipar + 2) % 3;
            res = (u[I + b * nc] - u[I - S[a] + b * nc]) - (u[I + a * nc] - u[I - S[b] + a * nc]);
        } else if (D == 3) {
            T w[3];
            for (int c = 0; c < 3; ++c) {
                const int a = (c + 1) % 3, b = (c + 2) % 3;
                w[c] = SUF(dudx)(u, g, I, b, a) - SUF(dudx)(u, g, I, a, b);
            }
            if (kind == 2) {
                res = (T)sqrt((double)(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
            } else if (kind == 3) {
                const double x[3] = {(double)i - 0.5 - par2[0], (double)j - 0.5 - par2[1], (double)k - 0.5 - par2[2]};
                const double th[3] = {par[1] * x[2] - par[2] * x[1], par[2] * x[0] - par[0] * x[2], par[0] * x[1] - par[1] * x[0]};
                const double n = sqrt(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
                res = n <= 2.220446049250313e-16 * n ? (T)0 : (T)((th[0] * (double)w[0] + th[1] * (double)w[1] + th[2] * (double)w[2]) / n);
            } else {
                double J[3][3], M[3][3];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) J[a][b] = (double)SUF(dudx)(u, g, I, a, b);
                for (int a = 0; a < 3; ++a)
                    for (int b = a; b < 3; ++b) {
                        double m = 0;
                        for (int c = 0; c < 3; ++c) {
                            const double sa = 0.5 * (J[a][c] + J[c][a]), sb = 0.5 * (J[c][b] + J[b][c]);
                            const double oa = 0.5 * (J[a][c] - J[c][a]), ob = 0.5 * (J[c][b] - J[b][c]);
                            m += sa * sb + oa * ob;
                        }
                        M[a][b] = m;
                    }
                res = (T)wlo_sym3_mid_eig(M[0][0], M[0][1], M[0][2], M[1][1], M[1][2], M[2][2]);
            }
        }
        out[I] = res;
    });
}

/* known-answer helpers exported for tests/test_oracle_pins.py */
double SUF(wlo_t_quick)(double u, double c, double d) { return (double)SUF(quick)((T)u, (T)c, (T)d); }
double SUF(wlo_t_vanleer)(double u, double c, double d) { return (double)SUF(vanleer)((T)u, (T)c, (T)d); }
double SUF(wlo_t_phi)(const T *f, long I) { return SUF(phi)(f, I, 1); }
double SUF(wlo_t_phiu)(const T *f, long I, double u) { return SUF(phiu)(f, I, 1, u); }
double SUF(wlo_t_phiuP)(const T *f, long Ip, long I, double u) { return SUF(phiuP)(f, Ip, I, 1, u); }
double SUF(wlo_t_phiuL)(const T *f, long I, double u) { return SUF(phiuL)(f, I, 1, u); }
double SUF(wlo_t_phiuR)(const T *f, long I, double u) { return SUF(phiuR)(f, I, 1, u); }

// wl_ops.h -- operator templates (one per reference operator), gfx950.
//
// v1 kernels: one thread per cell through the generic range kernels of wl_common.h; x is the unit-stride
// axis so every wavefront reads/writes 64 consecutive elements (coalesced 256-B/512-B segments); stencil
// neighbours are served by L1/L2 (per-XCD) and the 256 MiB Infinity Cache.  All per-cell arithmetic
// follows the reference operation by operation (citations per function) including its Float64
// promotions, so results match the CPU oracle to the last bit wherever no reduction is involved.
#pragma once
#include "wl_common.h"
#include "wl_convdiff.h"
#include "wl_stencil7.h"

namespace wl {

// device-resident solver scalars (no host round trip inside pcg!/residual!)
struct State {
    double rho, alpha, beta;  // pcg scalars, each holding a value already rounded to T
    double shift;             // residual! mean
    double r2;                // L2(p)
    double out[4];            // generic scalar outputs (cfl dt, dot, sum, ...)
    double red[4];            // reduction staging for the cross-rank all-reduce
    int active;               // pcg still running
    int do_shift;             // residual! must subtract the mean
    int nupd;                 // number of (x,r) updates pcg performed
    int r2_valid;             // the last pcg! update already produced r.r (solver! can skip the separate L2 pass)
    int xpend;                // pcg stopped at :138 with x += alpha*eps still owed (deferred-x form, see op_pcg)
    PcgS slots[2];            // pcg!'s scalars while its kernels evaluate them themselves (Gate kind 1..4, wl_stencil7.h)
};

template <class T> struct LevelT {
    G g;
    T *L, *D, *iD, *x, *eps, *r, *z;
    const T *rowc = nullptr;   // row constants of L / iD (wl_stencil7.h, k_lrow); nullptr = none (raw operator calls)
};

inline long span(const G &g) { return g.D == 3 ? (long)g.n[2] * g.s[2] : (long)g.n[1] * g.s[1]; }

// ------------------------------------------------------------------------------------------ util.jl
// BC!(a,A,saveexit,perdir)  src/util.jl:192-210 -- same sequence of plane loops as the reference, because
// later planes read ghost values written by earlier ones (edges/corners).
template <class T, int D> int op_bc_vec_fused(const G &g, T *a, const double *A, int saveexit, int permask, bool skip_x = false);
// The x part of BC!(u,U) on the INTERIOR rows is row-local when x is not periodic (normal component: u[0] = u[1] = U0,
// u[n-1] = U0 unless the exit is saved; tangential components: u[0] = u[1], u[n-1] = u[n-2]), so the kernel that has just
// produced a row of u (BDIM!, the velocity correction) writes those cells itself, into DRAM pages it is streaming anyway.
// As a launch of its own the three x planes -- one cache line per cell -- cost 0.1 ms at 512^3, the y and z planes
// 0.02 ms; every x-plane cell of a ghost ROW is covered by the y / z planes of the BC kernel (skip_x).
template <class T> struct XBc { int on, saveexit; T U0; };
template <class T> __device__ __forceinline__ void xbc_apply(const XBc<T> &b, T *u, long o, long sc, int i, int n0, VecA<T> (&uv)[3]) {
    constexpr int V = VecA<T>::V;
    if (!b.on) return;
    if (i == 1) { uv[0].v[0] = b.U0; u[o - 1] = b.U0; u[o - 1 + sc] = uv[1].v[0]; u[o - 1 + 2 * sc] = uv[2].v[0]; }
    if (i + V - 1 == n0 - 2) {
        if (!b.saveexit) u[o + V] = b.U0;
        u[o + V + sc] = uv[1].v[V - 1]; u[o + V + 2 * sc] = uv[2].v[V - 1];
    }
}
template <class T, int D>
int op_bc_vec(const G &g, T *a, const double *A, int saveexit, int permask, bool skip_x = false) {
    if (ctx().opt[7]) return op_bc_vec_fused<T, D>(g, a, A, saveexit, permask, skip_x);   // one launch (closed form), see below
    if (skip_x) return fail(WL_E_STATE, "BC!: x planes folded into the producer need the one-launch form", __FILE__, __LINE__);
    for (int c = 0; c < D; ++c)
        for (int j = 0; j < D; ++j) {
            T *ac = a + (long)c * g.sc;
            const long sj = g.s[j];
            const int N = (j == 2) ? g.nzg : g.n[j];   // plane numbers along z are global
            const G gg = g;
            if (((permask >> j) & 1) && j == 2 && g.dist) {
                if (g.zring) continue;   // ring of slabs: the halo exchange that follows BC! fills the z ghost planes
                return fail(WL_E_ARG, "periodic z on a z-slab decomposition needs grid.zring", __FILE__, __LINE__);
            }
            if ((permask >> j) & 1) {
                const long off = (long)(N - 2) * sj;
                WL_TRY(launch_range(WL_K_BC, r_slice(g, 0, j, 0), [=] __device__(int i, int jj, int k) {
                    const long I = gg.at(i, jj, k); ac[I] = ac[I + off]; }));
                WL_TRY(launch_range(WL_K_BC, r_slice(g, N - 1, j, 0), [=] __device__(int i, int jj, int k) {
                    const long I = gg.at(i, jj, k); ac[I] = ac[I - off]; }));
            } else if (c == j) {
                const T Ac = (T)A[c];
                for (int s = 0; s < 2; ++s)
                    WL_TRY(launch_range(WL_K_BC, r_slice(g, s, j, 0), [=] __device__(int i, int jj, int k) {
                        ac[gg.at(i, jj, k)] = Ac; }));
                if (!saveexit || c > 0)
                    WL_TRY(launch_range(WL_K_BC, r_slice(g, N - 1, j, 0), [=] __device__(int i, int jj, int k) {
                        ac[gg.at(i, jj, k)] = Ac; }));
            } else {
                WL_TRY(launch_range(WL_K_BC, r_slice(g, 0, j, 0), [=] __device__(int i, int jj, int k) {
                    const long I = gg.at(i, jj, k); ac[I] = ac[I + sj]; }));
                WL_TRY(launch_range(WL_K_BC, r_slice(g, N - 1, j, 0), [=] __device__(int i, int jj, int k) {
                    const long I = gg.at(i, jj, k); ac[I] = ac[I - sj]; }));
            }
        }
    return 0;
}

// BC!(a,A,saveexit,perdir) in ONE launch.  The reference applies its plane loops in the order (component i, direction j)
// and later loops read ghost values written by earlier ones; the value a cell ends up with can be evaluated directly by
// walking the passes of its component BACKWARDS: periodic j -> continue at the wrapped source index, tangential j ->
// continue at the clamped source index, normal j -> the Dirichlet value A[i] (planes 1,2,N; N skipped when saving the
// exit).  The walk ends on a cell no pass writes, so every thread reads only never-written cells: no ordering hazard.
template <class T, int D>
__global__ __launch_bounds__(256) void k_bc_vec_all(G g, T *a, T A0, T A1, T A2, int saveexit, int permask, int combo0) {
    // blockIdx.y -> (direction d, plane pl in {0,1,n-1}, component c): uniform per workgroup, decoded on the scalar unit;
    // blockIdx.x*256 + thread -> position in the plane, first remaining axis fastest (for the y and z planes consecutive
    // lanes write consecutive x).  One 32-bit division per thread (the flat decode of round 1 had eight: VALU-bound).
    const int ng[3] = {g.n[0], g.n[1], D > 2 ? g.nzg : 1};
    const int combo = (int)blockIdx.y + combo0;   // combo0 = 3*D: the x planes are not part of this launch
    const int c = combo % D, pl = (combo / D) % 3, d = combo / (3 * D);
    const int e1 = d == 0 ? 1 : 0, e2 = (D > 2) ? (d == 2 ? 1 : 2) : -1;
    const unsigned ext1 = (e1 == 2) ? (unsigned)(g.zhi - g.zlo + 1) : (unsigned)g.n[e1];
    const unsigned ext2 = (e2 < 0) ? 1u : ((e2 == 2) ? (unsigned)(g.zhi - g.zlo + 1) : (unsigned)g.n[e2]);
    const unsigned pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= ext1 * ext2) return;
    const unsigned q2 = pos / ext1, q1 = pos - q2 * ext1;
    int idx[3] = {0, 0, 0};   // GLOBAL indices
    idx[e1] = (int)q1 + (e1 == 2 ? g.zlo + g.kz0 : 0);
    if constexpr (D > 2) idx[e2] = (int)q2 + (e2 == 2 ? g.zlo + g.kz0 : 0);
    idx[d] = pl == 0 ? 0 : (pl == 1 ? 1 : ng[d] - 1);
    if (D > 2 && d == 2) {   // z planes: only the rank that owns them (nobody on a periodic ring: halo exchange fills them)
        const int kl = idx[2] - g.kz0;
        if (g.zring || kl < g.zlo || kl > g.zhi) return;
    }
    const long dst = g.at(idx[0], idx[1], D > 2 ? idx[2] - g.kz0 : 0) + (long)c * g.sc;
    const T Ac = c == 0 ? A0 : (c == 1 ? A1 : A2);
    bool dirichlet = false;
    for (int j = D - 1; j >= 0; --j) {
        const int n = ng[j];
        if ((permask >> j) & 1) {
            if (idx[j] == 0) idx[j] = n - 2;
            else if (idx[j] == n - 1) idx[j] = 1;
        } else if (j == c) {
            if (idx[j] <= 1 || (idx[j] == n - 1 && (!saveexit || c > 0))) { dirichlet = true; break; }
        } else {
            if (idx[j] == 0) idx[j] = 1;
            else if (idx[j] == n - 1) idx[j] = n - 2;
        }
    }
    a[dst] = dirichlet ? Ac : a[g.at(idx[0], idx[1], D > 2 ? idx[2] - g.kz0 : 0) + (long)c * g.sc];
}
template <class T, int D>
int op_bc_vec_fused(const G &g, T *a, const double *A, int saveexit, int permask, bool skip_x) {
    if (D > 2 && ((permask >> 2) & 1) && g.dist && !g.zring)
        return fail(WL_E_ARG, "periodic z on a z-slab decomposition needs grid.zring", __FILE__, __LINE__);
    long total = 0, big = 0;
    for (int q = skip_x ? 1 : 0; q < D; ++q) {
        long cells = 1;
        for (int e = 0; e < D; ++e) if (e != q) cells *= (e == 2 ? (long)(g.zhi - g.zlo + 1) : (long)g.n[e]);
        total += cells * 3 * D;
        big = cells > big ? cells : big;
    }
    if (big <= 0) return 0;
    if (big >= (1L << 31)) return fail(WL_E_ARG, "BC!: more than 2^31 cells in a boundary plane", __FILE__, __LINE__);
    Prof p(WL_K_BC, total);
    const int combo0 = skip_x ? 3 * D : 0;
    hipLaunchKernelGGL((k_bc_vec_all<T, D>), dim3((unsigned)((big + 255) / 256), (unsigned)(3 * D * D - combo0)), dim3(256), 0, ctx().stream,
                       g, a, (T)A[0], (T)A[1], (T)(D > 2 ? A[2] : 0.0), saveexit, permask, combo0);
    return (int)hipGetLastError();
}

// perBC!(a,perdir)  src/util.jl:227-231
// exchange = false: only the x/y copies; the caller exchanges the z halos itself (launch_stencil7_halo overlaps it)
template <class T, int D>
int op_bc_per(const G &g, T *a, int permask, bool exchange = true) {
    for (int j = 0; j < D; ++j)
        if ((permask >> j) & 1) {
            if (j == 2 && g.dist) {
                if (g.zring) continue;   // the exchange below is the periodic copy
                return fail(WL_E_ARG, "periodic z on a z-slab decomposition needs grid.zring", __FILE__, __LINE__);
            }
            const long off = (long)(g.n[j] - 2) * g.s[j];
            const G gg = g;
            WL_TRY(launch_range(WL_K_BC, r_slice(g, 0, j, 0), [=] __device__(int i, int jj, int k) {
                const long I = gg.at(i, jj, k); a[I] = a[I + off]; }));
            WL_TRY(launch_range(WL_K_BC, r_slice(g, g.n[j] - 1, j, 0), [=] __device__(int i, int jj, int k) {
                const long I = gg.at(i, jj, k); a[I] = a[I - off]; }));
        }
    // z-slab decomposition: the reference calls perBC! exactly where a stencil operand's ghosts must be current
    // (mult!, residual!, increment!, pcg!, end of solver!), which is also where the z halos must be exchanged
    return exchange ? halo_exchange<T>(g, a, 1, 1) : 0;
}

// exitBC!(u,u0,U,dt)  src/util.jl:216-222
template <class T, int D>
int op_exit_bc(const G &g, T *u, const T *u0, const double *U, double dt_, double *partials, State *st) {
    Range R;
    for (int d = 0; d < 3; ++d) {
        if (d >= D) { R.lo[d] = R.hi[d] = 0; }
        else if (d == 0) { R.lo[d] = R.hi[d] = g.n[0] - 1; }
        else { R.lo[d] = 1; R.hi[d] = g.n[d] - 2; }
    }
    clip_z(g, R, 1, g.nzg - 2);
    const T U1 = (T)U[0], Udt = U1 * (T)dt_;
    const G gg = g;
    int np = 0;
    WL_TRY((launch_range_red<1>(WL_K_BC, R, [=] __device__(int i, int j, int k, double(&acc)[1]) {
        const long I = gg.at(i, j, k);
        const T v = u0[I] - Udt * (u0[I] - u0[I - 1]);
        u[I] = v;
        acc[0] += (double)v;
    }, partials, RED_SUM, 0.0, &np)));
    const T cnt = (T)((long)(g.n[1] - 2) * (D > 2 ? (long)(g.nzg - 2) : 1L));  // length(exitR) of the whole domain
    WL_TRY((launch_finalize<1>(g.dist, partials, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
        st->out[0] = (double)((T)v[0] / cnt - U1); })));
    return launch_range(WL_K_BC, R, [=] __device__(int i, int j, int k) { u[gg.at(i, j, k)] -= (T)st->out[0]; });
}

// generic reductions: dot / sum / max / sum of squares (Float64 accumulation, fixed order)
// whole: every element of the array, ghost cells included (Base.sum / maximum / LinearAlgebra.dot of the reference); z-slab
// runs: the planes this rank owns
template <class T, int D, class F>
int op_reduce(const G &g, int kclass, int op, double init, F cell, double *partials, State *st, int slot, bool whole = false) {
    int np = 0;
    const G gg = g;
    WL_TRY((launch_range_red<1>(kclass, whole ? r_whole(g) : r_inside(g), [=] __device__(int i, int j, int k, double(&acc)[1]) {
        const double v = cell(gg.at(i, j, k));
        acc[0] = (op == RED_SUM) ? acc[0] + v : (v > acc[0] ? v : acc[0]);
    }, partials, op, init, &np)));
    return launch_finalize<1>(g.dist, partials, np, op, init, st->red, [=] __device__(const double *v) { st->out[slot] = v[0]; });
}

// ------------------------------------------------------------------------------------------ Flow.jl
// conv_diff!(r,u,Phi;nu,perdir)  src/Flow.jl:36-60 in GATHER form: every cell evaluates the fluxes through
// its own lower and upper faces and applies them in the reference's order (+lower then -upper, j=1..D),
// each interior flux rounded to T first exactly like the Phi scratch does (:45-47); boundary faces are added
// in Float64 like :54-55.  No Phi array, no 9x re-read of r, no write race.
// FUSE: also applies accelerate! (:68-70) and the first BDIM! loop (:133): f = u0 + dt*r - V on ALL cells.
// flux sum of component c at one cell, every operand addressed as ub[jd*csz + I +- k*st[jd]]: the velocity array itself
// (csz = g.sc, st = g.s) or an LDS patch of it with its own strides (k_convdiff_xghost below)
template <class T, int D>
__device__ __forceinline__ T cd_gather_rr(const T *ub, long csz, const long (&st)[3], long I, const int (&idx)[3], const int (&N)[3],
                                          bool zring, int permask, T nu, int c) {
    const T *ui = ub + (long)c * csz;
    const long si = st[c];
    T rr = 0;
_Pragma("unroll")
    for (int jd = 0; jd < D; ++jd) {
        const int Nj = N[jd];
        const bool ring = (jd == 2) && zring;   // periodic ring of slabs: every z face is an interior face
        if (!ring && idx[jd] > Nj - 2) continue;
        const T *uj = ub + (long)jd * csz;
        const long sj = st[jd];
        const bool per = ((permask >> jd) & 1) && !ring;
        {   // lower face of the cell: face index I
            const double uf = phi<T>(uj, I, si);
            const T nud = nu * (T)(ui[I] - ui[I - sj]);
            if (!ring && idx[jd] == 1) {
                if (!per) {
                    rr = (T)((double)rr + (phiuL<T>(ui, I, sj, uf) - (double)nud));
                } else {
                    const T P = (T)(phiuP<T>(ui, I + (long)(Nj - 4) * sj, I, sj, uf) - (double)nud);
                    rr += P;
                }
            } else {
                const T P = (T)(phiu<T>(ui, I, sj, uf) - (double)nud);
                rr += P;
            }
        }
        {   // upper face of the cell: face index I+sj
            const long J = I + sj;
            if (!ring && idx[jd] == Nj - 2) {
                if (!per) {
                    const double uf = phi<T>(uj, J, si);
                    const T nud = nu * (T)(ui[J] - ui[J - sj]);
                    rr = (T)((double)rr + (-phiuR<T>(ui, J, sj, uf) + (double)nud));
                } else {  // :60  r[I-d] -= Phi[CIj(j,I,2)] : the lower-boundary flux, wrapped
                    const long I2 = I - (long)(idx[jd] - 1) * sj;
                    const double uf = phi<T>(uj, I2, si);
                    const T nud = nu * (T)(ui[I2] - ui[I2 - sj]);
                    const T P = (T)(phiuP<T>(ui, I2 + (long)(Nj - 4) * sj, I2, sj, uf) - (double)nud);
                    rr -= P;
                }
            } else {
                const double uf = phi<T>(uj, J, si);
                const T nud = nu * (T)(ui[J] - ui[J - sj]);
                const T P = (T)(phiu<T>(ui, J, sj, uf) - (double)nud);
                rr -= P;
            }
        }
    }
    return rr;
}

// ---- sigma's ghost cells.  conv_diff!(a.f, a.u, a.σ) uses σ as its flux scratch Φ (Flow.jl:45,59,157,164): the loops over
// inside_u(N,j) = 3:N_j-1 x 2:N_k (TOP ghost included, util.jl:55-57) and, for a periodic j, over slice(N,2,j,2) leave Φ in
// σ's top ghost cells, and the reference's whole-array reductions see it there: maximum(a.σ) in CFL (Flow.jl:174), and z⋅ϵ
// of pcg! on level 1, where z ≡ σ (WaterLily.jl:77) and ϵ's ghosts are periodic copies (Poisson.jl:129,131).  The gather
// form of conv_diff! has no Φ, so those values are produced here.  A ghost cell I (every index >= 2, at least one == N)
// ends up with the flux of the LAST (i, j) pair of `for i ∈ 1:n, j ∈ 1:n` whose range contains it: i = n and the largest j
// with 3 <= I_j <= N_j-1 (I_j == 2 too when j is periodic: lowerBoundary!, Flow.jl:58-59).  A cell that no range contains
// keeps what it holds (the zero it was created with).  Indices below are 0-based; idx[2] and N[2] are global.
template <class T, int D>
__device__ __forceinline__ bool phi_stale(const T *u, long csz, const long (&st)[3], long I, const int (&idx)[3], const int (&N)[3],
                                          bool zring, int permask, T nu, T &out) {
    int js = -1;
    bool plow = false;
_Pragma("unroll")
    for (int j = 0; j < D; ++j) {
        const bool ring = (j == 2) && zring;   // ring of slabs: no z boundary, the halo planes hold the wrapped cells
        if ((idx[j] >= 2 && idx[j] <= N[j] - 2) || (ring && idx[j] == 1)) { js = j; plow = false; }
        else if (((permask >> j) & 1) && !ring && idx[j] == 1) { js = j; plow = true; }
    }
    if (js < 0) return false;
    const T *ui = u + (long)(D - 1) * csz, *uj = u + (long)js * csz;
    const long si = st[D - 1], sj = st[js];
    const double uf = phi<T>(uj, I, si);
    const T nud = nu * (T)(ui[I] - ui[I - sj]);
    out = plow ? (T)(phiuP<T>(ui, I + (long)(N[js] - 4) * sj, I, sj, uf) - (double)nud) : (T)(phiu<T>(ui, I, sj, uf) - (double)nud);
    return true;
}
// write them into sigma (the top planes of the shell; cells with an index 0 are never in a range)
template <class T, int D>
int op_sigma_ghosts(const G &g, T *sigma, const T *u, double nu_, int permask) {
    const G gg = g;
    const T nu = (T)nu_;
    const int top = (1 << 1) | (1 << 3) | (D > 2 ? (1 << 5) : 0);
    return launch_shell(WL_K_CONVDIFF, g, top, [=] __device__(int i, int j, int k) {
        const int idx[3] = {i, j, D > 2 ? gg.kg(k) : 0};
        if (i < 1 || j < 1 || (D > 2 && idx[2] < 1)) return;
        const int N[3] = {gg.n[0], gg.n[1], gg.nzg};
        const long st[3] = {gg.s[0], gg.s[1], gg.s[2]};
        const long I = gg.at(i, j, k);
        T v;
        if (phi_stale<T, D>(u, gg.sc, st, I, idx, N, gg.zring, permask, nu, v)) sigma[I] = v;
    });
}

template <class T, int D, bool FUSE, bool COPY = false>
int op_conv_diff_range(const G &g, const Range &R, T *r, const T *u, double nu_, int permask, const T *u0, const T *V,
                       double dt_, const double *acc, bool has_acc, T *u0out = nullptr) {
    const T nu = (T)nu_, dt = (T)dt_;
    double a3[3] = {0, 0, 0};
    if (has_acc) for (int d = 0; d < D; ++d) a3[d] = acc[d];
    const double a0 = a3[0], a1 = a3[1], a2 = a3[2];
    const G gg = g;
    return launch_range(WL_K_CONVDIFF, R, [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
        const int idx[3] = {i, j, gg.kg(k)};   // z index in global numbering
        const int N[3] = {gg.n[0], gg.n[1], gg.nzg};
        const long st[3] = {gg.s[0], gg.s[1], gg.s[2]};
        bool lowok = true;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) lowok = lowok && (idx[d] >= 1 || (d == 2 && gg.zring));
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
            T rr = 0;
            if (lowok) rr = cd_gather_rr<T, D>(u, gg.sc, st, I, idx, N, gg.zring, permask, nu, c);
            if (FUSE) {
                if (has_acc) rr = (T)((double)rr + (c == 0 ? a0 : (c == 1 ? a1 : a2)));
                const T uo = COPY ? u[I + (long)c * gg.sc] : u0[I + (long)c * gg.sc];
                if (COPY) u0out[I + (long)c * gg.sc] = uo;
                r[I + (long)c * gg.sc] = (uo + dt * rr) - V[I + (long)c * gg.sc];
            } else {
                r[I + (long)c * gg.sc] = rr;
            }
        }
    });
}

// The two x-ghost planes of a 3-D, non-periodic conv_diff! (i = 0: no flux at all; i = n0-1: y and z fluxes only,
// util.jl:55-57 "top ghost included") in ONE launch.  As a plain gather a thread of these planes issues ~100 loads, each
// lane on its own cache line (the plane is strided by the row pitch): 0.07 + 0.21 ms at 512^3 for 0.4 % of the cells.
// Here a 64(y) x 4(z) tile first stages what its cells read -- the three components in columns n0-2, n0-1 over the tile
// + 2 halo rows / planes -- in LDS (13 loads per thread, all in flight together) and evaluates the same expressions
// (cd_gather_rr) on the patch.
constexpr int XG_BJ = 64, XG_BK = 4, XG_H = 2;
constexpr int XG_ROWS = XG_BJ + 2 * XG_H, XG_PL = XG_BK + 2 * XG_H;
constexpr int XG_SY = 2, XG_SZ = XG_SY * XG_ROWS, XG_CS = XG_SZ * XG_PL;   // patch strides: x, y, z, component
template <class T, bool FUSE, bool COPY>
__global__ __launch_bounds__(XG_BJ *XG_BK) void k_convdiff_xghost(G g, T *__restrict__ r, const T *__restrict__ u, T nu, const T *u0,
                                                                  T *u0out, const T *__restrict__ V, T dt, double a0, double a1,
                                                                  double a2, bool has_acc, int ntj, int ntile, int klo, int khi) {
    __shared__ T pt[3 * XG_CS];
    const int role_top = (int)blockIdx.x < ntile;              // the heavier plane first
    const int tb = role_top ? (int)blockIdx.x : (int)blockIdx.x - ntile;
    const int j0 = XG_BJ * (tb % ntj), k0 = klo + XG_BK * (tb / ntj);
    const int tj = threadIdx.x & (XG_BJ - 1), tk = threadIdx.x / XG_BJ;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int j = j0 + tj, k = k0 + tk;
    const bool active = (j <= n1 - 1) && (k <= khi);
    const int i = role_top ? n0 - 1 : 0;
    if (role_top) {
        for (int e = threadIdx.x; e < 3 * XG_CS; e += XG_BJ * XG_BK) {
            const int c = e / XG_CS, e1 = e - c * XG_CS;
            const int p = e1 / XG_SZ, e2 = e1 - p * XG_SZ;
            const int row = e2 >> 1, x = e2 & 1;
            const int gj = min(max(j0 - XG_H + row, 0), n1 - 1), gk = min(max(k0 - XG_H + p, 0), n2 - 1);
            pt[e] = u[(long)c * g.sc + (long)(n0 - 2 + x) + g.s[1] * (long)gj + g.s[2] * (long)gk];
        }
        __syncthreads();
    }
    if (!active) return;
    const long I = g.at(i, j, k);
    const int idx[3] = {i, j, g.kg(k)};
    const int N[3] = {n0, n1, g.nzg};
    const long st[3] = {1, XG_SY, XG_SZ};
    const long Il = 1 + XG_SY * (long)(tj + XG_H) + XG_SZ * (long)(tk + XG_H);
    const bool lowok = role_top && idx[1] >= 1 && (idx[2] >= 1 || g.zring);
_Pragma("unroll")
    for (int c = 0; c < 3; ++c) {
        T rr = 0;
        if (lowok) rr = cd_gather_rr<T, 3>(pt, XG_CS, st, Il, idx, N, g.zring, 0, nu, c);
        const long q = I + (long)c * g.sc;
        if (FUSE) {
            if (has_acc) rr = (T)((double)rr + (c == 0 ? a0 : (c == 1 ? a1 : a2)));
            const T uo = COPY ? u[q] : u0[q];
            if (COPY) u0out[q] = uo;
            r[q] = (uo + dt * rr) - V[q];
        } else {
            r[q] = rr;
        }
    }
}
template <class T, bool FUSE, bool COPY>
int launch_convdiff_xghost(const G &g, T *r, const T *u, double nu_, const T *u0, const T *V, double dt_, const double *acc,
                           bool has_acc, T *u0out) {
    const Range R = r_whole(g);
    const int nk = R.hi[2] - R.lo[2] + 1;
    if (nk <= 0) return 0;
    double a3[3] = {0, 0, 0};
    if (has_acc) for (int d = 0; d < 3; ++d) a3[d] = acc[d];
    const int ntj = (g.n[1] + XG_BJ - 1) / XG_BJ, ntile = ntj * ((nk + XG_BK - 1) / XG_BK);
    Prof p(WL_K_CONVDIFF, 2L * g.n[1] * nk);
    hipLaunchKernelGGL((k_convdiff_xghost<T, FUSE, COPY>), dim3(2 * ntile), dim3(XG_BJ * XG_BK), 0, ctx().stream, g, r, u, (T)nu_, u0,
                       u0out, V, (T)dt_, a3[0], a3[1], a3[2], has_acc, ntj, ntile, R.lo[2], R.hi[2]);
    return (int)hipGetLastError();
}

// dispatch: D=3 non-periodic -> LDS-tiled marching kernel (wl_convdiff.h) + generic gather on the two x-ghost
// planes; everything else (2-D, periodic directions) -> generic gather kernel over the whole array.
// COPY: also perform `u0 .= u` (Flow.jl:154) for the cells written (u0out), the epilogue then uses u itself.
// exchange_u (z-slab runs): the 2-plane halo exchange of u that has to precede this call is issued HERE, on the comm
// stream, and the LDS kernel runs on the planes that read no halo plane (zlo+2 .. zhi-2) while it is in flight.
// does op_conv_diff take the LDS-tiled kernels for this grid?
template <int D> inline bool conv_diff_tiled(const G &g, int permask) {
    return D == 3 && ctx().opt[2] && (permask == 0 || (permask == 4 && g.zring)) && g.n[0] >= 5 && g.n[1] >= 5 && g.n[2] >= 5;
}
// FIN (1 predictor / 2 corrector, with `fin`): the tiled kernels also finish BDIM! on the body-free rows (CdFin, wl_convdiff.h);
// only where conv_diff_tiled() holds.  The corrector's `fin->unew` is the array `u0` points to (every cell is read by the thread
// that writes it), so the x-ghost planes -- which read u0 in cells the row kernels overwrite with BC! values -- go first.
template <class T, int D, bool FUSE, bool COPY = false, int FIN = 0>
int op_conv_diff(const G &g, T *r, const T *u, double nu_, int permask, const T *u0, const T *V, double dt_,
                 const double *acc, bool has_acc, T *u0out = nullptr, bool exchange_u = false, const CdFin<T> *fin = nullptr) {
    if constexpr (D == 3) {
        if (conv_diff_tiled<D>(g, permask)) {
            const CdFin<T> fn = (FIN && fin) ? *fin : CdFin<T>{nullptr, nullptr, 0, (T)0};
            if (FIN && (!fin || !fin->unew || !fin->rowfree || (const T *)fin->unew == u))
                return fail(WL_E_STATE, "conv_diff!: finishing BDIM! needs the row flags and an output that is not the stencil input", __FILE__, __LINE__);
            if (exchange_u && g.dist && overlap_on() && g.zhi - g.zlo + 1 >= 5) {
                WL_TRY((halo_begin<T>(g, const_cast<T *>(u), D, 2)));
                G gi = g;
                gi.zlo = g.zlo + 2; gi.zhi = g.zhi - 2;
                int rc = 0;
                if (FIN == 2) rc = launch_convdiff_xghost<T, FUSE, COPY>(gi, r, u, nu_, u0, V, dt_, acc, has_acc, u0out);
                if (!rc) rc = launch_convdiff3<T, FUSE, COPY, FIN>(gi, r, u, nu_, u0, u0out, V, dt_, acc, has_acc, fn);
                WL_TRY(halo_end());
                if (rc) return rc;
                G gl = g, gh = g;
                gl.zhi = g.zlo + 1; gh.zlo = g.zhi - 1;
                if (FIN == 2) {
                    WL_TRY((launch_convdiff_xghost<T, FUSE, COPY>(gl, r, u, nu_, u0, V, dt_, acc, has_acc, u0out)));
                    WL_TRY((launch_convdiff_xghost<T, FUSE, COPY>(gh, r, u, nu_, u0, V, dt_, acc, has_acc, u0out)));
                }
                WL_TRY((launch_convdiff3<T, FUSE, COPY, FIN>(gl, r, u, nu_, u0, u0out, V, dt_, acc, has_acc, fn)));
                WL_TRY((launch_convdiff3<T, FUSE, COPY, FIN>(gh, r, u, nu_, u0, u0out, V, dt_, acc, has_acc, fn)));
                if (FIN == 2) return 0;
            } else {
                if (exchange_u) WL_TRY((halo_exchange<T>(g, const_cast<T *>(u), D, 2)));
                if (FIN == 2) WL_TRY((launch_convdiff_xghost<T, FUSE, COPY>(g, r, u, nu_, u0, V, dt_, acc, has_acc, u0out)));
                WL_TRY((launch_convdiff3<T, FUSE, COPY, FIN>(g, r, u, nu_, u0, u0out, V, dt_, acc, has_acc, fn)));
                if (FIN == 2) return 0;
            }
            // (the two x-ghost planes -- strided, latency-bound, 0.07 + 0.21 ms at 512^3 -- were also tried on a side stream
            //  next to the LDS kernel: no gain, 29.73 vs 29.74 ms per step)
            return launch_convdiff_xghost<T, FUSE, COPY>(g, r, u, nu_, u0, V, dt_, acc, has_acc, u0out);
        }
    }
    if (FIN) return fail(WL_E_STATE, "conv_diff!: finishing BDIM! needs the tiled kernels", __FILE__, __LINE__);
    if (exchange_u) WL_TRY((halo_exchange<T>(g, const_cast<T *>(u), D, 2)));
    return op_conv_diff_range<T, D, FUSE, COPY>(g, r_whole(g), r, u, nu_, permask, u0, V, dt_, acc, has_acc, u0out);
}

// accelerate!  src/Flow.jl:68-70: r[..,i] .+= g_i on every element
template <class T, int D>
int op_accelerate(const G &g, T *r, const double *acc) {
    const double a0 = acc[0], a1 = acc[1], a2 = D > 2 ? acc[2] : 0.0;
    const G gg = g;
    return launch_range(WL_K_MISC, r_whole(g), [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
            T *p = r + I + (long)c * gg.sc;
            *p = (T)((double)*p + (c == 0 ? a0 : (c == 1 ? a1 : a2)));
        }
    });
}

// first BDIM! loop  src/Flow.jl:133 (all cells)
template <class T, int D>
int op_bdim1(const G &g, T *f, const T *u0, const T *V, double dt_) {
    const T dt = (T)dt_;
    const G gg = g;
    return launch_range(WL_K_BDIM, r_whole(g), [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
            const long q = I + (long)c * gg.sc;
            f[q] = (u0[q] + dt * f[q]) - V[q];
        }
    });
}
// second BDIM! loop  src/Flow.jl:134 with mu_ddn (:18-24).  MODE 0: u += ...  (the reference statement)
// MODE 1: predictor, u was zeroed by scale_u!(a,0) (:154) -> u = ... ; MODE 2: corrector, followed by
// scale_u!(a,0.5) (:166) -> u = 0.5*(u + ...), both roundings kept.
// general BDIM! statement on a compact list of rows (row = j + n1*k), one wavefront per 64-cell row segment
template <class T, int MODE>
// seg (optional; wl_flow_update's scan): seg[row*ntx + s] != 0 -- the 64-cell segment s of that row holds no body cell (mu1 = 0,
// V = 0, mu0 = 1 there): the statement is u (+)= f as in a body-free row, the 15 coefficient values and the 6 neighbours of f
// are not read.  A torus puts a band cell into 21 % of the x-rows of a 512^3 grid, but into a third of their segments.
__global__ __launch_bounds__(256) void k_bdim2_busy(G g, T *u, const T *uin, const T *f, const T *V, const T *mu0, const T *mu1,
                                                    const int *rows, int nrows, int ntx, XBc<T> xb, const unsigned char *seg) {   // uin: the u that is read (MODE 0, 2)
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)nrows * ntx) return;
    const int row = rows[w / ntx];
    const int i = 1 + (int)(w % ntx) * 64 + (threadIdx.x & 63);
    const bool sfree = seg && __builtin_amdgcn_readfirstlane((int)seg[(long)row * ntx + (w % ntx)]) != 0;
    if (i > g.n[0] - 2) return;
    const int j = row % g.n[1], k = row / g.n[1];
    const long I = g.at(i, j, k);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T *fc = f + (long)c * g.sc;
        const long q = I + (long)c * g.sc;
        double tmp;
        if (sfree) {
            tmp = (0.5 * 0.0 + 0.0) + (double)fc[I];
        } else {
            T s = 0;
#pragma unroll
            for (int jd = 0; jd < 3; ++jd) {
                const T *m1 = mu1 + (long)(c + 3 * jd) * g.sc;
                s += m1[I] * (fc[I + g.s[jd]] - fc[I - g.s[jd]]);
            }
            tmp = (0.5 * (double)s + (double)V[q]) + (double)(T)(mu0[q] * fc[I]);
        }
        T val;
        if (MODE == 1) val = (T)(0.0 + tmp);
        else { const T un = (T)((double)uin[q] + tmp); val = (MODE == 2) ? (T)((double)un * 0.5) : un; }
        if (xb.on) {   // the row's x-ghost cells (see XBc)
            if (i == 1) { if (c == 0) val = xb.U0; u[q - 1] = val; }
            if (i == g.n[0] - 2 && !(c == 0 && xb.saveexit)) u[q + 1] = (c == 0) ? xb.U0 : val;
        }
        u[q] = val;
    }
}

// the busy rows alone, out of place: the body-free rows were finished by conv_diff! (CdFin).  z-slab runs: mu_ddn reads
// f[I +- dz], so f's plane travels first -- on the comm stream, while the busy rows of the planes that read no halo plane are
// done (the list is sorted by plane: the first nlo rows lie in the first owned interior plane, the last nhi in the last one)
template <class T, int MODE>
int op_bdim2_busy(const G &g, T *u, const T *uin, const T *f, const T *V, const T *mu0, const T *mu1, const int *busy, int nbusy,
                  int nlo, int nhi, const XBc<T> &xb, const unsigned char *seg = nullptr) {
    const int ntx = (g.n[0] - 2 + 63) / 64;
    auto rows = [&](int first, int n) -> int {
        if (n <= 0) return 0;
        const long nw = (long)n * ntx;
        Prof p(WL_K_BDIM, (long)n * (g.n[0] - 2));
        hipLaunchKernelGGL((k_bdim2_busy<T, MODE>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, ctx().stream, g, u, uin, f, V, mu0, mu1,
                           busy + first, n, ntx, xb, seg);
        return (int)hipGetLastError();
    };
    Comm *cm = ctx().comm;
    if (!(g.dist && cm && cm->size > 1) || nlo + nhi > nbusy) return rows(0, nbusy);
    WL_TRY((halo_begin<T>(g, const_cast<T *>(f), 3, 1)));   // (in-stream when overlap is off)
    const int rc = rows(nlo, nbusy - nlo - nhi);
    WL_TRY(halo_end());
    if (rc) return rc;
    WL_TRY(rows(0, nlo));
    return rows(nbusy - nhi, nhi);
}
// `rowfree` (optional, mom_step! only): rowfree[j + n1*k] != 0 means mu1 == 0, V == 0 and mu0 == 1 on x-row (j,k), so
// the statement reduces to u (+)= f -- same value, 15 coefficient reads and 6 neighbour reads per cell skipped.
// exchange_f (z-slab runs): the 1-plane halo exchange of f that mu_ddn needs is issued HERE on the comm stream; the pass
// over the body-free rows (which reads no neighbour of f) runs while it is in flight, the busy rows after it.
template <class T, int D, int MODE>
int op_bdim2(const G &g, T *u, const T *f, const T *V, const T *mu0, const T *mu1, const unsigned char *rowfree = nullptr,
             const int *busy = nullptr, int nbusy = 0, bool exchange_f = false, const XBc<T> *xbc = nullptr, bool *xdone = nullptr,
             const unsigned char *seg = nullptr) {
    const G gg = g;
    if (!ctx().opt[3]) rowfree = nullptr;
    if (xdone) *xdone = false;
    XBc<T> xb{0, 0, (T)0};
    if (xbc && xbc->on && rowfree && busy) xb = *xbc;   // both kernels below take part, or neither
    if (exchange_f) WL_TRY((halo_begin<T>(g, const_cast<T *>(f), D, 1)));   // (in-stream when overlap is off)
    bool skip_free = false;   // the free rows were already done by the vector pass
    if constexpr (D == 3) {
        // two passes: (1) 16-B vector kernel streams u (+)= f on the body-free rows, (2) the scalar range kernel below
        // evaluates the general statement on the busy rows only (coalesced per cell; ~5 % of the rows for a sphere)
        if (rowfree && stencil7_ok<T>(g) && rowvec_fits<T>(g)) {
            using VA = VecA<T>;
            struct Dat { VA f[3], u[3]; int free; };
            auto busy_rows = [&]() -> int {   // the general statement on the compact list of busy rows (built by wl_flow_update)
                if (nbusy == 0) return 0;
                const int ntx = (g.n[0] - 2 + 63) / 64;
                const long nw = (long)nbusy * ntx;
                Prof p(WL_K_BDIM, (long)nbusy * (g.n[0] - 2));
                hipLaunchKernelGGL((k_bdim2_busy<T, MODE>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, ctx().stream, g, u, (const T *)u, f, V,
                                   mu0, mu1, busy, nbusy, ntx, xb, seg);
                return (int)hipGetLastError();
            };
            const int rc = launch_rowvec<T, 0, false>(WL_K_BDIM, g,
                [=] __device__(long o, int j, int k, const Pre &) {
                    Dat d;
                    d.free = rowfree[j + gg.n[1] * k];
                    if (d.free) {
_Pragma("unroll")
                        for (int c = 0; c < 3; ++c) {
                            d.f[c] = VA::load(f + o + (long)c * gg.sc);
                            if (MODE != 1) d.u[c] = VA::load(u + o + (long)c * gg.sc);
                        }
                    }
                    return d;
                },
                [=] __device__(long o, int i, int, int, const Dat &d, const auto &, double *, const Pre &) {
                    if (!d.free) return;
                    VA uv[3];
_Pragma("unroll")
                    for (int c = 0; c < 3; ++c) {
                        uv[c] = d.u[c];
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) {
                            const double tmp = (0.5 * 0.0 + 0.0) + (double)d.f[c].v[v];
                            if (MODE == 1) uv[c].v[v] = (T)(0.0 + tmp);
                            else { const T un = (T)((double)uv[c].v[v] + tmp); uv[c].v[v] = (MODE == 2) ? (T)((double)un * 0.5) : un; }
                        }
                    }
                    xbc_apply<T>(xb, u, o, gg.sc, i, gg.n[0], uv);
_Pragma("unroll")
                    for (int c = 0; c < 3; ++c) uv[c].store(u + o + (long)c * gg.sc);
                }, (const T *)nullptr, nullptr, nullptr);
            WL_TRY(halo_end());
            if (rc > 0) return rc;
            if (rc == 0) skip_free = true;
            if (skip_free && busy) {
                if (xdone) *xdone = xb.on != 0;
                return busy_rows();
            }
            if (xb.on) return fail(WL_E_STATE, "BDIM!: x-ghost fold without the vector pass", __FILE__, __LINE__);
        }
    }
    WL_TRY(halo_end());
    return launch_range(WL_K_BDIM, r_inside(g), [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
        if (rowfree && rowfree[j + gg.n[1] * k]) {   // wave-uniform: a wavefront never spans two rows
            if (skip_free) return;
_Pragma("unroll")
            for (int c = 0; c < D; ++c) {
                const long q = I + (long)c * gg.sc;
                const double tmp = (0.5 * 0.0 + 0.0) + (double)f[q];
                if (MODE == 1) u[q] = (T)(0.0 + tmp);
                else { const T un = (T)((double)u[q] + tmp); u[q] = (MODE == 2) ? (T)((double)un * 0.5) : un; }
            }
            return;
        }
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
            const T *fc = f + (long)c * gg.sc;
            T s = 0;
_Pragma("unroll")
            for (int jd = 0; jd < D; ++jd) {
                const T *m1 = mu1 + (long)(c + D * jd) * gg.sc;
                s += m1[I] * (fc[I + gg.s[jd]] - fc[I - gg.s[jd]]);
            }
            const long q = I + (long)c * gg.sc;
            const double tmp = (0.5 * (double)s + (double)V[q]) + (double)(T)(mu0[q] * fc[I]);
            if (MODE == 1) {
                u[q] = (T)(0.0 + tmp);
            } else {
                const T un = (T)((double)u[q] + tmp);
                u[q] = (MODE == 2) ? (T)((double)un * 0.5) : un;
            }
        }
    });
}

// body-free row flags (see op_bdim2): one wavefront scans one x-row of the 15 coefficient arrays
template <class T, int D>
__global__ __launch_bounds__(256) void k_rowflags(G g, const T *V, const T *mu0, const T *mu1, unsigned char *flags, unsigned char *seg,
                                                  bool xper) {   // seg (optional): the same test per 64-cell segment of the row
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.n[1] * (D > 2 ? g.n[2] : 1);
    if (row >= nrows) return;
    const int j = (int)(row % g.n[1]), k = (int)(row / g.n[1]);
    const long base = g.at(0, j, k);
    bool busy = false;
    const int ntx = (g.n[0] - 2 + 63) / 64;
    // interior cells only (BDIM! does not touch ghosts).  With a non-periodic x the inflow face (i=1, component x)
    // carries the zero of BC!(mu0,0) (Flow.jl:119) in EVERY row; inside mom_step! the BC!(u) that follows BDIM!
    // (Flow.jl:159,166) overwrites u_x on that plane, so its mu0 is not allowed to mark the row busy.
    for (int q = 0; q < ntx; ++q) {              // (every lane takes part in every ballot)
        const int i = 1 + lane + 64 * q;
        bool b = false;
        if (i <= g.n[0] - 2) {
            const long I = base + i;
            for (int c = 0; c < D; ++c) {
                const bool inflow_face = (i == 1 && c == 0 && !xper);
                b = b || (V[I + (long)c * g.sc] != (T)0) || (!inflow_face && mu0[I + (long)c * g.sc] != (T)1);
                for (int d = 0; d < D; ++d) b = b || (mu1[I + (long)(c + D * d) * g.sc] != (T)0);
            }
        }
        const unsigned long long anyq = __ballot(b);
        if (seg && lane == 0) seg[row * ntx + q] = anyq ? 0 : 1;
        busy = busy || b;
    }
    const unsigned long long any = __ballot(busy);
    if (lane == 0) flags[j + (long)g.n[1] * k] = any ? 0 : 1;
}
template <class T, int D>
int op_rowflags(const G &g, const T *V, const T *mu0, const T *mu1, unsigned char *flags, int permask, unsigned char *seg = nullptr) {
    const long nrows = (long)g.n[1] * (D > 2 ? g.n[2] : 1);
    Prof p(WL_K_MISC, g.cells());
    hipLaunchKernelGGL((k_rowflags<T, D>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, V, mu0, mu1, flags, seg,
                       (bool)(permask & 1));
    return (int)hipGetLastError();
}

// scale_u!  src/Flow.jl:170
template <class T, int D>
int op_scale_u(const G &g, T *u, double scale) {
    const G gg = g;
    return launch_range(WL_K_SCALE, r_inside(g), [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
_Pragma("unroll")
        for (int c = 0; c < D; ++c) {
            T *p = u + I + (long)c * gg.sc;
            *p = (T)((double)*p * scale);
        }
    });
}

// a .*= s  /  a ./= s over the whole scalar array (src/Flow.jl:139,144).  `dbl`: the scalar is Float64
// (w = 0.5 makes dt = w*dt a Float64, :138) so the operation is done in Float64 and rounded.
// The array is one contiguous span [a, a + n_last*s_last) of the caller's allocation (ghosts and, in a padded layout,
// the row padding included -- padding is never read by anything), so this is a flat 16-B-vector streaming kernel.
// Two such operations can be chained in ONE pass (`two`): the second is applied to the rounded result of the first.
// mom_step! uses it between its two projections: the predictor's `x ./= dt` (Flow.jl:144) and the corrector's
// `x .*= 0.5dt` (:139) touch nothing in between that reads x, so they are one stream over x instead of two.
struct ScaleOp { double s; bool divide, dbl; };
template <class T> __device__ __forceinline__ T scale_one(T x, const ScaleOp &o) {
    if (o.dbl) return o.divide ? (T)((double)x / o.s) : (T)((double)x * o.s);
    return o.divide ? x / (T)o.s : x * (T)o.s;
}
template <class T>
__global__ __launch_bounds__(256) void k_scale_flat(T *a, long n, long head, ScaleOp o1, ScaleOp o2, bool two, int rev) {
    constexpr int V = 16 / sizeof(T);
    auto one = [&](T x) -> T {
        const T y = scale_one<T>(x, o1);
        return two ? scale_one<T>(y, o2) : y;
    };
    const long nv = (n - head) / V;                 // whole aligned vectors after the unaligned head
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
    for (long q0 = t0; q0 < nv; q0 += nt) {
        const long q = rev ? nv - 1 - q0 : q0;            // (alternating sweep direction: see sweep_rev)
        VecA<T> v = VecA<T>::load(a + head + q * V);
#pragma unroll
        for (int e = 0; e < V; ++e) v.v[e] = one(v.v[e]);
        v.store(a + head + q * V);
    }
    if (t0 < head) a[t0] = one(a[t0]);              // unaligned head (< V elements)
    const long tail = head + nv * V + t0;           // and tail
    if (t0 < V && tail < n) a[tail] = one(a[tail]);
}
template <class T, int D>
int op_scale_all(const G &g, T *a, double s, bool divide, bool dbl, const ScaleOp *then = nullptr) {
    constexpr int V = 16 / sizeof(T);
    const long n = span(g);
    const long mis = (long)((reinterpret_cast<uintptr_t>(a) & 15) / sizeof(T));
    long head = mis ? V - mis : 0;
    if (head > n) head = n;
    const long nv = (n - head) / V;
    long nb = (nv + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    Prof p(WL_K_SCALE, r_whole(g).count());
    const ScaleOp o1{s, divide, dbl};
    hipLaunchKernelGGL((k_scale_flat<T>), dim3((unsigned)nb), dim3(256), 0, ctx().stream, a, n, head, o1, then ? *then : o1, then != nullptr, sweep_rev());
    return (int)hipGetLastError();
}

// @inside z[I] = div(I,u)  src/Flow.jl:11-17,139
// operands of a face-difference pass (div, CFL): the V cells of the three components at o and their upper neighbours
template <class T> struct FaceDat { VecA<T> x0, y0, y1, z0, z1; T xr; };
template <class T> __device__ __forceinline__ FaceDat<T> face_load(const G &g, const T *u, long o, int j, int k) {
    constexpr int V = VecA<T>::V;
    FaceDat<T> d;
    const int i = (int)(o - g.s[1] * j - g.s[2] * k);
    d.x0 = VecA<T>::load(u + o);
    d.xr = (T)0;
    if ((threadIdx.x & 63) == 63 || i + V > g.n[0] - 2) d.xr = u[o + V];        // the cell beyond this lane's vector
    d.y0 = VecA<T>::load(u + o + g.sc); d.y1 = VecA<T>::load(u + o + g.sc + g.s[1]);
    d.z0 = VecA<T>::load(u + o + 2 * g.sc); d.z1 = VecA<T>::load(u + o + 2 * g.sc + g.s[2]);
    return d;
}
// upper x neighbour of element v of the lane's vector: next element, next lane's first element, or the loaded cell
template <class T> __device__ __forceinline__ void face_xup(const G &g, const FaceDat<T> &d, int i, T (&xu)[VecA<T>::V]) {
    constexpr int V = VecA<T>::V;
    T nxt = lane_dn1(d.x0.v[0]);
    if ((threadIdx.x & 63) == 63 || i + V > g.n[0] - 2) nxt = d.xr;
#pragma unroll
    for (int v = 0; v < V; ++v) xu[v] = (v == V - 1) ? nxt : d.x0.v[v == V - 1 ? v : v + 1];
}
template <class T, int D>
int op_div(const G &g, T *z, const T *u, int klo = 0, int khi = -1) {   // [klo,khi]: optional local plane sub-range
    const G gg = g;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(g) && ctx().opt[5]) {   // 16-B vector form (same sums in the same order)
            using VA = VecA<T>;
            const int rc = launch_rowvec<T, 0, false>(WL_K_DIV, g,
                [=] __device__(long o, int j, int k, const Pre &) { return face_load<T>(gg, u, o, j, k); },
                [=] __device__(long o, int i, int, int, const FaceDat<T> &d, const auto &, double *, const Pre &) {
                    T xu[VA::V];
                    face_xup<T>(gg, d, i, xu);
                    VA out;
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) {
                        T s = 0;
                        s += xu[v] - d.x0.v[v];
                        s += d.y1.v[v] - d.y0.v[v];
                        s += d.z1.v[v] - d.z0.v[v];
                        out.v[v] = s;
                    }
                    out.store(z + o);
                }, (const T *)nullptr, nullptr, nullptr, Gate(), klo, khi);
            if (rc >= 0) return rc;
        }
    }
    Range R = r_inside(g);
    if (khi >= klo) { R.lo[2] = klo > R.lo[2] ? klo : R.lo[2]; R.hi[2] = khi < R.hi[2] ? khi : R.hi[2]; }
    return launch_range(WL_K_DIV, R, [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
        T s = 0;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) s += u[I + gg.s[d] + (long)d * gg.sc] - u[I + (long)d * gg.sc];
        z[I] = s;
    });
}

// 16-B vectorised form of the velocity correction for D=3 (same layout requirements as wl_stencil7.h).
// rowc (optional): row constants of L (wl_stencil7.h): in a coefficient-uniform row L is not loaded.
template <class T>
__global__ __launch_bounds__(64 * S7_BY) void k_correct3(G g, T *__restrict__ u, const T *__restrict__ L, const T *__restrict__ x,
                                                  const T *__restrict__ rowc, int ntx, int tpp, int nblk, int clen, int klo, int khi,
                                                  XBc<T> xb, int rev) {
    constexpr int V = Vec16<T>::V;
    using VA = VecA<T>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    int lb, pslot;
    tile_of(b, nblk, rev, lb, pslot);
    (void)pslot;
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int i = 1 + (pt % ntx) * 64 * V + lane * V, j = 1 + (pt / ntx) * S7_BY + wv;
    const int k0 = klo + ch * clen, k1 = min(khi + 1, k0 + clen);
    if (i > g.n[0] - 2 || j > g.n[1] - 2 || k0 >= k1) return;
    const long sy = g.s[1], sz = g.s[2], sc = g.sc;
    const long col = (long)i + sy * (long)j;
    const int ju = __builtin_amdgcn_readfirstlane(j);
    const long n1 = g.n[1];
    // software pipeline: the operands of plane k+1 are requested before plane k is computed and stored
    struct Pl { VA xc, xy, u0, u1, u2; T left; };
    auto request = [&](int k) {
        const long o = col + sz * k;
        Pl p;
        p.xc = VA::load(x + o); p.xy = VA::load(x + o - sy);
        p.left = (T)0;
        if (lane == 0) p.left = x[o - 1];
        p.u0 = VA::load(u + o); p.u1 = VA::load(u + o + sc); p.u2 = VA::load(u + o + 2 * sc);
        return p;
    };
    VA xm = VA::load(x + col + sz * (k0 - 1));
    // two operand sets alternate (loop unrolled by two): a set is never copied while its loads are outstanding
    auto step = [&](int k, const Pl &cur, const RowK<T> &rc, Pl &nxt, RowK<T> &rcn) {
        const long o = col + sz * k;
        const int kn = min(k + 1, k1 - 1);
        nxt = request(kn);
        rcn = rowk_load<T>(rowc, ju + n1 * kn);
        T left = lane_up1(cur.xc.v[V - 1]);
        if (lane == 0) left = cur.left;
        auto plane = [&](auto FAST) {   // two copies: in a coefficient-uniform row nothing can load L (see RowKU)
            constexpr bool F = decltype(FAST)::value;
            VA l0, l1, l2;
            if (F) {   // the lower faces of the row are all c, except the x-boundary face of cell 1
                l0 = VA::splat(rc.c); l1 = l0; l2 = l0;
                if (i == 1) l0.v[0] = rc.lxf;
            } else {
                l0 = VA::load(L + o); l1 = VA::load(L + o + sc); l2 = VA::load(L + o + 2 * sc);
            }
            VA uv[3] = {cur.u0, cur.u1, cur.u2};
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T xl = (v == 0) ? left : cur.xc.v[v == 0 ? 0 : v - 1];
                uv[0].v[v] -= l0.v[v] * (cur.xc.v[v] - xl);
                uv[1].v[v] -= l1.v[v] * (cur.xc.v[v] - cur.xy.v[v]);
                uv[2].v[v] -= l2.v[v] * (cur.xc.v[v] - xm.v[v]);
            }
            xbc_apply<T>(xb, u, o, sc, i, g.n[0], uv);   // the row's x-ghost cells of the BC! that follows (see XBc)
            uv[0].store(u + o); uv[1].store(u + o + sc); uv[2].store(u + o + 2 * sc);
        };
        if (rc.uni()) plane(std::true_type{}); else plane(std::false_type{});
        xm = cur.xc;
    };
    Pl A = request(k0), B;
    RowK<T> rA = rowk_load<T>(rowc, ju + n1 * k0), rB = rowk_none<T>();
    for (int k = k0; k < k1; k += 2) {
        step(k, A, rA, B, rB);
        if (k + 1 < k1) step(k + 1, B, rB, A, rA);
    }
}

// u[I,i] -= L[I,i]*d_i x  src/Flow.jl:141-143 (three loops fused: they touch disjoint components)
template <class T, int D>
int op_correct(const G &g, T *u, const T *L, const T *x, const T *rowc = nullptr, const XBc<T> *xbc = nullptr, bool *xdone = nullptr) {
    if (xdone) *xdone = false;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(g)) {
            constexpr int V = Vec16<T>::V;
            const Range R = r_inside(g);
            if (R.count() <= 0) return 0;
            const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + S7_BY - 1) / S7_BY;
            const int tpp = ((ntx * nty + 7) / 8) * 8;
            const int nown = R.hi[2] - R.lo[2] + 1;
            int clen, nchunk;
            chunking(tpp, nown, 0, 16, &clen, &nchunk);   // 16 K workgroups, like the streaming kernels (4 K: 5 % slower at 512^3)
            Prof p(WL_K_CORRECT, R.count());
            const XBc<T> xb = (xbc && xbc->on) ? *xbc : XBc<T>{0, 0, (T)0};
            hipLaunchKernelGGL((k_correct3<T>), dim3(tpp * nchunk), dim3(64 * S7_BY), 0, ctx().stream, g, u, L, x, rowc, ntx, tpp,
                               tpp * nchunk, clen, R.lo[2], R.hi[2], xb, sweep_rev());
            if (xdone) *xdone = xb.on != 0;
            return (int)hipGetLastError();
        }
    }
    const G gg = g;
    return launch_range(WL_K_CORRECT, r_inside(g), [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
        const T xc = x[I];
_Pragma("unroll")
        for (int d = 0; d < D; ++d) {
            const long q = I + (long)d * gg.sc;
            u[q] -= L[q] * (xc - x[I - gg.s[d]]);
        }
    });
}

// CFL  src/Flow.jl:172-182: sigma = flux_out (Float64 via max(0.,.)), dt = min(10, inv(max(sigma)+5nu)).
// The max runs over the whole array like the reference's maximum(a.sigma): the interior pass below plus the ghost shell,
// where sigma holds the flux scratch of the last conv_diff! (sigma doubles as Phi; op_sigma_ghosts).
// exchange_u (z-slab runs, mom_step!): the end-of-step 2-plane halo exchange of u (it feeds this kernel's u[I+dz] on the last
// owned plane and the next step's conv_diff!) is issued here on the comm stream; all owned planes but the last are
// processed while it is in flight.
template <class T, int D>
int op_cfl(const G &g, T *sigma, const T *u, double nu_, double *partials, State *st, bool exchange_u = false) {
    const G gg = g;
    int np = 0;
    auto body = [=] __device__(int i, int j, int k, double(&acc)[1]) {
        const long I = gg.at(i, j, k);
        double s = 0;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) {
            const double a = (double)u[I + gg.s[d] + (long)d * gg.sc], b = -(double)u[I + (long)d * gg.sc];
            s += (a > 0 ? a : 0.0) + (b > 0 ? b : 0.0);
        }
        const T sg = (T)s;
        sigma[I] = sg;
        acc[0] = (double)sg > acc[0] ? (double)sg : acc[0];
    };
    const Range R = r_inside(g);
    if (exchange_u && D == 3 && g.dist && overlap_on() && R.hi[2] - R.lo[2] + 1 >= 2) {
        WL_TRY((halo_begin<T>(g, const_cast<T *>(u), D, 2)));
        Range Ra = R, Rb = R;
        Ra.hi[2] = R.hi[2] - 1;
        Rb.lo[2] = R.hi[2];
        int n1 = 0, n2 = 0;
        const int rc = launch_range_red<1>(WL_K_CFL, Ra, body, partials, RED_MAX, -1e300, &n1);
        WL_TRY(halo_end());
        if (rc) return rc;
        WL_TRY((launch_range_red<1>(WL_K_CFL, Rb, body, partials + n1, RED_MAX, -1e300, &n2)));
        np = n1 + n2;
    } else {
        if (exchange_u) WL_TRY((halo_exchange<T>(g, const_cast<T *>(u), D, 2)));
        int rcv = -1;
        if constexpr (D == 3) {
            if (stencil7_ok<T>(g) && ctx().opt[5]) {   // 16-B vector form: same sums in the same order, max over the same cells
                using VA = VecA<T>;
                auto ld = [=] __device__(long o, int j, int k, const Pre &) { return face_load<T>(gg, u, o, j, k); };
                auto st2 = [=] __device__(long o, int i, int, int, const FaceDat<T> &d, const auto &, double *acc, const Pre &) {
                    T xu[VA::V];
                    face_xup<T>(gg, d, i, xu);
                    VA out;
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) {
                        double s = 0;
                        { const double a = (double)xu[v], b = -(double)d.x0.v[v]; s += (a > 0 ? a : 0.0) + (b > 0 ? b : 0.0); }
                        { const double a = (double)d.y1.v[v], b = -(double)d.y0.v[v]; s += (a > 0 ? a : 0.0) + (b > 0 ? b : 0.0); }
                        { const double a = (double)d.z1.v[v], b = -(double)d.z0.v[v]; s += (a > 0 ? a : 0.0) + (b > 0 ? b : 0.0); }
                        const T sg = (T)s;
                        out.v[v] = sg;
                        acc[0] = (double)sg > acc[0] ? (double)sg : acc[0];
                    }
                    out.store(sigma + o);
                };
                rcv = launch_rowvec<T, 1, false, decltype(ld), decltype(st2), RED_MAX>(WL_K_CFL, g, ld, st2, (const T *)nullptr, partials, &np);
                if (rcv > 0) return rcv;
            }
        }
        if (rcv != 0) WL_TRY((launch_range_red<1>(WL_K_CFL, R, body, partials, RED_MAX, -1e300, &np)));
    }
    {   // maximum(a.σ) runs over the WHOLE array (Flow.jl:174): σ's ghost cells hold conv_diff!'s flux scratch (op_sigma_ghosts)
        int nps = 0;
        const T *sg = sigma;
        WL_TRY((launch_shell_red(WL_K_CFL, g, 63, [=] __device__(int i, int j, int k, double(&acc)[1]) {
            const double v = (double)sg[gg.at(i, j, k)];
            acc[0] = v > acc[0] ? v : acc[0];
        }, partials + np, RED_MAX, -1e300, &nps)));
        np += nps;
    }
    const T nu5 = (T)5 * (T)nu_;
    return launch_finalize<1>(g.dist, partials, np, RED_MAX, -1e300, st->red, [=] __device__(const double *v) {
        const T d = (T)1 / ((T)v[0] + nu5);
        st->out[0] = (double)(d < (T)10 ? d : (T)10);
    });
}

// ------------------------------------------------------------------------------------------ Poisson.jl
// set_diag!  src/Poisson.jl:42-54 (the two @inside loops fused; same values)
// rows (optional, D == 3): per x-row flags (j + n1*k); only flagged rows are recomputed (update! after a measure! that
// rewrote a known set of rows of L)
template <class T, int D>
int op_set_diag(const G &g, T *Dg, T *iD, const T *L, const unsigned char *rows = nullptr) {
    const G gg = g;
    const T eps2 = (T)2 * Lim<T>::eps;
    return launch_range(WL_K_SETDIAG, r_inside(g), [=] __device__(int i, int j, int k) {
        if (rows && !rows[j + gg.n[1] * k]) return;   // wave-uniform: a wavefront never spans two rows
        const long I = gg.at(i, j, k);
        T s = 0;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) s -= (L[I + (long)d * gg.sc] + L[I + gg.s[d] + (long)d * gg.sc]);
        Dg[I] = s;
        iD[I] = (s * s < eps2) ? (T)0 : (T)1 / s;
    });
}

// mult!(p,x)  src/Poisson.jl:62-68
template <class T, int D>
int op_mult(const LevelT<T> &p, T *x, int permask) {
    WL_TRY((op_bc_per<T, D>(p.g, x, permask)));
    WL_HIP(hipMemsetAsync(p.z, 0, (size_t)span(p.g) * sizeof(T), ctx().stream));
    const LevelT<T> q = p;
    return launch_range(WL_K_PCG_MULT, r_inside(p.g), [=] __device__(int i, int j, int k) {
        const long I = q.g.at(i, j, k);
        q.z[I] = mult1<T, D>(q.g, q.L, q.D, x, I);
    });
}

// residual!  src/Poisson.jl:91-97
// epilogue of the residual kernel with the right-hand side evaluated on the fly: z[I] = div(I,u) (Flow.jl:139, project!)
// is not read from p.z but formed from the six face values -- same differences, same order as op_div -- so the pass
// that writes z and the read of z both disappear (512^3: div 0.42 ms + residual 0.43 ms -> one kernel).  p.z is left
// unwritten: nothing reads it before pcg! overwrites it (it is pcg!'s scratch for A*eps).
template <class T> struct ResidualDivEpi {
    static constexpr bool HAS_LD = true;
    using VA = VecA<T>;
    struct Dat { VA x0, y0, y1, z1; T xr; };   // the face values of plane k except the lower z face, which is carried
    using Carry = VA;                          // u_z of plane k (the upper z face of plane k-1's cells: loaded once)
    G gu; const T *u; const T *iD; T *r; int n0;
    __device__ __forceinline__ Dat ld(long o, int i, int, int) const {
        Dat d;
        d.x0 = VA::load(u + o);
        d.xr = (T)0;
        if ((threadIdx.x & 63) == 63 || i + VA::V > n0 - 2) d.xr = u[o + VA::V];        // the cell beyond this lane's vector
        d.y0 = VA::load(u + o + gu.sc); d.y1 = VA::load(u + o + gu.sc + gu.s[1]);
        d.z1 = VA::load(u + o + 2 * gu.sc + gu.s[2]);
        return d;
    }
    __device__ __forceinline__ Carry first(long o, int, int) const { return VA::load(u + o + 2 * gu.sc); }
    __device__ __forceinline__ Carry next(const Dat &d) const { return d.z1; }
    template <class RKT>
    __device__ __forceinline__ void operator()(long o, int i, int, int, const VA &ax, const VA &, const Dat &d, const Carry &z0, const RKT &rk,
                                               double *acc, const Pre &) const {
        const VA id = row_iD<T>(rk, iD, o, i, n0);
        T nxt = lane_dn1(d.x0.v[0]);
        if ((threadIdx.x & 63) == 63 || i + VA::V > n0 - 2) nxt = d.xr;
        VA rv;
_Pragma("unroll")
        for (int v = 0; v < VA::V; ++v) {
            const T xu = (v == VA::V - 1) ? nxt : d.x0.v[v == VA::V - 1 ? v : v + 1];
            T s = 0;
            s += xu - d.x0.v[v];
            s += d.y1.v[v] - d.y0.v[v];
            s += d.z1.v[v] - z0.v[v];
            rv.v[v] = (id.v[v] == 0) ? (T)0 : s - ax.v[v];
            acc[0] += (double)rv.v[v];
        }
        rv.store(r + o);
    }
};
// divu / gu (optional): take z = div(u) from the velocity field u laid out like the level (see above).  On a z-slab the kernel is
// split like every 7-point launch (inner planes while x's halo planes travel, then the two boundary planes, whose upper one
// also reads u's halo plane); begun: the caller has already started that exchange (flow_project sends u and x in one batch).
template <class T, int D>
int op_residual(const LevelT<T> &p, int permask, double *partials, State *st, const T *divu = nullptr, const G *gu = nullptr,
                bool begun = false) {
    WL_TRY((op_bc_per<T, D>(p.g, p.x, permask, false)));
    const LevelT<T> q = p;
    int np = 0;
    int rcv = -1;
    bool exchanged = false;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(p.g)) {
            using VA = VecA<T>;
            exchanged = true;
            if (divu) {
                if (gu->s[1] != p.g.s[1] || gu->s[2] != p.g.s[2] || gu->n[0] != p.g.n[0] || gu->n[2] != p.g.n[2])
                    return fail(WL_E_ARG, "residual with div(u): layouts differ", __FILE__, __LINE__);
                rcv = launch_stencil7_halo<T, 1>(WL_K_RESIDUAL, p.g, p.x, SrcArray<T>{p.x}, p.L, p.rowc, (const T *)nullptr, (const T *)nullptr,
                                                 ResidualDivEpi<T>{*gu, divu, q.iD, q.r, q.g.n[0]}, partials, &np, Gate(), begun);
                if (rcv != 0) return rcv > 0 ? rcv : fail(WL_E_STATE, "residual with div(u): launch rejected", __FILE__, __LINE__);
            } else {
            rcv = launch_stencil7_halo<T, 1>(WL_K_RESIDUAL, p.g, p.x, SrcArray<T>{p.x}, p.L, p.rowc, (const T *)nullptr, p.z,
                [=] __device__(long o, int i, int, int, const VA &ax, const VA &, const VA &, const VA &zz, const auto &rk, double *acc, const Pre &) {
                const VA id = row_iD<T>(rk, q.iD, o, i, q.g.n[0]);
                VA rv;
_Pragma("unroll")
                for (int v = 0; v < VA::V; ++v) {
                    rv.v[v] = (id.v[v] == 0) ? (T)0 : zz.v[v] - ax.v[v];
                    acc[0] += (double)rv.v[v];
                }
                rv.store(q.r + o);
            }, partials, &np);
            if (rcv > 0) return rcv;
            }
        }
    }
    if (divu && rcv != 0) return fail(WL_E_STATE, "residual with div(u): needs the 3-D vector kernels", __FILE__, __LINE__);
    if (!exchanged) WL_TRY((halo_exchange<T>(p.g, p.x, 1, 1)));
    if (rcv != 0)
    WL_TRY((launch_range_red<1>(WL_K_RESIDUAL, r_inside(p.g), [=] __device__(int i, int j, int k, double(&acc)[1]) {
        const long I = q.g.at(i, j, k);
        const T v = (q.iD[I] == 0) ? (T)0 : q.z[I] - mult1r<T, D>(q.g, q.L, q.x, I);
        q.r[I] = v;
        acc[0] += (double)v;
    }, partials, RED_SUM, 0.0, &np)));
    const T cnt = (T)p.g.interior_cells();
    const T eps2 = (T)2 * Lim<T>::eps;
    WL_TRY((launch_finalize<1>(p.g.dist, partials, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
        const T s = (T)v[0] / cnt;
        st->shift = (double)s;
        st->do_shift = !((s < 0 ? -s : s) <= eps2);
    })));
    // (the shift pass is left at once by every workgroup unless the finalize above asked for it)
    return launch_range_if(WL_K_RESIDUAL, r_inside(p.g), &st->do_shift, [=] __device__(int i, int j, int k) {
        const long I = q.g.at(i, j, k);
        q.r[I] = q.r[I] - (T)st->shift;
    });
}

// increment!  src/Poisson.jl:99-103
template <class T, int D>
int op_increment(const LevelT<T> &p, int permask) {
    WL_TRY((op_bc_per<T, D>(p.g, p.eps, permask, false)));
    const LevelT<T> q = p;
    bool exchanged = false;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(p.g)) {
            using VA = VecA<T>;
            exchanged = true;
            const int rcv = launch_stencil7_halo<T, 0>(WL_K_INCREMENT, p.g, p.eps, SrcArray<T>{p.eps}, p.L, p.rowc, p.r, p.x,
                [=] __device__(long o, int, int, int, const VA &ae, const VA &ec, const VA &r0, const VA &x0, const auto &, double *, const Pre &) {
                VA rv = r0, xv = x0;
_Pragma("unroll")
                for (int v = 0; v < VA::V; ++v) { rv.v[v] = rv.v[v] - ae.v[v]; xv.v[v] = xv.v[v] + ec.v[v]; }
                rv.store(q.r + o);
                xv.store(q.x + o);
            }, nullptr, nullptr);
            if (rcv >= 0) return rcv;
        }
    }
    if (!exchanged) WL_TRY((halo_exchange<T>(p.g, p.eps, 1, 1)));
    return launch_range(WL_K_INCREMENT, r_inside(p.g), [=] __device__(int i, int j, int k) {
        const long I = q.g.at(i, j, k);
        q.r[I] = q.r[I] - mult1r<T, D>(q.g, q.L, q.eps, I);
        q.x[I] = q.x[I] + q.eps[I];
    });
}

// Jacobi!  src/Poisson.jl:110-113
template <class T, int D>
int op_jacobi(const LevelT<T> &p, int it, int permask) {
    const LevelT<T> q = p;
    for (int n = 0; n < it; ++n) {
        WL_TRY(launch_range(WL_K_JACOBI, r_inside(p.g), [=] __device__(int i, int j, int k) {
            const long I = q.g.at(i, j, k);
            q.eps[I] = q.r[I] * q.iD[I];
        }));
        WL_TRY((op_increment<T, D>(p, permask)));
    }
    return 0;
}

// ---- fused V-cycle smoothers (non-periodic grids; used by Vcycle! only, the stand-alone operators stay as above)
// Jacobi!(it=1) = [eps = r*iD ; r -= A eps ; x += eps]  (src/Poisson.jl:110-113,99-103) in ONE pass: eps is
// evaluated on the fly at the 7 stencil points (ghost eps is 0 because ghost r and iD are 0), the new residual
// goes OUT OF PLACE into `rout` (the eps buffer, whose content is dead inside Vcycle!), so neighbours still
// read the old r: no race, eps is never written.  Same per-cell operations => same bits as the two-pass form.
template <class T, int D>
int op_smooth_fused(const LevelT<T> &p, T *rout) {
    // z-slab runs: eps at halo cells = r*iD of the neighbour rank, so r is exchanged (overlapped with the inner planes)
    const LevelT<T> q = p;
    bool exchanged = false;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(p.g)) {
            using VA = VecA<T>;
            exchanged = true;
            const int rcv = launch_stencil7_halo<T, 0>(WL_K_SMOOTH, p.g, p.r, SrcJacobi<T>{p.r, p.iD, p.g.n[0]}, p.L, p.rowc,
                p.r, p.x, [=] __device__(long o, int, int, int, const VA &ae, const VA &ec, const VA &r0, const VA &x0, const auto &, double *, const Pre &) {
                    VA rv = r0, xv = x0;
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) { rv.v[v] = rv.v[v] - ae.v[v]; xv.v[v] = xv.v[v] + ec.v[v]; }
                    rv.store(rout + o);
                    xv.store(q.x + o);
                }, nullptr, nullptr);
            if (rcv >= 0) return rcv;
        }
    }
    if (!exchanged) WL_TRY((halo_exchange<T>(p.g, p.r, 1, 1)));
    return launch_range(WL_K_SMOOTH, r_inside(p.g), [=] __device__(int i, int j, int k) {
        const long I = q.g.at(i, j, k);
        T lo[D], hi[D];
_Pragma("unroll")
        for (int d = 0; d < D; ++d) { lo[d] = q.L[I + (long)d * q.g.sc]; hi[d] = q.L[I + q.g.s[d] + (long)d * q.g.sc]; }
        T dg = 0;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) dg -= (lo[d] + hi[d]);
        const T e0 = q.r[I] * q.iD[I];
        T s = e0 * dg;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) {
            const long sd = q.g.s[d];
            s += (q.r[I - sd] * q.iD[I - sd]) * lo[d] + (q.r[I + sd] * q.iD[I + sd]) * hi[d];
        }
        rout[I] = q.r[I] - s;
        q.x[I] = q.x[I] + e0;
    });
}
// prolongate!(fine.eps, coarse.x) ; increment!(fine)  (src/MultiLevelPoisson.jl:80-81) in ONE pass: eps at a
// fine cell is the coarse x of its parent (down(I), :2), 0 on ghost cells; residual read from `rin`, written to
// p.r (the pair of fused kernels moves r: R -> E -> R, so the buffers end up in their own roles).
// pcg_np (optional, solver! only): the pcg! call that follows on this level starts with eps = r*iD, rho = r.eps
// (Poisson.jl:125-126) -- one more pass over the r this kernel has just computed.  When pcg_np is given (and the level
// takes the vector kernel) that start is done HERE: eps (into p.eps, which is `rin`'s buffer: each thread overwrites
// only the element it has already read) and the partials of rho (into `partials`); *pcg_np = their count, else -1.
template <class T, int D>
int op_prolong_increment_fused(const LevelT<T> &p, const T *rin, const G &gc, const T *cx, double *partials = nullptr,
                               int *pcg_np = nullptr) {
    WL_TRY((halo_exchange<T>(gc, const_cast<T *>(cx), 1, 1)));   // z-slab coarse level: parents in the halo plane
    const LevelT<T> q = p;
    const G C = gc;
    if (pcg_np) *pcg_np = -1;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(p.g)) {
            using VA = VecA<T>;
            const SrcProlong<T> src{cx, C, p.g.n[0], p.g.n[1], p.g.nzg, p.g.kz0};
            if (pcg_np && partials && ctx().opt[10] && ctx().opt[13] && ctx().opt[5]) {
                T *e0 = p.eps;
                const int tpp = (((p.g.n[0] - 2 + 64 * VA::V - 1) / (64 * VA::V)) * ((p.g.n[1] - 2 + 3) / 4) + 7) / 8 * 8;
                Gate gcap;
                if (tpp <= WL_PCG_PARTIALS) gcap.zcap = WL_PCG_PARTIALS / tpp;   // few partials: pcg!'s first mult kernel may sum them itself
                int np = 0;
                const int rcv = launch_stencil7<T, 1>(WL_K_PROLONG, p.g, src, p.L, p.rowc, rin, p.x,
                    [=] __device__(long o, int i, int, int, const VA &ae, const VA &ec, const VA &r0, const VA &x0, const auto &rk, double *acc, const Pre &) {
                        VA rv = r0, xv = x0;
                        const VA id = row_iD<T>(rk, q.iD, o, i, q.g.n[0]);
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) { rv.v[v] = rv.v[v] - ae.v[v]; xv.v[v] = xv.v[v] + ec.v[v]; }
                        rv.store(q.r + o);
                        xv.store(q.x + o);
                        VA ev;
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) { ev.v[v] = rv.v[v] * id.v[v]; acc[0] += (double)rv.v[v] * (double)ev.v[v]; }
                        ev.store(e0 + o);
                    }, partials, &np, gcap);
                if (rcv > 0) return rcv;
                if (rcv == 0) { *pcg_np = np; return 0; }
            }
            const int rcv = launch_stencil7<T, 0>(WL_K_PROLONG, p.g, src, p.L, p.rowc, rin, p.x,
                [=] __device__(long o, int, int, int, const VA &ae, const VA &ec, const VA &r0, const VA &x0, const auto &, double *, const Pre &) {
                    VA rv = r0, xv = x0;
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) { rv.v[v] = rv.v[v] - ae.v[v]; xv.v[v] = xv.v[v] + ec.v[v]; }
                    rv.store(q.r + o);
                    xv.store(q.x + o);
                }, nullptr, nullptr);
            if (rcv >= 0) return rcv;
        }
    }
    return launch_range(WL_K_PROLONG, r_inside(p.g), [=] __device__(int i, int j, int k) {
        const long I = q.g.at(i, j, k);
        const int gi[3] = {i, j, D > 2 ? q.g.kg(k) : 0};          // global fine index
        const int ng[3] = {q.g.n[0], q.g.n[1], q.g.nzg};
        auto epsat = [&](int d, int off) -> T {                    // eps at the neighbour `off` along d
            int f[3] = {gi[0], gi[1], gi[2]};
            f[d] += off;
            if (f[d] < 1 || f[d] > ng[d] - 2) return (T)0;         // ghost cell of the undecomposed array
            return cx[C.at((f[0] + 1) / 2, (f[1] + 1) / 2, D > 2 ? (f[2] + 1) / 2 - C.kz0 : 0)];
        };
        T lo[D], hi[D];
_Pragma("unroll")
        for (int d = 0; d < D; ++d) { lo[d] = q.L[I + (long)d * q.g.sc]; hi[d] = q.L[I + q.g.s[d] + (long)d * q.g.sc]; }
        T dg = 0;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) dg -= (lo[d] + hi[d]);
        const T e0 = epsat(0, 0);
        T s = e0 * dg;
_Pragma("unroll")
        for (int d = 0; d < D; ++d) s += epsat(d, -1) * lo[d] + epsat(d, +1) * hi[d];
        q.r[I] = rin[I] - s;
        q.x[I] = q.x[I] + e0;
    });
}

// pcg!  src/Poisson.jl:123-143 with device-resident rho/alpha/beta and the four early exits turned into a
// device flag: once `active` drops, the remaining (already enqueued) kernels are no-ops, so no host sync.
// Fusions: [mult + z.eps], [x,r update + z=r*iD + r.z], [direction]; identical per-cell arithmetic.
// want_r2: the caller (solver!) needs L2(p) = r.r right after this call; it is accumulated by the last update kernel.
// Deferred x (wl_set_option(8), default on): in iterations 1..it-1 the update x += alpha*eps (:133) moves from the
// update kernel into the direction kernel that follows it (which streams eps anyway): one array pass less per
// iteration (10T instead of 11T for update+direction).  Same per-cell expression, alpha unchanged in between; the :138
// exit leaves st->xpend so that the direction kernel still applies the owed x update and nothing else.
// (A variant that folded the direction update into the next mult kernel -- eps_new evaluated on the fly at the 7 stencil
// points, one array pass less on paper -- was measured slower and removed: see DESIGN.md "measured dead ends".)
//
// The function has two independent halves.  WHICH KERNELS run (16-B vector or scalar range kernels; z' stored or
// recomputed; z = A*eps stored or formed twice) is decided by the switches below.  HOW A DOT PRODUCT IS FINISHED and
// pcg!'s scalar logic applied (Poisson.jl:127,131-132,135,137-139) is one of three forms, PcgDots:
//   FINALIZE   a one-workgroup launch after every reducing kernel sums its partials and updates the State; the kernels gate
//              on st->active (every level when the layout rules out the vector kernels, and the levels above 2^25 cells);
//   IN_KERNEL  no such launches: the NEXT kernel sums the producer's <= 1024 partials itself in every workgroup and applies the
//              scalar logic (Gate kind 1..3), the state hops between two slots; one finalize at the end publishes it
//              (single rank, levels of <= 2^25 cells: -2.8 % per step at 256^3 and below);
//   SLAB       z-slab levels: the producer's partials are reduced and summed over the ranks (mailbox or ncclAllReduce) into ONE
//              value, which the consuming kernel's gate treats like a single partial.
enum class PcgForm { FINALIZE, IN_KERNEL, SLAB };
template <class T> struct PcgDots {
    PcgForm form;
    State *st;
    double *partials;
    bool dist, xdef, want_r2;
    T eps10;
    int f32, zcap;
    double *P0, *PA, *PB;   // partials of rho (init) / z.eps (mult) / r.z' or r.r (update); one buffer in the FINALIZE form
    int cur = 0;            // the slot of st->slots holding the current state (IN_KERNEL, SLAB)
    PcgDots(PcgForm f, State *st_, double *partials_, bool dist_, bool xdef_, bool want_r2_, T eps10_, int zcap_)
        : form(f), st(st_), partials(partials_), dist(dist_), xdef(xdef_), want_r2(want_r2_), eps10(eps10_), f32(sizeof(T) == 4), zcap(zcap_) {
        const bool own = f != PcgForm::FINALIZE;
        P0 = partials; PA = own ? partials + WL_MAXB : partials; PB = own ? partials + 2 * WL_MAXB : partials;
    }
    Gate base() const { Gate g; g.eps10 = (double)eps10; g.f32 = f32; g.zcap = zcap; return g; }
    // SLAB: the producer's partials become the one value summed over the ranks
    int ready(const double *&part, int &n) const {
        if (form != PcgForm::SLAB) return 0;
        Prof pr(WL_K_SCALAR, 0);
        WL_TRY((reduce_allreduce<1>(part, n, (int)RED_SUM, 0.0, st->red)));
        part = st->red;
        n = 1;
        return 0;
    }
    // ---- after the init kernel (rho = r.z, :126-127)
    int after_init(int np) const {
        if (form != PcgForm::FINALIZE) return 0;
        State *s = st;
        const T e10 = eps10;
        return launch_finalize<1>(dist, partials, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
            const T rho = (T)v[0];
            s->rho = (double)rho;
            s->nupd = 0;
            s->r2_valid = 0;
            s->xpend = 0;
            s->active = !((rho < 0 ? -rho : rho) < e10);
        });
    }
    // ---- the gate of mult number n (it finishes rho when n == 1), then what follows the kernel (alpha, :131-132)
    int gate_mult(int n, int np0, Gate &g) {
        g = base();
        if (form == PcgForm::FINALIZE) { g.active = &st->active; return 0; }
        if (n == 1) {
            const double *gp = P0; int gn = np0;
            WL_TRY(ready(gp, gn));
            g.kind = 1; g.part = gp; g.np = gn; g.out = &st->slots[0]; cur = 0;
        } else { g.kind = 4; g.in = &st->slots[cur]; }
        return 0;
    }
    int after_mult(int np) const {
        if (form != PcgForm::FINALIZE) return 0;
        State *s = st;
        return launch_finalize<1>(dist, partials, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
            s->xpend = 0;   // any owed x update was applied by the direction kernel before this mult
            if (!s->active) return;
            const T alpha = (T)s->rho / (T)v[0];
            const double aa = (double)(alpha < 0 ? -alpha : alpha);
            s->alpha = (double)alpha;
            if (aa < 1e-2 || aa > 1e2) s->active = 0;  // :132
        });
    }
    // ---- the update kernel's gate (it finishes z.eps), then what follows it (rho2 / beta, :135,137-139; r.r after the last)
    int gate_update(int npA, Gate &g) {
        g = base();
        if (form == PcgForm::FINALIZE) { g.active = &st->active; g.s0 = &st->alpha; return 0; }
        const double *gp = PA; int gn = npA;
        WL_TRY(ready(gp, gn));
        g.kind = 2; g.part = gp; g.np = gn; g.in = &st->slots[cur]; g.out = &st->slots[cur ^ 1];
        return 0;
    }
    void launched_update() { if (form != PcgForm::FINALIZE) cur ^= 1; }
    int after_update(int np, bool last, bool xnow) const {
        State *s = st;
        const bool wr2 = want_r2;
        if (form != PcgForm::FINALIZE) {
            if (!last) return 0;
            const int cs = cur;   // the one finalize of the call: finishes r.r (:135) and publishes the state for the host / L2
            return launch_finalize<1>(dist, PB, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
                PcgS sl = s->slots[cs];
                if (sl.active) {
                    sl.nupd += 1;
                    if (wr2) { sl.r2 = (double)(T)v[0]; sl.r2_valid = 1; }
                    sl.active = 0;
                }
                s->rho = sl.rho; s->alpha = sl.alpha; s->beta = sl.beta;
                s->active = sl.active; s->xpend = sl.xpend; s->nupd = sl.nupd;
                s->r2_valid = sl.r2_valid;
                if (sl.r2_valid) s->r2 = sl.r2;
            });
        }
        const T e10 = eps10;
        return launch_finalize<1>(dist, partials, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
            if (!s->active) return;
            s->nupd += 1;
            if (last) {  // :135
                if (wr2) { s->r2 = (double)(T)v[0]; s->r2_valid = 1; }
                s->active = 0;
                return;
            }
            const T rho2 = (T)v[0];
            if ((rho2 < 0 ? -rho2 : rho2) < e10) { s->active = 0; s->xpend = !xnow; return; }  // :138
            s->beta = (double)(rho2 / (T)s->rho);
            s->rho = (double)rho2;
        });
    }
    // ---- the direction kernel's gate (it finishes r.z')
    int gate_direction(int npB, Gate &g) {
        g = base();
        if (form == PcgForm::FINALIZE) {
            g.active = &st->active; g.also = xdef ? &st->xpend : nullptr; g.s0 = &st->alpha; g.s1 = &st->beta;
            return 0;
        }
        const double *gp = PB; int gn = npB;
        WL_TRY(ready(gp, gn));
        g.kind = 3; g.part = gp; g.np = gn; g.in = &st->slots[cur]; g.out = &st->slots[cur ^ 1]; g.also_x = 1;
        return 0;
    }
    void launched_direction() { if (form != PcgForm::FINALIZE) cur ^= 1; }
};

template <class T, int D>
int op_pcg(const LevelT<T> &p, int it, int permask, double *partials, State *st, bool want_r2 = false,
           int pre_np = -1,     // pre_np >= 0: eps = r*iD and the partials of rho are already there (op_prolong_increment_fused)
           bool ghost_z = false) {   // z's ghost cells may hold non-zero values (level 1: z is flow.sigma, see op_sigma_ghosts)
    const LevelT<T> q = p;
    const Range R = r_inside(p.g);
    using VA = VecA<T>;
    const int n0 = p.g.n[0];
    // ---- which kernels
    // streaming pcg kernels in 16-B vector form where the layout allows (wl_set_option(5,0) = scalar range kernels)
    bool vec = false;
    if constexpr (D == 3) vec = ctx().opt[5] != 0 && stencil7_ok<T>(p.g);
    const bool xdef = ctx().opt[8] != 0;
    // z' = r*iD (:136) is not stored (wl_set_option(13), default on): the direction kernel recomputes it from r and iD
    // (iD is a row constant away from the body), one array write + one read less per iteration; z keeps A*eps.
    const bool zrec = ctx().opt[13] != 0;
    // z = A*eps is not stored either (wl_set_option(19); 3-D vector kernels): the update kernel is a second 7-point kernel
    // over eps that forms the same A*eps again (same expression, same operands => same bits) and applies
    // r -= alpha*(A*eps) in its epilogue.  Per iteration one array write (mult) and one array read (update) are replaced
    // by a second read of eps (with its halo rows and planes).  Measured: 512^3 mult 0.303 -> 0.224 ms but update
    // 0.344 -> 0.449 ms (the 7-point form of the update runs at 4.1 TB/s): no gain; 256^3 as the finest level: -2.5 % per
    // step; levels of 128^3 cells and below are latency-bound and lose 5-10 % of a pcg! call to the heavier update kernel
    // (tools/midlevels.py 19=3,0: 194 / 175, 110 / 101, 77 / 72 us).  Default (1): levels of 2^22 .. 2^26 cells; 3 = every
    // level below 2^26; 2 = every level; 0 = never.
    bool zst = false;
    if constexpr (D == 3)
        zst = (ctx().opt[19] == 2 || (ctx().opt[19] == 3 && R.count() < (1L << 26)) || (ctx().opt[19] == 1 && R.count() < (1L << 26) && R.count() >= (1L << 22))) &&
              ctx().opt[5] != 0 && zrec && stencil7_ok<T>(p.g);
    // ---- how the dot products are finished (PcgDots above).  The in-kernel sums want few partials: the kernels of such a
    // call cut z into at most `zcap` chunks (Gate::zcap); levels above 2^25 cells keep the finalize launches (option 15 = 1;
    // 2 = never): there the cap costs the streaming kernels more than the launches it saves (512^3: +0.8 %).
    const int tpp_v = D == 3 ? (((p.g.n[0] - 2 + 64 * VA::V - 1) / (64 * VA::V)) * ((p.g.n[1] - 2 + 3) / 4) + 7) / 8 * 8 : 0;
    const bool distr = p.g.dist && ctx().comm && ctx().comm->size > 1;
    const bool gates = vec && xdef && zrec && R.count() > 0 && tpp_v > 0 &&
                       (distr ? ctx().opt[15] != 0
                              : ((ctx().opt[15] == 2 || (ctx().opt[15] == 1 && R.count() <= (1L << 25))) && tpp_v <= WL_PCG_PARTIALS));
    const PcgForm form = !gates ? PcgForm::FINALIZE : (distr ? PcgForm::SLAB : PcgForm::IN_KERNEL);
    const int zcap = form == PcgForm::IN_KERNEL ? std::max(WL_PCG_PARTIALS / std::max(tpp_v, 1), 1) : 0;
    PcgDots<T> dots(form, st, partials, p.g.dist, xdef, want_r2, (T)10 * Lim<T>::eps, zcap);
    const bool own = form != PcgForm::FINALIZE;   // the vector kernels must take the launch: the gates live in them
    int np = 0, np0 = 0, npA = 0;

    // ---- :125-127  eps = z = r*iD ; rho = r.z
    int rv0 = -1;
    if (pre_np >= 0) { rv0 = 0; np = np0 = pre_np; }
    else if (vec) {
        Gate gate_init = dots.base();
        rv0 = launch_rowvec<T, 1, true>(WL_K_PCG_INIT, p.g,
            [=] __device__(long o, int, int, const Pre &) { return VA::load(q.r + o); },
            [=] __device__(long o, int i, int, int, const VA &rr, const auto &rk, double *acc, const Pre &) {
                const VA id = row_iD<T>(rk, q.iD, o, i, n0);
                VA zv;
_Pragma("unroll")
                for (int v = 0; v < VA::V; ++v) { zv.v[v] = rr.v[v] * id.v[v]; acc[0] += (double)rr.v[v] * (double)zv.v[v]; }
                if (!zrec) zv.store(q.z + o);
                zv.store(q.eps + o);
            }, p.rowc, dots.P0, &np, gate_init);
        if (rv0 > 0) return rv0;
        np0 = np;
    }
    if (own && rv0 != 0) return fail(WL_E_STATE, "pcg: vector init kernel rejected", __FILE__, __LINE__);
    if (rv0 != 0)
    WL_TRY((launch_range_red<1>(WL_K_PCG_INIT, R, [=] __device__(int i, int j, int k, double(&acc)[1]) {
        const long I = q.g.at(i, j, k);
        const T v = q.r[I] * q.iD[I];
        if (!zrec) q.z[I] = v;
        q.eps[I] = v;
        acc[0] += (double)q.r[I] * (double)v;
    }, partials, RED_SUM, 0.0, &np)));
    WL_TRY(dots.after_init(np));

    for (int n = 1; n <= it; ++n) {
        const bool last = (n == it);
        const bool xnow = last || !xdef;   // x += alpha*eps in the update kernel (else in the direction kernel)
        // ---- :129-131  perBC!(eps) ; z = A eps ; z.eps
        WL_TRY((op_bc_per<T, D>(p.g, p.eps, permask, false)));  // (x/y copies; the z-slab halo exchange follows)
        bool exchanged = false;
        int rcv = -1;
        if constexpr (D == 3) {
            if (stencil7_ok<T>(p.g)) {
                exchanged = true;
                Gate gate_mult;
                WL_TRY(dots.gate_mult(n, np0, gate_mult));
                rcv = launch_stencil7_halo<T, 1>(WL_K_PCG_MULT, p.g, p.eps, SrcArray<T>{p.eps}, p.L, p.rowc, (const T *)nullptr, (const T *)nullptr,
                    [=] __device__(long o, int, int, int, const VA &ae, const VA &ec, const VA &, const VA &, const auto &, double *acc, const Pre &) {
                    if (!zst) ae.store(q.z + o);
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) acc[0] += (double)ae.v[v] * (double)ec.v[v];
                }, dots.PA, &np, gate_mult);
                if (rcv > 0) return rcv;
                if (own && rcv != 0) return fail(WL_E_STATE, "pcg: vector mult kernel rejected", __FILE__, __LINE__);
            }
        }
        if (!exchanged) WL_TRY((halo_exchange<T>(p.g, p.eps, 1, 1)));
        if (rcv != 0)
        WL_TRY((launch_range_red<1>(WL_K_PCG_MULT, R, [=] __device__(int i, int j, int k, double(&acc)[1]) {
            if (!st->active) return;
            const long I = q.g.at(i, j, k);
            const T v = mult1r<T, D>(q.g, q.L, q.eps, I);
            q.z[I] = v;
            acc[0] += (double)v * (double)q.eps[I];
        }, partials, RED_SUM, 0.0, &np)));
        if (ghost_z && permask != 0) {
            // z⋅ϵ is a whole-array dot (Poisson.jl:131).  ϵ's ghost cells are zero except in the planes perBC! fills (:129), and
            // there z may be non-zero: on level 1 z is flow.σ, whose top ghost cells keep conv_diff!'s flux scratch.  Surface sum
            // over the periodic planes, appended to the partials of the interior sum.
            int planes = 0, nps = 0;
            for (int j = 0; j < D; ++j) if ((permask >> j) & 1) planes |= 3 << (2 * j);
            const G gg = p.g;
            WL_TRY((launch_shell_red(WL_K_PCG_MULT, p.g, planes, [=] __device__(int i, int j, int k, double(&acc)[1]) {
                const long I = gg.at(i, j, k);
                acc[0] += (double)q.z[I] * (double)q.eps[I];
            }, (rcv == 0 ? dots.PA : partials) + np, RED_SUM, 0.0, &nps, own ? 32 : 256)));
            np += nps;
        }
        npA = np;
        WL_TRY(dots.after_mult(np));

        // ---- :133-137  x += alpha eps (now or deferred) ; r -= alpha z ; z' = r*iD ; r.z'  (last iteration: r.r)
        int rvu = -1;
        Gate gate_upd;
        WL_TRY(dots.gate_update(npA, gate_upd));
        if constexpr (D == 3) {
            if (zst) {   // 7-point kernel over eps: Ae == the z the mult kernel would have stored; a = r, b = x (when x is due)
                auto upd_epi = [=] __device__(long o, int i, int, int, const VA &ae, const VA &ec, const VA &r0, const VA &x0, const auto &rk, double *acc, const Pre &pre) {
                    const T alpha = (T)pre.s0;
                    VA rr = r0;
                    if (xnow) {
                        VA xv = x0;
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) xv.v[v] += alpha * ec.v[v];
                        xv.store(q.x + o);
                    }
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) rr.v[v] = rr.v[v] - alpha * ae.v[v];
                    rr.store(q.r + o);
                    if (!last) {
                        const VA id = row_iD<T>(rk, q.iD, o, i, n0);
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) { const T zn = rr.v[v] * id.v[v]; acc[0] += (double)rr.v[v] * (double)zn; }
                    } else if (want_r2) {
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) acc[0] += (double)rr.v[v] * (double)rr.v[v];
                    }
                };
                const T *xin = xnow ? (const T *)q.x : (const T *)nullptr;
                rvu = launch_stencil7<T, 1>(WL_K_PCG_UPDATE, p.g, SrcArray<T>{p.eps}, p.L, p.rowc, (const T *)q.r, xin, upd_epi, dots.PB, &np, gate_upd);
                if (rvu != 0) return rvu > 0 ? rvu : fail(WL_E_STATE, "pcg: 7-point update kernel rejected", __FILE__, __LINE__);
                dots.launched_update();
            }
        }
        if (vec && rvu != 0) {
            struct UD { VA r, z, x, e; };
            rvu = launch_rowvec<T, 1, true>(WL_K_PCG_UPDATE, p.g,
                [=] __device__(long o, int, int, const Pre &) {
                    UD d;
                    d.r = VA::load(q.r + o);
                    d.z = VA::load(q.z + o);
                    if (xnow) { d.x = VA::load(q.x + o); d.e = VA::load(q.eps + o); }
                    return d;
                },
                [=] __device__(long o, int i, int, int, const UD &d, const auto &rk, double *acc, const Pre &pre) {
                    const T alpha = (T)pre.s0;
                    VA rr = d.r;
                    if (xnow) {
                        VA xv = d.x;
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) xv.v[v] += alpha * d.e.v[v];
                        xv.store(q.x + o);
                    }
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) rr.v[v] = rr.v[v] - alpha * d.z.v[v];
                    rr.store(q.r + o);
                    if (!last) {
                        const VA id = row_iD<T>(rk, q.iD, o, i, n0);
                        VA zn;
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) { zn.v[v] = rr.v[v] * id.v[v]; acc[0] += (double)rr.v[v] * (double)zn.v[v]; }
                        if (!zrec) zn.store(q.z + o);   // (else) the direction kernel recomputes r*iD: z' is never stored
                    } else if (want_r2) {
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) acc[0] += (double)rr.v[v] * (double)rr.v[v];
                    }
                }, p.rowc, dots.PB, &np, gate_upd);
            if (rvu > 0) return rvu;
            if (own && rvu != 0) return fail(WL_E_STATE, "pcg: vector update kernel rejected", __FILE__, __LINE__);
            if (rvu == 0) dots.launched_update();
        }
        if (rvu != 0)
        WL_TRY((launch_range_red<1>(WL_K_PCG_UPDATE, R, [=] __device__(int i, int j, int k, double(&acc)[1]) {
            if (!st->active) return;
            const long I = q.g.at(i, j, k);
            const T alpha = (T)st->alpha;
            if (xnow) q.x[I] += alpha * q.eps[I];
            const T rn = q.r[I] - alpha * q.z[I];
            q.r[I] = rn;
            if (!last) {
                const T zn = rn * q.iD[I];
                if (!zrec) q.z[I] = zn;
                acc[0] += (double)rn * (double)zn;
            } else if (want_r2) {
                acc[0] += (double)rn * (double)rn;
            }
        }, partials, RED_SUM, 0.0, &np)));
        WL_TRY(dots.after_update(np, last, xnow));
        if (last) break;

        // ---- :140  eps = beta eps + z'  (+ the deferred x += alpha eps)
        int rvd = -1;
        Gate gate_dir;
        WL_TRY(dots.gate_direction(np, gate_dir));
        if (vec) {
            struct DD { VA e, x, z; };
            rvd = launch_rowvec<T, 0, true>(WL_K_PCG_DIR, p.g,
                // gate: runs when pcg is active, or (deferred x) when only the x update of the :138 exit is owed
                [=] __device__(long o, int, int, const Pre &pre) {
                    DD d;
                    d.e = VA::load(q.eps + o);
                    if (xdef) d.x = VA::load(q.x + o);
                    if (pre.act) d.z = VA::load((zrec ? q.r : q.z) + o);
                    return d;
                },
                [=] __device__(long o, int i, int, int, const DD &d, const auto &rk, double *, const Pre &pre) {
                    VA ev = d.e, zv = VA::splat((T)0);
                    if (pre.act) {
                        zv = d.z;
                        if (zrec) {   // z' = r*iD (:136) recomputed
                            const VA id = row_iD<T>(rk, q.iD, o, i, n0);
_Pragma("unroll")
                            for (int v = 0; v < VA::V; ++v) zv.v[v] = d.z.v[v] * id.v[v];
                        }
                    }
                    if (xdef) {   // :133, deferred from the update kernel
                        const T alpha = (T)pre.s0;
                        VA xv = d.x;
_Pragma("unroll")
                        for (int v = 0; v < VA::V; ++v) xv.v[v] += alpha * ev.v[v];
                        xv.store(q.x + o);
                    }
                    if (!pre.act) return;
                    const T beta = (T)pre.s1;
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) ev.v[v] = beta * ev.v[v] + zv.v[v];
                    ev.store(q.eps + o);
                }, p.rowc, nullptr, nullptr, gate_dir);
            if (rvd > 0) return rvd;
            if (own && rvd != 0) return fail(WL_E_STATE, "pcg: vector direction kernel rejected", __FILE__, __LINE__);
            if (rvd == 0) dots.launched_direction();
        }
        if (rvd != 0)
        WL_TRY(launch_range(WL_K_PCG_DIR, R, [=] __device__(int i, int j, int k) {
            const int act = st->active;
            if (!act && !(xdef && st->xpend)) return;
            const long I = q.g.at(i, j, k);
            if (xdef) {
                q.x[I] += (T)st->alpha * q.eps[I];
                if (!act) return;
            }
            q.eps[I] = (T)st->beta * q.eps[I] + (zrec ? q.r[I] * q.iD[I] : q.z[I]);
        }));
    }
    return 0;
}

// L2(p) = r.r  src/Poisson.jl:146 -> st->r2.  after_pcg: skip the work when the pcg! call just before already
// produced it (st->r2_valid, see op_pcg want_r2); the kernels are still enqueued (no host decision) but return at once.
template <class T, int D>
int op_L2(const LevelT<T> &p, double *partials, State *st, bool after_pcg = false) {
    const LevelT<T> q = p;
    int np = 0;
    int rcv = -1;
    if constexpr (D == 3) {
        if (stencil7_ok<T>(p.g) && ctx().opt[5]) {   // 16-B streaming form; when the pcg! before it already produced r.r the
            using VA = VecA<T>;                      // gate closes and every workgroup leaves at once
            Gate gate;
            if (after_pcg) { gate.active = &st->r2_valid; gate.inv = 1; }
            rcv = launch_rowvec<T, 1, false>(WL_K_DOT, p.g,
                [=] __device__(long o, int, int, const Pre &) { return VA::load(q.r + o); },
                [=] __device__(long, int, int, int, const VA &rr, const auto &, double *acc, const Pre &) {
_Pragma("unroll")
                    for (int v = 0; v < VA::V; ++v) acc[0] += (double)rr.v[v] * (double)rr.v[v];
                }, (const T *)nullptr, partials, &np, gate);
            if (rcv > 0) return rcv;
        }
    }
    if (rcv != 0)
    WL_TRY((launch_range_red<1>(WL_K_DOT, r_inside(p.g), [=] __device__(int i, int j, int k, double(&acc)[1]) {
        if (after_pcg && st->r2_valid) return;
        const double v = (double)q.r[q.g.at(i, j, k)];
        acc[0] += v * v;
    }, partials, RED_SUM, 0.0, &np)));
    return launch_finalize<1>(p.g.dist, partials, np, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
        if (after_pcg && st->r2_valid) return;
        st->r2 = (double)(T)v[0];
    });
}

// ------------------------------------------------------------------------------------------ MultiLevelPoisson.jl
// interior coarse cells whose 2^D children (up(I), MultiLevelPoisson.jl:1) are all OWNED fine cells of this rank.
// Not decomposed: exactly inside(a).  Also valid when the coarse array is replicated and the fine one a slab.
inline Range r_children(const G &ga, const G &gb) {
    Range R = r_inside(ga);
    if (ga.D == 3) {
        const int flo = (gb.zlo + gb.kz0) > 1 ? (gb.zlo + gb.kz0) : 1;
        const int fhi = (gb.zhi + gb.kz0) < gb.nzg - 2 ? (gb.zhi + gb.kz0) : gb.nzg - 2;
        const int clo = (flo + 2) / 2 - ga.kz0, chi = fhi / 2 - ga.kz0;   // 2c-1 >= flo, 2c <= fhi
        if (clo > R.lo[2]) R.lo[2] = clo;
        if (chi < R.hi[2]) R.hi[2] = chi;
    }
    return R;
}

// restrictL!  src/MultiLevelPoisson.jl:10-16,26-32
template <class T, int D>
int op_restrictL(const G &ga, T *a, const G &gb, const T *b, int permask) {
    for (int c = 0; c < D; ++c) {
        T *ac = a + (long)c * ga.sc;
        const T *bc = b + (long)c * gb.sc;
        const G A = ga, B = gb;
        WL_TRY(launch_range(WL_K_RESTRICTL, r_children(ga, gb), [=] __device__(int i, int j, int k) {
            const int cc[3] = {i, j, A.kg(k)};
            int lo[3], hi[3];
_Pragma("unroll")
            for (int d = 0; d < 3; ++d) {
                if (d >= D) { lo[d] = hi[d] = 0; }
                else { lo[d] = 2 * cc[d] - 1; hi[d] = (d == c) ? lo[d] : 2 * cc[d]; }
            }
            lo[2] -= B.kz0; hi[2] -= B.kz0;   // fine planes: global -> local
            T s = 0;
            for (int kk = lo[2]; kk <= hi[2]; ++kk)
                for (int jj = lo[1]; jj <= hi[1]; ++jj)
                    for (int ii = lo[0]; ii <= hi[0]; ++ii) s += bc[B.at(ii, jj, kk)];
            ac[A.at(i, j, k)] = (T)(0.5 * (double)s);
        }));
    }
    (void)permask;
    return 0;   // BC!(a,0) (MultiLevelPoisson.jl:31) is applied by the caller after the slab hand-over, see coarse_L_finish
}
// second half of restrictL!: (multi-GPU hand-over of the owned planes), then BC!(aL, 0) (:31), then halos
template <class T, int D>
int coarse_L_finish(const G &ga, T *a, const G &gb, int permask) {
    Comm *cm = ctx().comm;
    if (cm && cm->size > 1 && gb.dist && !ga.dist) {
        // fine level decomposed, coarse level replicated: all-gather the planes each rank restricted
        const int nzl = (gb.nzg - 2) / cm->size / 2;   // coarse planes per rank
        for (int c = 0; c < D; ++c) {
            int rc = cm->allgather(a + (long)c * ga.sc + ga.s[2], (size_t)nzl * ga.s[2] * sizeof(T));
            if (rc) return rc;
        }
    }
    const double zero[3] = {0, 0, 0};
    WL_TRY((op_bc_vec<T, D>(ga, a, zero, 0, permask)));
    return halo_exchange<T>(ga, a, D, 1);
}

// restrict!  src/MultiLevelPoisson.jl:3-9,33 : coarse = SUM of the 2^D children (x fastest)
// zero_x (optional, Vcycle!): the coarse solution array, zeroed for the same cells (`fill!(coarse.x, 0)`,
// MultiLevelPoisson.jl:75: its ghost cells are never written on a non-periodic, undecomposed level, so they stay zero)
template <class T, int D>
int op_restrict(const G &ga, T *a, const G &gb, const T *b, T *zero_x = nullptr) {
    const G A = ga, B = gb;
    return launch_range(WL_K_RESTRICT, r_children(ga, gb), [=] __device__(int i, int j, int k) {
        if (zero_x) zero_x[A.at(i, j, k)] = (T)0;
        T s = 0;
        const int k0 = D > 2 ? 2 * A.kg(k) - 1 - B.kz0 : 0, k1 = D > 2 ? 2 * A.kg(k) - B.kz0 : 0;
        for (int kk = k0; kk <= k1; ++kk)
            for (int jj = 2 * j - 1; jj <= 2 * j; ++jj)
                for (int ii = 2 * i - 1; ii <= 2 * i; ++ii) s += b[B.at(ii, jj, kk)];
        a[A.at(i, j, k)] = s;
    });
}

// prolongate!  src/MultiLevelPoisson.jl:2,34 : fine[I] = coarse[down(I)]
template <class T, int D>
int op_prolongate(const G &ga, T *a, const G &gb, const T *b) {
    const G A = ga, B = gb;
    return launch_range(WL_K_PROLONG, r_inside(ga), [=] __device__(int i, int j, int k) {
        a[A.at(i, j, k)] = b[B.at((i + 1) / 2, (j + 1) / 2, D > 2 ? (A.kg(k) + 1) / 2 - B.kz0 : 0)];
    });
}

}  // namespace wl

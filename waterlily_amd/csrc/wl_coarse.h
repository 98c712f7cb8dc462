// wl_coarse.h -- the bottom of the V-cycle in ONE launch.
//
// Multigrid levels with <= 4096 interior cells (16^3 and below) are pure launch latency when run as ~36 kernels per
// level and iteration.  Here a single 1024-thread workgroup executes, for levels l0..last, exactly what
// Vcycle!(ml; l=l0) followed by smooth!(levels[l0]) does (src/MultiLevelPoisson.jl:70-82, src/Poisson.jl:123-143):
//   down: Jacobi!+increment! (fused, out of place) -> restrict! -> fill!(x,0)      for l = l0 .. last-1
//   up  : pcg!(l+1) -> prolongate!+increment! (fused)                              for l = last-1 .. l0
//   then pcg!(l0)
// Phases are separated by __syncthreads() (workgroup-scope release/acquire: all data is produced and consumed by
// this one workgroup); dot products are block reductions in Float64; the pcg! scalars and early exits live in
// registers/LDS.  Per-cell arithmetic is that of the multi-launch kernels; only the summation grouping of the dot
// products differs.  Requirements: non-periodic, levels not decomposed (they are replicated in multi-GPU runs).
#pragma once
#include "wl_ops.h"

namespace wl {

constexpr int CV_THREADS = 1024, CV_MAXLEV = 8, CV_MAXCELLS = 4096;

template <class T> struct CoarseArgs {
    int nlev;
    LevelT<T> lev[CV_MAXLEV];
};

__device__ __forceinline__ double cv_block_sum(double v, double *sm) {
    v = wave_red(v, RED_SUM);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();                       // protect sm against the previous use
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CV_THREADS / 64; ++i) s += sm[i];   // fixed order, identical in every thread
    return s;
}

// interior cell number c -> offset
template <int D> __device__ __forceinline__ long cv_cell(const G &g, int c, int &i, int &j, int &k) {
    const int nx = g.n[0] - 2, ny = g.n[1] - 2;
    i = 1 + c % nx;
    j = 1 + (c / nx) % ny;
    k = D > 2 ? 1 + c / (nx * ny) : 0;
    return g.at(i, j, k);
}
template <int D> __device__ __forceinline__ int cv_ncells(const G &g) {
    return (g.n[0] - 2) * (g.n[1] - 2) * (D > 2 ? g.n[2] - 2 : 1);
}

template <class T, int D> __device__ void cv_smooth(const LevelT<T> &p) {   // r -> eps buffer (out of place), x += eps
    const int nc = cv_ncells<D>(p.g);
    for (int c = threadIdx.x; c < nc; c += CV_THREADS) {
        int i, j, k;
        const long I = cv_cell<D>(p.g, c, i, j, k);
        T lo[D], hi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { lo[d] = p.L[I + (long)d * p.g.sc]; hi[d] = p.L[I + p.g.s[d] + (long)d * p.g.sc]; }
        T dg = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) dg -= (lo[d] + hi[d]);
        const T e0 = p.r[I] * p.iD[I];
        T s = e0 * dg;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const long sd = p.g.s[d];
            s += (p.r[I - sd] * p.iD[I - sd]) * lo[d] + (p.r[I + sd] * p.iD[I + sd]) * hi[d];
        }
        p.eps[I] = p.r[I] - s;
        p.x[I] = p.x[I] + e0;
    }
    __syncthreads();
}
template <class T, int D> __device__ void cv_restrict(const LevelT<T> &c, const LevelT<T> &f, const T *fr) {
    const int ntot = c.g.n[0] * c.g.n[1] * c.g.n[2];
    for (int q = threadIdx.x; q < ntot; q += CV_THREADS) {   // fill!(coarse.x, 0) on the whole array (dense or padded)
        const int i = q % c.g.n[0], j = (q / c.g.n[0]) % c.g.n[1], k = q / (c.g.n[0] * c.g.n[1]);
        c.x[c.g.at(i, j, k)] = 0;
    }
    const int nc = cv_ncells<D>(c.g);
    for (int q = threadIdx.x; q < nc; q += CV_THREADS) {
        int i, j, k;
        const long I = cv_cell<D>(c.g, q, i, j, k);
        T s = 0;
        const int k0 = D > 2 ? 2 * k - 1 : 0, k1 = D > 2 ? 2 * k : 0;
        for (int kk = k0; kk <= k1; ++kk)
            for (int jj = 2 * j - 1; jj <= 2 * j; ++jj)
                for (int ii = 2 * i - 1; ii <= 2 * i; ++ii) s += fr[f.g.at(ii, jj, kk)];
        c.r[I] = s;
    }
    __syncthreads();
}
template <class T, int D> __device__ void cv_prolong_inc(const LevelT<T> &p, const LevelT<T> &c) {   // r' in eps buffer -> r
    const int nc = cv_ncells<D>(p.g);
    for (int q = threadIdx.x; q < nc; q += CV_THREADS) {
        int i, j, k;
        const long I = cv_cell<D>(p.g, q, i, j, k);
        const int gi[3] = {i, j, k};
        auto epsat = [&](int d, int off) -> T {
            int f[3] = {gi[0], gi[1], gi[2]};
            f[d] += off;
            if (f[d] < 1 || f[d] > p.g.n[d] - 2) return (T)0;
            return c.x[c.g.at((f[0] + 1) / 2, (f[1] + 1) / 2, D > 2 ? (f[2] + 1) / 2 : 0)];
        };
        T lo[D], hi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { lo[d] = p.L[I + (long)d * p.g.sc]; hi[d] = p.L[I + p.g.s[d] + (long)d * p.g.sc]; }
        T dg = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) dg -= (lo[d] + hi[d]);
        const T e0 = epsat(0, 0);
        T s = e0 * dg;
#pragma unroll
        for (int d = 0; d < D; ++d) s += epsat(d, -1) * lo[d] + epsat(d, +1) * hi[d];
        p.r[I] = p.eps[I] - s;
        p.x[I] = p.x[I] + e0;
    }
    __syncthreads();
}
// pcg!(p; it=6)  src/Poisson.jl:123-143
template <class T, int D> __device__ void cv_pcg(const LevelT<T> &p, double *sm) {
    const int nc = cv_ncells<D>(p.g);
    const T eps10 = (T)10 * Lim<T>::eps;
    double acc = 0;
    for (int q = threadIdx.x; q < nc; q += CV_THREADS) {
        int i, j, k;
        const long I = cv_cell<D>(p.g, q, i, j, k);
        const T v = p.r[I] * p.iD[I];
        p.z[I] = v; p.eps[I] = v;
        acc += (double)p.r[I] * (double)v;
    }
    T rho = (T)cv_block_sum(acc, sm);          // (also a barrier: eps is complete)
    if ((rho < 0 ? -rho : rho) < eps10) return;   // uniform in the workgroup
    for (int n = 1; n <= 6; ++n) {
        acc = 0;
        for (int q = threadIdx.x; q < nc; q += CV_THREADS) {
            int i, j, k;
            const long I = cv_cell<D>(p.g, q, i, j, k);
            const T v = mult1r<T, D>(p.g, p.L, p.eps, I);
            p.z[I] = v;
            acc += (double)v * (double)p.eps[I];
        }
        const T alpha = rho / (T)cv_block_sum(acc, sm);
        const double aa = (double)(alpha < 0 ? -alpha : alpha);
        if (aa < 1e-2 || aa > 1e2) return;
        const bool last = (n == 6);
        acc = 0;
        for (int q = threadIdx.x; q < nc; q += CV_THREADS) {
            int i, j, k;
            const long I = cv_cell<D>(p.g, q, i, j, k);
            p.x[I] += alpha * p.eps[I];
            const T rn = p.r[I] - alpha * p.z[I];
            p.r[I] = rn;
            if (!last) {
                const T zn = rn * p.iD[I];
                p.z[I] = zn;
                acc += (double)rn * (double)zn;
            }
        }
        if (last) { __syncthreads(); return; }
        const T rho2 = (T)cv_block_sum(acc, sm);
        if ((rho2 < 0 ? -rho2 : rho2) < eps10) return;
        const T beta = rho2 / rho;
        for (int q = threadIdx.x; q < nc; q += CV_THREADS) {
            int i, j, k;
            const long I = cv_cell<D>(p.g, q, i, j, k);
            p.eps[I] = beta * p.eps[I] + p.z[I];
        }
        rho = rho2;
        __syncthreads();
    }
}

// pcg!(p; it=6) with the level held ON CHIP (wl_set_option(31), default on): a thread keeps r, x, z, eps, iD and the face
// coefficients of its <= 4 cells in registers for the whole call; only eps -- the one operand neighbours read -- lives in
// LDS (ghost cells 0: non-periodic levels).  The phases of an iteration then wait for LDS (~0.1 us) instead of for stores
// to reach L2 and come back (~1 us each, three per iteration).  Same per-cell expressions, same order of every sum as
// cv_pcg => bit-identical x, r, z, eps (all four are written back at whichever exit is taken).
constexpr int CV_CPT = CV_MAXCELLS / CV_THREADS;   // cells per thread
constexpr int CV_LDS = 7936;                       // elements of the LDS copy of eps (with ghosts): 18^3 = 5832, 66^2 = 4356 fit
template <int D> __device__ __forceinline__ bool cv_fits_lds(const G &g) {
    return g.n[0] * g.n[1] * (D > 2 ? g.n[2] : 1) <= CV_LDS && cv_ncells<D>(g) <= CV_MAXCELLS;
}
// CPT: cells per thread the instance is built for (1: levels of <= 1024 cells, 4: up to 4096); LREG: the 2*D face coefficients of
// a cell stay in registers too (else they are re-read each iteration: read-only, cache hits -- Float64 with four cells per
// thread would spill 450 bytes per lane otherwise)
template <class T, int D, int CPT, bool LREG> __device__ void cv_pcg_onchip(const LevelT<T> &p, double *sm, T *el) {
    const G &g = p.g;
    const int nc = cv_ncells<D>(g);
    const int n0 = g.n[0], n1 = g.n[1];
    const int ls[3] = {1, n0, n0 * n1};
    const int ntot = n0 * n1 * (D > 2 ? g.n[2] : 1);
    const T eps10 = (T)10 * Lim<T>::eps;
    for (int q = threadIdx.x; q < ntot; q += CV_THREADS) el[q] = (T)0;
    __syncthreads();
    constexpr int NL = LREG ? CPT : 1;
    long I[CPT];
    int li[CPT];
    bool has[CPT];
    T r[CPT], x[CPT], z[CPT], e[CPT], id[CPT], lo[NL][D], hi[NL][D];
    double acc = 0;
#pragma unroll
    for (int m = 0; m < CPT; ++m) {
        const int c = (int)threadIdx.x + m * CV_THREADS;
        has[m] = c < nc;
        I[m] = 0; li[m] = 0;
        r[m] = x[m] = z[m] = e[m] = id[m] = (T)0;
        if (has[m]) {
            int i, j, k;
            I[m] = cv_cell<D>(g, c, i, j, k);
            li[m] = i + n0 * (j + n1 * k);
            r[m] = p.r[I[m]]; x[m] = p.x[I[m]]; id[m] = p.iD[I[m]];
            if (LREG) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    lo[LREG ? m : 0][d] = p.L[I[m] + (long)d * g.sc];
                    hi[LREG ? m : 0][d] = p.L[I[m] + g.s[d] + (long)d * g.sc];
                }
            }
            const T v = r[m] * id[m];
            z[m] = v; e[m] = v;
            el[li[m]] = v;
            acc += (double)r[m] * (double)v;
        }
    }
    T rho = (T)cv_block_sum(acc, sm);          // (also a barrier: the LDS copy of eps is complete)
    bool go = !((rho < 0 ? -rho : rho) < eps10);
    for (int n = 1; go && n <= 6; ++n) {       // (every exit of pcg! is a `break`: one write-back below)
        acc = 0;
#pragma unroll
        for (int m = 0; m < CPT; ++m)
            if (has[m]) {
                T l0[D], h0[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    l0[d] = LREG ? lo[LREG ? m : 0][d] : p.L[I[m] + (long)d * g.sc];
                    h0[d] = LREG ? hi[LREG ? m : 0][d] : p.L[I[m] + g.s[d] + (long)d * g.sc];
                }
                T dg = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) dg -= (l0[d] + h0[d]);
                T s = e[m] * dg;
#pragma unroll
                for (int d = 0; d < D; ++d) s += el[li[m] - ls[d]] * l0[d] + el[li[m] + ls[d]] * h0[d];
                z[m] = s;
                acc += (double)s * (double)e[m];
            }
        const T alpha = rho / (T)cv_block_sum(acc, sm);
        const double aa = (double)(alpha < 0 ? -alpha : alpha);
        if (aa < 1e-2 || aa > 1e2) break;
        const bool last = (n == 6);
        acc = 0;
#pragma unroll
        for (int m = 0; m < CPT; ++m)
            if (has[m]) {
                x[m] += alpha * e[m];
                const T rn = r[m] - alpha * z[m];
                r[m] = rn;
                if (!last) {
                    const T zn = rn * id[m];
                    z[m] = zn;
                    acc += (double)rn * (double)zn;
                }
            }
        if (last) break;
        const T rho2 = (T)cv_block_sum(acc, sm);
        if ((rho2 < 0 ? -rho2 : rho2) < eps10) break;
        const T beta = rho2 / rho;
#pragma unroll
        for (int m = 0; m < CPT; ++m)
            if (has[m]) {
                e[m] = beta * e[m] + z[m];
                el[li[m]] = e[m];             // (every read of the old eps happened before the two block sums above)
            }
        rho = rho2;
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < CPT; ++m)
        if (has[m]) { p.x[I[m]] = x[m]; p.r[I[m]] = r[m]; p.z[I[m]] = z[m]; p.eps[I[m]] = e[m]; }
}

template <class T, int D, bool ONCHIP>
__global__ __launch_bounds__(CV_THREADS) void k_coarse_vcycle(CoarseArgs<T> a) {
    __shared__ double sm[CV_THREADS / 64];
    __shared__ T el[ONCHIP ? CV_LDS : 1];
    auto pcg = [&](const LevelT<T> &p) {
        if (ONCHIP && cv_fits_lds<D>(p.g)) {                                     // (uniform in the workgroup)
            if (cv_ncells<D>(p.g) <= CV_THREADS) cv_pcg_onchip<T, D, 1, true>(p, sm, el);
            else cv_pcg_onchip<T, D, CV_CPT, sizeof(T) == 4>(p, sm, el);
        } else cv_pcg<T, D>(p, sm);
    };
    const int last = a.nlev - 1;
    for (int l = 0; l < last; ++l) {                 // down
        cv_smooth<T, D>(a.lev[l]);
        cv_restrict<T, D>(a.lev[l + 1], a.lev[l], a.lev[l].eps);
    }
    for (int l = last - 1; l >= 0; --l) {            // up
        pcg(a.lev[l + 1]);
        __syncthreads();
        cv_prolong_inc<T, D>(a.lev[l], a.lev[l + 1]);
    }
    pcg(a.lev[0]);
}

}  // namespace wl

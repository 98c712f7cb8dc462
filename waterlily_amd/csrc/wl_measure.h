// wl_measure.h -- measure!(flow, body; t, eps) (src/Body.jl:31-53) for PARAMETRIC bodies as hand-written kernels.
//
// The reference evaluates user closures (sdf, map) with ForwardDiff inside its fill loop (src/AutoBody.jl:115-131).
// Closures cannot cross a C ABI; a body whose sdf belongs to a closed-form family (sphere/circle, torus, plate) composed with
// an affine map xi = A(t) x + b(t) (rigid translation / rotation / scaling) can: the host passes the family id, its
// parameters and A, b, dA/dt, db/dt, A^-1 at the measured time (wl_body_desc), and the device evaluates
//     d = sdf(xi), grad_x d = A^T grad_xi sdf,  m = |grad|, d /= m, n = grad / m,  V = -A^-1 (dA/dt x + db/dt)
// exactly as AutoBody.jl:117-130 does with dual numbers.  Arbitrary closures keep the torch path (waterlily_amd.body).
//
// Precision: like the reference's closures, whose Float64 constants promote every operation (README.md:41-44,118-120:
// `radius, center = m/8, m/2-1`), everything is evaluated in Float64 from the exact half-integer positions loc(i,I)
// and rounded to T on store; the band test d^2 < (2+eps)^2 (Body.jl:35) is made on the rounded sigma = T(d).
//
// Traffic: the reference rewrites all 15 coefficient arrays (V .= 0; mu0 .= 1; mu1 .= 0, then the band) = 15T + sigma
// per cell = 8.6 GB at 512^3.  Here an x-row is REWRITTEN only if it holds a band / inside cell now or held one at the
// previous measure! ("touched" rows: ~6 % for the 512^3 sphere); every other row already contains (1, 0, 0).  Only
// sigma (1T) is written everywhere.  Three launches:
//   k_measure_rows : sigma = T(sdf(centre)) for every interior cell, per row the number of band cells and a touched flag
//   k_scan_rows    : exclusive scan of the per-row band counts (one workgroup) -> position of each row's band cells
//   k_measure_fill : the 15 arrays of the touched rows + the list of band cells, in row-major (deterministic) order
// plus k_body_nds (Metrics.jl:84-87) on that list for the force metrics.
#pragma once
#include "wl_common.h"

namespace wl {

struct LeafDev {
    int family, ident;        // wl_body_desc.family; ident: the map is the identity
    int op;                   // how this leaf joins the running composite (WL_BODY_OP_*; ignored for leaf 0)
    double p[8];
    double A[9], b[3], dA[9], db[3], Ainv[9];   // row-major 3x3
};
// Up to WL_BODY_MAXLEAF leaves combined left to right like the reference's `Bodies` (src/AutoBody.jl:57-93, and the
// AutoBody operators :22-34): union takes the smaller distance, intersection the larger, minus max(d, -d_leaf); geometry
// (gradient, map, velocity) is that of the ACTIVE leaf, its sdf negated after a minus.
struct BodyDev {
    int n;
    LeafDev leaf[WL_BODY_MAXLEAF];
};

template <int D> __device__ __forceinline__ void body_map(const LeafDev &B, const double (&x)[D], double (&xi)[D]) {
    if (B.ident) {
#pragma unroll
        for (int a = 0; a < D; ++a) xi[a] = x[a];
        return;
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
        double s = B.b[a];
#pragma unroll
        for (int c = 0; c < D; ++c) s += B.A[3 * a + c] * x[c];
        xi[a] = s;
    }
}
// sdf(xi) and its gradient with respect to xi (closed forms of what ForwardDiff.gradient returns)
template <int D> __device__ __forceinline__ double leaf_sdf(const LeafDev &B, const double (&xi)[D], double (&g)[D], bool want_grad) {
    if (B.family == WL_BODY_TORUS && D == 3) {   // norm((e1, norm((e2,e3)) - R)) - r
        const double e0 = xi[0] - B.p[0], e1 = xi[1] - B.p[1], e2 = xi[D - 1] - B.p[2];
        const double s = sqrt(e1 * e1 + e2 * e2), q = s - B.p[3];
        const double rho = sqrt(e0 * e0 + q * q);
        if (want_grad) { g[0] = e0 / rho; g[1] = (q / rho) * (e1 / s); g[D - 1] = (q / rho) * (e2 / s); }
        return rho - B.p[4];
    }
    if (B.family == WL_BODY_PLATE) {   // sqrt(sum(abs2, xi - (clamp(xi0,-a,a), 0[, 0]))) - thk   (maintests.jl:375)
        const double a = B.p[0];
        double e[D], s2 = 0;
        e[0] = xi[0] - (xi[0] < -a ? -a : (xi[0] > a ? a : xi[0]));
#pragma unroll
        for (int c = 1; c < D; ++c) e[c] = xi[c];
#pragma unroll
        for (int c = 0; c < D; ++c) s2 += e[c] * e[c];
        const double rho = sqrt(s2);
        if (want_grad) {   // d e0/d xi0 = 1 - clamp' = 0 inside the span, 1 outside (0*e0/rho keeps the NaN of rho = 0)
            g[0] = (fabs(xi[0]) > a) ? e[0] / rho : 0.0 * e[0] / rho;
#pragma unroll
            for (int c = 1; c < D; ++c) g[c] = e[c] / rho;
        }
        return rho - B.p[1];
    }
    // WL_BODY_SPHERE: sqrt(sum(abs2, xi - c)) - R;  WL_BODY_CYLINDER: the same over the axes with p[4+a] != 0 (the
    // others do not enter: an infinite cylinder / a slab-less disc along them)
    double e[D], s2 = 0;
    const bool cyl = B.family == WL_BODY_CYLINDER;
#pragma unroll
    for (int a = 0; a < D; ++a) { e[a] = (cyl && B.p[4 + a] == 0.0) ? 0.0 : xi[a] - B.p[a]; s2 += e[a] * e[a]; }
    const double rho = sqrt(s2);
    if (want_grad) {
#pragma unroll
        for (int a = 0; a < D; ++a) g[a] = e[a] / rho;
    }
    return rho - B.p[3];
}
// the composite's distance at x and which leaf is active (sdf_map_d / reduce_sdf_map, src/AutoBody.jl:73-93); sgn = -1
// when the active leaf entered through a minus
template <int D> __device__ __forceinline__ double body_sdf(const BodyDev &B, const double (&x)[D], int &act, double &sgn) {
    double xi[D], g[D];
    body_map<D>(B.leaf[0], x, xi);
    double d = leaf_sdf<D>(B.leaf[0], xi, g, false);
    act = 0; sgn = 1.0;
    for (int q = 1; q < B.n; ++q) {
        body_map<D>(B.leaf[q], x, xi);
        const double dq = leaf_sdf<D>(B.leaf[q], xi, g, false);
        const int op = B.leaf[q].op;
        if (op == WL_BODY_OP_UNION) { if (dq < d) { d = dq; act = q; sgn = 1.0; } }
        else if (op == WL_BODY_OP_MINUS) { if (-dq > d) { d = -dq; act = q; sgn = -1.0; } }
        else { if (dq > d) { d = dq; act = q; sgn = 1.0; } }
    }
    return d;
}
template <int D> __device__ __forceinline__ double body_sdf(const BodyDev &B, const double (&x)[D]) {
    int act; double sgn;
    return body_sdf<D>(B, x, act, sgn);
}
// measure(body, x, t; fastd2)  src/AutoBody.jl:115-131 (for `Bodies`: :107-110 -- measure of the active leaf's sdf and map)
template <int D>
__device__ __forceinline__ void body_measure(const BodyDev &BB, const double (&x)[D], double fastd2, double &d, double (&n)[D], double (&V)[D]) {
    double xi[D], gx[D];
#pragma unroll
    for (int a = 0; a < D; ++a) { n[a] = 0; V[a] = 0; }
    int act; double sgn;
    d = body_sdf<D>(BB, x, act, sgn);
    if (d * d > fastd2) return;                       // :118
    const LeafDev &B = BB.leaf[act];
    body_map<D>(B, x, xi);
    d = sgn * leaf_sdf<D>(B, xi, gx, true);
    double g[D];
    bool nan = false;
#pragma unroll
    for (int c = 0; c < D; ++c) {                     // chain rule: grad_x = A^T grad_xi
        double s = 0;
        if (B.ident) s = gx[c];
        else {
#pragma unroll
            for (int a = 0; a < D; ++a) s += B.A[3 * a + c] * gx[a];
        }
        s *= sgn;
        g[c] = s;
        nan = nan || (s != s);
    }
    if (nan) return;                                  // :120
    double m2 = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) m2 += g[c] * g[c];
    const double m = sqrt(m2);                        // :124
    d /= m;
#pragma unroll
    for (int c = 0; c < D; ++c) n[c] = g[c] / m;
    if (!B.ident) {                                   // :128-130  V = -J \ dot
        double dot[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double s = B.db[a];
#pragma unroll
            for (int c = 0; c < D; ++c) s += B.dA[3 * a + c] * x[c];
            dot[a] = s;
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double s = 0;
#pragma unroll
            for (int c = 0; c < D; ++c) s += B.Ainv[3 * a + c] * dot[c];
            V[a] = -s;
        }
    }
}
// Body.jl:56-61 (Float64)
__device__ __forceinline__ double wl_kern(double d) { return 0.5 + 0.5 * cos(M_PI * d); }
__device__ __forceinline__ double wl_kern0(double d) { return 0.5 + 0.5 * d + 0.5 * sin(M_PI * d) / M_PI; }
__device__ __forceinline__ double wl_kern1(double d) {
    return 0.25 * (1 - d * d) - 0.5 * (d * sin(M_PI * d) + (1 + cos(M_PI * d)) / M_PI) / M_PI;
}
__device__ __forceinline__ double wl_clamp1(double v) { return v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v); }

// cell centre loc(0,I) for 0-based indices (util.jl:160: I .- 1.5 in 1-based terms)
template <int D> __device__ __forceinline__ void cell_loc(const G &g, int i, int j, int k, double (&x)[D]) {
    x[0] = (double)i - 0.5;
    x[1] = (double)j - 0.5;
    if (D == 3) x[D - 1] = (double)(k + g.kz0) - 0.5;
}

// one wavefront per x-row (j,k): sigma, band count, touched flag.  Rows outside inside(p) get count 0 / untouched.
template <class T, int D>
__global__ __launch_bounds__(256) void k_measure_rows(G g, BodyDev B, T *sigma, T d2, int *rowcount, unsigned char *touched) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.n[1] * (D == 3 ? g.n[2] : 1);
    if (row >= nrows) return;
    const int j = (int)(row % g.n[1]), k = (int)(row / g.n[1]);
    const int kg = k + g.kz0;
    int cnt = 0;
    bool any = false;
    const bool inside_row = j >= 1 && j <= g.n[1] - 2 && (D < 3 || (kg >= 1 && kg <= g.nzg - 2));
    if (inside_row) {
        for (int i = 1 + lane; i <= g.n[0] - 2; i += 64) {
            double x[D];
            cell_loc<D>(g, i, j, k, x);
            const T d = (T)body_sdf<D>(B, x);
            sigma[g.at(i, j, k)] = d;                                  // Body.jl:34
            const bool band = d * d < d2;                              // :35 (in T)
            cnt += band ? 1 : 0;
            any = any || band || d < (T)0;
        }
    }
    const unsigned long long bm = __ballot(any);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0) { rowcount[row] = cnt; touched[row] = bm ? 1 : 0; }
}
// exclusive scan of n ints by ONE workgroup of 1024 threads: off[i] = sum_{q<i} cnt[q]; off[n] = total
__global__ __launch_bounds__(1024) void k_scan_rows(const int *cnt, long *off, long n) {
    __shared__ long part[1024];
    const long per = (n + 1023) / 1024, lo = (long)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    long s = 0;
    for (long q = lo; q < hi; ++q) s += cnt[q];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long run = 0;
        for (int q = 0; q < 1024; ++q) { const long v = part[q]; part[q] = run; run += v; }
        off[n] = run;
    }
    __syncthreads();
    long run = part[threadIdx.x];
    for (long q = lo; q < hi; ++q) { off[q] = run; run += cnt[q]; }
}
// the 15 coefficient arrays of the rows that are touched now or were at the previous measure! (`prev`; full: all rows),
// and the band-cell list `cand` (LOCAL dense column-major linear indices i + n0*(j + n1*k)) at rowoff[row] + rank in row
template <class T, int D>
__global__ __launch_bounds__(256) void k_measure_fill(G g, BodyDev B, const T *sigma, T d2, double fast2, double eps, T *mu0, T *mu1, T *V,
                                                      const unsigned char *touched, const unsigned char *prev, bool full,
                                                      const long *rowoff, long *cand) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.n[1] * (D == 3 ? g.n[2] : 1);
    if (row >= nrows) return;
    const int j = (int)(row % g.n[1]), k = (int)(row / g.n[1]);
    const int kg = k + g.kz0;
    const bool inside_row = j >= 1 && j <= g.n[1] - 2 && (D < 3 || (kg >= 1 && kg <= g.nzg - 2));
    if (!inside_row) return;
    if (!full && !touched[row] && !prev[row]) return;                  // the row already holds (1, 0, 0)
    long pos = rowoff[row];
    for (int i0 = 1; i0 <= g.n[0] - 2; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i <= g.n[0] - 2;
        const long I = g.at(ok ? i : 1, j, k);
        const T d = ok ? sigma[I] : (T)1e30;
        const bool band = ok && d * d < d2;
        const unsigned long long bm = __ballot(band);
        if (band) {
            const int rank = __popcll(bm & ((1ull << lane) - 1ull));
            cand[pos + rank] = (long)i + (long)g.n[0] * ((long)j + (long)g.n[1] * k);
        }
        pos += __popcll(bm);
        if (!ok) continue;
        if (band) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double x[D], dc, n[D], Vc[D];
                cell_loc<D>(g, i, j, k, x);
                x[c] -= 0.5;                                           // face location loc(c,I) (util.jl:160)
                body_measure<D>(B, x, fast2, dc, n, Vc);               // Body.jl:37
                const double q = wl_clamp1(dc / eps);
                V[I + (long)c * g.sc] = (T)Vc[c];                      // :38
                mu0[I + (long)c * g.sc] = (T)wl_kern0(q);              // :39
                const double k1 = eps * wl_kern1(q);
#pragma unroll
                for (int jd = 0; jd < D; ++jd) mu1[I + (long)(c + D * jd) * g.sc] = (T)(k1 * n[jd]);   // :41
            }
        } else {
            const T m = d < (T)0 ? (T)0 : (T)1;                        // :45-47
#pragma unroll
            for (int c = 0; c < D; ++c) {
                V[I + (long)c * g.sc] = (T)0;
                mu0[I + (long)c * g.sc] = m;
#pragma unroll
                for (int jd = 0; jd < D; ++jd) mu1[I + (long)(c + D * jd) * g.sc] = (T)0;
            }
        }
    }
}
// nds(body, loc(0,I), t) (Metrics.jl:84-87, Float64) for the listed cells: out[b*D + c]
template <int D>
__global__ __launch_bounds__(256) void k_body_nds(G g, BodyDev B, const long *cand, long n, double *out) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const long lin = cand[b];
    const int i = (int)(lin % g.n[0]), j = (int)((lin / g.n[0]) % g.n[1]), k = (int)(lin / ((long)g.n[0] * g.n[1]));
    double x[D], d, nn[D], V[D];
    cell_loc<D>(g, i, j, k, x);
    body_measure<D>(B, x, 1.0, d, nn, V);
    const double w = wl_kern(wl_clamp1(d));
#pragma unroll
    for (int c = 0; c < D; ++c) out[b * D + c] = nn[c] * w;
}
// body-free row flags from the touched flags (what k_rowflags finds by reading the 15 arrays after BC!(mu0,0), BC!(V,0)):
// untouched interior rows hold (mu0,mu1,V) = (1,0,0) except where BC! zeroed the boundary-normal mu0 on the first
// interior plane of a non-periodic direction (y: j == 1, z: kg == 1; x: cell i = 1, which k_rowflags exempts too)
template <int D>
__global__ __launch_bounds__(256) void k_rowflags_touched(G g, const unsigned char *touched, unsigned char *flags, int permask) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const long nrows = (long)g.n[1] * (D == 3 ? g.n[2] : 1);
    if (row >= nrows) return;
    const int j = (int)(row % g.n[1]), k = (int)(row / g.n[1]);
    const int kg = k + g.kz0;
    bool fre = !touched[row];
    if (j == 1 && !((permask >> 1) & 1)) fre = false;
    if (D == 3 && kg == 1 && !((permask >> 2) & 1)) fre = false;
    flags[row] = fre ? 1 : 0;
}

// rows whose coefficient arrays this measure! rewrote: touched now or at the previous measure! (all of them the first time).
// keep: an earlier measure!'s flags have not been consumed by update!(pois) yet (two measure! calls in a row): OR into them.
__global__ __launch_bounds__(256) static void k_rows_changed(const unsigned char *touched, const unsigned char *prev, bool all, bool keep,
                                                            unsigned char *changed, long nrows) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r < nrows) changed[r] = (all || touched[r] || prev[r] || (keep && changed[r])) ? 1 : 0;
}
// level-0 rows whose D / iD / row constants read a changed row of L: the row itself and its lower y and z neighbours.
// Periodic y / z (single device): the ghost row above the last interior row is the periodic copy of the FIRST interior row
// (BC!(mu0,0,perdir)), which never carries a flag of its own, so the last interior row also looks at the first one.
__global__ __launch_bounds__(256) static void k_rows_dirty(const unsigned char *changed, unsigned char *dirty, int n1, int n2, int yper,
                                                          int zper) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= (long)n1 * n2) return;
    const int j = (int)(r % n1), k = (int)(r / n1);
    bool d = changed[r] != 0;
    if (j + 1 < n1) d = d || changed[r + 1];
    if (k + 1 < n2) d = d || changed[r + n1];
    if (yper && j == n1 - 2) d = d || changed[1 + (long)n1 * k];
    if (zper && k == n2 - 2) d = d || changed[j + (long)n1 * 1];
    dirty[r] = d ? 1 : 0;
}

}  // namespace wl

// wl_stencil7.h -- the 7-point variable-coefficient operator  A e = D e + sum_d L[I,d] e[I-d] + L[I+d,d] e[I+d]
// (src/Poisson.jl:69-75) for D=3 as a 16-byte-vectorised z-marching kernel.
//
//   * one lane owns V = 16/sizeof(T) consecutive x cells (float4 / double2): every global access is an aligned
//     16-B vector -> 1 KiB per wave instruction, the coalescing sweet spot of gfx950;
//   * a wavefront spans 64*V cells of one row; a 256-thread workgroup = 4 rows; the workgroup marches along z
//     with a 3-deep register window of e (k-1,k,k+1) and a 2-deep window of L_z, so e and L_z are loaded ONCE
//     per cell; x neighbours come from the adjacent lane (wave shuffle) -- only the two edge lanes of a row
//     touch memory for them; y neighbours are aligned vector loads of the rows above/below (L1/L2 hits: the
//     neighbouring wave of the same workgroup streams that row);
//   * the diagonal is recomputed from the six face coefficients (same operations as set_diag!, Poisson.jl:48-54);
//   * an epilogue functor turns A e into the operator at hand: z=Ae & z.e (pcg!), r-=Ae & x+=e (increment!),
//     r = z-Ax (residual!), with per-thread Float64 partials reduced exactly like the range kernels.
// Per-cell arithmetic and its order are those of mult()/set_diag! => bit-identical to the generic kernels.
// Requirements (else the caller falls back to the generic range kernel): D==3, (n0-2) % V == 0 and every row's
// first interior element 16-B aligned (true for the padded layout of the Python host).
#pragma once
#include "wl_common.h"

namespace wl {

template <class T> struct Vec16;
template <> struct Vec16<float> { static constexpr int V = 4; using type = float4; };
template <> struct Vec16<double> { static constexpr int V = 2; using type = double2; };

template <class T> struct VecA {   // array view of a 16-B vector
    static constexpr int V = Vec16<T>::V;
    T v[V];
    __device__ __forceinline__ static VecA load(const T *p) {
        VecA r;
        const typename Vec16<T>::type q = *reinterpret_cast<const typename Vec16<T>::type *>(p);
        __builtin_memcpy(r.v, &q, 16);
        return r;
    }
    __device__ __forceinline__ void store(T *p) const {
        typename Vec16<T>::type q;
        __builtin_memcpy(&q, v, 16);
        *reinterpret_cast<typename Vec16<T>::type *>(p) = q;
    }
};

constexpr int S7_BY = 4;   // rows (= wavefronts) per workgroup of the helper kernels (k_correct3)

// ---- stencil operand sources: vec(o,i,j,k) = the V cells starting at (i,j,k) [offset o], scal = one cell
template <class T> struct SrcArray {          // e is an array (pcg!: eps, residual!: x, increment!: eps)
    const T *e;
    __device__ __forceinline__ VecA<T> vec(long o, int, int, int) const { return VecA<T>::load(e + o); }
    __device__ __forceinline__ T scal(long o, int, int, int) const { return e[o]; }
};
template <class T> struct SrcJacobi {         // e = r*iD evaluated on the fly (Jacobi!, src/Poisson.jl:111)
    const T *r, *iD;
    __device__ __forceinline__ VecA<T> vec(long o, int, int, int) const {
        const VecA<T> a = VecA<T>::load(r + o), b = VecA<T>::load(iD + o);
        VecA<T> c;
#pragma unroll
        for (int v = 0; v < VecA<T>::V; ++v) c.v[v] = a.v[v] * b.v[v];
        return c;
    }
    __device__ __forceinline__ T scal(long o, int, int, int) const { return r[o] * iD[o]; }
};
template <class T> struct SrcDirection {      // e = beta*eps + r*iD : pcg!'s new search direction on the fly (Poisson.jl:136,140)
    const T *e, *r, *iD;
    const double *beta;                        // device scalar, already rounded to T
    __device__ __forceinline__ VecA<T> vec(long o, int, int, int) const {
        const T b = (T)*beta;
        const VecA<T> ev = VecA<T>::load(e + o), rv = VecA<T>::load(r + o), dv = VecA<T>::load(iD + o);
        VecA<T> c;
#pragma unroll
        for (int v = 0; v < VecA<T>::V; ++v) c.v[v] = b * ev.v[v] + rv.v[v] * dv.v[v];
        return c;
    }
    __device__ __forceinline__ T scal(long o, int, int, int) const { return (T)*beta * e[o] + r[o] * iD[o]; }
};
template <class T> struct SrcProlong {        // e[I] = coarse x[down(I)] inside, 0 on ghosts (MultiLevelPoisson.jl:2,34)
    const T *cx;
    G C;            // coarse grid
    int n0, n1, nzg, kz0;   // fine extents (global along z) and the fine grid's kz0
    __device__ __forceinline__ T scal(long, int i, int j, int k) const {
        const int kg = k + kz0;
        if (i < 1 || i > n0 - 2 || j < 1 || j > n1 - 2 || kg < 1 || kg > nzg - 2) return (T)0;
        return cx[C.at((i + 1) / 2, (j + 1) / 2, (kg + 1) / 2 - C.kz0)];
    }
    __device__ __forceinline__ VecA<T> vec(long, int i, int j, int k) const {   // i is odd (1 + V*m), i+V-1 <= n0-2
        VecA<T> c;
        const int kg = k + kz0;
        if (j < 1 || j > n1 - 2 || kg < 1 || kg > nzg - 2) {
#pragma unroll
            for (int v = 0; v < VecA<T>::V; ++v) c.v[v] = 0;
            return c;
        }
        const T *row = cx + C.at(0, (j + 1) / 2, (kg + 1) / 2 - C.kz0);
#pragma unroll
        for (int v = 0; v < VecA<T>::V; v += 2) { const T p = row[(i + v + 1) / 2]; c.v[v] = p; c.v[v + 1] = p; }
        return c;
    }
};

template <class T, int NRED, int BY, class SRC, class EPI>
__global__ __launch_bounds__(64 * BY) void k_stencil7(G g, SRC src, const T *__restrict__ L, EPI epi,
                                                  double *partials, int ntx, int tpp, int nblk, int clen, int klo,
                                                  int khi) {
    constexpr int V = Vec16<T>::V;
    using VA = VecA<T>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int lb = (nblk & 7) ? b : (b & 7) * (nblk >> 3) + (b >> 3);   // XCD-contiguous logical id
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int nxi = g.n[0] - 2, nyi = g.n[1] - 2;
    const int i = 1 + (pt % ntx) * 64 * V + lane * V;
    const int j = 1 + (pt / ntx) * BY + wv;
    const int k0 = klo + ch * clen, k1 = min(khi + 1, k0 + clen);
    double acc[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int q = 0; q < (NRED > 0 ? NRED : 1); ++q) acc[q] = 0.0;
    const bool active = (i <= nxi) && (j <= nyi) && (k0 < k1);
    if (active) {   // (no barriers below: inactive lanes may simply skip; shuffles only pair active lanes)
        const bool first = (lane == 0), last = (lane == 63) || (i + V > nxi);
        const long sy = g.s[1], sz = g.s[2], sc = g.sc;
        const long col = (long)i + sy * (long)j;
        const T *Lx = L, *Ly = L + sc, *Lz = L + 2 * sc;
        VA em = src.vec(col + sz * (k0 - 1), i, j, k0 - 1), ec = src.vec(col + sz * k0, i, j, k0);
        VA lzc = VA::load(Lz + col + sz * k0);
        for (int k = k0; k < k1; ++k) {
            const long o = col + sz * k;
            const VA ep = src.vec(o + sz, i, j, k + 1), lzp = VA::load(Lz + o + sz);
            const VA ym = src.vec(o - sy, i, j - 1, k), yp = src.vec(o + sy, i, j + 1, k);
            const VA lx = VA::load(Lx + o), ly0 = VA::load(Ly + o), ly1 = VA::load(Ly + o + sy);
            // x neighbours of the vector ends: adjacent lane, or memory at the two ends of the row segment
            T left = __shfl_up(ec.v[V - 1], 1, 64), right = __shfl_down(ec.v[0], 1, 64), lxr = __shfl_down(lx.v[0], 1, 64);
            if (first) left = src.scal(o - 1, i - 1, j, k);
            if (last) { right = src.scal(o + V, i + V, j, k); lxr = Lx[o + V]; }
            VA ae;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T xm = (v == 0) ? left : ec.v[v == 0 ? 0 : v - 1];
                const T xp = (v == V - 1) ? right : ec.v[v == V - 1 ? v : v + 1];
                const T lxlo = lx.v[v], lxhi = (v == V - 1) ? lxr : lx.v[v == V - 1 ? v : v + 1];
                T dg = 0;
                dg -= (lxlo + lxhi);
                dg -= (ly0.v[v] + ly1.v[v]);
                dg -= (lzc.v[v] + lzp.v[v]);
                T s = ec.v[v] * dg;
                s += xm * lxlo + xp * lxhi;
                s += ym.v[v] * ly0.v[v] + yp.v[v] * ly1.v[v];
                s += em.v[v] * lzc.v[v] + ep.v[v] * lzp.v[v];
                ae.v[v] = s;
            }
            epi(o, ae, ec, acc);
            em = ec; ec = ep; lzc = lzp;
        }
    }
    if (NRED > 0) {
        block_red<(NRED > 0 ? NRED : 1), BY>(acc, RED_SUM);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NRED; ++q) partials[(long)q * gridDim.x + blockIdx.x] = acc[q];
        }
    }
}

// ---- 16-B vectorised streaming (no neighbours): f(o, j, k, acc) is called once per V-cell vector at element offset o.
// Same row mapping as k_stencil7 (lane = V cells of a row, wavefront = row segment, workgroup = 4 rows marching in z).
template <class T, int NRED, class F>
__global__ __launch_bounds__(256) void k_rowvec(G g, F f, double *partials, int ntx, int tpp, int nblk, int clen, int klo,
                                                int khi) {
    constexpr int V = Vec16<T>::V;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int lb = (nblk & 7) ? b : (b & 7) * (nblk >> 3) + (b >> 3);
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int i = 1 + (pt % ntx) * 64 * V + lane * V, j = 1 + (pt / ntx) * 4 + wv;
    const int k0 = klo + ch * clen, k1 = min(khi + 1, k0 + clen);
    double acc[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int q = 0; q < (NRED > 0 ? NRED : 1); ++q) acc[q] = 0.0;
    if (i <= g.n[0] - 2 && j <= g.n[1] - 2) {
        const long col = (long)i + g.s[1] * (long)j;
        for (int k = k0; k < k1; ++k) f(col + g.s[2] * k, j, k, acc);
    }
    if (NRED > 0) {
        block_red<(NRED > 0 ? NRED : 1), 4>(acc, RED_SUM);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NRED; ++q) partials[(long)q * gridDim.x + blockIdx.x] = acc[q];
        }
    }
}
template <class T, int NRED, class F>
inline int launch_rowvec(int kclass, const G &g, F f, double *partials, int *np) {
    constexpr int V = Vec16<T>::V;
    Range R = r_inside(g);
    if (np) *np = 0;
    if (R.count() <= 0) return 0;
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + 3) / 4;
    const int tpp = ((ntx * nty + 7) / 8) * 8;
    const int nown = R.hi[2] - R.lo[2] + 1;
    int want = WL_MAXB / tpp;
    if (want < 1) want = 1;
    if (want > nown) want = nown;
    const int clen = (nown + want - 1) / want, nchunk = (nown + clen - 1) / clen;
    const int nblk = tpp * nchunk;
    if (nblk > WL_MAXB) return -1;
    if (np) *np = nblk;
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_rowvec<T, NRED, F>), dim3(nblk), dim3(256), 0, ctx().stream, g, f, partials, ntx, tpp, nblk, clen,
                       R.lo[2], R.hi[2]);
    return (int)hipGetLastError();
}

// can the vector kernel run on this level?
template <class T> inline bool stencil7_ok(const G &g, const T *e, const T *L) {
    constexpr int V = Vec16<T>::V;
    if (!ctx().opt[0]) return false;
    if (g.D != 3 || (g.n[0] - 2) % V != 0 || g.n[0] - 2 < V) return false;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return al(e + 1) && al(L + 1) && (g.s[1] * sizeof(T)) % 16 == 0 && (g.s[2] * sizeof(T)) % 16 == 0 &&
           (g.sc * sizeof(T)) % 16 == 0;
}

// launch over the owned interior planes; *np = number of partials per reduced value (0 if nothing to do).
// BY = rows (wavefronts) per workgroup: 4 (256 threads) or 8 (512 threads; fewer y-halo rows re-read at tile edges),
// selected by wl_set_option(4, .).
template <class T, int NRED, int BY, class SRC, class EPI>
inline int launch_stencil7_by(int kclass, const G &g, SRC src, const T *L, EPI epi, double *partials, int *np) {
    constexpr int V = Vec16<T>::V;
    Range R = r_inside(g);
    if (np) *np = 0;
    if (R.count() <= 0) return 0;
    const int klo = R.lo[2], khi = R.hi[2];
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + BY - 1) / BY;
    const int tpp = ((ntx * nty + 7) / 8) * 8;
    const int nown = khi - klo + 1;
    int want = WL_MAXB / tpp;
    if (want < 1) want = 1;
    if (want > nown) want = nown;
    const int clen = (nown + want - 1) / want;
    const int nchunk = (nown + clen - 1) / clen;
    const int nblk = tpp * nchunk;
    if (nblk > WL_MAXB) return -1;   // plane too large for the partial buffer: caller falls back
    if (np) *np = nblk;
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_stencil7<T, NRED, BY, SRC, EPI>), dim3(nblk), dim3(64 * BY), 0, ctx().stream, g, src, L, epi, partials,
                       ntx, tpp, nblk, clen, klo, khi);
    return (int)hipGetLastError();
}
template <class T, int NRED, class SRC, class EPI>
inline int launch_stencil7(int kclass, const G &g, SRC src, const T *L, EPI epi, double *partials, int *np) {
    if (ctx().opt[4]) return launch_stencil7_by<T, NRED, 8>(kclass, g, src, L, epi, partials, np);
    return launch_stencil7_by<T, NRED, 4>(kclass, g, src, L, epi, partials, np);
}

}  // namespace wl

// wl_stencil7.h -- the 7-point variable-coefficient operator  A e = D e + sum_d L[I,d] e[I-d] + L[I+d,d] e[I+d]
// (src/Poisson.jl:69-75) for D=3 as a 16-byte-vectorised z-marching kernel.
//
//   * one lane owns V = 16/sizeof(T) consecutive x cells (float4 / double2): every global access is a 16-B vector
//     -> 1 KiB per wave instruction, the coalescing sweet spot of gfx950 (16-B aligned in the padded layout);
//   * a wavefront spans 64*V cells of one row; a 256-thread workgroup = 4 rows; the workgroup marches along z
//     with a 3-deep register window of e (k-1,k,k+1) and a 2-deep window of L_z, so e and L_z are loaded ONCE
//     per cell; x neighbours come from the adjacent lane (wave shuffle) -- only the two edge lanes of a row
//     touch memory for them; y neighbours are aligned vector loads of the rows above/below (L1/L2 hits: the
//     neighbouring wave of the same workgroup streams that row);
//   * the diagonal is recomputed from the six face coefficients (same operations as set_diag!, Poisson.jl:48-54);
//   * coefficient-uniform rows: away from the body (and from the domain faces) every face coefficient of a row is
//     the same number c (1 on the finest level, 2^l below: restrictL! sums four unit faces and halves).  wl_mg_update
//     records c per row (NaN = not uniform, see k_lrow); in such a row the kernel does not load L at all (3 of the 5
//     array passes of mult) and uses c -- the very values the loads would have returned, so results are unchanged;
//   * an epilogue functor epi(o, j, k, Ae, e, acc, pre) turns A e into the operator at hand: z=Ae & z.e (pcg!), r-=Ae & x+=e (increment!),
//     r = z-Ax (residual!), with per-thread Float64 partials reduced exactly like the range kernels.
// Per-cell arithmetic and its order are those of mult()/set_diag! => bit-identical to the generic kernels.
// Requirements (else the caller falls back to the generic range kernel): D==3 and (n0-2) % V == 0.
#pragma once
#include "wl_common.h"

namespace wl {

// 16-byte vectors with ELEMENT alignment: the compiler still emits one global_load/store_dwordx4 per access (gfx950
// runs with unaligned global access enabled), so rows of the reference's dense layout (pitch N+2, not 16-B aligned)
// take the same kernels as the padded layout; a misaligned wave access merely touches one more 128-B line.
typedef float wl_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double wl_d2u __attribute__((ext_vector_type(2), aligned(8)));
template <class T> struct Vec16;
template <> struct Vec16<float> { static constexpr int V = 4; using type = wl_f4u; };
template <> struct Vec16<double> { static constexpr int V = 2; using type = wl_d2u; };

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

// Device-side gate and scalars of a solver kernel, read ONCE per thread before its z loop (uniform addresses): inside
// the loop they would be re-loaded every plane (the stores in between may alias them as far as the compiler knows),
// each time draining the loads in flight.
//   kind 0: the body runs when no gate is given, when *active != 0, or when `also` is given and *also != 0; s0/s1 are
//           optional device scalars (alpha, beta) handed to the functor.
//   kind 1..4 (pcg! without the one-workgroup finalize launches): the kernel ITSELF finishes the dot product of the
//           kernel before it -- every workgroup sums that kernel's per-workgroup partials in the fixed order of k_finalize
//           (<= 1024 doubles, L2-resident) and applies pcg!'s scalar logic (src/Poisson.jl:127,131-132,137-139) to the
//           state it reads from slot `in`; the first thread of the grid stores the new state to the OTHER slot `out`
//           (nobody reads that one during this kernel), from where the next kernel picks it up.
//           1: after the init kernel (rho, :126-127)   2: after mult (alpha, :131-132)
//           3: after a non-final update (rho2, beta, :137-139)   4: read the state only
struct PcgS { double rho, alpha, beta, r2; int active, xpend, nupd, r2_valid; };
struct Gate {
    const int *active = nullptr, *also = nullptr;
    const double *s0 = nullptr, *s1 = nullptr;
    int kind = 0;
    const double *part = nullptr;   // partials of the producing kernel
    int np = 0;
    const PcgS *in = nullptr;
    PcgS *out = nullptr;
    double eps10 = 0.0;             // 10 eps(T)
    int f32 = 1;                    // the solver's element type is Float32 (scalars are rounded to it)
    int also_x = 0;                 // run also when only the deferred x update is owed (direction kernel)
};
struct Pre { int act; double s0, s1; };
// pcg!'s scalar logic, shared by the in-kernel form (gate_open) and the k_finalize epilogues of op_pcg
__device__ __forceinline__ double pcg_rnd(double x, int f32) { return f32 ? (double)(float)x : x; }
__device__ __forceinline__ double pcg_div(double a, double b, int f32) { return f32 ? (double)((float)a / (float)b) : a / b; }
__device__ __forceinline__ void pcg_after_init(PcgS &s, double v, double eps10, int f32) {
    const double rho = pcg_rnd(v, f32);
    s.rho = rho; s.alpha = 0.0; s.beta = 0.0; s.r2 = 0.0;
    s.nupd = 0; s.r2_valid = 0; s.xpend = 0;
    s.active = !((rho < 0 ? -rho : rho) < eps10);
}
__device__ __forceinline__ void pcg_after_mult(PcgS &s, double v, int f32) {
    s.xpend = 0;   // any owed x update was applied by the direction kernel before this mult
    if (!s.active) return;
    const double alpha = pcg_div(s.rho, pcg_rnd(v, f32), f32);
    const double aa = alpha < 0 ? -alpha : alpha;
    s.alpha = alpha;
    if (aa < 1e-2 || aa > 1e2) s.active = 0;   // :132
}
__device__ __forceinline__ void pcg_after_update(PcgS &s, double v, double eps10, int f32, bool xnow) {   // non-final iteration
    if (!s.active) return;
    s.nupd += 1;
    const double rho2 = pcg_rnd(v, f32);
    if ((rho2 < 0 ? -rho2 : rho2) < eps10) { s.active = 0; s.xpend = !xnow; return; }   // :138
    s.beta = pcg_div(rho2, s.rho, f32);
    s.rho = rho2;
}
// sum of np partials, same order as k_finalize; 256-thread workgroups, every thread must call; all threads get the value
__device__ __forceinline__ double block_sum_partials(const double *part, int np) {
    __shared__ double bc;
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < np; i += 256) acc[0] = acc[0] + part[i];
    block_red<1>(acc, RED_SUM);
    if (threadIdx.x == 0) bc = acc[0];
    __syncthreads();
    return bc;
}
__device__ __forceinline__ bool gate_open(const Gate &gt, Pre &pre) {
    pre.act = 1; pre.s0 = 0.0; pre.s1 = 0.0;
    bool run = true;
    if (gt.kind == 0) {
        if (gt.active) { pre.act = *gt.active; run = pre.act || (gt.also && *gt.also); }
        if (gt.s0) pre.s0 = *gt.s0;
        if (gt.s1) pre.s1 = *gt.s1;
        return run;
    }
    PcgS s;
    if (gt.kind == 1) {
        pcg_after_init(s, block_sum_partials(gt.part, gt.np), gt.eps10, gt.f32);
    } else {
        s = *gt.in;
        if (gt.kind == 2) pcg_after_mult(s, block_sum_partials(gt.part, gt.np), gt.f32);
        else if (gt.kind == 3) pcg_after_update(s, block_sum_partials(gt.part, gt.np), gt.eps10, gt.f32, false);
    }
    if (gt.out && blockIdx.x == 0 && threadIdx.x == 0) *gt.out = s;
    pre.act = s.active; pre.s0 = s.alpha; pre.s1 = s.beta;
    return s.active || (gt.also_x && s.xpend);
}

// ---- stencil operand sources: vec(o,i,j,k) = the V cells starting at (i,j,k) [offset o], scal = one cell
// init(pre): called once per thread before the z loop with the gate's scalars (only SrcDirection uses them)
template <class T> struct SrcArray {          // e is an array (pcg!: eps, residual!: x, increment!: eps)
    const T *e;
    __device__ __forceinline__ void init(const Pre &) {}
    __device__ __forceinline__ VecA<T> vec(long o, int, int, int) const { return VecA<T>::load(e + o); }
    __device__ __forceinline__ T scal(long o, int, int, int) const { return e[o]; }
};
// iD of the V cells at (i..i+V-1, j, k): the row constant where the row is coefficient-uniform (k_lrow; the two end
// cells of a row are excluded from that guarantee, so the vectors holding them are loaded), else the array.
// j and k must be wave-uniform (they are: a wavefront works on one row).
// Row constants (k_lrow): RC_N values per x-row (j,k) of a level: [c, lxf, lxl, idc, idf, idl, -, -]
//   c   = the one value of every face coefficient of the row (NaN: the row is not uniform, use the arrays),
//   lxf, lxl = Lx at the two x-boundary faces (i = 1 and n0-1), idc = iD of the cells 2..n0-3, idf/idl = iD of the end cells.
constexpr int RC_N = 8;
template <class T> struct RowC { T c, lxf, lxl, idc; };
template <class T> __device__ __forceinline__ RowC<T> load_rowc(const T *p) {
    RowC<T> r;
    if constexpr (sizeof(T) == 4) {
        const VecA<T> a = VecA<T>::load(p);
        r.c = a.v[0]; r.lxf = a.v[1]; r.lxl = a.v[2]; r.idc = a.v[3];
    } else {
        const VecA<T> a = VecA<T>::load(p), b = VecA<T>::load(p + 2);
        r.c = a.v[0]; r.lxf = a.v[1]; r.lxl = b.v[0]; r.idc = b.v[1];
    }
    return r;
}
template <class T>
__device__ __forceinline__ VecA<T> load_iD(const T *iD, const T *rowc, int n0, int n1, long o, int i, int j, int k) {
    constexpr int V = VecA<T>::V;
    if (rowc) {
        const T *rc = rowc + RC_N * ((long)__builtin_amdgcn_readfirstlane(j) + (long)n1 * __builtin_amdgcn_readfirstlane(k));
        const T c = rc[0];
        if (c == c) {
            const T idc = rc[3];
            VecA<T> b;
#pragma unroll
            for (int v = 0; v < V; ++v) b.v[v] = idc;
            if (i == 1) b.v[0] = rc[4];
            if (i + V - 1 == n0 - 2) b.v[V - 1] = rc[5];
            return b;
        }
    }
    return VecA<T>::load(iD + o);
}

template <class T> struct SrcJacobi {         // e = r*iD evaluated on the fly (Jacobi!, src/Poisson.jl:111)
    const T *r, *iD;
    const T *rowc;                             // row constants (nullptr: none)
    int n0, n1;
    __device__ __forceinline__ void init(const Pre &) {}
    __device__ __forceinline__ VecA<T> vec(long o, int i, int j, int k) const {
        const VecA<T> a = VecA<T>::load(r + o), b = load_iD<T>(iD, rowc, n0, n1, o, i, j, k);
        VecA<T> c;
#pragma unroll
        for (int v = 0; v < VecA<T>::V; ++v) c.v[v] = a.v[v] * b.v[v];
        return c;
    }
    __device__ __forceinline__ T scal(long o, int, int, int) const { return r[o] * iD[o]; }
};
template <class T> struct SrcDirection {      // e = beta*eps + r*iD : pcg!'s new search direction on the fly (Poisson.jl:136,140)
    const T *e, *r, *iD;
    const T *rowc;                             // row constants (nullptr: none)
    int n0, n1;
    T b;                                       // beta: taken from the gate's second scalar by init()
    __device__ __forceinline__ void init(const Pre &pre) { b = (T)pre.s1; }
    __device__ __forceinline__ VecA<T> vec(long o, int i, int j, int k) const {
        const VecA<T> ev = VecA<T>::load(e + o), rv = VecA<T>::load(r + o), dv = load_iD<T>(iD, rowc, n0, n1, o, i, j, k);
        VecA<T> c;
#pragma unroll
        for (int v = 0; v < VecA<T>::V; ++v) c.v[v] = b * ev.v[v] + rv.v[v] * dv.v[v];
        return c;
    }
    __device__ __forceinline__ T scal(long o, int, int, int) const { return b * e[o] + r[o] * iD[o]; }
};
template <class T> struct SrcProlong {        // e[I] = coarse x[down(I)] inside, 0 on ghosts (MultiLevelPoisson.jl:2,34)
    const T *cx;
    G C;            // coarse grid
    int n0, n1, nzg, kz0;   // fine extents (global along z) and the fine grid's kz0
    __device__ __forceinline__ void init(const Pre &) {}
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
__global__ __launch_bounds__(64 * BY) void k_stencil7(G g, SRC src, const T *__restrict__ L, const T *__restrict__ rowc, EPI epi,
                                                  double *partials, int ntx, int tpp, int nblk, int clen, int klo,
                                                  int khi, Gate gate) {
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
    Pre pre;
    const bool run = gate_open(gate, pre);
    src.init(pre);
    const bool active = run && (i <= nxi) && (j <= nyi) && (k0 < k1);
    if (active) {   // (no barriers below: inactive lanes may simply skip; shuffles only pair active lanes)
        const bool first = (lane == 0), last = (lane == 63) || (i + V > nxi);
        const long sy = g.s[1], sz = g.s[2], sc = g.sc;
        const long col = (long)i + sy * (long)j;
        const T *Lx = L, *Ly = L + sc, *Lz = L + 2 * sc;
        // register window of e: planes k-1, k, k+1; plane k+2 is requested in iteration k and consumed in k+1, so that
        // two planes of e are in flight per wavefront (in a coefficient-uniform row e is the only HBM stream)
        VA em = src.vec(col + sz * (k0 - 1), i, j, k0 - 1), ec = src.vec(col + sz * k0, i, j, k0);
        VA ep = src.vec(col + sz * (k0 + 1), i, j, k0 + 1);
        VA lzc = VA::load(Lz + col + sz * k0);
        const int ju = __builtin_amdgcn_readfirstlane(j);   // a wavefront works on ONE row: row constants are scalar loads
        const T *rcp = rowc ? rowc + RC_N * ((long)ju + (long)g.n[1] * k0) : nullptr;
        const long rcs = RC_N * (long)g.n[1];
        RowC<T> rn;
        rn.c = rn.lxf = rn.lxl = rn.idc = (T)0;
        if (rcp) rn = load_rowc<T>(rcp);
        for (int k = k0; k < k1; ++k) {
            const long o = col + sz * k;
            const RowC<T> rc = rn;
            const T c = rc.c;
            const bool uni = rcp && (c == c);
            if (rcp) { rcp += rcs; rn = load_rowc<T>(rcp); }   // next plane's row constants (plane k1 <= n2-1 exists)
            const VA ym = src.vec(o - sy, i, j - 1, k), yp = src.vec(o + sy, i, j + 1, k);
            VA lx, ly0, ly1, lzp;
            T lxr;
            if (uni) {   // all faces of this row are c, except possibly the two x-boundary faces (in the row constants)
#pragma unroll
                for (int v = 0; v < V; ++v) { lx.v[v] = c; ly0.v[v] = c; ly1.v[v] = c; lzp.v[v] = c; }
                if (i == 1) lx.v[0] = rc.lxf;
                lxr = (i + V > nxi) ? rc.lxl : c;
            } else {
                lx = VA::load(Lx + o); ly0 = VA::load(Ly + o); ly1 = VA::load(Ly + o + sy); lzp = VA::load(Lz + o + sz);
                lxr = __shfl_down(lx.v[0], 1, 64);
                if (last) lxr = Lx[o + V];
            }
            // x neighbours of the vector ends: adjacent lane, or memory at the two ends of the row segment
            T left = __shfl_up(ec.v[V - 1], 1, 64), right = __shfl_down(ec.v[0], 1, 64);
            if (first) left = src.scal(o - 1, i - 1, j, k);
            if (last) right = src.scal(o + V, i + V, j, k);
            // issued last: nothing in this iteration waits for it (plane k1 is the last one that exists for this chunk)
            const int kn = (k + 2 <= k1) ? k + 2 : k1;
            const VA en = src.vec(col + sz * kn, i, j, kn);
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
            epi(o, j, k, ae, ec, acc, pre);
            em = ec; ec = ep; ep = en; lzc = lzp;
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
                                                int khi, Gate gate) {
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
    Pre pre;
    const bool run = gate_open(gate, pre);
    if (run && i <= g.n[0] - 2 && j <= g.n[1] - 2) {
        const long col = (long)i + g.s[1] * (long)j;
        for (int k = k0; k < k1; ++k) f(col + g.s[2] * k, j, k, acc, pre);
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
inline int launch_rowvec(int kclass, const G &g, F f, double *partials, int *np, Gate gate = Gate()) {
    constexpr int V = Vec16<T>::V;
    Range R = r_inside(g);
    if (np) *np = 0;
    if (R.count() <= 0) return 0;
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + 3) / 4;
    const int tpp = ((ntx * nty + 7) / 8) * 8;
    const int nown = R.hi[2] - R.lo[2] + 1;
    int want = WL_MAXB / tpp;
    if (ctx().opt[12] > 0 && ctx().opt[12] < want) want = ctx().opt[12];
    if (want < 1) want = 1;
    if (want > nown) want = nown;
    const int clen = (nown + want - 1) / want, nchunk = (nown + clen - 1) / clen;
    const int nblk = tpp * nchunk;
    if (nblk > WL_MAXB) return -1;
    if (np) *np = nblk;
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_rowvec<T, NRED, F>), dim3(nblk), dim3(256), 0, ctx().stream, g, f, partials, ntx, tpp, nblk, clen,
                       R.lo[2], R.hi[2], gate);
    return (int)hipGetLastError();
}

// Row constants of a Poisson level (see the header comment): one wavefront scans one x-row.  c is recorded when
//   Lx[i] == c for the interior faces i = 2..n0-2 (faces 1 and n0-1 are the domain boundary, or its periodic image),
//   Ly[i,j] == Ly[i,j+1] == Lz[i,k] == Lz[i,k+1] == c for every interior i, and iD[i] == idc for i = 2..n0-3
// (the two end cells have a different diagonal when the boundary faces are not c), else NaN; layout: RC_N above.
template <class T>
__global__ __launch_bounds__(256) void k_lrow(G g, const T *__restrict__ L, const T *__restrict__ iD, T *rowc) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.n[1] * g.n[2];
    if (row >= nrows) return;
    const int j = (int)(row % g.n[1]), k = (int)(row / g.n[1]);
    const T nan = __builtin_nanf("");
    T c = nan, idc = nan;
    bool ok = (j >= 1 && j <= g.n[1] - 2 && k >= g.zlo && k <= g.zhi && k >= 1 && k <= g.n[2] - 2);
    if (ok) {
        const long base = g.at(0, j, k);
        const T *Lx = L, *Ly = L + g.sc, *Lz = L + 2 * g.sc;
        c = Ly[base + 1];
        idc = iD[base + g.n[0] / 2];
        ok = (c == c) && (idc == idc);
        for (int i = 1 + lane; i <= g.n[0] - 2; i += 64) {
            const long I = base + i;
            ok = ok && Ly[I] == c && Ly[I + g.s[1]] == c && Lz[I] == c && Lz[I + g.s[2]] == c;
            if (i >= 2) ok = ok && Lx[I] == c;
            if (i >= 2 && i <= g.n[0] - 3) ok = ok && iD[I] == idc;
        }
    }
    const unsigned long long bad = __ballot(!ok);
    if (lane == 0) {
        T *rc = rowc + RC_N * row;
        rc[0] = bad ? nan : c;
        rc[3] = idc;
        if (!bad) {
            const long base = g.at(0, j, k);
            rc[1] = L[base + 1]; rc[2] = L[base + g.n[0] - 1];
            rc[4] = iD[base + 1]; rc[5] = iD[base + g.n[0] - 2];
        } else { rc[1] = rc[2] = rc[4] = rc[5] = nan; }
        rc[6] = rc[7] = (T)0;
    }
}
template <class T> inline int op_lrow(const G &g, const T *L, const T *iD, T *rowc) {
    const long nrows = (long)g.n[1] * g.n[2];
    Prof p(WL_K_MISC, g.cells());
    hipLaunchKernelGGL((k_lrow<T>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, L, iD, rowc);
    return (int)hipGetLastError();
}

// can the vector kernel run on this level?  (e, L: the arrays involved -- kept for call-site symmetry; any element-
// aligned pointer and any strides will do)
template <class T> inline bool stencil7_ok(const G &g, const T *e, const T *L) {
    constexpr int V = Vec16<T>::V;
    (void)e; (void)L;
    if (!ctx().opt[0]) return false;
    return g.D == 3 && (g.n[0] - 2) % V == 0 && g.n[0] - 2 >= V;
}

// launch over the owned interior planes; *np = number of partials per reduced value (0 if nothing to do).
// BY = rows (wavefronts) per workgroup: 4 (256 threads) or 8 (512 threads; fewer y-halo rows re-read at tile edges),
// selected by wl_set_option(4, .).
template <class T, int NRED, int BY, class SRC, class EPI>
inline int launch_stencil7_by(int kclass, const G &g, SRC src, const T *L, const T *rowc, EPI epi, double *partials, int *np, Gate gate,
                              int kov_lo = 0, int kov_hi = -1) {
    constexpr int V = Vec16<T>::V;
    Range R = r_inside(g);
    if (np) *np = 0;
    if (kov_hi >= kov_lo) { R.lo[2] = kov_lo > R.lo[2] ? kov_lo : R.lo[2]; R.hi[2] = kov_hi < R.hi[2] ? kov_hi : R.hi[2]; }   // plane sub-range
    if (R.count() <= 0) return 0;
    const int klo = R.lo[2], khi = R.hi[2];
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + BY - 1) / BY;
    const int tpp = ((ntx * nty + 7) / 8) * 8;
    const int nown = khi - klo + 1;
    int want = WL_MAXB / tpp;
    if (ctx().opt[11] > 0 && ctx().opt[11] < want) want = ctx().opt[11];
    if (want < 1) want = 1;
    if (want > nown) want = nown;
    const int clen = (nown + want - 1) / want;
    const int nchunk = (nown + clen - 1) / clen;
    const int nblk = tpp * nchunk;
    if (nblk > WL_MAXB) return -1;   // plane too large for the partial buffer: caller falls back
    if (np) *np = nblk;
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_stencil7<T, NRED, BY, SRC, EPI>), dim3(nblk), dim3(64 * BY), 0, ctx().stream, g, src, L, rowc, epi, partials,
                       ntx, tpp, nblk, clen, klo, khi, gate);
    return (int)hipGetLastError();
}
template <class T, int NRED, class SRC, class EPI>
inline int launch_stencil7(int kclass, const G &g, SRC src, const T *L, const T *rowc, EPI epi, double *partials, int *np,
                           Gate gate = Gate(), int kov_lo = 0, int kov_hi = -1) {
    if (ctx().opt[4]) return launch_stencil7_by<T, NRED, 8>(kclass, g, src, L, rowc, epi, partials, np, gate, kov_lo, kov_hi);
    return launch_stencil7_by<T, NRED, 4>(kclass, g, src, L, rowc, epi, partials, np, gate, kov_lo, kov_hi);
}
// Launch with two epilogue operand arrays: EPI(o, j, k, Ae, e, a, b, acc, pre) receives the V values of ea and eb at o
// (e.g. r and x of increment!), loaded next to the stencil operands.  (A software-pipelined variant of the kernel --
// loads of plane k+1/k+2 issued one iteration ahead, sources split into load/arithmetic halves -- was measured at
// 512^3 and 256^3: no gain, 124-152 VGPRs; the kernels are not latency-bound.  See DESIGN.md.)
template <class T, int NRED, class SRC, class EPI>
inline int launch_stencil7ab(int kclass, const G &g, SRC src, const T *L, const T *rowc, const T *ea, const T *eb, EPI epi,
                             double *partials, int *np, Gate gate = Gate(), int kov_lo = 0, int kov_hi = -1) {
    using VA = VecA<T>;
    return launch_stencil7<T, NRED>(kclass, g, src, L, rowc,
        [=] __device__(long o, int j, int k, const VA &ae, const VA &ec, double *acc, const Pre &pre) {
            VA a = ec, b = ec;
            if (ea) a = VA::load(ea + o);
            if (eb) b = VA::load(eb + o);
            epi(o, j, k, ae, ec, a, b, acc, pre);
        }, partials, np, gate, kov_lo, kov_hi);
}

// The same launch on a z-slab level whose operand `hal` (one halo plane per side) has to be exchanged first
// (where the reference calls perBC! on the operand).  With a comm stream available (halo_begin) the planes that do
// not read a halo plane are computed WHILE the exchange is in flight; the first and last owned plane follow once it
// has landed.  Reduction partials of the three launches are laid end to end (NRED <= 1).  Not decomposed / overlap
// off / fewer than 3 planes: exchange in-stream, one launch.
template <class T, int NRED, class SRC, class EPI>
inline int launch_stencil7_halo(int kclass, const G &g, T *hal, SRC src, const T *L, const T *rowc, const T *ea, const T *eb,
                                EPI epi, double *partials, int *np, Gate gate = Gate()) {
    static_assert(NRED <= 1, "partials of the split launches are concatenated: one reduced value at most");
    const Range R = r_inside(g);
    const int lo = R.lo[2], hi = R.hi[2];
    if (!g.dist || !overlap_on() || hi - lo + 1 < 3) {
        WL_TRY((halo_exchange<T>(g, hal, 1, 1)));
        return launch_stencil7ab<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials, np, gate);
    }
    WL_TRY((halo_begin<T>(g, hal, 1, 1)));
    ctx().n_overlapped += 1;
    int n1 = 0, n2 = 0, n3 = 0;
    int rc = launch_stencil7ab<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials, &n1, gate, lo + 1, hi - 1);
    const int rce = halo_end();
    if (rc) return rc;      // (-1: not applicable -- the exchange has been waited for, the caller's fallback may run)
    if (rce) return rce;
    rc = launch_stencil7ab<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials ? partials + n1 : nullptr, &n2, gate, lo, lo);
    if (rc) return rc > 0 ? rc : fail(WL_E_STATE, "split 7-point launch: boundary plane rejected", __FILE__, __LINE__);
    rc = launch_stencil7ab<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials ? partials + n1 + n2 : nullptr, &n3, gate, hi, hi);
    if (rc) return rc > 0 ? rc : fail(WL_E_STATE, "split 7-point launch: boundary plane rejected", __FILE__, __LINE__);
    if (np) *np = n1 + n2 + n3;
    return 0;
}

}  // namespace wl

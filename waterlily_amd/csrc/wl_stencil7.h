// wl_stencil7.h -- the 7-point variable-coefficient operator  A e = D e + sum_d L[I,d] e[I-d] + L[I+d,d] e[I+d]
// (src/Poisson.jl:69-75) for D=3 as a 16-byte-vectorised, software-pipelined z-marching kernel, and the 16-byte
// streaming kernel (no neighbours) that shares its row mapping.
//
//   * one lane owns V = 16/sizeof(T) consecutive x cells (float4 / double2): every global access is a 16-B vector
//     -> 1 KiB per wave instruction, the coalescing sweet spot of gfx950 (16-B aligned in the padded layout);
//   * a wavefront spans 64*V cells of R consecutive rows (R = 1 or 2 rows per thread); a 256-thread workgroup =
//     4 wavefronts = 4R rows; the workgroup marches along z with a 3-deep register window of e per row, so e is loaded
//     ONCE per cell; x neighbours come from the adjacent lane (wave shuffle), y neighbours between the rows of one
//     thread are registers; only the two rows bordering a wavefront's strip are loaded again (L1/L2 hits when the
//     neighbouring wavefront of the workgroup streams them; the workgroup's outermost halo rows are the only bytes
//     fetched twice from HBM: 2 per 4R rows);
//   * SOFTWARE PIPELINE: everything iteration k consumes was requested in iteration k-1 (own rows of plane k+1, halo
//     rows and edge cells of plane k) -- a source is split into raw() (issue the loads) and xf() (turn what arrived into
//     e, e.g. r*iD for the fused Jacobi smoother), so no instruction waits for a load issued in the same iteration
//     except the epilogue operands, which are requested first and consumed last;
//   * the diagonal is recomputed from the six face coefficients (same operations as set_diag!, Poisson.jl:48-54);
//   * coefficient-uniform rows: away from the body (and from the domain faces) every face coefficient of a row is
//     the same number c (1 on the finest level, 2^l below: restrictL! sums four unit faces and halves).  wl_mg_update
//     records c per row (NaN = not uniform, see k_lrow) together with the iD values of the row; in such a row the
//     kernels load neither L (3 of the 5 array passes of mult) nor iD and use the recorded numbers -- the very values
//     the loads would have returned, so results are unchanged.  The row constants are read through the CONSTANT address
//     space with wave-uniform addresses => scalar loads (s_load_dwordx8, scalar cache, lgkmcnt): they neither occupy
//     vector-memory issue slots nor force the vector loads in flight to drain, and are prefetched one plane ahead;
//   * an epilogue functor turns A e into the operator at hand: z=Ae & z.e (pcg!), r-=Ae & x+=e (increment!),
//     r = z-Ax (residual!), with per-thread Float64 partials reduced exactly like the range kernels.
// Per-cell arithmetic and its order are those of mult()/set_diag! => bit-identical to the generic kernels.
// Requirements (else the caller falls back to the generic range kernel): D==3 and (n0-2) % V == 0.
#pragma once
#include <type_traits>

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
    __device__ __forceinline__ static VecA splat(T x) {
        VecA r;
#pragma unroll
        for (int q = 0; q < V; ++q) r.v[q] = x;
        return r;
    }
};

constexpr int S7_BY = 4;   // wavefronts per workgroup of the vector kernels

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
    int inv = 0;                    // kind 0: the body runs when *active == 0 instead
    int rev = 0;                    // every XCD walks its range of tiles backwards (set by the launchers, see sweep_rev)
    int zcap = 0;                   // host side: > 0 = cut z into at most this many chunks (few partials for the consumer's sum)
};
// budget of per-workgroup partials a pcg! kernel may sum itself (Gate kind 1..3): the grid of the kernels of such a call is
// capped at this many workgroups (swept in round 3 on a 128^3 level: 517 / 326 / 228 / 195 / 232 us per pcg! call for
// 128 / 256 / 512 / 1024 / 2048)
constexpr int WL_PCG_PARTIALS = 1024;
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
        if (gt.active) { pre.act = gt.inv ? !*gt.active : *gt.active; run = pre.act || (gt.also && *gt.also); }
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

// ---- row constants (k_lrow): RC_N values per x-row (j,k) of a level: [c, idc, idf, idl, lxf, lxl, -, -]
//   c   = the one value of every face coefficient of the row (NaN: the row is not uniform, use the arrays),
//   idc = iD of the cells 2..n0-3, idf/idl = iD of the two end cells, lxf/lxl = Lx at the two x-boundary faces
//   (i = 1 and n0-1).  A wavefront works on whole rows, so the row index is wave-uniform and the constants are fetched
//   with scalar loads through the constant address space (written by k_lrow in an earlier kernel, never by a consumer).
constexpr int RC_N = 8;
template <class T> struct RowK {
    T c, idc, idf, idl, lxf, lxl;
    __device__ __forceinline__ bool uni() const { return c == c; }
};
// the same constants of a row KNOWN (at compile time) to be uniform: consumers that branch on uni() lose the branch and
// the array loads behind it.  The kernels test uniformity ONCE per plane for all rows involved and run one of two
// copies of the plane body -- in the all-uniform copy no instruction can load L or iD, so the loads in flight (the
// prefetch of the next plane) are never drained on behalf of a rarely-taken path (the compiler cannot bound the age of
// a load that MAY have been issued into a register: it waits for vmcnt(0)).
template <class T> struct RowKU : RowK<T> {
    __device__ __forceinline__ RowKU(const RowK<T> &r) : RowK<T>(r) {}
    __device__ __forceinline__ constexpr bool uni() const { return true; }
};
template <class T> __device__ __forceinline__ RowK<T> rowk_none() {
    RowK<T> r;
    r.c = __builtin_nanf("");
    r.idc = r.idf = r.idl = r.lxf = r.lxl = (T)0;
    return r;
}
// row = j + n1*k, wave-uniform (callers pass values that went through readfirstlane); rowc == nullptr -> "not uniform"
template <class T> __device__ __forceinline__ RowK<T> rowk_load(const T *rowc, long row) {
    if (!rowc) return rowk_none<T>();
    typedef T wl_rc8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) wl_rc8 *cptr;
    const wl_rc8 a = *(cptr)(rowc + RC_N * row);
    RowK<T> r;
    r.c = a[0]; r.idc = a[1]; r.idf = a[2]; r.idl = a[3]; r.lxf = a[4]; r.lxl = a[5];
    return r;
}
// iD of the V cells starting at interior cell i of a row with constants rk (uniform row), or from the array
template <class T, class RKT>
__device__ __forceinline__ VecA<T> row_iD(const RKT &rk, const T *iD, long o, int i, int n0) {
    constexpr int V = VecA<T>::V;
    if (rk.uni()) {
        VecA<T> b = VecA<T>::splat(rk.idc);
        if (i == 1) b.v[0] = rk.idf;
        if (i + V - 1 == n0 - 2) b.v[V - 1] = rk.idl;
        return b;
    }
    return VecA<T>::load(iD + o);
}
// iD of the single cell ii (any index 0..n0-1) of such a row; ghost cells carry iD = 0 (set_diag! writes inside only)
template <class T, class RKT> __device__ __forceinline__ T row_iD1(const RKT &rk, const T *iD, long o, int ii, int n0) {
    if (rk.uni()) return (ii < 1 || ii > n0 - 2) ? (T)0 : (ii == 1 ? rk.idf : (ii == n0 - 2 ? rk.idl : rk.idc));
    return iD[o];
}

// ---- stencil operand sources.  raw(o,i,j,k) issues the loads of the V cells starting at (i,j,k) [offset o];
// xf(raw, rk, o, i) turns them into e once they have arrived (rk = row constants of that row); sraw / sxf do the same
// for ONE cell (the x neighbour of a wavefront's first / last lane).  NEED_ID: xf uses the row's iD constants.
template <class T> struct SrcArray {          // e is an array (pcg!: eps, residual!: x, increment!: eps)
    static constexpr bool NEED_ID = false;
    static constexpr int ROWS_AUTO = 2;        // rows per thread the default picks on big levels (measured at 512^3)
    static constexpr bool EA_IS_RAW = false;   // the epilogue operand array `ea` is NOT the array raw() reads
    using Raw = VecA<T>;
    const T *e;
    __device__ __forceinline__ Raw raw(long o, int, int, int) const { return VecA<T>::load(e + o); }
    template <class RKT> __device__ __forceinline__ VecA<T> xf(const Raw &a, const RKT &, long, int) const { return a; }
    __device__ __forceinline__ T sraw(long o, int, int, int) const { return e[o]; }
    template <class RKT> __device__ __forceinline__ T sxf(T a, const RKT &, long, int) const { return a; }
};
template <class T> struct SrcJacobi {         // e = r*iD evaluated on the fly (Jacobi!, src/Poisson.jl:111)
    static constexpr bool NEED_ID = true;
    static constexpr int ROWS_AUTO = 2;
    static constexpr bool EA_IS_RAW = true;    // the fused smoother's epilogue operand `a` is r itself: the raw centre vector
                                               // of the window is handed over instead of loading r a second time
    using Raw = VecA<T>;
    const T *r, *iD;
    int n0;
    __device__ __forceinline__ Raw raw(long o, int, int, int) const { return VecA<T>::load(r + o); }
    template <class RKT> __device__ __forceinline__ VecA<T> xf(const Raw &a, const RKT &rk, long o, int i) const {
        const VecA<T> b = row_iD<T>(rk, iD, o, i, n0);
        VecA<T> c;
#pragma unroll
        for (int v = 0; v < VecA<T>::V; ++v) c.v[v] = a.v[v] * b.v[v];
        return c;
    }
    __device__ __forceinline__ T sraw(long o, int, int, int) const { return r[o]; }
    template <class RKT> __device__ __forceinline__ T sxf(T a, const RKT &rk, long o, int ii) const { return a * row_iD1<T>(rk, iD, o, ii, n0); }
};
template <class T> struct SrcProlong {        // e[I] = coarse x[down(I)] inside, 0 on ghosts (MultiLevelPoisson.jl:2,34)
    static constexpr bool NEED_ID = false;
    static constexpr int ROWS_AUTO = 1;        // (two rows per thread measured 12 % slower for this source at 512^3)
    static constexpr bool EA_IS_RAW = false;
    using Raw = VecA<T>;
    const T *cx;
    G C;            // coarse grid
    int n0, n1, nzg, kz0;   // fine extents (global along z) and the fine grid's kz0
    __device__ __forceinline__ T sraw(long, int i, int j, int k) const {
        const int kg = k + kz0;
        if (i < 1 || i > n0 - 2 || j < 1 || j > n1 - 2 || kg < 1 || kg > nzg - 2) return (T)0;
        return cx[C.at((i + 1) / 2, (j + 1) / 2, (kg + 1) / 2 - C.kz0)];
    }
    __device__ __forceinline__ Raw raw(long, int i, int j, int k) const {   // i is odd (1 + V*m), i+V-1 <= n0-2
        const int kg = k + kz0;
        if (j < 1 || j > n1 - 2 || kg < 1 || kg > nzg - 2) return VecA<T>::splat((T)0);
        VecA<T> c;
        const T *row = cx + C.at(0, (j + 1) / 2, (kg + 1) / 2 - C.kz0);
#pragma unroll
        for (int v = 0; v < VecA<T>::V; v += 2) { const T p = row[(i + v + 1) / 2]; c.v[v] = p; c.v[v + 1] = p; }
        return c;
    }
    template <class RKT> __device__ __forceinline__ VecA<T> xf(const Raw &a, const RKT &, long, int) const { return a; }
    template <class RKT> __device__ __forceinline__ T sxf(T a, const RKT &, long, int) const { return a; }
};

// An epilogue may bring its own operand loader (a functor with HAS_LD, `Dat ld(o, i, j, k)` and
// `operator()(o, i, j, k, Ae, e, dat, carry, rk, acc, pre)`): ld is called where ea / eb are requested, its result reaches the
// epilogue at the end of the iteration (op_residual with the divergence evaluated on the fly).  Lambdas have none.
// The loader may also keep one value per row from plane to plane (`Carry first(o, j, k0)` before the first plane,
// `Carry next(dat)` after each): an operand of plane k+1 that the epilogue of plane k has already loaded (the upper z face
// of a face difference is the lower one of the next plane).
template <class E, class = void> struct EpiLd { static constexpr bool ON = false; struct Dat {}; struct Carry {}; };
template <class E> struct EpiLd<E, std::enable_if_t<E::HAS_LD>> {
    static constexpr bool ON = true;
    using Dat = typename E::Dat;
    using Carry = typename E::Carry;
};

// One launch of the 7-point kernel.  R rows per thread.  ea / eb: optional epilogue operand arrays, loaded next to the
// stencil operands (requested at the top of the iteration, consumed by the epilogue at its end).
// epi(o, i, j, k, Ae, e, a, b, rk, acc, pre): rk = row constants of the cell's own row in plane k.
template <class T, int NRED, int R, class SRC, class EPI>
__global__ __launch_bounds__(64 * S7_BY) void k_stencil7(G g, SRC src, const T *__restrict__ L, const T *rowc, const T *ea, const T *eb,
                                                        EPI epi, double *partials, int ntx, int tpp, int nblk, int clen, int klo,
                                                        int khi, Gate gate) {
    constexpr int V = Vec16<T>::V;
    using VA = VecA<T>;
    using Raw = typename SRC::Raw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    int lb, pslot;
    tile_of(b, nblk, gate.rev, lb, pslot);                             // XCD-contiguous logical id
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int n0 = g.n[0], n1 = g.n[1], nxi = n0 - 2, nyi = n1 - 2;
    const int i = 1 + (pt % ntx) * 64 * V + lane * V;
    const int jb = __builtin_amdgcn_readfirstlane(1 + (pt / ntx) * (S7_BY * R) + wv * R);   // first row of this wavefront's strip
    const int k0 = klo + ch * clen, k1 = min(khi + 1, k0 + clen);
    double acc[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int q = 0; q < (NRED > 0 ? NRED : 1); ++q) acc[q] = 0.0;
    Pre pre;
    const bool run = gate_open(gate, pre);
    // (n1-2) % R == 0 (launch condition): a strip is either whole or absent
    const bool active = run && (i <= nxi) && (jb <= nyi) && (k0 < k1);
    if (active) {   // (no barriers below: inactive lanes may simply skip; shuffles only pair active lanes)
        const bool first = (lane == 0), last = (lane == 63) || (i + V > nxi);
        const long sy = g.s[1], sz = g.s[2], sc = g.sc;
        const long col = (long)i + sy * (long)jb;           // row q of the strip: col + q*sy
        const T *Lx = L, *Ly = L + sc, *Lz = L + 2 * sc;
        const long rrow = (long)jb;                         // row-constant index of (row q, plane k): rrow + q + n1*k
        auto RK = [&](int q, int k) { return rowk_load<T>(rowc, rrow + q + (long)n1 * k); };
        const int kmax = k1;                                // last plane that exists for this chunk (k1 <= n2-1)

        // Everything iteration k consumes that is still IN FLIGHT when the iteration starts: requested by iteration k-1.
        // Two such sets alternate (the loop is unrolled by two), so a set is never copied while its loads are outstanding
        // -- a register move of an in-flight load would wait for it at the bottom of the producing iteration.
        struct Fly {
            Raw own[R];          // raw own rows of plane k+1
            RowK<T> rk[R];       // their row constants (plane k+1)
            Raw hlo, hhi;        // raw halo rows (jb-1, jb+R) of plane k
            RowK<T> rkl, rkh;    // and their row constants (only when the source needs iD)
            T lf[R], rg[R];      // raw x-neighbour cells of the strip's first / last lane, plane k
        };
        auto request = [&](Fly &f, int k) {   // the set consumed by iteration k
            const int kn = min(k + 1, kmax);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                f.own[q] = src.raw(col + q * sy + sz * kn, i, jb + q, kn);
                f.rk[q] = RK(q, kn);
                f.lf[q] = (T)0; f.rg[q] = (T)0;
                if (first) f.lf[q] = src.sraw(col + q * sy + sz * k - 1, i - 1, jb + q, k);
                if (last) f.rg[q] = src.sraw(col + q * sy + sz * k + V, i + V, jb + q, k);
            }
            f.hlo = src.raw(col - sy + sz * k, i, jb - 1, k);
            f.hhi = src.raw(col + R * sy + sz * k, i, jb + R, k);
            f.rkl = rowk_none<T>(); f.rkh = rowk_none<T>();
            if (SRC::NEED_ID) { f.rkl = RK(-1, k); f.rkh = RK(R, k); }
        };

        // ---- prologue: e of planes k0-1, k0 and the first in-flight set
        VA em[R], ec[R];
        Raw rawc[R];                                        // raw own rows of plane k (kept only when the epilogue wants them)
        RowK<T> rk0[R];                                     // own-row constants of plane k
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long cq = col + q * sy;
            const Raw a = src.raw(cq + sz * (k0 - 1), i, jb + q, k0 - 1), c = src.raw(cq + sz * k0, i, jb + q, k0);
            rawc[q] = c;
            const RowK<T> rkm = RK(q, k0 - 1);
            rk0[q] = RK(q, k0);
            em[q] = src.xf(a, rkm, cq + sz * (k0 - 1), i);
            ec[q] = src.xf(c, rk0[q], cq + sz * k0, i);
        }
        typename EpiLd<EPI>::Carry cy[R];
        if constexpr (EpiLd<EPI>::ON) {
#pragma unroll
            for (int q = 0; q < R; ++q) cy[q] = epi.first(col + q * sy + sz * k0, jb + q, k0);
        }
        Fly A, B;
        request(A, k0);
        VA lzc[R];              // upper z-face coefficients of plane k-1 (rows that loaded L), see (3)
        bool lzok[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { lzc[q] = VA::splat((T)0); lzok[q] = false; }

        auto step = [&](int k, Fly &cur, Fly &nxt) {
            const long ok = col + sz * k;
            const int kn = min(k + 1, kmax);
            // ---- (1) requests: epilogue operands of this plane first (consumed last), then the set of the NEXT iteration
            VA av[R], bv[R];
            typename EpiLd<EPI>::Dat ex[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (SRC::EA_IS_RAW) av[q] = rawc[q];
                else if (ea) av[q] = VA::load(ea + ok + q * sy);
                if (eb) bv[q] = VA::load(eb + ok + q * sy);
                if constexpr (EpiLd<EPI>::ON) ex[q] = epi.ld(ok + q * sy, i, jb + q, k);
            }
            request(nxt, kn);
            // ---- (2)+(3) in two copies: every row involved in this plane is coefficient-uniform (no L / iD load can occur),
            //      or the general form
            bool fast = true;
#pragma unroll
            for (int q = 0; q < R; ++q) fast = fast && rk0[q].uni() && (!SRC::NEED_ID || cur.rk[q].uni());
            if (SRC::NEED_ID) fast = fast && cur.rkl.uni() && cur.rkh.uni();
            auto plane = [&](auto FAST) {
                constexpr bool F = decltype(FAST)::value;
                using RKT = typename std::conditional<F, RowKU<T>, RowK<T>>::type;
                // (2) what arrived: e of plane k+1 (own rows), of the halo rows and edge cells of plane k
                VA ep[R];
#pragma unroll
                for (int q = 0; q < R; ++q) ep[q] = src.xf(cur.own[q], RKT(cur.rk[q]), col + q * sy + sz * kn, i);
                const VA ylo = src.xf(cur.hlo, RKT(cur.rkl), ok - sy, i), yhi = src.xf(cur.hhi, RKT(cur.rkh), ok + R * sy, i);
                // (3) the stencil, row by row.  In rows that load L, a face coefficient shared with the cell processed just before
                // is loaded ONCE: the lower y face of row q is the upper y face of row q-1 (same iteration), the lower z face
                // of plane k the upper z face of plane k-1 (kept from the previous iteration) -- 3 to 3.5 vector loads of L
                // per row instead of 5 (the second read of Lz came one plane later, mostly from HBM again: bodies that cut
                // many rows, e.g. the torus of C5, paid 0.7 of an array pass for it).  A coefficient-uniform neighbour has
                // that face equal to its constant c, by the definition of uniform.
                VA lyc;                 // upper y face of the previous row of this strip
                bool lyok = false;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const long o = ok + q * sy;
                    const RKT rc(rk0[q]);
                    VA lx, ly0, ly1, lz0, lz1;
                    T lxr;
                    if (rc.uni()) {   // all faces of this row are c, except possibly the two x-boundary faces (in the row constants)
                        lx = VA::splat(rc.c); ly0 = lx; ly1 = lx; lz0 = lx; lz1 = lx;
                        if (i == 1) lx.v[0] = rc.lxf;
                        lxr = (i + V > nxi) ? rc.lxl : rc.c;
                        if (!F) { lyc = ly1; lyok = true; lzc[q] = lz1; }
                        lzok[q] = !F;   // (the all-uniform copy of the plane body keeps no vector state: a later general plane reloads)
                    } else {
                        lx = VA::load(Lx + o);
                        if (lyok) ly0 = lyc; else ly0 = VA::load(Ly + o);
                        ly1 = VA::load(Ly + o + sy);
                        if (lzok[q]) lz0 = lzc[q]; else lz0 = VA::load(Lz + o);
                        lz1 = VA::load(Lz + o + sz);
                        lyc = ly1; lyok = true;
                        lzc[q] = lz1; lzok[q] = true;
                        lxr = lane_dn1(lx.v[0]);
                        if (last) lxr = Lx[o + V];
                    }
                    // x neighbours of the vector ends: adjacent lane, or the (prefetched) cell beyond the strip
                    T left = lane_up1(ec[q].v[V - 1]), right = lane_dn1(ec[q].v[0]);
                    if (first) left = src.sxf(cur.lf[q], rc, o - 1, i - 1);
                    if (last) right = src.sxf(cur.rg[q], rc, o + V, i + V);
                    const VA &ym = (q == 0) ? ylo : ec[q == 0 ? 0 : q - 1];
                    const VA &yp = (q == R - 1) ? yhi : ec[q == R - 1 ? q : q + 1];
                    VA ae;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T xm = (v == 0) ? left : ec[q].v[v == 0 ? 0 : v - 1];
                        const T xp = (v == V - 1) ? right : ec[q].v[v == V - 1 ? v : v + 1];
                        const T lxlo = lx.v[v], lxhi = (v == V - 1) ? lxr : lx.v[v == V - 1 ? v : v + 1];
                        T dg = 0;
                        dg -= (lxlo + lxhi);
                        dg -= (ly0.v[v] + ly1.v[v]);
                        dg -= (lz0.v[v] + lz1.v[v]);
                        T s = ec[q].v[v] * dg;
                        s += xm * lxlo + xp * lxhi;
                        s += ym.v[v] * ly0.v[v] + yp.v[v] * ly1.v[v];
                        s += em[q].v[v] * lz0.v[v] + ep[q].v[v] * lz1.v[v];
                        ae.v[v] = s;
                    }
                    if constexpr (EpiLd<EPI>::ON) { epi(o, i, jb + q, k, ae, ec[q], ex[q], cy[q], rc, acc, pre); cy[q] = epi.next(ex[q]); }
                    else epi(o, i, jb + q, k, ae, ec[q], (ea || SRC::EA_IS_RAW) ? av[q] : ec[q], eb ? bv[q] : ec[q], rc, acc, pre);
                }
#pragma unroll
                for (int q = 0; q < R; ++q) { em[q] = ec[q]; ec[q] = ep[q]; }
            };
            if (fast) plane(std::true_type{}); else plane(std::false_type{});
            // ---- (4) the constants of plane k+1 (arrived: xf used them) become those of the next iteration's own plane
#pragma unroll
            for (int q = 0; q < R; ++q) { rk0[q] = cur.rk[q]; if (SRC::EA_IS_RAW) rawc[q] = cur.own[q]; }
        };
        for (int k = k0; k < k1; k += 2) {
            step(k, A, B);
            if (k + 1 < k1) step(k + 1, B, A);
        }
    }
    if (NRED > 0) {
        block_red<(NRED > 0 ? NRED : 1), S7_BY>(acc, RED_SUM);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NRED; ++q) partials[(long)q * gridDim.x + pslot] = acc[q];
        }
    }
}

// ---- 16-B vectorised streaming (no neighbours), software-pipelined: ld(o, j, k, pre) requests the operands of the V
// cells at element offset o and returns them; st(o, i, j, k, data, rk, acc, pre) consumes them one iteration later (the
// loads of plane k+1 are in flight while plane k is computed and stored).  RK: hand the row constants to st.
// Same row mapping as k_stencil7 with R = 1 (lane = V cells of a row, wavefront = row segment, workgroup = 4 rows).
template <class T, int NRED, bool RK, class LD, class ST, int OP = RED_SUM>
__global__ __launch_bounds__(256) void k_rowvec(G g, LD ld, ST st, const T *rowc, double *partials, int ntx, int tpp, int nblk,
                                                int clen, int klo, int khi, Gate gate) {
    constexpr int V = Vec16<T>::V;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x;
    int lb, pslot;
    tile_of(b, nblk, gate.rev, lb, pslot);
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int i = 1 + (pt % ntx) * 64 * V + lane * V;
    const int j = __builtin_amdgcn_readfirstlane(1 + (pt / ntx) * 4 + wv);
    const int k0 = klo + ch * clen, k1 = min(khi + 1, k0 + clen);
    double acc[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int q = 0; q < (NRED > 0 ? NRED : 1); ++q) acc[q] = (OP == RED_MAX) ? -1e300 : 0.0;
    Pre pre;
    const bool run = gate_open(gate, pre);
    if (run && i <= g.n[0] - 2 && j <= g.n[1] - 2 && k0 < k1) {
        const long col = (long)i + g.s[1] * (long)j;
        const long n1 = g.n[1];
        using Dat = decltype(ld(col, j, k0, pre));
        // (two copies of the consumer: in a coefficient-uniform row no instruction can load iD -- see RowKU)
        auto consume = [&](int k, const Dat &cur, const RowK<T> &rkc) {
            if (RK && rkc.uni()) st(col + g.s[2] * k, i, j, k, cur, RowKU<T>(rkc), acc, pre);
            else st(col + g.s[2] * k, i, j, k, cur, rkc, acc, pre);
        };
        auto rkl = [&](int k) { return RK ? rowk_load<T>(rowc, j + n1 * k) : rowk_none<T>(); };
        // two operand sets alternate (loop unrolled by two): a set is never copied while its loads are outstanding
        Dat dA = ld(col + g.s[2] * k0, j, k0, pre), dB;
        RowK<T> rkA = rkl(k0), rkB = rowk_none<T>();
        auto step = [&](int k, const Dat &cur, const RowK<T> &rkc, Dat &nxt, RowK<T> &rkn) {
            const int kn = min(k + 1, k1 - 1);
            nxt = ld(col + g.s[2] * kn, j, kn, pre);
            rkn = rkl(kn);
            consume(k, cur, rkc);
        };
        for (int k = k0; k < k1; k += 2) {
            step(k, dA, rkA, dB, rkB);
            if (k + 1 < k1) step(k + 1, dB, rkB, dA, rkA);
        }
    }
    if (NRED > 0) {
        block_red<(NRED > 0 ? NRED : 1), 4>(acc, OP);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NRED; ++q) partials[(long)q * gridDim.x + pslot] = acc[q];
        }
    }
}

// chunking of the marching axis: `tpp` workgroups per plane, `nown` planes; the grid is tpp*nchunk <= the target (also the
// number of reduction partials); cap > 0 limits the number of chunks (in-kernel partial sums want few partials)
inline void chunking(int tpp, int nown, int cap, int target_k, int *clen, int *nchunk) {
    int target = target_k * 1024;                 // wl_set_option(16 / 17): grid size of the 7-point / streaming kernels in units of 1024 workgroups
    if (target < 1024) target = 1024;
    if (target > WL_MAXB) target = WL_MAXB;
    int want = target / tpp;
    if (cap > 0 && cap < want) want = cap;
    if (want < 1) want = 1;
    if (want > nown) want = nown;
    *clen = (nown + want - 1) / want;
    *nchunk = (nown + *clen - 1) / *clen;
}

// will launch_rowvec accept this level?  (one plane's workgroups must fit the partial buffer)
template <class T> inline bool rowvec_fits(const G &g) {
    constexpr int V = Vec16<T>::V;
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + 3) / 4;
    return ((ntx * nty + 7) / 8) * 8 <= WL_MAXB;
}
template <class T, int NRED, bool RK, class LD, class ST, int OP = RED_SUM>
inline int launch_rowvec(int kclass, const G &g, LD ld, ST st, const T *rowc, double *partials, int *np, Gate gate = Gate(),
                         int kov_lo = 0, int kov_hi = -1) {
    constexpr int V = Vec16<T>::V;
    Range R = r_inside(g);
    if (np) *np = 0;
    if (kov_hi >= kov_lo) { R.lo[2] = kov_lo > R.lo[2] ? kov_lo : R.lo[2]; R.hi[2] = kov_hi < R.hi[2] ? kov_hi : R.hi[2]; }
    if (R.count() <= 0) return 0;
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + 3) / 4;
    const int tpp = ((ntx * nty + 7) / 8) * 8;
    int clen, nchunk;
    chunking(tpp, R.hi[2] - R.lo[2] + 1, gate.zcap, ctx().opt[17], &clen, &nchunk);
    const int nblk = tpp * nchunk;
    if (nblk > WL_MAXB) return -1;
    if (np) *np = nblk;
    Prof p(kclass, R.count());
    gate.rev = sweep_rev();
    hipLaunchKernelGGL((k_rowvec<T, NRED, RK, LD, ST, OP>), dim3(nblk), dim3(256), 0, ctx().stream, g, ld, st, RK ? rowc : nullptr, partials,
                       ntx, tpp, nblk, clen, R.lo[2], R.hi[2], gate);
    return (int)hipGetLastError();
}

// Row constants of a Poisson level (see the header comment): one wavefront scans one x-row.  c is recorded when
//   Lx[i] == c for the interior faces i = 2..n0-2 (faces 1 and n0-1 are the domain boundary, or its periodic image),
//   Ly[i,j] == Ly[i,j+1] == Lz[i,k] == Lz[i,k+1] == c for every interior i, and iD[i] == idc for i = 2..n0-3
// (the two end cells have a different diagonal when the boundary faces are not c), else NaN; layout: RC_N above.
template <class T>
__global__ __launch_bounds__(256) void k_lrow(G g, const T *__restrict__ L, const T *__restrict__ iD, T *rowc, const unsigned char *rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)g.n[1] * g.n[2];
    if (row >= nrows) return;
    if (rows && !rows[row]) return;   // only the flagged rows are re-examined (their constants are rewritten)
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
        rc[1] = idc;
        if (!bad) {
            const long base = g.at(0, j, k);
            rc[2] = iD[base + 1]; rc[3] = iD[base + g.n[0] - 2];
            rc[4] = L[base + 1]; rc[5] = L[base + g.n[0] - 1];
        } else { rc[2] = rc[3] = rc[4] = rc[5] = nan; }
        rc[6] = rc[7] = (T)0;
    }
}
template <class T> inline int op_lrow(const G &g, const T *L, const T *iD, T *rowc, const unsigned char *rows = nullptr) {
    const long nrows = (long)g.n[1] * g.n[2];
    Prof p(WL_K_MISC, g.cells());
    hipLaunchKernelGGL((k_lrow<T>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, L, iD, rowc, rows);
    return (int)hipGetLastError();
}

// can the vector kernels run on this level?  (any element-aligned pointers and any strides will do)
template <class T> inline bool stencil7_ok(const G &g) {
    constexpr int V = Vec16<T>::V;
    if (!ctx().opt[0]) return false;
    return g.D == 3 && (g.n[0] - 2) % V == 0 && g.n[0] - 2 >= V;
}

// launch over the owned interior planes (or the plane sub-range [kov_lo,kov_hi]); *np = number of partials per reduced
// value (0 if nothing to do).  Returns -1 when the level does not fit the launch (caller falls back).
template <class T, int NRED, int R, class SRC, class EPI>
inline int launch_stencil7_r(int kclass, const G &g, SRC src, const T *L, const T *rowc, const T *ea, const T *eb, EPI epi,
                             double *partials, int *np, Gate gate, int kov_lo, int kov_hi) {
    constexpr int V = Vec16<T>::V;
    Range Rg = r_inside(g);
    if (np) *np = 0;
    if (kov_hi >= kov_lo) { Rg.lo[2] = kov_lo > Rg.lo[2] ? kov_lo : Rg.lo[2]; Rg.hi[2] = kov_hi < Rg.hi[2] ? kov_hi : Rg.hi[2]; }
    if (Rg.count() <= 0) return 0;
    const int klo = Rg.lo[2], khi = Rg.hi[2];
    const int ntx = (g.n[0] - 2 + 64 * V - 1) / (64 * V), nty = (g.n[1] - 2 + S7_BY * R - 1) / (S7_BY * R);
    const int tpp = ((ntx * nty + 7) / 8) * 8;
    int clen, nchunk;
    chunking(tpp, khi - klo + 1, gate.zcap, ctx().opt[16], &clen, &nchunk);
    const int nblk = tpp * nchunk;
    if (nblk > WL_MAXB) return -1;   // plane too large for the partial buffer: caller falls back
    if (np) *np = nblk;
    Prof p(kclass, Rg.count());
    gate.rev = sweep_rev();
    hipLaunchKernelGGL((k_stencil7<T, NRED, R, SRC, EPI>), dim3(nblk), dim3(64 * S7_BY), 0, ctx().stream, g, src, L, rowc, ea, eb, epi,
                       partials, ntx, tpp, nblk, clen, klo, khi, gate);
    return (int)hipGetLastError();
}
// rows per thread: wl_set_option(4, .): 1 or 2 forced; 0 (default) = the source's preference (2 for array / Jacobi
// sources, 1 for the prolongation source) on levels of at least 2^26 interior cells whose y extent is even (half the
// halo-row traffic; the larger register window costs occupancy, which only the big levels can afford to trade), else 1
template <class T, int NRED, class SRC, class EPI>
inline int launch_stencil7(int kclass, const G &g, SRC src, const T *L, const T *rowc, const T *ea, const T *eb, EPI epi,
                           double *partials, int *np, Gate gate = Gate(), int kov_lo = 0, int kov_hi = -1) {
    const int want = ctx().opt[4];
    const bool even = ((g.n[1] - 2) % 2) == 0;
    const bool two = even && (want == 2 || (want == 0 && SRC::ROWS_AUTO == 2 && r_inside(g).count() >= (1L << 26)));
    if (two) return launch_stencil7_r<T, NRED, 2>(kclass, g, src, L, rowc, ea, eb, epi, partials, np, gate, kov_lo, kov_hi);
    return launch_stencil7_r<T, NRED, 1>(kclass, g, src, L, rowc, ea, eb, epi, partials, np, gate, kov_lo, kov_hi);
}

// The same launch on a z-slab level whose operand `hal` (one halo plane per side) has to be exchanged first
// (where the reference calls perBC! on the operand).  With a comm stream available (halo_begin) the planes that do
// not read a halo plane are computed WHILE the exchange is in flight; the first and last owned plane follow once it
// has landed.  Reduction partials of the three launches are laid end to end (NRED <= 1).  Not decomposed / overlap
// off / fewer than 3 planes: exchange in-stream, one launch.
// begun (optional): the exchange this launch depends on was ALREADY started by the caller (halo_begin / halo_begin2, possibly
// together with other arrays) -- it is only waited for here (hal is not exchanged again).
template <class T, int NRED, class SRC, class EPI>
inline int launch_stencil7_halo(int kclass, const G &g, T *hal, SRC src, const T *L, const T *rowc, const T *ea, const T *eb,
                                EPI epi, double *partials, int *np, Gate gate = Gate(), bool begun = false) {
    static_assert(NRED <= 1, "partials of the split launches are concatenated: one reduced value at most");
    const Range R = r_inside(g);
    const int lo = R.lo[2], hi = R.hi[2];
    if (!g.dist || !overlap_on() || hi - lo + 1 < 3) {
        if (begun) WL_TRY(halo_end());           // (in-stream or already waited for: a no-op then)
        else WL_TRY((halo_exchange<T>(g, hal, 1, 1)));
        return launch_stencil7<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials, np, gate);
    }
    if (!begun) WL_TRY((halo_begin<T>(g, hal, 1, 1)));
    ctx().n_overlapped += 1;
    int n1 = 0, n2 = 0, n3 = 0;
    int rc = launch_stencil7<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials, &n1, gate, lo + 1, hi - 1);
    const int rce = halo_end();
    if (rc) return rc;      // (-1: not applicable -- the exchange has been waited for, the caller's fallback may run)
    if (rce) return rce;
    rc = launch_stencil7<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials ? partials + n1 : nullptr, &n2, gate, lo, lo);
    if (rc) return rc > 0 ? rc : fail(WL_E_STATE, "split 7-point launch: boundary plane rejected", __FILE__, __LINE__);
    rc = launch_stencil7<T, NRED>(kclass, g, src, L, rowc, ea, eb, epi, partials ? partials + n1 + n2 : nullptr, &n3, gate, hi, hi);
    if (rc) return rc > 0 ? rc : fail(WL_E_STATE, "split 7-point launch: boundary plane rejected", __FILE__, __LINE__);
    if (np) *np = n1 + n2 + n3;
    return 0;
}

}  // namespace wl

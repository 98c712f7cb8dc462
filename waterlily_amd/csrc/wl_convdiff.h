// wl_convdiff.h -- conv_diff! (src/Flow.jl:36-60) for D=3, non-periodic, as ONE LDS-tiled z-marching kernel.
//
// Layout of the work (gfx950):
//   * a 256-thread workgroup owns a 64(x) x 4(y) column of cells and marches along z; x in [1, n0-2] so that a
//     power-of-two grid is covered by whole 64-lane wavefronts (the two x-ghost planes go through the generic
//     gather kernel: 0.4 % of the cells);
//   * per z-plane the three velocity components of the tile + a 2-cell halo are staged in LDS (triple-buffered:
//     planes k-1, k for the stencils, k+1 being filled) -- every u element is fetched from HBM/L2 once per
//     workgroup instead of ~25x per cell; z-neighbours of the thread's own column live in a 4-deep register
//     window; global loads for plane k+1/k+2 are issued before the flux arithmetic of plane k (latency hidden);
//   * a thread evaluates the x and y fluxes through both faces of its cell and the z flux through the LOWER
//     face only: the upper z flux of cell k is the lower z flux of cell k+1, computed one iteration later,
//     so the partially summed cell is carried in registers for one iteration (15 flux evaluations per cell
//     instead of 18);
//   * all boundary special cases (src/Flow.jl:54-55: one-sided fluxes, Float64 accumulation on boundary faces,
//     "top ghost included" ranges of inside_u, util.jl:55-57) are operand/flag selects: straight-line code.
// The arithmetic per flux and the order of the six +/- updates per cell are exactly those of the reference
// (and of the generic gather kernel in wl_ops.h), so results are bit-identical to the oracle.
#pragma once
#include <type_traits>

#include "wl_common.h"

namespace wl {

constexpr int CD_BX = 64, CD_BY = 4, CD_H = 2;
constexpr int CD_W = CD_BX + 2 * CD_H;   // 68 columns
constexpr int CD_R = CD_BY + 2 * CD_H;   // 8 rows
constexpr int CD_HALO = CD_W * CD_R - CD_BX * CD_BY;  // 288 halo cells per component and plane

// flux through one face (src/Flow.jl:6,8,9 + the diffusive term of :45,:54,:55), returned in Float64:
//   interior face : phiu  - nu*d      lower boundary: phiuL - nu*d      upper boundary: phiuR - nu*d
// fm2,fm1,f0,fp1 = f[I-2s], f[I-s], f[I], f[I+s] of the transported component, uf = face velocity.
// GEN = false: the caller knows that no face of this workgroup's tile is a domain-boundary face in this plane
// (lowbnd = topbnd = bnd = false everywhere): the compiler drops the central-flux and Float64-accumulation variants
// instead of evaluating them next to the interior ones and selecting (about a third of the VALU work of a cell).
template <class T, bool GEN = true>
__device__ __forceinline__ double cd_flux(T fm2, T fm1, T f0, T fp1, double uf, T nu, bool lowbnd, bool topbnd) {
    if (!GEN) { lowbnd = false; topbnd = false; }
    const bool neg = uf < 0;
    const bool up = topbnd ? !neg : (uf > 0);      // phiuR takes the upwind-from-below triple unless u<0
    const T q = quick<T>(up ? fm2 : fp1, up ? fm1 : f0, up ? f0 : fm1);
    const double cen = (double)((T)(f0 + fm1) * (T)0.5);
    const bool central = (lowbnd && uf > 0) || (topbnd && neg);
    const double flux = uf * (central ? cen : (double)q);
    const T nud = nu * (T)(f0 - fm1);
    return flux - (double)nud;
}
template <class T, bool GEN = true> __device__ __forceinline__ T cd_add(T r, double F, bool bnd) {   // r += flux (lower face)
    if (!GEN) return r + (T)F;
    return bnd ? (T)((double)r + F) : r + (T)F;
}
template <class T, bool GEN = true> __device__ __forceinline__ T cd_sub(T r, double F, bool bnd) {   // r -= flux (upper face)
    if (!GEN) return r - (T)F;
    return bnd ? (T)((double)r - F) : r - (T)F;
}

// FIN (mom_step! only): the kernel also FINISHES BDIM! (Flow.jl:134, then scale_u! :166) on the body-free x-rows -- mu1 = 0,
// V = 0, mu0 = 1 along the row (`rowfree`, op_bdim2), where the statement is u (+)= f: the new velocity of such a row goes
// to `unew` straight from the registers that hold f, V is not read there (f = (u0 + dt r) - 0; the load stays unconditional but
// goes to one fixed address: a value selected around a load would be moved while the load is in flight, and that move waits
// for every load issued before it), and the row's x-ghost cells
// of the BC! that follows are written too (XBc).  The separate pass over the body-free rows (read f, read u, write u:
// 6T / 9T per cell) disappears; the busy rows keep their own kernel.  `unew` must not be the array the stencil reads:
//   FIN = 1, predictor: u was zeroed (scale_u!(a,0), :154)  -> unew = (0 + f);   u0 IS the stencil input, nothing is copied
//   FIN = 2, corrector: u' = the stencil input              -> unew = ((u' + f) * 0.5), both roundings kept
// (the values of op_bdim2's pass over the body-free rows, see cd_fin => bit-identical)
template <class T> struct CdFin { T *unew; const unsigned char *rowfree; int xon; T U0; };
// op_bdim2 evaluates the statement as the reference's promotions dictate: tmp = (0.5*0 + 0) + Float64(f); u = T(0 + tmp) or
// un = T(Float64(u') + tmp), u = T(Float64(un) * 0.5).  The same values in T arithmetic (this kernel is short of VALU cycles,
// and Float64 adds and conversions run at half rate): tmp = f with -0 turned into +0, i.e. f + 0; a sum of two T values
// rounded to Float64 and then to T equals the sum rounded to T at once (53 >= 2*24 + 2 bits); a product with 0.5 is exact in
// both types, and where it is not (subnormal result) both paths round the same exact value to nearest-even.
template <class T, int FIN> __device__ __forceinline__ T cd_fin(T uold, T fv) {
    const T tmp = fv + (T)0;
    if (FIN == 1) return tmp;
    const T un = uold + tmp;
    return un * (T)0.5;
}
// store a finished cell of component c (+ the row's x-ghost cells: Dirichlet U0 for the normal component on the planes 0, 1
// and n0-1, the neighbour's value for the other two -- util.jl:200-207 for a non-periodic x without the convective exit)
template <class T, bool XEDGE = true> __device__ __forceinline__ void cd_fin_store(const CdFin<T> &fn, long o, int c, int i, int n0, T val) {
    if (XEDGE && fn.xon) {   // (XEDGE = false: the caller knows that its tile holds neither i = 1 nor i = n0-2)
        if (i == 1) { if (c == 0) val = fn.U0; fn.unew[o - 1] = val; }
        if (i == n0 - 2) fn.unew[o + 1] = (c == 0) ? fn.U0 : val;
    }
    fn.unew[o] = val;
}

// COPY (predictor, Flow.jl:154): `a.u0 .= a.u` is folded in -- the kernel reads u, writes u0out = u for every cell it
// owns and uses that value in the BDIM epilogue (saves the separate 6T copy pass).
// One launch covers up to CD_NSEG boxes of tiles ("segments": a range of planes x a range of tile rows, each with its own
// chunking): the shell around the shared-flux kernel's box -- the two boundary plane pairs, the first and the last tile
// rows -- is four small boxes, which as four launches in a row cost more than the work in them (4 x 33 us at 512^3).
constexpr int CD_NSEG = 4;
struct CdSeg { int b0, nblk, tpp, clen, ty0, ntile, zlo, zhi; };   // b0: first block of the segment (multiple of 8)
struct CdSegs { int n; CdSeg s[CD_NSEG]; };
template <class T, bool FUSE, bool COPY, int FIN>
__global__ __launch_bounds__(CD_BX *CD_BY) void k_convdiff3(G g, T *__restrict__ r, const T *__restrict__ u, T nu,
                                                           const T *u0, T *u0out, const T *__restrict__ V, T dt,
                                                           double a0, double a1, double a2, bool has_acc, int ntx,
                                                           CdSegs segs, CdFin<T> fn) {
    __shared__ T sm[3][3][CD_R][CD_W];  // [plane slot][component][row][col]
    const int tx = threadIdx.x & (CD_BX - 1), ty = threadIdx.x / CD_BX;
    CdSeg sg = segs.s[0];
#pragma unroll
    for (int q = 1; q < CD_NSEG; ++q)
        if (q < segs.n && (int)blockIdx.x >= segs.s[q].b0) sg = segs.s[q];   // (scalar selects: uniform per workgroup)
    const int nblk = sg.nblk, tpp = sg.tpp, clen = sg.clen, ty0 = sg.ty0, ntile = sg.ntile, zlo = sg.zlo, zhi = sg.zhi;
    const int b = (int)blockIdx.x - sg.b0;
    const int lb = (nblk & 7) ? b : (b & 7) * (nblk >> 3) + (b >> 3);  // XCD-contiguous logical id
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int i0 = 1 + CD_BX * (pt % ntx), j0 = CD_BY * (ty0 + pt / ntx);   // ty0: first tile row of this segment
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2], nzg = g.nzg, kz0 = g.kz0;
    const int k0 = zlo + ch * clen, k1 = min(zhi + 1, k0 + clen);   // this chunk of the segment's (owned) planes
    if (k0 > zhi || j0 >= n1 || pt >= ntile) return;  // uniform per workgroup (padding tiles)
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i <= n0 - 2) && (j <= n1 - 1);
    const int ic = min(i, n0 - 1), jc = min(j, n1 - 1);
    const long col = (long)ic + g.s[1] * (long)jc;  // own column offset (clamped for idle threads)
    const long sz = g.s[2], sc = g.sc;

    // halo duties of this thread: halo cell h1 = t, and h2 = t + 256 when < 288
    int hl[2] = {0, 0};
    long hg[2] = {0, 0};
    int nh = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int h = (int)threadIdx.x + q * 256;
        if (h < CD_HALO) {
            int row, cl;
            if (h < 4 * CD_W) { const int rr = h / CD_W; row = rr < 2 ? rr : rr + CD_BY; cl = h - rr * CD_W; }
            else { const int qq = h - 4 * CD_W; row = CD_H + (qq >> 2); const int c4 = qq & 3; cl = c4 < 2 ? c4 : c4 + CD_BX; }
            const int gx = min(max(i0 - CD_H + cl, 0), n0 - 1), gy = min(max(j0 - CD_H + row, 0), n1 - 1);
            hl[nh] = row * CD_W + cl;
            hg[nh] = (long)gx + g.s[1] * (long)gy;
            ++nh;
        }
    }
    const int own_l = (ty + CD_H) * CD_W + tx + CD_H;
    auto clampk = [n2](int k) { return min(max(k, 0), n2 - 1); };
    auto SM = [&](int slot, int c) -> T * { return &sm[slot][c][0][0]; };

    // ---- prologue: register window k0-2..k0+1 of the own column, LDS planes k0-1 and k0
    T W[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) W[c][q] = u[col + sz * clampk(k0 - 2 + q) + sc * c];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int kk = clampk(k0 - 1 + pl);
        const int slot = (k0 - 1 + pl + 3) % 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            SM(slot, c)[own_l] = W[c][1 + pl];
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nh) SM(slot, c)[hl[q]] = u[hg[q] + sz * kk + sc * c];
        }
    }
    __syncthreads();

    T part[3] = {0, 0, 0};     // carried partial sums of cell k-1 (after x, y and lower-z fluxes)
    T cu0[3] = {0, 0, 0}, cV[3] = {0, 0, 0};
    bool carry = false, cfin = false;
    const bool jlow = j >= 1;
    const bool jin = (j >= 1) && (j <= n1 - 2);
    // FIN: the row flag of plane k is requested one iteration ahead (a load consumed in the iteration that issues it would
    // drain every load in flight: vmcnt counts in order)
    const unsigned char *ffp = fn.rowfree + jc;
    unsigned char ffl = 0;
    if (FIN) ffl = ffp[(long)n1 * clampk(k0)];
    const int kend = min(k1, n2 - 1);

    for (int k = k0; k <= kend; ++k) {
        const bool own = k < k1;
        // ---- A. issue the global loads of the next planes first
        T nxt[3], hv[2][3], e0[3], eV[3];
        const int kn = clampk(k + 2), kh = clampk(k + 1);
        const int kg = k + kz0;                 // plane number in the undecomposed array
        // FIN: is cell k an interior cell of a body-free row?  (wave-uniform: a wavefront works on one row)
        bool fin = false;
        if (FIN) { fin = own && jin && kg >= 1 && kg <= nzg - 2 && ffl != 0; ffl = ffp[(long)n1 * kh]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            nxt[c] = u[col + sz * kn + sc * c];
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nh) hv[q][c] = u[hg[q] + sz * kh + sc * c];
            if (FUSE) {
                e0[c] = (COPY || FIN == 1) ? W[c][2] : u0[col + sz * k + sc * c];
                eV[c] = V[(fin ? 0L : col + sz * k) + sc * c];   // (body-free row: V = 0 is not read -- see vsel below)
            }
            if (COPY && own && active) u0out[col + sz * k + sc * c] = W[c][2];
        }
        // ---- B. fluxes of plane k
        const int s1 = (k + 3) % 3, s0 = (k + 2) % 3;  // LDS slots of planes k and k-1
        const bool ring = g.zring;              // periodic ring of slabs: every z face is an interior face
        const bool klow = ring || kg >= 1;
        const bool lowok = jlow && klow;
        const bool zlb = !ring && (kg == 1), ztb = !ring && (kg == nzg - 1);
        T rr[3] = {0, 0, 0};
        double Fz[3];
        // boundary faces of this tile in this plane? (x: first / last tile of a row; y: the tiles holding j = 1 and
        // j = n1-2; z: the planes kg = 1 and kg = nzg-1) -- uniform over the workgroup
        const bool bface = (i0 == 1) || (i0 + CD_BX > n0 - 2) || (j0 <= 1) || (j0 + CD_BY > n1 - 2) || zlb || ztb;
        auto fluxes = [&](auto gen) {
            constexpr bool GEN = decltype(gen)::value;
    #pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T *P = SM(s1, c) + own_l;  // plane k, component c, centred on the own cell
                // ---- z: lower face of cell k (also the upper face of the carried cell k-1)
                double ufz;
                if (c == 0) ufz = (double)((T)(W[2][2] + SM(s1, 2)[own_l - 1]) * (T)0.5);
                else if (c == 1) ufz = (double)((T)(W[2][2] + SM(s1, 2)[own_l - CD_W]) * (T)0.5);
                else ufz = (double)((T)(W[2][2] + W[2][1]) * (T)0.5);
                Fz[c] = cd_flux<T, GEN>(W[c][0], W[c][1], W[c][2], W[c][3], ufz, nu, zlb, ztb);
                if (own && lowok) {
                    // ---- x: lower face i, upper face i+1
                    const T xm2 = P[-2], xm1 = P[-1], x0 = W[c][2], xp1 = P[1], xp2 = P[2];
                    double ufl, ufu;
                    const T *PX = SM(s1, 0) + own_l;
                    if (c == 0) { ufl = (double)((T)(PX[0] + PX[-1]) * (T)0.5); ufu = (double)((T)(PX[1] + PX[0]) * (T)0.5); }
                    else if (c == 1) { ufl = (double)((T)(PX[0] + PX[-CD_W]) * (T)0.5); ufu = (double)((T)(PX[1] + PX[1 - CD_W]) * (T)0.5); }
                    else { ufl = (double)((T)(PX[0] + W[0][1]) * (T)0.5); ufu = (double)((T)(PX[1] + SM(s0, 0)[own_l + 1]) * (T)0.5); }
                    const bool xlb = (i == 1), xtb = (i == n0 - 2);
                    rr[c] = cd_add<T, GEN>(rr[c], cd_flux<T, GEN>(xm2, xm1, x0, xp1, ufl, nu, xlb, false), xlb);
                    rr[c] = cd_sub<T, GEN>(rr[c], cd_flux<T, GEN>(xm1, x0, xp1, xp2, ufu, nu, false, xtb), xtb);
                    // ---- y: lower face j, upper face j+1   (only rows j <= n1-2 take part, util.jl:55-57)
                    if (j <= n1 - 2) {
                        const T ym2 = P[-2 * CD_W], ym1 = P[-CD_W], yp1 = P[CD_W], yp2 = P[2 * CD_W];
                        const T *PY = SM(s1, 1) + own_l;
                        double vfl, vfu;
                        if (c == 0) { vfl = (double)((T)(PY[0] + PY[-1]) * (T)0.5); vfu = (double)((T)(PY[CD_W] + PY[CD_W - 1]) * (T)0.5); }
                        else if (c == 1) { vfl = (double)((T)(PY[0] + PY[-CD_W]) * (T)0.5); vfu = (double)((T)(PY[CD_W] + PY[0]) * (T)0.5); }
                        else { vfl = (double)((T)(PY[0] + W[1][1]) * (T)0.5); vfu = (double)((T)(PY[CD_W] + SM(s0, 1)[own_l + CD_W]) * (T)0.5); }
                        const bool ylb = (j == 1), ytb = (j == n1 - 2);
                        rr[c] = cd_add<T, GEN>(rr[c], cd_flux<T, GEN>(ym2, ym1, x0, yp1, vfl, nu, ylb, false), ylb);
                        rr[c] = cd_sub<T, GEN>(rr[c], cd_flux<T, GEN>(ym1, x0, yp1, yp2, vfu, nu, false, ytb), ytb);
                    }
                    // ---- z lower face of the own cell (cells k <= n2-2 take part in z)
                    if (ring || kg <= nzg - 2) rr[c] = cd_add<T, GEN>(rr[c], Fz[c], zlb);
                }
            }
        };
        if (bface) fluxes(std::true_type{}); else fluxes(std::false_type{});
        // ---- C. finish and store: the carried cell k-1 gets its upper z flux; cell k is stored now when it
        //         takes no upper z flux (k == 0, k == n2-1, or j == 0), else it is carried
        auto emit = [&](int kk, const T(&val)[3], const T(&q0)[3], const T(&qV)[3], bool fin_, int wq) {   // wq: W slot of plane kk
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T v = val[c];
                const long o = col + sz * kk + sc * c;
                if (FUSE) {
                    if (has_acc) v = (T)((double)v + (c == 0 ? a0 : (c == 1 ? a1 : a2)));
                    const T fv = (q0[c] + dt * v) - ((FIN && fin_) ? (T)0 : qV[c]);   // vsel
                    r[o] = fv;
                    if (FIN) { if (fin_) cd_fin_store<T>(fn, o, c, i, n0, cd_fin<T, FIN>(wq == 1 ? W[c][1] : W[c][2], fv)); }
                } else {
                    r[o] = v;
                }
            }
        };
        if (carry) {
            T done[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) done[c] = cd_sub<T>(part[c], Fz[c], ztb);
            if (active) emit(k - 1, done, cu0, cV, cfin, 1);
            carry = false;
        }
        if (own) {
            const bool needs_up = lowok && (ring || kg <= nzg - 2);
            if (needs_up) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { part[c] = rr[c]; cu0[c] = e0[c]; cV[c] = eV[c]; }
                carry = true; cfin = fin;
            } else if (active) {
                emit(k, rr, e0, eV, fin, 2);
            }
        }
        // ---- D. rotate the register window, publish plane k+1 into its LDS slot
        const int s2 = (k + 4) % 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            W[c][0] = W[c][1]; W[c][1] = W[c][2]; W[c][2] = W[c][3]; W[c][3] = nxt[c];
            SM(s2, c)[own_l] = W[c][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nh) SM(s2, c)[hl[q]] = hv[q][c];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// SHARED-FLUX form for the tiles and planes whose y and z faces are all interior faces (all but the first / last tile
// row and the two planes next to each z boundary): the flux through a face is evaluated ONCE and used by both cells it
// separates, as the reference's scatter form does (Flow.jl:46-47: r[I] += Phi[I]; r[I-d] -= Phi[I]) -- 9 flux
// evaluations per cell instead of 15 (the kernel is VALU-bound: DESIGN.md section 4).
//   * a thread evaluates the flux through the LOWER x, y and z face of its cell only;
//   * its upper x flux is the lower x flux of lane+1 (wave shuffle), its upper y flux that of the thread one row up
//     (through a small LDS flux buffer), its upper z flux that of its own next plane (carried, as before);
//   * the fluxes of the tile's outermost upper faces -- x face i0+64 of the 4 rows, y face j0+4 of the 64 columns -- belong
//     to no cell of the tile: each of the four wavefronts evaluates one more flux per plane for them (wavefronts 0..2 the
//     y face of one component each, wavefront 3 the twelve x-face values), all operands from the LDS tile -- 10 evaluations
//     per wavefront and plane, balanced (a fifth "edge" wavefront was measured first: 320-thread workgroups pack only 16
//     working wavefronts on a CU instead of 24 and the kernel ran 70 % SLOWER than the per-cell gather form);
//   * so that ONE barrier per plane still suffices, a cell is finished one iteration late: iteration k evaluates the
//     lower fluxes of plane k and publishes them; iteration k+1 (after the plane barrier) collects the three upper fluxes
//     of plane k and applies all six in the reference's order  ((((0 + Fx) - Fx') + Fy) - Fy') + Fz) - Fz'  .
// x-boundary tiles (first / last tile of a row) stay on this path: the one boundary face of such a tile takes the
// one-sided flux (GENX) and, for the upper boundary, the Float64 accumulation of Flow.jl:55.
// Same arithmetic per flux, same order of the six updates => bit-identical to k_convdiff3 and to the oracle.
// BY = rows of cells per tile = wavefronts per workgroup (4 or 8: an 8-row tile stages 68x12 cells per 64x8 owned instead
// of 68x8 per 64x4 -- the halo share of the u reads drops from 2.1x to 1.6x)
template <class T, int BY> struct CdsShared {
    T sm[3][3][BY + 2 * CD_H][CD_W];   // [plane slot][component][row][col]
    T fyb[2][3][BY][CD_BX];            // lower-y fluxes of the tile's cells, by plane parity
    T yedge[2][3][CD_BX];              // y flux through face j0+BY
    double xedge[2][3][BY];            // x flux through face ie (Float64: the upper x boundary accumulates in it)
};
template <class T, bool FUSE, bool COPY, bool GENX, int BY, int FIN>
__device__ __forceinline__ void convdiff3s_tile(CdsShared<T, BY> &S_, const G &g, T *__restrict__ r, const T *__restrict__ u, T nu, const T *u0,
                                                T *u0out, const T *__restrict__ V, T dt, double a0, double a1, double a2, bool has_acc,
                                                int i0, int j0, int ie, int k0, int k1, const CdFin<T> &fn) {
    auto &sm = S_.sm; auto &fyb = S_.fyb; auto &yedge = S_.yedge; auto &xedge = S_.xedge;
    const int tx = threadIdx.x & (CD_BX - 1), ty = threadIdx.x / CD_BX;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const bool xlow = GENX && (i0 == 1), xtop = GENX && (ie == n0 - 1);
    const int i = i0 + tx, j = j0 + ty;
    const bool active = (i <= n0 - 2);
    const int ic = min(i, n0 - 1);
    const long col = (long)ic + g.s[1] * (long)j;  // own column offset (clamped for idle lanes)
    const long sz = g.s[2], sc = g.sc;

    // halo duties of this thread: halo cell h1 = t, and h2 = t + NT when that is still a halo cell
    constexpr int NT = CD_BX * BY, NHALO = CD_W * (BY + 2 * CD_H) - CD_BX * BY;
    int hl[2] = {0, 0};
    long hg[2] = {0, 0};
    int nh = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int h = (int)threadIdx.x + q * NT;
        if (h < NHALO) {
            int row, cl;
            if (h < 4 * CD_W) { const int rr = h / CD_W; row = rr < 2 ? rr : rr + BY; cl = h - rr * CD_W; }
            else { const int qq = h - 4 * CD_W; row = CD_H + (qq >> 2); const int c4 = qq & 3; cl = c4 < 2 ? c4 : c4 + CD_BX; }
            const int gx = min(max(i0 - CD_H + cl, 0), n0 - 1), gy = min(max(j0 - CD_H + row, 0), n1 - 1);
            hl[nh] = row * CD_W + cl;
            hg[nh] = (long)gx + g.s[1] * (long)gy;
            ++nh;
        }
    }
    const int own_l = (ty + CD_H) * CD_W + tx + CD_H;
    auto clampk = [n2](int k) { return min(max(k, 0), n2 - 1); };
    auto SM = [&](int slot, int c) -> T * { return &sm[slot][c][0][0]; };

    // ---- prologue: register window k0-2..k0+1 of the own column, LDS planes k0-1 and k0
    T W[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) W[c][q] = u[col + sz * clampk(k0 - 2 + q) + sc * c];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int kk = clampk(k0 - 1 + pl);
        const int slot = (k0 - 1 + pl + 3) % 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            SM(slot, c)[own_l] = W[c][1 + pl];
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nh) SM(slot, c)[hl[q]] = u[hg[q] + sz * kk + sc * c];
        }
    }
    __syncthreads();

    // carried state of cell k-1: its three lower fluxes (rounded to T like the reference's Phi scratch), BDIM operands
    T cfx[3] = {0, 0, 0}, cfy[3] = {0, 0, 0}, cfz[3] = {0, 0, 0}, cu0[3] = {0, 0, 0}, cV[3] = {0, 0, 0};
    bool carry = false, cfin = false;
    // FIN: the row flag of plane k is requested one iteration ahead (a load consumed in the iteration that issues it would
    // drain every load in flight: vmcnt counts in order)
    const unsigned char *ffp = fn.rowfree + j;
    unsigned char ffl = 0;
    if (FIN) ffl = ffp[(long)n1 * k0];

    for (int k = k0; k <= k1; ++k) {
        const bool own = k < k1;
        const int par = k & 1;
        // ---- A. issue the global loads of the next planes first
        T nxt[3], hv[2][3], e0[3] = {0, 0, 0}, eV[3] = {0, 0, 0};
        const int kn = clampk(k + 2), kh = clampk(k + 1);
        // FIN: is (j, k) a body-free row?  (every cell of this kernel is an interior cell; wave-uniform: a wavefront = one row)
        bool fin = false;
        if (FIN) { fin = own && ffl != 0; ffl = ffp[(long)n1 * kh]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            nxt[c] = u[col + sz * kn + sc * c];
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nh) hv[q][c] = u[hg[q] + sz * kh + sc * c];
            if (FUSE && own) {
                e0[c] = (COPY || FIN == 1) ? W[c][2] : u0[col + sz * k + sc * c];
                eV[c] = V[(fin ? 0L : col + sz * k) + sc * c];   // (body-free row: V = 0 is not read -- see vsel below)
            }
            if (COPY && own && active) u0out[col + sz * k + sc * c] = W[c][2];
        }
        // ---- B0. lower z fluxes of plane k (own column: registers + one LDS operand) = upper z fluxes of the carried cell
        const int s1 = (k + 3) % 3, s0 = (k + 2) % 3;  // LDS slots of planes k and k-1
        T fx[3] = {0, 0, 0}, fy[3] = {0, 0, 0}, fz[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double ufz;
            if (c == 0) ufz = (double)((T)(W[2][2] + SM(s1, 2)[own_l - 1]) * (T)0.5);
            else if (c == 1) ufz = (double)((T)(W[2][2] + SM(s1, 2)[own_l - CD_W]) * (T)0.5);
            else ufz = (double)((T)(W[2][2] + W[2][1]) * (T)0.5);
            fz[c] = (T)cd_flux<T, false>(W[c][0], W[c][1], W[c][2], W[c][3], ufz, nu, false, false);
        }
        // ---- C. finish cell k-1 FIRST: its upper fluxes are the lower fluxes of the neighbours of plane k-1 (published last
        //         iteration) and fz above.  Its stores then have the flux arithmetic of plane k to drain in: the plane loop ends
        //         on a wait for every outstanding memory operation (the halo values go to LDS), stores included
        if (carry) {
            const int pp = par ^ 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T fxh = lane_dn1(cfx[c]);
                const double xe = xedge[pp][c][ty];
                const bool lastc = (i + 1 == ie);
                if (lastc) fxh = (T)xe;
                const T fyh = (ty < BY - 1) ? fyb[pp][c][ty + 1][tx] : yedge[pp][c][tx];
                T v = (T)0 + cfx[c];
                v = (GENX && xtop && lastc) ? (T)((double)v - xe) : v - fxh;
                v = v + cfy[c];
                v = v - fyh;
                v = v + cfz[c];
                v = v - fz[c];
                if (active) {
                    const long o = col + sz * (k - 1) + sc * c;
                    if (FUSE) {
                        if (has_acc) v = (T)((double)v + (c == 0 ? a0 : (c == 1 ? a1 : a2)));
                        const T fv = (cu0[c] + dt * v) - ((FIN && cfin) ? (T)0 : cV[c]);   // vsel
                        r[o] = fv;
                        if (FIN) { if (cfin) cd_fin_store<T, GENX>(fn, o, c, i, n0, cd_fin<T, FIN>(W[c][1], fv)); }   // (W[.][1]: plane k-1 of u)
                    } else {
                        r[o] = v;
                    }
                }
            }
        }
        // ---- B. lower x and y fluxes of plane k (plane k1 only finishes cell k1-1: nothing to do)
        if (own) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T *P = SM(s1, c) + own_l;  // plane k, component c, centred on the own cell
                const T xm2 = P[-2], xm1 = P[-1], x0 = W[c][2], xp1 = P[1];
                const T *PX = SM(s1, 0) + own_l;
                double ufl;
                if (c == 0) ufl = (double)((T)(PX[0] + PX[-1]) * (T)0.5);
                else if (c == 1) ufl = (double)((T)(PX[0] + PX[-CD_W]) * (T)0.5);
                else ufl = (double)((T)(PX[0] + W[0][1]) * (T)0.5);
                fx[c] = (T)cd_flux<T, GENX>(xm2, xm1, x0, xp1, ufl, nu, xlow && (i == 1), false);
                const T ym2 = P[-2 * CD_W], ym1 = P[-CD_W], yp1 = P[CD_W];
                const T *PY = SM(s1, 1) + own_l;
                double vfl;
                if (c == 0) vfl = (double)((T)(PY[0] + PY[-1]) * (T)0.5);
                else if (c == 1) vfl = (double)((T)(PY[0] + PY[-CD_W]) * (T)0.5);
                else vfl = (double)((T)(PY[0] + W[1][1]) * (T)0.5);
                fy[c] = (T)cd_flux<T, false>(ym2, ym1, x0, yp1, vfl, nu, false, false);
                fyb[par][c][ty][tx] = fy[c];
            }
            // ---- B'. the tile's outermost upper faces, one extra evaluation per wavefront (all operands from LDS):
            //      wavefronts 0..2: y face j0+BY of column tx for component ty; wavefront 3 (lanes 0..3BY-1): x face ie of row lane/3
            if (ty < 3) {
                const int c = ty;
                const int el = (BY + CD_H) * CD_W + tx + CD_H;        // LDS offset of cell (i0+tx, j0+BY)
                const T *P = SM(s1, c) + el;
                const T *PY = SM(s1, 1) + el;
                double vf;
                if (c == 0) vf = (double)((T)(PY[0] + PY[-1]) * (T)0.5);
                else if (c == 1) vf = (double)((T)(PY[0] + PY[-CD_W]) * (T)0.5);
                else vf = (double)((T)(PY[0] + SM(s0, 1)[el]) * (T)0.5);
                yedge[par][c][tx] = (T)cd_flux<T, false>(P[-2 * CD_W], P[-CD_W], P[0], P[CD_W], vf, nu, false, false);
            } else if (ty == 3 && tx < 3 * BY) {
                const int rr = tx / 3, c = tx - 3 * rr;
                const int xl = (rr + CD_H) * CD_W + (ie - i0) + CD_H;   // LDS offset of cell (ie, j0+rr)
                const T *P = SM(s1, c) + xl;
                const T *PX = SM(s1, 0) + xl;
                double uf;
                if (c == 0) uf = (double)((T)(PX[0] + PX[-1]) * (T)0.5);
                else if (c == 1) uf = (double)((T)(PX[0] + PX[-CD_W]) * (T)0.5);
                else uf = (double)((T)(PX[0] + SM(s0, 0)[xl]) * (T)0.5);
                const T fp1 = (ie + 1 <= n0 - 1) ? P[1] : P[0];          // (unused by the one-sided flux at the upper boundary)
                xedge[par][c][rr] = cd_flux<T, GENX>(P[-2], P[-1], P[0], fp1, uf, nu, false, xtop);
            }
        }
        carry = own; cfin = fin;
#pragma unroll
        for (int c = 0; c < 3; ++c) { cfx[c] = fx[c]; cfy[c] = fy[c]; cfz[c] = fz[c]; cu0[c] = e0[c]; cV[c] = eV[c]; }
        // ---- D. rotate the register window, publish plane k+1 into its LDS slot
        const int s2 = (k + 4) % 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            W[c][0] = W[c][1]; W[c][1] = W[c][2]; W[c][2] = W[c][3]; W[c][3] = nxt[c];
            SM(s2, c)[own_l] = W[c][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nh) SM(s2, c)[hl[q]] = hv[q][c];
        }
        __syncthreads();
    }
}
// One launch covers every x tile of the interior tile rows; a tile holding a domain x-boundary face (first / last of a
// row: uniform per workgroup) runs the copy of the plane loop with the one-sided flux variants (GENX), the others the plain one.
template <class T, bool FUSE, bool COPY, int BY, int FIN>
__global__ __launch_bounds__(CD_BX *BY) void k_convdiff3s(G g, T *__restrict__ r, const T *__restrict__ u, T nu, const T *u0, T *u0out,
                                                         const T *__restrict__ V, T dt, double a0, double a1, double a2, bool has_acc,
                                                         int ntx, int tpp, int nblk, int clen, int jbase, int klo, int khi, int ntile, int rev,
                                                         CdFin<T> fn) {
    // (dynamic LDS: CdsShared<double, 8> is 86.6 KB -- more than the 64 KB a static __shared__ object may take)
    extern __shared__ __attribute__((aligned(16))) unsigned char cds_raw[];
    CdsShared<T, BY> &S_ = *reinterpret_cast<CdsShared<T, BY> *>(cds_raw);
    const int b = blockIdx.x;
    int lb, pslot;
    tile_of(b, nblk, rev, lb, pslot);                                  // XCD-contiguous logical id
    (void)pslot;
    const int ch = lb / tpp, pt = lb - ch * tpp;
    const int n0 = g.n[0], n1 = g.n[1];
    const int i0 = 1 + CD_BX * (pt % ntx), j0 = jbase + BY * (pt / ntx);   // jbase: first row of this launch's tile rows
    const int k0 = klo + ch * clen, k1 = min(khi + 1, k0 + clen);      // this chunk of planes (all z faces interior)
    if (k0 > khi || pt >= ntile || j0 + BY - 1 > n1 - 3 || i0 > n0 - 2) return;   // uniform per workgroup (padding tiles)
    const int ie = min(i0 + CD_BX, n0 - 1);       // the x face beyond the tile's last cell
    if (i0 == 1 || ie == n0 - 1) convdiff3s_tile<T, FUSE, COPY, true, BY, FIN>(S_, g, r, u, nu, u0, u0out, V, dt, a0, a1, a2, has_acc, i0, j0, ie, k0, k1, fn);
    else convdiff3s_tile<T, FUSE, COPY, false, BY, FIN>(S_, g, r, u, nu, u0, u0out, V, dt, a0, a1, a2, has_acc, i0, j0, ie, k0, k1, fn);
}

// host side.  One tile row = CD_BY rows of cells starting at j0 = CD_BY*ty; the shared-flux kernel takes the tile rows
// whose cells have interior y faces only (2 <= j0, j0+CD_BY-1 <= n1-3) on the planes whose cells have interior z faces
// only (global plane 2 .. nzg-3; every plane on a periodic ring of slabs); k_convdiff3 takes the first / last tile row
// and the boundary planes; the two x-ghost planes go through the generic gather kernel (caller).
struct CdBox { int zlo, zhi, ty0, nty; };   // planes [zlo, zhi] x tile rows [ty0, ty0+nty)
template <class T, bool FUSE, bool COPY, int FIN>
int launch_convdiff3_old(const G &g, const CdBox *box, int nbox, T *r, const T *u, double nu_, const T *u0, T *u0out, const T *V,
                         double dt_, const double (&a3)[3], bool has_acc, const CdFin<T> &fn) {
    const int ntx = (g.n[0] - 2 + CD_BX - 1) / CD_BX;
    CdSegs segs;
    segs.n = 0;
    int nblk = 0;
    long cells = 0;
    for (int q = 0; q < nbox && segs.n < CD_NSEG; ++q) {
        const CdBox &bx = box[q];
        if (bx.nty <= 0 || bx.zhi < bx.zlo) continue;
        const int tpp = ((ntx * bx.nty + 7) / 8) * 8;  // padded to a multiple of 8 for the XCD mapping (idle tail tiles)
        const int nown = bx.zhi - bx.zlo + 1;
        int want = WL_GRID / tpp;
        if (want < 1) want = 1;
        if (want > nown) want = nown;
        int clen = (nown + want - 1) / want;
        if (clen < 8) clen = nown < 8 ? nown : 8;   // a chunk re-reads ~4 planes ahead of its first one: keep chunks >= 8 planes
        const int nchunk = (nown + clen - 1) / clen;
        segs.s[segs.n++] = CdSeg{nblk, tpp * nchunk, tpp, clen, bx.ty0, ntx * bx.nty, bx.zlo, bx.zhi};
        nblk += tpp * nchunk;
        cells += (long)g.n[0] * (long)(bx.nty * CD_BY) * nown;
    }
    if (!segs.n) return 0;
    for (int q = segs.n; q < CD_NSEG; ++q) segs.s[q] = segs.s[0];
    Prof p(WL_K_CONVDIFF, cells);
    hipLaunchKernelGGL((k_convdiff3<T, FUSE, COPY, FIN>), dim3(nblk), dim3(CD_BX * CD_BY), 0, ctx().stream, g, r, u, (T)nu_, u0,
                       u0out, V, (T)dt_, a3[0], a3[1], a3[2], has_acc, ntx, segs, fn);
    return (int)hipGetLastError();
}
template <class T, bool FUSE, bool COPY, int FIN = 0>
int launch_convdiff3(const G &g, T *r, const T *u, double nu_, const T *u0, T *u0out, const T *V, double dt_,
                     const double *acc, bool has_acc, const CdFin<T> &fn = CdFin<T>{nullptr, nullptr, 0, (T)0}) {
    static_assert(FIN == 0 || (FUSE && !COPY), "FIN finishes BDIM!: it needs the fused epilogue and copies nothing");
    double a3[3] = {0, 0, 0};
    if (has_acc) for (int d = 0; d < 3; ++d) a3[d] = acc[d];
    const int ntx = (g.n[0] - 2 + CD_BX - 1) / CD_BX, nty_all = (g.n[1] + CD_BY - 1) / CD_BY;
    // tile rows with interior y faces only: ty in [tlo, thi]
    const int tlo = 1;
    int thi = (g.n[1] - 3 - (CD_BY - 1)) / CD_BY;      // CD_BY*ty + CD_BY-1 <= n1-3
    // planes with interior z faces only, clipped to the owned planes
    int klo = g.zlo, khi = g.zhi;
    if (!g.zring) { klo = max(klo, 2 - g.kz0); khi = min(khi, g.nzg - 3 - g.kz0); }
    // Float32: 8-row tiles (43 KB of LDS per 512-thread workgroup).  Float64 stays on 4-row tiles: 8 rows would need 86.6 KB =
    // ONE workgroup of 8 wavefronts per CU where the 4-row tiles run three of 4 -- built and measured slower in round 3
    // (3.30 -> 3.55 ms per launch at 512^3; DESIGN.md, measured dead ends), removed in round 4
    const bool use8 = sizeof(T) == 4;
    int thi_s = thi;
    if (use8 && ((thi_s - tlo + 1) & 1) && thi_s > tlo) --thi_s;   // 8-row tiles: an odd tile row goes to the shell
    const bool shared = ctx().opt[18] != 0 && thi_s >= tlo && khi >= klo;
    if (!shared) {
        const CdBox all{g.zlo, g.zhi, 0, nty_all};
        return launch_convdiff3_old<T, FUSE, COPY, FIN>(g, &all, 1, r, u, nu_, u0, u0out, V, dt_, a3, has_acc, fn);
    }
    thi = thi_s;
    // (1)+(2) the shell, ONE launch of the per-cell kernel: boundary planes (every tile row), and on the interior planes
    // the first / last tile rows
    const CdBox shell[4] = {{g.zlo, klo - 1, 0, nty_all}, {khi + 1, g.zhi, 0, nty_all},
                            {klo, khi, 0, tlo}, {klo, khi, thi + 1, nty_all - (thi + 1)}};
    WL_TRY((launch_convdiff3_old<T, FUSE, COPY, FIN>(g, shell, 4, r, u, nu_, u0, u0out, V, dt_, a3, has_acc, fn)));
    // (3) interior planes, interior tile rows: shared-flux kernel; x-boundary tiles in a launch of their own (GENX)
    const int nty = thi - tlo + 1;
    const int nown = khi - klo + 1;
    // 8-row tiles (Float32: 43 KB of LDS per workgroup; Float64: 86.6 KB, optional), 4-row tiles for the rows that do not
    // fill one
    const int rev = sweep_rev();
    auto launch = [&](auto BYc, int jbase, int ntr) -> int {   // ntr tile rows of BY rows starting at row jbase
        constexpr int BY = decltype(BYc)::value;
        if (ntr <= 0) return 0;
        const int ntile = ntx * ntr;
        const int tpp = ((ntile + 7) / 8) * 8;
        int want = WL_GRID / tpp;
        if (want < 1) want = 1;
        if (want > nown) want = nown;
        const int clen = (nown + want - 1) / want, nchunk = (nown + clen - 1) / clen;
        const int nblk = tpp * nchunk;
        Prof p(WL_K_CONVDIFF, (long)g.n[0] * (long)(ntr * BY) * nown);
        constexpr size_t lds = sizeof(CdsShared<T, BY>);
        static_assert(lds <= 64 * 1024, "conv_diff tile: more LDS than a launch gets without opting in");
        hipLaunchKernelGGL((k_convdiff3s<T, FUSE, COPY, BY, FIN>), dim3(nblk), dim3(CD_BX * BY), lds, ctx().stream, g, r, u, (T)nu_, u0, u0out, V,
                           (T)dt_, a3[0], a3[1], a3[2], has_acc, ntx, tpp, nblk, clen, jbase, klo, khi, ntile, rev, fn);
        return (int)hipGetLastError();
    };
    const int rows = nty * CD_BY, jb0 = tlo * CD_BY;
    if constexpr (sizeof(T) == 4) {
        const int n8 = rows / 8;
        WL_TRY(launch(std::integral_constant<int, 8>{}, jb0, n8));
        return launch(std::integral_constant<int, 4>{}, jb0 + 8 * n8, (rows - 8 * n8) / 4);
    }
    return launch(std::integral_constant<int, 4>{}, jb0, nty);
}

}  // namespace wl

// wl_common.h -- shared host/device infrastructure of libwlhip (gfx950 only).
//
//  * Ctx        : the one stream everything is enqueued on, error text, launch counters, hipEvent timing
//  * Range      : an index box [lo,hi] and the generic range kernels (the native stand-in for the
//                 reference's `@loop ... over I in R`, src/util.jl:119-141): 64 lanes of a wavefront
//                 run along x (unit stride => coalesced), 4 rows per 256-thread workgroup, bounded
//                 grid-stride so that reductions produce a FIXED number of per-block partials
//                 (deterministic summation order, no float atomics).
//  * device math: median/quick/phi/... with the reference's Float32->Float64 promotions
//                 (src/Flow.jl:1-34); compiled with -ffp-contract=off like the oracle so that each
//                 operation rounds identically.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <cstdlib>

#include "../../include/wlhip.h"

namespace wl {

// ------------------------------------------------------------------------------------------ context
struct TimedEvt { hipEvent_t a, b; int64_t cells; };

// One communicator per process (z-slab neighbours + world collectives).  All buffers are DEVICE pointers and all
// operations are enqueued on ctx().stream (RCCL) or staged through pinned host memory (host-callback twin).
struct Comm {
    int rank = 0, size = 1;
    // collectives issued since the last wl_prof_reset (wl_prof_comm): all-reduces, exchange batches, send/recv pairs,
    // all-gathers, halo bytes sent, all-gather bytes contributed
    int64_t cnt[6] = {0, 0, 0, 0, 0, 0};
    int depth = 0;   // open group_begin()s
    virtual ~Comm() {}
    int allreduce(double *dev, int n, int op) { cnt[0] += 1; return do_allreduce(dev, n, op); }   // op: 0 sum, 1 max; in place
    int sendrecv(const void *send_lo, void *recv_lo, const void *send_hi, void *recv_hi, size_t bytes, int peer_lo, int peer_hi) {
        if (depth == 0) cnt[1] += 1;
        cnt[2] += (send_lo ? 1 : 0) + (send_hi ? 1 : 0);
        cnt[4] += (int64_t)bytes * ((send_lo ? 1 : 0) + (send_hi ? 1 : 0));
        return do_sendrecv(send_lo, recv_lo, send_hi, recv_hi, bytes, peer_lo, peer_hi);
    }
    int allgather(void *buf, size_t bytes_per_rank) { cnt[3] += 1; cnt[5] += (int64_t)bytes_per_rank; return do_allgather(buf, bytes_per_rank); }   // in place, rank r at r*bytes
    // several sendrecv calls issued as ONE batch (one latency, not one each)
    int group_begin() { if (depth++ == 0) cnt[1] += 1; return do_group_begin(); }
    int group_end() { --depth; return do_group_end(); }
    virtual int do_allreduce(double *dev, int n, int op) = 0;
    virtual int do_sendrecv(const void *send_lo, void *recv_lo, const void *send_hi, void *recv_hi, size_t bytes, int peer_lo,
                            int peer_hi) = 0;
    virtual int do_allgather(void *buf, size_t bytes_per_rank) = 0;
    virtual int do_group_begin() { return 0; }
    virtual int do_group_end() { return 0; }
};

// Mailbox all-reduce (csrc/wl_api.hip: wl_comm_mailbox): the scalars of a z-slab run -- one or two dot products per pcg!
// iteration, each a dependency of the next kernel -- are summed through a small block of pinned HOST memory shared by the
// ranks of the node (POSIX shared memory, registered with HIP): a rank posts its value with one system-scope store and
// reads the others' with one poll each, ~2 PCIe round trips instead of a general-purpose collective (kernel launch +
// ring / tree protocol) per 8 bytes.  Values are combined in RANK ORDER, so every rank computes bit-identical scalars.
struct MboxSlot { double v[4]; unsigned long long seq; unsigned long long pad[3]; };   // 64 B: one line per (parity, rank)
struct Mailbox {
    MboxSlot *host = nullptr, *dev = nullptr;   // [2][nranks] slots, host mapping and the device pointer to it
    size_t bytes = 0;
    unsigned long long seq = 0;                 // all-reduces issued so far (identical on every rank: SPMD)
    int *err_host = nullptr, *err_dev = nullptr;   // set by a kernel that gave up waiting for a peer
};

struct Ctx {
    Mailbox *mbox = nullptr;
    // wl_set_option (include/wlhip.h); keys 11, 12, 20, 21, 24, 25, 28, 29 are retired (WL_OPT_LIVE)
    int opt[32] = {1, 1, 1, 1, 0, 2, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 4, 16, 1, 1, 0, 0, 1, 1, 0, 0, 600, 1, 0, 0, 1, 1};
    double wall_khz = 0.0;             // rate of the device's wall clock (mailbox time-outs), read when the mailbox is made
    Comm *comm = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    int64_t launches[WL_K_COUNT] = {0};
    int64_t cells[WL_K_COUNT] = {0};
    int sweep = 0;                     // direction of the last marching launch (sweep_rev)
    int prof_class = -1;
    int64_t prof_min_cells = 0;
    std::vector<TimedEvt> evts;
    std::vector<hipEvent_t> pool;
    // halo exchange overlapped with interior compute (halo_begin / halo_end)
    int overlap = -1;                  // -1: read WL_OVERLAP on first use (default on), 0 off, 1 on
    hipStream_t cstream = nullptr;     // non-blocking comm stream (does not synchronise with the null stream)
    hipEvent_t ev_prod = nullptr, ev_halo = nullptr;
    bool halo_pending = false;
    int64_t n_overlapped = 0;          // stencil launches split around an exchange (wl_prof_overlapped)
    int64_t n_alloc = 0, alloc_bytes = 0;   // device + pinned-host allocations the library has made (wl_prof_allocs)
};
Ctx &ctx();
// every allocation of the library goes through these two (counted: a steady time step must not allocate, test/alloctest.jl)
inline hipError_t wl_dev_alloc(void **p, size_t n) { ctx().n_alloc += 1; ctx().alloc_bytes += (int64_t)n; return hipMalloc(p, n); }
inline hipError_t wl_host_alloc(void **p, size_t n, unsigned flags) { ctx().n_alloc += 1; ctx().alloc_bytes += (int64_t)n; return hipHostMalloc(p, n, flags); }
// live wl_set_option keys: 0-10, 13-19, 22, 23, 26, 27, 30, 31
constexpr unsigned WL_OPT_LIVE = 0xCCCFE7FFu;
int fail(int code, const char *what, const char *file, int line);

#define WL_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) return ::wl::fail((int)e__, #expr, __FILE__, __LINE__); \
    } while (0)
#define WL_TRY(...)                    \
    do {                               \
        int r__ = (__VA_ARGS__);       \
        if (r__ != 0) return r__;      \
    } while (0)
#define WL_REQUIRE(cond, msg)                                                  \
    do {                                                                       \
        if (!(cond)) return ::wl::fail(WL_E_ARG, msg, __FILE__, __LINE__);     \
    } while (0)

// RAII bracket: counts the launch, and records hipEvents around it when this class is being timed.
struct Prof {
    bool timed = false;
    hipEvent_t a{}, b{};
    int64_t ncell;
    Prof(int kclass, int64_t ncells);
    ~Prof();
};

// ------------------------------------------------------------------------------------------ grid
struct G {  // device-side copy of wl_grid (+ derived values)
    int D;
    int n[3];
    long s[3];
    long sc;
    int nzg;           // global z extent incl. ghosts (== n[2] when not decomposed)
    int kz0;           // global z index of local plane 0
    int zlo, zhi;      // owned local planes (inclusive)
    bool dist;         // z-slab of a decomposed array
    bool zring;        // z periodic across the slabs (ring of ranks): no z boundary anywhere
    __host__ __device__ int kg(int k) const { return k + kz0; }
    __host__ __device__ long at(int i, int j, int k) const { return (long)i + s[1] * (long)j + s[2] * (long)k; }
    long cells() const { return (long)n[0] * n[1] * n[2]; }
    long interior_cells() const {  // of the UNDECOMPOSED array (src/Poisson.jl:94 length(inside(r)))
        long c = 1;
        for (int d = 0; d < D; ++d) c *= (long)((d == 2 ? nzg : n[d]) - 2);
        return c;
    }
};
inline G mkG(const wl_grid *g) {
    G o;
    o.D = g->D;
    for (int d = 0; d < 3; ++d) { o.n[d] = g->n[d]; o.s[d] = g->s[d]; }
    o.sc = g->sc;
    if (g->D == 3 && g->nzg > 0) { o.nzg = g->nzg; o.kz0 = g->kz0; o.zlo = g->own_lo; o.zhi = g->own_hi; o.dist = true; o.zring = g->zring != 0; }
    else { o.nzg = g->n[2]; o.kz0 = 0; o.zlo = 0; o.zhi = g->n[2] - 1; o.dist = false; o.zring = false; }
    return o;
}
int check_grid(const wl_grid *g);

struct Range {
    int lo[3], hi[3];
    long count() const {
        long c = 1;
        for (int d = 0; d < 3; ++d) {
            if (hi[d] < lo[d]) return 0;
            c *= (long)(hi[d] - lo[d] + 1);
        }
        return c;
    }
};
// z extents of every range are expressed in GLOBAL plane numbers and clipped to the planes this rank owns
inline void clip_z(const G &g, Range &r, int glo, int ghi) {
    if (g.D < 3) { r.lo[2] = r.hi[2] = 0; return; }
    const int lo = glo - g.kz0, hi = ghi - g.kz0;
    r.lo[2] = lo > g.zlo ? lo : g.zlo;
    r.hi[2] = hi < g.zhi ? hi : g.zhi;   // may come out empty (hi < lo): launch_range skips it
}
inline Range r_inside(const G &g) {  // src/util.jl:47
    Range r;
    for (int d = 0; d < 3; ++d) {
        if (d < g.D) { r.lo[d] = 1; r.hi[d] = g.n[d] - 2; } else { r.lo[d] = r.hi[d] = 0; }
    }
    clip_z(g, r, 1, g.nzg - 2);
    return r;
}
inline Range r_whole(const G &g) {
    Range r;
    for (int d = 0; d < 3; ++d) { r.lo[d] = 0; r.hi[d] = g.n[d] - 1; }
    clip_z(g, r, 0, g.nzg - 1);
    return r;
}
// src/util.jl:180-182 slice(dims,i,j,low) with 0-based plane index `i0` and lower bound `low0`
// (i0 and low0 are GLOBAL indices along z)
inline Range r_slice(const G &g, int i0, int j, int low0) {
    Range r;
    for (int d = 0; d < 3; ++d) {
        if (d >= g.D) { r.lo[d] = r.hi[d] = 0; }
        else if (d == j) { r.lo[d] = r.hi[d] = i0; }
        else { r.lo[d] = low0; r.hi[d] = g.n[d] - 1; }
    }
    if (g.D == 3) {
        if (j == 2) clip_z(g, r, i0, i0);
        else clip_z(g, r, low0, g.nzg - 1);
    }
    return r;
}

// ------------------------------------------------------------------------------------------ launch
constexpr int WL_BX = 64;    // lanes along the fast axis = one wavefront
constexpr int WL_BY = 4;     // rows per workgroup
constexpr int WL_GRID = 4096;   // default grid size of the marching kernels (256 CUs x 16 workgroups)
constexpr int WL_MAXB = 16384;  // hard grid cap = max number of reduction partials per value (scratch is sized for it)

struct Tiling {
    int a, b, c;       // axis permutation: a = fast axis (first axis with extent > 1), c = marching axis
    int na, nb, nc;    // extents along a, b, c
    int nta;           // tiles along a
    int tpp;           // tiles per (a,b) plane = nta * ceil(nb / WL_BY)
    int nchunk;        // the marching axis is cut into nchunk pieces of `clen` planes
    int clen;
    int nblk;          // launch grid = min(tpp, cap) * nchunk   (multiple of 8 whenever possible)
    int ptb;           // plane tiles are strided by this many blocks (== min(tpp, cap))
    int lo[3];
};
// Mapping (gfx950): a workgroup owns one 64x4 tile of the (a,b) plane and MARCHES along c, so the c-1/c
// stencil operands it needs were brought into its CU's L1/L2 by its own previous iteration; the c axis is cut
// into chunks only to create enough workgroups (>= ~2048) to fill 256 CUs.  Hardware deals workgroups
// round-robin over the 8 XCDs (b and b+8 share an L2), so the logical block id is un-swizzled to give every
// XCD a CONTIGUOUS range of plane tiles: neighbouring tiles then share their halo rows through one L2.
inline Tiling mk_tiling(const Range &R) {
    Tiling t;
    int ext[3] = {R.hi[0] - R.lo[0] + 1, R.hi[1] - R.lo[1] + 1, R.hi[2] - R.lo[2] + 1};
    t.a = 0;
    if (ext[0] == 1) t.a = (ext[1] > 1) ? 1 : (ext[2] > 1 ? 2 : 0);
    t.b = (t.a + 1) % 3;
    t.c = (t.a + 2) % 3;
    if (t.b > t.c) { int x = t.b; t.b = t.c; t.c = x; }
    t.na = ext[t.a]; t.nb = ext[t.b]; t.nc = ext[t.c];
    t.nta = (t.na + WL_BX - 1) / WL_BX;
    t.tpp = t.nta * ((t.nb + WL_BY - 1) / WL_BY);
    t.ptb = ((t.tpp + 7) / 8) * 8;              // multiple of 8 so the XCD un-swizzle applies (idle tail blocks)
    if (t.ptb > WL_GRID) t.ptb = WL_GRID;
    int want = WL_GRID / t.ptb;                 // chunks that keep the grid <= WL_GRID
    if (want < 1) want = 1;
    if (want > t.nc) want = t.nc;
    t.clen = (t.nc + want - 1) / want;
    if (t.clen < 1) t.clen = 1;
    t.nchunk = (t.nc + t.clen - 1) / t.clen;
    t.nblk = t.ptb * t.nchunk;
    for (int d = 0; d < 3; ++d) t.lo[d] = R.lo[d];
    return t;
}
inline int grid_for(const Tiling &t) { return t.nblk; }

// logical block id: XCD x (= blockIdx % 8) receives the contiguous range [x*nblk/8, (x+1)*nblk/8)
__device__ inline int logical_block(int nblk) {
    const int b = blockIdx.x;
    return (nblk & 7) ? b : (b & 7) * (nblk >> 3) + (b >> 3);
}

#define WL_TILE_LOOP(t, BODY)                                                              \
    {                                                                                      \
        const int tx__ = threadIdx.x & (WL_BX - 1), ty__ = threadIdx.x / WL_BX;            \
        const int lb__ = logical_block(t.nblk);                                            \
        const int ch__ = lb__ / t.ptb, p0__ = lb__ - ch__ * t.ptb;                         \
        const int c0__ = ch__ * t.clen, c1__ = min(t.nc, c0__ + t.clen);                   \
        for (int pt__ = p0__; pt__ < t.tpp; pt__ += t.ptb) {                               \
            const int ta__ = pt__ % t.nta, tb__ = pt__ / t.nta;                            \
            const int ia__ = ta__ * WL_BX + tx__, ib__ = tb__ * WL_BY + ty__;              \
            if (ia__ < t.na && ib__ < t.nb) {                                              \
                int idx[3];                                                                \
                idx[t.a] = t.lo[t.a] + ia__;                                               \
                idx[t.b] = t.lo[t.b] + ib__;                                               \
                for (int ic__ = c0__; ic__ < c1__; ++ic__) {                               \
                    idx[t.c] = t.lo[t.c] + ic__;                                           \
                    BODY                                                                   \
                }                                                                          \
            }                                                                              \
        }                                                                                  \
    }

template <class F>
__global__ __launch_bounds__(WL_BX *WL_BY) void k_range(Tiling t, F f) {
    WL_TILE_LOOP(t, f(idx[0], idx[1], idx[2]);)
}
// the same launch, left at once by every workgroup unless *flag (a device flag written by an earlier kernel of the stream)
template <class F>
__global__ __launch_bounds__(WL_BX *WL_BY) void k_range_if(Tiling t, const int *flag, F f) {
    if (!*flag) return;
    WL_TILE_LOOP(t, f(idx[0], idx[1], idx[2]);)
}

enum RedOp { RED_SUM = 0, RED_MAX = 1 };

__device__ inline double wave_red(double v, int op) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_down(v, o, 64);
        v = (op == RED_SUM) ? v + w : (w > v ? w : v);
    }
    return v;
}
// block reduction of NV values per thread over NW wavefronts; result valid in thread 0
template <int NV, int NW = WL_BY>
__device__ inline void block_red(double (&v)[NV], int op) {
    __shared__ double sm[NV][NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double x = wave_red(v[q], op);
        if (lane == 0) sm[q][w] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double x = sm[q][0];
            for (int i = 1; i < NW; ++i) x = (op == RED_SUM) ? x + sm[q][i] : (sm[q][i] > x ? sm[q][i] : x);
            v[q] = x;
        }
    }
    __syncthreads();
}

// range kernel whose functor accumulates into NV per-thread doubles: f(i,j,k,acc).
// partials[q*gridDim.x + blockIdx.x] receives the block's reduction of acc[q].
template <int NV, class F>
__global__ __launch_bounds__(WL_BX *WL_BY) void k_range_red(Tiling t, F f, double *partials, int op, double init) {
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = init;
    WL_TILE_LOOP(t, f(idx[0], idx[1], idx[2], acc);)
    block_red<NV>(acc, op);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) partials[(long)q * gridDim.x + blockIdx.x] = acc[q];
    }
}

// Final stage: ONE workgroup (1024 threads: up to 16384 partials per value, 128 KB, in ~16 loads per thread) reduces
// `np` partials per value in a fixed order, then runs the scalar epilogue `fin(vals)` in thread 0 (device-resident
// solver scalars: no host round trip).
constexpr int WL_FIN_T = 1024;
template <int NV, class FIN>
__global__ __launch_bounds__(WL_FIN_T) void k_finalize(const double *partials, int np, int op, double init, FIN fin) {
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double a = init;
        for (int i = threadIdx.x; i < np; i += WL_FIN_T) {
            double w = partials[(long)q * np + i];
            a = (op == RED_SUM) ? a + w : (w > a ? w : a);
        }
        acc[q] = a;
    }
    block_red<NV, WL_FIN_T / 64>(acc, op);
    if (threadIdx.x == 0) fin(acc);
}
// distributed variant: local reduction -> red[], (all-reduce over ranks), then the scalar epilogue
template <int NV>
__global__ __launch_bounds__(WL_FIN_T) void k_reduce_only(const double *partials, int np, int op, double init, double *red) {
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double a = init;
        for (int i = threadIdx.x; i < np; i += WL_FIN_T) {
            double w = partials[(long)q * np + i];
            a = (op == RED_SUM) ? a + w : (w > a ? w : a);
        }
        acc[q] = a;
    }
    block_red<NV, WL_FIN_T / 64>(acc, op);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[q] = acc[q];
    }
}
template <class FIN> __global__ void k_apply(const double *red, FIN fin) { fin(red); }

// local reduction (exactly k_reduce_only) + the exchange through the mailbox: red[q] = op over the ranks, in rank order.
// Slot (seq & 1, rank) is written by its owner only; a rank can be at most one all-reduce ahead of the slowest one (it
// needs that rank's value of the current round to finish it), so two parities suffice.  A wait is bounded in WALL-CLOCK
// time (wl_set_option(26) seconds on the device's constant-rate clock, default 600; 0 = wait for ever, like a collective
// would): a lane that gives up raises the error flag (the host turns it into an error at its next synchronisation) and the
// values become NaN -- the grid always drains.  Ranks may legitimately be apart by the length of rank-asymmetric host work
// (a first-call compile, geometry, file output): the default leaves minutes for that.
constexpr int WL_MBOX_MAXRANKS = 64;
template <int NV>
__global__ __launch_bounds__(WL_FIN_T) void k_reduce_mbox(const double *partials, int np, int op, double init, double *red, MboxSlot *mb,
                                                          int rank, int nranks, unsigned long long seq, int *err, long long tick_limit) {
    double acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double a = init;
        for (int i = threadIdx.x; i < np; i += WL_FIN_T) {
            double w = partials[(long)q * np + i];
            a = (op == RED_SUM) ? a + w : (w > a ? w : a);
        }
        acc[q] = a;
    }
    block_red<NV, WL_FIN_T / 64>(acc, op);
    __shared__ double got[WL_MBOX_MAXRANKS][NV];
    MboxSlot *slots = mb + (size_t)(seq & 1ull) * (size_t)nranks;
    if (threadIdx.x == 0) {   // post
        MboxSlot *mine = slots + rank;
#pragma unroll
        for (int q = 0; q < NV; ++q) __hip_atomic_store(&mine->v[q], acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mine->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if ((int)threadIdx.x < nranks) {   // collect: one lane per peer (the own slot included: same path, same order)
        MboxSlot *s = slots + threadIdx.x;
        const long long t0 = wall_clock64();
        bool ok = true;
        while (__hip_atomic_load(&s->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (tick_limit > 0 && wall_clock64() - t0 > tick_limit) { ok = false; break; }
            __builtin_amdgcn_s_sleep(16);
        }
#pragma unroll
        for (int q = 0; q < NV; ++q)
            got[threadIdx.x][q] = ok ? __hip_atomic_load(&s->v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __builtin_nan("");
        if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double a = got[0][q];
            for (int r = 1; r < nranks; ++r) a = (op == RED_SUM) ? a + got[r][q] : (got[r][q] > a ? got[r][q] : a);
            red[q] = a;
        }
    }
}
// local partials -> red[] = the value over all ranks: through the mailbox when there is one, else k_reduce_only + the
// communicator's all-reduce (RCCL / host callbacks)
template <int NV> inline int reduce_allreduce(const double *partials, int np, int op, double init, double *red) {
    Comm *cm = ctx().comm;
    Mailbox *mb = ctx().mbox;
    if (mb && cm->size <= WL_MBOX_MAXRANKS) {
        cm->cnt[0] += 1;
        mb->seq += 1;
        hipLaunchKernelGGL((k_reduce_mbox<NV>), dim3(1), dim3(WL_FIN_T), 0, ctx().stream, partials, np, op, init, red, mb->dev, cm->rank,
                           cm->size, mb->seq, mb->err_dev, (long long)((double)(ctx().opt[26] > 0 ? ctx().opt[26] : 0) * ctx().wall_khz * 1e3));
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((k_reduce_only<NV>), dim3(1), dim3(WL_FIN_T), 0, ctx().stream, partials, np, op, init, red);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return cm->allreduce(red, NV, op);
}

template <class F>
inline int launch_range(int kclass, const Range &R, F f) {
    if (R.count() <= 0) return 0;
    Tiling t = mk_tiling(R);
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_range<F>), dim3(grid_for(t)), dim3(WL_BX * WL_BY), 0, ctx().stream, t, f);
    return (int)hipGetLastError();
}
template <class F>
inline int launch_range_if(int kclass, const Range &R, const int *flag, F f) {
    if (R.count() <= 0) return 0;
    Tiling t = mk_tiling(R);
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_range_if<F>), dim3(grid_for(t)), dim3(WL_BX * WL_BY), 0, ctx().stream, t, flag, f);
    return (int)hipGetLastError();
}
// returns the number of partials per value through *np
template <int NV, class F>
inline int launch_range_red(int kclass, const Range &R, F f, double *partials, int op, double init, int *np) {
    if (R.count() <= 0) { *np = 0; return 0; }  // empty local range (a rank that owns no such plane)
    Tiling t = mk_tiling(R);
    const int nb = grid_for(t);
    *np = nb;
    Prof p(kclass, R.count());
    hipLaunchKernelGGL((k_range_red<NV, F>), dim3(nb), dim3(WL_BX * WL_BY), 0, ctx().stream, t, f, partials, op, init);
    return (int)hipGetLastError();
}
// ---- the ghost SHELL of an array: the cells with at least one index 0 or n-1.  The reference reduces whole arrays
// (maximum(a.σ), Flow.jl:174; z⋅ϵ, Poisson.jl:131), and two of them carry non-zero ghost values (σ keeps conv_diff!'s flux
// scratch in its top ghost cells, ϵ its periodic copies), so those reductions visit the shell too -- surface work.
// Plane p = 2*d + side (side 0: index 0, side 1: index n_d-1) covers, exactly once over all planes, the cells with
// idx[d] on that side, idx[e] in 1..n_e-2 for e < d and idx[e] anywhere for e > d.  z-slab runs: a rank visits the planes
// it owns; the two z ghost planes belong to the first / last rank (a periodic ring: only the top one, which the last
// rank holds as its upper halo plane -- the bottom one is never written by anybody and holds zeros in the reference).
struct Shell {
    G g;
    int kmin, kmax;      // local plane range of the x / y planes
    int zbot, ztop;      // local index of the global z ghost planes 0 / nzg-1 when this rank visits them, else -1
    int plane[6];        // the planes of this launch (2*d + side)
    int np;
};
inline Shell mk_shell(const G &g, int planemask) {
    Shell s;
    s.g = g;
    if (g.D < 3) { s.kmin = s.kmax = 0; s.zbot = s.ztop = -1; }
    else {
        s.zbot = (!g.zring && g.kg(g.zlo) == 0) ? g.zlo : -1;
        s.ztop = (g.kg(g.zhi) == g.nzg - 1) ? g.zhi : ((g.zring && g.kg(g.zhi) == g.nzg - 2 && g.zhi + 1 <= g.n[2] - 1) ? g.zhi + 1 : -1);
        s.kmin = g.zlo;
        s.kmax = s.ztop >= 0 ? s.ztop : g.zhi;
    }
    s.np = 0;
    for (int p = 0; p < 2 * g.D; ++p) {
        if (!((planemask >> p) & 1)) continue;
        if (p == 4 && s.zbot < 0) continue;
        if (p == 5 && s.ztop < 0) continue;
        s.plane[s.np++] = p;
    }
    for (int q = s.np; q < 6; ++q) s.plane[q] = 0;
    return s;
}
// the cells of plane blockIdx.y, first remaining axis fastest (y / z planes: consecutive lanes = consecutive x)
#define WL_SHELL_LOOP(s, BODY)                                                                              \
    {                                                                                                        \
        const int p__ = s.plane[blockIdx.y], d__ = p__ >> 1, side__ = p__ & 1;                               \
        const int e1__ = d__ == 0 ? 1 : 0, e2__ = d__ == 2 ? 1 : 2;                                          \
        int lo__[3], hi__[3];                                                                                \
        for (int e = 0; e < 3; ++e) {                                                                        \
            if (e >= s.g.D) { lo__[e] = hi__[e] = 0; }                                                       \
            else if (e == 2) { lo__[e] = s.kmin; hi__[e] = s.kmax; }                                         \
            else if (e < d__) { lo__[e] = 1; hi__[e] = s.g.n[e] - 2; }                                       \
            else { lo__[e] = 0; hi__[e] = s.g.n[e] - 1; }                                                    \
        }                                                                                                    \
        if (d__ == 2) lo__[2] = hi__[2] = side__ ? s.ztop : s.zbot;                                          \
        else lo__[d__] = hi__[d__] = side__ ? s.g.n[d__] - 1 : 0;                                            \
        const long ext1__ = hi__[e1__] - lo__[e1__] + 1, ext2__ = (s.g.D > 2 || d__ == 2) ? hi__[e2__] - lo__[e2__] + 1 : 1; \
        for (long pos__ = (long)blockIdx.x * 256 + threadIdx.x; pos__ < ext1__ * ext2__; pos__ += (long)gridDim.x * 256) { \
            const long q2__ = pos__ / ext1__, q1__ = pos__ - q2__ * ext1__;                                  \
            int idx[3] = {lo__[0], lo__[1], lo__[2]};                                                        \
            idx[e1__] = lo__[e1__] + (int)q1__;                                                              \
            if (s.g.D > 2) idx[e2__] = lo__[e2__] + (int)q2__;                                               \
            BODY                                                                                             \
        }                                                                                                    \
    }
template <class F> __global__ __launch_bounds__(256) void k_shell(Shell s, F f) { WL_SHELL_LOOP(s, f(idx[0], idx[1], idx[2]);) }
template <class F> __global__ __launch_bounds__(256) void k_shell_red(Shell s, F f, double *partials, int op, double init) {
    double acc[1] = {init};
    WL_SHELL_LOOP(s, f(idx[0], idx[1], idx[2], acc);)
    block_red<1>(acc, op);
    if (threadIdx.x == 0) partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = acc[0];
}
inline int shell_blocks(const G &g, int cap) {
    long big = 1;
    for (int d = 0; d < g.D; ++d) {
        long c = 1;
        for (int e = 0; e < g.D; ++e) if (e != d) c *= (long)g.n[e];
        big = c > big ? c : big;
    }
    long nb = (big + 255) / 256;
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}
// f(i, j, k) over the chosen planes of the shell (k local)
template <class F> inline int launch_shell(int kclass, const G &g, int planemask, F f) {
    const Shell s = mk_shell(g, planemask);
    if (s.np == 0) return 0;
    Prof p(kclass, 0);
    hipLaunchKernelGGL((k_shell<F>), dim3(shell_blocks(g, 256), s.np), dim3(256), 0, ctx().stream, s, f);
    return (int)hipGetLastError();
}
// f(i, j, k, acc) reduced over the chosen planes: *np partials (<= 6 * cap) are written from partials[0] on
template <class F> inline int launch_shell_red(int kclass, const G &g, int planemask, F f, double *partials, int op, double init, int *np,
                                               int cap = 256) {
    const Shell s = mk_shell(g, planemask);
    *np = 0;
    if (s.np == 0) return 0;
    const int nb = shell_blocks(g, cap);
    *np = nb * s.np;
    Prof p(kclass, 0);
    hipLaunchKernelGGL((k_shell_red<F>), dim3(nb, s.np), dim3(256), 0, ctx().stream, s, f, partials, op, init);
    return (int)hipGetLastError();
}

// `dist`: the reduced quantity lives on a z-slab decomposition -> all-reduce over the ranks before `fin`.
// `red`: device scratch of >= NV doubles.  fin(const double *vals) runs in one device thread.
template <int NV, class FIN>
inline int launch_finalize(bool dist, const double *partials, int np, int op, double init, double *red, FIN fin) {
    Prof p(WL_K_SCALAR, 0);
    Comm *cm = ctx().comm;
    if (!dist || !cm || cm->size == 1) {
        hipLaunchKernelGGL((k_finalize<NV, FIN>), dim3(1), dim3(WL_FIN_T), 0, ctx().stream, partials, np, op, init, fin);
        return (int)hipGetLastError();
    }
    int rc = reduce_allreduce<NV>(partials, np, op, init, red);
    if (rc) return rc;
    hipLaunchKernelGGL((k_apply<FIN>), dim3(1), dim3(1), 0, ctx().stream, (const double *)red, fin);
    return (int)hipGetLastError();
}
// fill the z-halo planes of `ncomp` components from the neighbouring ranks (no-op when not decomposed)
template <class T> inline int halo_exchange(const G &g, T *a, int ncomp, int depth) {
    Comm *cm = ctx().comm;
    if (!g.dist || !cm || cm->size == 1) return 0;
    const size_t bytes = (size_t)depth * (size_t)g.s[2] * sizeof(T);
    const bool lo = g.zring || cm->rank > 0, hi = g.zring || cm->rank < cm->size - 1;
    const int plo = lo ? (cm->rank - 1 + cm->size) % cm->size : -1, phi = hi ? (cm->rank + 1) % cm->size : -1;
    int rc = cm->group_begin();
    if (rc) return rc;
    for (int c = 0; c < ncomp; ++c) {
        T *b = a + (long)c * g.sc;
        rc = cm->sendrecv(lo ? b + (long)g.zlo * g.s[2] : nullptr, lo ? b + (long)(g.zlo - depth) * g.s[2] : nullptr,
                          hi ? b + (long)(g.zhi - depth + 1) * g.s[2] : nullptr,
                          hi ? b + (long)(g.zhi + 1) * g.s[2] : nullptr, bytes, plo, phi);
        if (rc) { (void)cm->group_end(); return rc; }
    }
    return cm->group_end();
}

// ---- overlapped form: halo_begin enqueues the exchange on the comm stream (after everything enqueued so far on the
// compute stream), halo_end makes the compute stream wait for it.  Kernels launched in between run concurrently with
// the transfer and must neither read the halo planes of `a` nor write its outermost owned planes.  The comm stream
// is non-blocking, so the two streams are ordered by these two events only.  Every RCCL call of the process is still
// totally ordered (exchange -> halo_end -> later all-reduces on the compute stream -> next halo_begin): the
// communicator never sees two operations in flight.  WL_OVERLAP=0 in the environment restores in-stream exchanges.
inline bool overlap_on() {
    Ctx &c = ctx();
    if (c.overlap < 0) {
        const char *e = getenv("WL_OVERLAP");
        c.overlap = (e && e[0] == '0') ? 0 : 1;
    }
    return c.overlap == 1 && c.comm && c.comm->size > 1;
}
// b (optional): a second array in the SAME batch (project! on z-slabs: the plane of u that div reads and the planes of x that
// residual! reads travel together: one latency, one event pair)
template <class T> inline int halo_begin2(const G &ga, T *a, int ncompa, int deptha, const G &gb, T *b, int ncompb, int depthb) {
    Ctx &c = ctx();
    if (!ga.dist || !c.comm || c.comm->size == 1) return 0;
    auto both = [&]() -> int {                  // one group: one exchange batch
        int rc = c.comm->group_begin();
        if (rc) return rc;
        rc = halo_exchange<T>(ga, a, ncompa, deptha);
        if (!rc && b) rc = halo_exchange<T>(gb, b, ncompb, depthb);
        const int rce = c.comm->group_end();
        return rc ? rc : rce;
    };
    if (!overlap_on()) return both();
    if (!c.cstream) {
        WL_HIP(hipStreamCreateWithFlags(&c.cstream, hipStreamNonBlocking));
        WL_HIP(hipEventCreateWithFlags(&c.ev_prod, hipEventDisableTiming));
        WL_HIP(hipEventCreateWithFlags(&c.ev_halo, hipEventDisableTiming));
    }
    WL_HIP(hipEventRecord(c.ev_prod, c.stream));
    WL_HIP(hipStreamWaitEvent(c.cstream, c.ev_prod, 0));
    hipStream_t compute = c.stream;
    c.stream = c.cstream;                       // the transport enqueues on ctx().stream
    const int rc = both();
    c.stream = compute;
    if (rc) return rc;
    WL_HIP(hipEventRecord(c.ev_halo, c.cstream));
    c.halo_pending = true;
    return 0;
}
template <class T> inline int halo_begin(const G &g, T *a, int ncomp, int depth) {
    return halo_begin2<T>(g, a, ncomp, depth, g, (T *)nullptr, 0, 0);
}
inline int halo_end() {
    Ctx &c = ctx();
    if (!c.halo_pending) return 0;
    c.halo_pending = false;
    WL_HIP(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
    return 0;
}

// Consecutive marching kernels sweep in OPPOSITE directions (wl_set_option(30)): each XCD keeps its range of tiles -- its slab
// of the z axis, the same for every kernel -- but starts where the kernel before it stopped, on the lines that kernel has
// just read or written and that still sit in the XCD's L2 and in the 256 MB Infinity Cache (a 512^3 Float32 array is
// 537 MB: without the reversal every kernel begins on the lines that were evicted first).  The tile a workgroup takes
// changes, the slot its reduction partial goes to stays that tile's: sums and fields are bit-identical either way.
inline int sweep_rev() {
    Ctx &c = ctx();
    if (!c.opt[30]) return 0;
    c.sweep ^= 1;
    return c.sweep;
}
// logical tile of physical workgroup b, and the slot of that tile's partial (= the workgroup that takes it when rev = 0)
__device__ __forceinline__ void tile_of(int b, int nblk, int rev, int &lb, int &pslot) {
    if (nblk & 7) { lb = b; pslot = b; return; }
    const int per = nblk >> 3, q = b >> 3;
    const int qq = rev ? per - 1 - q : q;
    lb = (b & 7) * per + qq;
    pslot = (qq << 3) | (b & 7);
}

// ---- neighbour-lane exchange by DPP wave shifts (gfx9 family: wave_shr:1 / wave_shl:1): one VALU move per 32 bits,
// no trip through the LDS crossbar (ds_bpermute) and no lgkmcnt wait behind it.  lane_up1(x): lane i receives lane i-1's
// x (== __shfl_up(x,1)); lane_dn1(x): lane i receives lane i+1's (== __shfl_down(x,1)).  Lane 0 / lane 63 (and lanes
// whose source lane is inactive) receive 0: callers overwrite those lanes with the value beyond the wavefront.
__device__ __forceinline__ int dpp_up1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ int dpp_dn1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }   // wave_shl:1
__device__ __forceinline__ float lane_up1(float x) { return __builtin_bit_cast(float, dpp_up1(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ float lane_dn1(float x) { return __builtin_bit_cast(float, dpp_dn1(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ double lane_up1(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)dpp_up1((int)(unsigned)b), hi = (unsigned)dpp_up1((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lane_dn1(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)dpp_dn1((int)(unsigned)b), hi = (unsigned)dpp_dn1((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------------------------------ device math
template <class T> struct Lim;
template <> struct Lim<float> { static constexpr float eps = FLT_EPSILON; };
template <> struct Lim<double> { static constexpr double eps = DBL_EPSILON; };

// src/Flow.jl:25-34: the reference's branchy median(a,b,c) returns the mathematical median for every
// non-NaN input (ties return one of the equal values), so it is evaluated branch-free: one v_med3_f32 for
// Float32, min/max for Float64.  Only the sign of a zero result can differ from the branchy form.
__device__ inline float median3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ inline double median3(double a, double b, double c) {
    return fmax(fmin(a, b), fmin(fmax(a, b), c));
}
// x/6 correctly rounded WITHOUT the ~10-instruction IEEE division sequence (Markstein): q0 = rn(x*z), z = rn(1/6);
// r = x - 6*q0 exactly (fma); q = rn(q0 + r*z).  The correction term is off by < 2^-24 ulp(q), and x/6 can never be
// closer than ulp(q)/6 to a rounding midpoint (x is an even, 6*midpoint an odd multiple of ulp(q)), so q == rn(x/6)
// for every x whose quotient is a normal number: bit-identical to the oracle's true division.
__device__ __forceinline__ float div6(float x) {
    const float z = 1.0f / 6.0f;
    const float q0 = x * z;
    const float r = __fmaf_rn(-6.0f, q0, x);
    return __fmaf_rn(r, z, q0);
}
__device__ __forceinline__ double div6(double x) {
    const double z = 1.0 / 6.0;
    const double q0 = x * z;
    const double r = __fma_rn(-6.0, q0, x);
    return __fma_rn(r, z, q0);
}
// src/Flow.jl:4
template <class T> __device__ inline T quick(T u, T c, T d) {
    T a1 = div6(((T)5 * c + (T)2 * d) - u);
    T a2 = median3((T)10 * c - (T)9 * u, c, d);
    return median3(a1, c, a2);
}
// src/Flow.jl:3 : T add, then *0.5 in Float64
// (the *0.5 is done in T before widening: scaling by a power of two is exact, so the Float64 value is the same)
template <class T> __device__ inline double phi(const T *f, long I, long s) { return (double)((T)(f[I] + f[I - s]) * (T)0.5); }
// src/Flow.jl:6
// (the upwind triple is selected first, then ONE quick is evaluated: same value, half the arithmetic)
template <class T> __device__ inline double phiu(const T *f, long I, long s, double u) {
    const bool up = u > 0;
    const T c1 = f[I - s], c0 = f[I];
    const T uu = up ? f[I - 2 * s] : f[I + s];
    return u * (double)quick<T>(uu, up ? c1 : c0, up ? c0 : c1);
}
// src/Flow.jl:7
template <class T> __device__ inline double phiuP(const T *f, long Ip, long I, long s, double u) {
    const bool up = u > 0;
    const T c1 = f[I - s], c0 = f[I];
    const T uu = up ? f[Ip] : f[I + s];
    return u * (double)quick<T>(uu, up ? c1 : c0, up ? c0 : c1);
}
// src/Flow.jl:8
template <class T> __device__ inline double phiuL(const T *f, long I, long s, double u) {
    return u > 0 ? u * phi<T>(f, I, s) : u * (double)quick<T>(f[I + s], f[I], f[I - s]);
}
// src/Flow.jl:9
template <class T> __device__ inline double phiuR(const T *f, long I, long s, double u) {
    return u < 0 ? u * phi<T>(f, I, s) : u * (double)quick<T>(f[I - 2 * s], f[I - s], f[I]);
}
// src/Poisson.jl:69-75 mult(I,L,D,x)
template <class T, int D> __device__ inline T mult1(const G &g, const T *L, const T *Dg, const T *x, long I) {
    T s = x[I] * Dg[I];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long sd = g.s[d];
        const T *Ld = L + (long)d * g.sc;
        s += x[I - sd] * Ld[I] + x[I + sd] * Ld[I + sd];
    }
    return s;
}

// mult with the diagonal RECOMPUTED from the six face coefficients the stencil loads anyway
// (D[I] = -sum_d (L[I,d]+L[I+d,d]), src/Poisson.jl:48-54: same operations as set_diag!, so the same bits as
// the stored D as long as update! ran after the last change of L).  Saves one array read per cell.
template <class T, int D> __device__ inline T mult1r(const G &g, const T *L, const T *x, long I) {
    T lo[D], hi[D];
_Pragma("unroll")
    for (int d = 0; d < D; ++d) {
        const T *Ld = L + (long)d * g.sc;
        lo[d] = Ld[I];
        hi[d] = Ld[I + g.s[d]];
    }
    T dg = 0;
_Pragma("unroll")
    for (int d = 0; d < D; ++d) dg -= (lo[d] + hi[d]);
    T s = x[I] * dg;
_Pragma("unroll")
    for (int d = 0; d < D; ++d) s += x[I - g.s[d]] * lo[d] + x[I + g.s[d]] * hi[d];
    return s;
}

// dispatch on (dtype, D)
#define WL_DISPATCH(t, D, CALL)                                          \
    do {                                                                 \
        if ((t) == WL_F32 && (D) == 3) { using T = float; constexpr int DD = 3; return CALL; }   \
        if ((t) == WL_F32 && (D) == 2) { using T = float; constexpr int DD = 2; return CALL; }   \
        if ((t) == WL_F64 && (D) == 3) { using T = double; constexpr int DD = 3; return CALL; }  \
        if ((t) == WL_F64 && (D) == 2) { using T = double; constexpr int DD = 2; return CALL; }  \
        return ::wl::fail(WL_E_ARG, "unsupported dtype/dimension", __FILE__, __LINE__);          \
    } while (0)

}  // namespace wl

// wl_api.hip -- C ABI (include/wlhip.h) + host orchestration: Vcycle!, solver!, project!, mom_step!.
#include <cstdarg>
#include <cstring>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <rccl/rccl.h>

#include "wl_ops.h"
#include "wl_coarse.h"
#include "wl_measure.h"

namespace wl {

// ------------------------------------------------------------------------------------------ communicators
// RCCL over xGMI: neighbour send/recv pairs and world collectives, enqueued on the compute stream (so they are
// ordered with the kernels that produce/consume the planes without host synchronisation).
struct RcclComm : Comm {
    ncclComm_t nc = nullptr;    // world collectives (all-reduce, all-gather), always on the compute stream
    ncclComm_t nch = nullptr;   // halo send/recv: a split of `nc`, so that exchanges running on the comm stream (halo_begin)
                                // never share a communicator with a collective on the compute stream; == nullptr: use nc
    ~RcclComm() override {
        if (nch) ncclCommDestroy(nch);
        if (nc) ncclCommDestroy(nc);
    }
    int chk(ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return 0;
        ctx().err = std::string("rccl: ") + what + ": " + ncclGetErrorString(r);
        return WL_E_STATE;
    }
    int do_allreduce(double *dev, int n, int op) override {
        return chk(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, nc, ctx().stream), "allreduce");
    }
    int do_sendrecv(const void *slo, void *rlo, const void *shi, void *rhi, size_t bytes, int plo, int phi) override {
        int rc = chk(ncclGroupStart(), "groupStart");
        if (rc) return rc;
        // order: my upper planes go up and fill the lower halo of peer_hi, whose first receive from me is its recv_lo, ...
        // (with a 2-rank ring both peers are the same rank: k-th send must meet the k-th receive of that peer)
        ncclComm_t h = nch ? nch : nc;
        // every call is checked; on an error the group is still closed (an open group would swallow every later call)
        if (!rc && shi) rc = chk(ncclSend(shi, bytes, ncclChar, phi, h, ctx().stream), "send(up)");
        if (!rc && rlo) rc = chk(ncclRecv(rlo, bytes, ncclChar, plo, h, ctx().stream), "recv(from below)");
        if (!rc && slo) rc = chk(ncclSend(slo, bytes, ncclChar, plo, h, ctx().stream), "send(down)");
        if (!rc && rhi) rc = chk(ncclRecv(rhi, bytes, ncclChar, phi, h, ctx().stream), "recv(from above)");
        const std::string first = ctx().err;
        const int rce = chk(ncclGroupEnd(), "groupEnd(sendrecv)");
        if (rc) { ctx().err = first; return rc; }
        return rce;
    }
    int do_allgather(void *buf, size_t bytes) override {
        return chk(ncclAllGather((const char *)buf + (size_t)rank * bytes, buf, bytes, ncclChar, nc, ctx().stream), "allgather");
    }
    int do_group_begin() override { return chk(ncclGroupStart(), "groupStart"); }   // NCCL groups nest
    int do_group_end() override { return chk(ncclGroupEnd(), "groupEnd"); }
};

// Host-callback twin (tests: 2+ ranks sharing one GPU, transport = torch.distributed gloo).  Every operation
// synchronises the stream and stages through pinned host memory: correct, not fast.
struct HostComm : Comm {
    wl_host_sendrecv_fn sr; wl_host_allreduce_fn ar; wl_host_allgather_fn ag; void *user;
    char *pin = nullptr; size_t cap = 0;
    ~HostComm() override { if (pin) (void)hipHostFree(pin); }
    int need(size_t b) {
        if (b <= cap) return 0;
        if (pin) (void)hipHostFree(pin);
        cap = b * 2;
        return (int)wl_host_alloc((void **)&pin, cap, hipHostMallocDefault);
    }
    int do_allreduce(double *dev, int n, int op) override {
        double v[8];
        WL_HIP(hipMemcpyAsync(v, dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        if (ar(user, v, n, op)) return fail(WL_E_STATE, "host allreduce callback failed", __FILE__, __LINE__);
        WL_HIP(hipMemcpyAsync(dev, v, sizeof(double) * n, hipMemcpyHostToDevice, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        return 0;
    }
    int do_sendrecv(const void *slo, void *rlo, const void *shi, void *rhi, size_t bytes, int plo, int phi) override {
        WL_TRY(need(4 * bytes));
        char *hs_lo = pin, *hr_lo = pin + bytes, *hs_hi = pin + 2 * bytes, *hr_hi = pin + 3 * bytes;
        if (slo) WL_HIP(hipMemcpyAsync(hs_lo, slo, bytes, hipMemcpyDeviceToHost, ctx().stream));
        if (shi) WL_HIP(hipMemcpyAsync(hs_hi, shi, bytes, hipMemcpyDeviceToHost, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        if (sr(user, slo ? hs_lo : nullptr, slo ? hr_lo : nullptr, shi ? hs_hi : nullptr, shi ? hr_hi : nullptr, (int64_t)bytes, plo, phi))
            return fail(WL_E_STATE, "host sendrecv callback failed", __FILE__, __LINE__);
        if (rlo) WL_HIP(hipMemcpyAsync(rlo, hr_lo, bytes, hipMemcpyHostToDevice, ctx().stream));
        if (rhi) WL_HIP(hipMemcpyAsync(rhi, hr_hi, bytes, hipMemcpyHostToDevice, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        return 0;
    }
    int do_allgather(void *buf, size_t bytes) override {
        const size_t tot = bytes * (size_t)size;
        WL_TRY(need(tot));
        WL_HIP(hipMemcpyAsync(pin + (size_t)rank * bytes, (char *)buf + (size_t)rank * bytes, bytes, hipMemcpyDeviceToHost, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        if (ag(user, pin, (int64_t)bytes)) return fail(WL_E_STATE, "host allgather callback failed", __FILE__, __LINE__);
        WL_HIP(hipMemcpyAsync(buf, pin, tot, hipMemcpyHostToDevice, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        return 0;
    }
};

// Loopback twin (measurement: bench.py --comm loopback): ONE process plays rank `rank` of `size`.  Its neighbours are taken to be
// copies of itself (a periodic stack of this slab): the planes it would send up arrive as the planes from below and vice
// versa (device copies on the stream), a sum over the ranks is `size` times the local value, an all-gather repeats the local
// segment.  Every kernel, split launch and reduction of a rank of the N-GPU run is issued; nothing waits for a wire.
__global__ void k_loop_scale(double *v, int n, double f) { if ((int)threadIdx.x < n) v[threadIdx.x] *= f; }
struct LoopComm : Comm {
    int do_allreduce(double *dev, int n, int op) override {
        if (op != 0) return 0;
        hipLaunchKernelGGL(k_loop_scale, dim3(1), dim3(64), 0, ctx().stream, dev, n, (double)size);
        return (int)hipGetLastError();
    }
    int do_sendrecv(const void *slo, void *rlo, const void *shi, void *rhi, size_t bytes, int, int) override {
        const void *from_below = shi ? shi : slo, *from_above = slo ? slo : shi;
        if (rlo && from_below) WL_HIP(hipMemcpyAsync(rlo, from_below, bytes, hipMemcpyDeviceToDevice, ctx().stream));
        if (rhi && from_above) WL_HIP(hipMemcpyAsync(rhi, from_above, bytes, hipMemcpyDeviceToDevice, ctx().stream));
        return 0;
    }
    int do_allgather(void *buf, size_t bytes) override {
        for (int r = 0; r < size; ++r)
            if (r != rank) WL_HIP(hipMemcpyAsync((char *)buf + (size_t)r * bytes, (const char *)buf + (size_t)rank * bytes, bytes, hipMemcpyDeviceToDevice, ctx().stream));
        return 0;
    }
};

Ctx &ctx() {
    static Ctx c;
    return c;
}
int fail(int code, const char *what, const char *file, int line) {
    char buf[512];
    const char *hs = (code > 0 && code < 10000) ? hipGetErrorString((hipError_t)code) : "";
    snprintf(buf, sizeof buf, "wlhip error %d at %s:%d: %s %s", code, file, line, what, hs);
    ctx().err = buf;
    return code ? code : WL_E_ARG;
}

Prof::Prof(int kclass, int64_t ncells) : ncell(ncells) {
    Ctx &c = ctx();
    c.launches[kclass] += 1;
    c.cells[kclass] += ncells;
    if (kclass == c.prof_class && ncells >= c.prof_min_cells) {
        auto get = [&c]() {
            hipEvent_t e;
            if (!c.pool.empty()) { e = c.pool.back(); c.pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get(); b = get();
        timed = true;
        (void)hipEventRecord(a, c.stream);
    }
}
Prof::~Prof() {
    if (timed) {
        Ctx &c = ctx();
        (void)hipEventRecord(b, c.stream);
        c.evts.push_back({a, b, ncell});
    }
}

int check_grid(const wl_grid *g) {
    if (!g) return fail(WL_E_ARG, "null grid", __FILE__, __LINE__);
    if (g->D != 2 && g->D != 3) return fail(WL_E_ARG, "grid.D must be 2 or 3", __FILE__, __LINE__);
    if (g->s[0] != 1) return fail(WL_E_ARG, "grid.s[0] must be 1", __FILE__, __LINE__);
    for (int d = 0; d < g->D; ++d)
        if (g->n[d] < 3) return fail(WL_E_ARG, "grid extents must be >= 3 (one ghost layer per side)", __FILE__, __LINE__);
    if (g->D == 2 && g->n[2] != 1) return fail(WL_E_ARG, "grid.n[2] must be 1 for D==2", __FILE__, __LINE__);
    if (g->s[1] < g->n[0]) return fail(WL_E_ARG, "grid.s[1] < n[0]", __FILE__, __LINE__);
    if (g->D == 3 && g->s[2] < g->s[1] * g->n[1]) return fail(WL_E_ARG, "grid.s[2] < s[1]*n[1]", __FILE__, __LINE__);
    const int64_t sp = g->D == 3 ? g->s[2] * g->n[2] : g->s[1] * g->n[1];
    if (g->sc < sp) return fail(WL_E_ARG, "grid.sc smaller than one component", __FILE__, __LINE__);
    if (g->D == 3 && g->nzg > 0) {
        if (g->own_lo < 0 || g->own_hi >= g->n[2] || g->own_lo > g->own_hi)
            return fail(WL_E_ARG, "grid.own_lo/own_hi outside the local planes", __FILE__, __LINE__);
        if (g->kz0 + g->own_lo < 0 || g->kz0 + g->own_hi > g->nzg - 1)
            return fail(WL_E_ARG, "owned planes outside the global array", __FILE__, __LINE__);
        if (g->zring && (g->kz0 + g->own_lo < 1 || g->kz0 + g->own_hi > g->nzg - 2))
            return fail(WL_E_ARG, "zring: the z ghost planes must not be owned", __FILE__, __LINE__);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ handles
struct Scratch {
    double *partials = nullptr;  // 4 * WL_MAXB doubles
    State *st = nullptr;         // device
    State *hst = nullptr;        // pinned host mirror
    int init() {
        WL_HIP(wl_dev_alloc((void **)&partials, sizeof(double) * 4 * WL_MAXB));
        WL_HIP(wl_dev_alloc((void **)&st, sizeof(State)));
        WL_HIP(hipMemset(st, 0, sizeof(State)));
        WL_HIP(wl_host_alloc((void **)&hst, sizeof(State), hipHostMallocDefault));
        memset(hst, 0, sizeof(State));
        return 0;
    }
    void release() {
        if (partials) (void)hipFree(partials);
        if (st) (void)hipFree(st);
        if (hst) (void)hipHostFree(hst);
        partials = nullptr; st = nullptr; hst = nullptr;
    }
    // bring the device scalars to the host (synchronises the stream)
    int fetch() {
        WL_HIP(hipMemcpyAsync(hst, st, sizeof(State), hipMemcpyDeviceToHost, ctx().stream));
        WL_HIP(hipStreamSynchronize(ctx().stream));
        if (ctx().mbox && *ctx().mbox->err_host)
            return fail(WL_E_STATE, "mailbox all-reduce: gave up waiting for a peer rank (is every rank still running?)", __FILE__, __LINE__);
        return 0;
    }
};

Scratch &global_scratch(int *rc) {
    static Scratch s;
    static bool ok = false;
    *rc = 0;
    if (!ok) { *rc = s.init(); ok = (*rc == 0); }
    return s;
}

}  // namespace wl

using namespace wl;

struct wl_mg {
    wl_dtype t;
    int D;
    int nlev;
    int permask;
    std::vector<wl_level_desc> lev;
    Scratch sc;
    std::vector<void *> rowc;            // per level: row constants of L and iD (k_lrow), RC_N values per (j,k) row; D==3 only
    unsigned char *dirty = nullptr;      // level-0 row flags of wl_mg_update_changed
    bool log_on = false;                 // wl_mg_log: record {n, Linf, L2} per solver iteration
    std::vector<double> log;
    int alloc_rowc() {
        const size_t es = t == WL_F32 ? 4 : 8;
        rowc.assign(nlev, nullptr);
        for (int l = 0; l < nlev; ++l) {
            const wl_grid &g = lev[l].g;
            if (g.D != 3) continue;
            WL_HIP(wl_dev_alloc(&rowc[l], (size_t)g.n[1] * (size_t)g.n[2] * RC_N * es));
        }
        return 0;
    }
    void free_scratch() {
        for (void *p : rowc) if (p) (void)hipFree(p);
        rowc.clear();
        if (dirty) (void)hipFree(dirty);
        dirty = nullptr;
    }
};
struct wl_flow {
    wl_dtype t;
    wl_flow_desc d;
    Scratch sc;
    unsigned char *rowfree = nullptr;   // body-free row flags (wl_flow_update); nullptr until built
    unsigned char *rowbuf = nullptr;
    unsigned char *segbuf = nullptr;    // the same per 64-cell segment of a row (3-D; wl_flow_update's scan only)
    bool seg_valid = false;
    const unsigned char *segfree() const { return (seg_valid && ctx().opt[3]) ? segbuf : nullptr; }
    int *busy = nullptr;                // compact list of the busy interior rows (j + n1*k), device
    int nbusy = 0;
    int nbusy_lo = 0, nbusy_hi = 0;      // how many of them lie in the first / last owned interior plane (the list is sorted by plane)
    size_t busy_cap = 0;
    // native measure! (wl_measure.h): per-row band counts / offsets, rows touched by this and by the previous measure!
    int *rowcount = nullptr;
    long *rowoff = nullptr;
    unsigned char *touched = nullptr, *prev = nullptr;
    unsigned char *changed = nullptr;   // rows whose coefficient arrays the last native measure! rewrote (touched now or before)
    bool changed_valid = false;
    bool changed_pending = false;       // `changed` holds rows no update!(pois) has consumed yet: the next measure! ORs into it
    bool prev_valid = false;            // `prev` describes the arrays' current content (else: rewrite every row)
    long nband = -1;                    // result of the last wl_measure_rows (-1: none pending)
};

template <class T> static LevelT<T> lvl(const wl_mg *m, int l) {
    const wl_level_desc &d = m->lev[l];
    LevelT<T> o;
    o.g = mkG(&d.g);
    o.L = (T *)d.L; o.D = (T *)d.D; o.iD = (T *)d.iD; o.x = (T *)d.x; o.eps = (T *)d.eps; o.r = (T *)d.r; o.z = (T *)d.z;
    o.rowc = (ctx().opt[9] && l < (int)m->rowc.size()) ? (const T *)m->rowc[l] : nullptr;
    return o;
}

// ------------------------------------------------------------------------------------------ MG orchestration
// dirty (optional, D == 3): flags of the level-0 x-rows whose D / iD / row constants can have changed (see
// wl_mg_update_changed); every other row of level 0 keeps what it has.  Levels >= 1 are rebuilt in full (1/8, 1/64 ...).
template <class T, int D> static int mg_update(wl_mg *m, const unsigned char *dirty = nullptr) {
    {
        LevelT<T> p = lvl<T>(m, 0);
        WL_TRY((op_set_diag<T, D>(p.g, p.D, p.iD, p.L, dirty)));
        WL_TRY((halo_exchange<T>(p.g, p.iD, 1, 1)));   // z-slab: the fused smoother evaluates r*iD in the halo planes
        if (D == 3 && m->rowc[0]) WL_TRY((op_lrow<T>(p.g, p.L, p.iD, (T *)m->rowc[0], dirty)));
    }
    for (int l = 1; l < m->nlev; ++l) {
        LevelT<T> a = lvl<T>(m, l), b = lvl<T>(m, l - 1);
        WL_TRY((op_restrictL<T, D>(a.g, a.L, b.g, b.L, m->permask)));
        WL_TRY((coarse_L_finish<T, D>(a.g, a.L, b.g, m->permask)));
        WL_TRY((op_set_diag<T, D>(a.g, a.D, a.iD, a.L)));
        WL_TRY((halo_exchange<T>(a.g, a.iD, 1, 1)));
        if (D == 3 && m->rowc[l]) WL_TRY((op_lrow<T>(a.g, a.L, a.iD, (T *)m->rowc[l])));
    }
    return 0;
}

// Vcycle!  src/MultiLevelPoisson.jl:70-82
// pcg_np (solver! only): receives the partial count when the closing prolongate!+increment! also did the start of the
// pcg! that the caller runs next on level l (eps = r*iD, rho partials), else -1
template <class T, int D> static int mg_vcycle(wl_mg *m, int l, int *pcg_np = nullptr) {
    LevelT<T> fine = lvl<T>(m, l), coarse = lvl<T>(m, l + 1);
    const bool fused = (m->permask == 0) && ctx().opt[1];   // periodic ghosts of eps are copies, not zeros: keep the two-pass form
    if (fused) WL_TRY((op_smooth_fused<T, D>(fine, fine.eps)));     // r' lives in the eps buffer until the way up
    else WL_TRY((op_jacobi<T, D>(fine, 1, m->permask)));
    // fill!(coarse.x, 0) (MultiLevelPoisson.jl:75) rides in the restriction kernel where the level's ghost cells cannot
    // hold anything but zero (no periodic copy, no halo exchange); otherwise it is a memset of the whole array
    const bool zero_in_restrict = (m->permask == 0) && !coarse.g.dist && !fine.g.dist;
    WL_TRY((op_restrict<T, D>(coarse.g, coarse.r, fine.g, fused ? fine.eps : fine.r, zero_in_restrict ? coarse.x : (T *)nullptr)));
    if (ctx().comm && ctx().comm->size > 1 && fine.g.dist && !coarse.g.dist) {
        // hand-over to the replicated coarse levels: every rank restricted the children it owns
        const int nzl = (fine.g.nzg - 2) / ctx().comm->size / 2;
        WL_TRY(ctx().comm->allgather(coarse.r + coarse.g.s[2], (size_t)nzl * coarse.g.s[2] * sizeof(T)));
    }
    if (!zero_in_restrict) {
        Prof p(WL_K_MISC, coarse.g.cells());
        WL_HIP(hipMemsetAsync(coarse.x, 0, (size_t)span(coarse.g) * sizeof(T), ctx().stream));
    }
    // levels <= 4096 cells: the rest of the recursion + smooth!(coarse) as ONE single-workgroup launch (wl_coarse.h)
    const long tail_cells = ctx().opt[6] == 1 ? CV_MAXCELLS : ctx().opt[6];   // option 6: 0 off, 1 default, else threshold
    bool tail = ctx().opt[6] && fused && !coarse.g.dist && coarse.g.interior_cells() <= tail_cells && (m->nlev - (l + 1)) <= CV_MAXLEV;
    if (tail) {
        CoarseArgs<T> ca;
        ca.nlev = m->nlev - (l + 1);
        for (int q = 0; q < ca.nlev; ++q) {
            ca.lev[q] = lvl<T>(m, l + 1 + q);
            if (ca.lev[q].g.dist) tail = false;
        }
        if (tail) {
            Prof p(WL_K_SMOOTH, coarse.g.cells());
            if (ctx().opt[31]) hipLaunchKernelGGL((k_coarse_vcycle<T, D, true>), dim3(1), dim3(CV_THREADS), 0, ctx().stream, ca);
            else hipLaunchKernelGGL((k_coarse_vcycle<T, D, false>), dim3(1), dim3(CV_THREADS), 0, ctx().stream, ca);
            WL_HIP(hipGetLastError());
        }
    }
    if (!tail) {
        int pre = -1;
        if (l + 2 < m->nlev) WL_TRY((mg_vcycle<T, D>(m, l + 1, pcg_np ? &pre : nullptr)));
        WL_TRY((op_pcg<T, D>(coarse, 6, m->permask, m->sc.partials, m->sc.st, false, pre)));
    }
    if (pcg_np) *pcg_np = -1;
    if (fused) return op_prolong_increment_fused<T, D>(fine, fine.eps, coarse.g, coarse.x, m->sc.partials, pcg_np);
    WL_TRY((op_prolongate<T, D>(fine.g, fine.eps, coarse.g, coarse.x)));
    return op_increment<T, D>(fine, m->permask);
}

// solver!  src/MultiLevelPoisson.jl:87-99 (nlev>1) / src/Poisson.jl:162-172 (nlev==1).
// One host synchronisation per iteration: the r2 < tol test (:95).
// divu / gu: the right-hand side is div(u) of this velocity field, evaluated inside residual! (project!, single device)
// L∞(p) = maximum(abs, p.r)  src/Poisson.jl:147 -> st->out[1]
template <class T, int D> static int mg_Linf(wl_mg *m, int l) {
    LevelT<T> p = lvl<T>(m, l);
    const T *r = p.r;
    return op_reduce<T, D>(p.g, WL_K_DOT, RED_MAX, 0.0, [=] __device__(long I) { const double v = (double)r[I]; return v < 0 ? -v : v; },
                           m->sc.partials, m->sc.st, 1);
}
// one row of the reference's solver log: `@log ", $n, $(L∞(p)), $r₂\n"` (Poisson.jl:164,167; MultiLevelPoisson.jl:90,94)
template <class T, int D> static int mg_log_row(wl_mg *m, int n, bool have_r2) {
    if (!have_r2) WL_TRY((op_L2<T, D>(lvl<T>(m, 0), m->sc.partials, m->sc.st, false)));
    WL_TRY((mg_Linf<T, D>(m, 0)));
    WL_TRY(m->sc.fetch());
    m->log.push_back((double)n); m->log.push_back(m->sc.hst->out[1]); m->log.push_back(m->sc.hst->r2);
    return 0;
}
template <class T, int D> static int mg_solve(wl_mg *m, double tol, int itmx, int *n_iter, const T *divu = nullptr, const G *gu = nullptr,
                                              bool halo_begun = false) {
    LevelT<T> p = lvl<T>(m, 0);
    WL_TRY((op_residual<T, D>(p, m->permask, m->sc.partials, m->sc.st, divu, gu, halo_begun)));
    int n = 0;
    if (m->log_on) WL_TRY((mg_log_row<T, D>(m, 0, false)));
    while (n < itmx) {
        int pre = -1;
        if (m->nlev > 1) WL_TRY((mg_vcycle<T, D>(m, 0, &pre)));
        WL_TRY((op_pcg<T, D>(p, 6, m->permask, m->sc.partials, m->sc.st, true, pre, true)));   // (level 1: z ≡ flow.σ)
        WL_TRY((op_L2<T, D>(p, m->sc.partials, m->sc.st, true)));
        WL_TRY(m->sc.fetch());
        ++n;
        const double r2 = m->sc.hst->r2;
        if (m->log_on) WL_TRY((mg_log_row<T, D>(m, n, true)));
        if (r2 < tol) break;
    }
    WL_TRY((op_bc_per<T, D>(p.g, p.x, m->permask)));
    if (n_iter) *n_iter = n;
    return 0;
}

// project!  src/Flow.jl:137-145
// exchange_u (z-slab runs, mom_step!): the 1-plane halo exchange of u that div needs (it reads u[I+dz]) is issued here on
// the comm stream; x*=dt and div on all owned planes but the last run while it is in flight.
// head_done: `x .*= dt` of this call was already applied (chained onto the previous projection's `x ./= dt`);
// tail_then: instead of this call's own `x ./= dt` pass, run that and the NEXT projection's `x .*= dt'` as one stream.
template <class T> static ScaleOp project_scale(double dt_, double w) {
    const bool dbl = (w != 1.0);
    return ScaleOp{dbl ? w * (double)(T)dt_ : (double)(T)dt_, false, dbl};
}
// uvel: the velocity array to project (mom_step! may hold the predicted velocity in the flow's u0 array; default a->d.u)
template <class T, int D> static int flow_project(wl_flow *a, wl_mg *b, double dt_, double w, int *n_iter, bool exchange_u = false,
                                                  bool head_done = false, const ScaleOp *tail_then = nullptr, const XBc<T> *xbc = nullptr,
                                                  bool *xdone = nullptr, T *uvel = nullptr) {
    const G g = mkG(&a->d.g);
    T *const uv = uvel ? uvel : (T *)a->d.u;
    LevelT<T> p = lvl<T>(b, 0);
    const ScaleOp sc = project_scale<T>(dt_, w);
    const double dts = sc.s;
    const bool dbl = sc.dbl;
    const Range R = r_inside(g);
    // 3-D vector kernels: z = div(u) is formed inside residual! (wl_set_option(22)); p.z stays unwritten
    // (rowvec_fits: a plane's workgroups fit the partial buffer -- the launch below cannot be rejected for its size)
    const bool fused_div = D == 3 && ctx().opt[22] && ctx().opt[5] && stencil7_ok<T>(g) && stencil7_ok<T>(p.g) && rowvec_fits<T>(p.g) && b->permask == 0 &&
                           g.s[1] == p.g.s[1] && g.s[2] == p.g.s[2] && g.n[0] == p.g.n[0] && g.n[2] == p.g.n[2] && g.dist == p.g.dist;
    bool begun = false;
    if (fused_div && g.dist) {
        // z-slabs: the plane of u that div reads above the last owned plane and the planes of x that residual! reads travel in ONE
        // batch on the comm stream (x must carry its `.*= dt` first); the fused kernel runs on the inner planes meanwhile
        if (!head_done) WL_TRY((op_scale_all<T, D>(g, p.x, dts, false, dbl)));
        if (exchange_u) { WL_TRY((halo_begin2<T>(g, uv, D, 1, p.g, p.x, 1, 1))); begun = true; }
    } else if (exchange_u && D == 3 && g.dist && overlap_on() && R.hi[2] - R.lo[2] + 1 >= 2) {
        WL_TRY((halo_begin<T>(g, uv, D, 1)));
        int rc = head_done ? 0 : op_scale_all<T, D>(g, p.x, dts, false, dbl);
        if (!rc) rc = op_div<T, D>(g, p.z, (const T *)uv, R.lo[2], R.hi[2] - 1);
        WL_TRY(halo_end());   // (always joined, also on an error above)
        if (rc) return rc;
        WL_TRY((op_div<T, D>(g, p.z, (const T *)uv, R.hi[2], R.hi[2])));
    } else {
        if (exchange_u) WL_TRY((halo_exchange<T>(g, uv, D, 1)));
        if (!fused_div) WL_TRY((op_div<T, D>(g, p.z, (const T *)uv)));
        if (!head_done) WL_TRY((op_scale_all<T, D>(g, p.x, dts, false, dbl)));
    }
    WL_TRY((mg_solve<T, D>(b, 1e-4, 32, n_iter, fused_div ? (const T *)uv : nullptr, &g, begun)));
    WL_TRY((op_correct<T, D>(g, uv, p.L, p.x, p.rowc, xbc, xdone)));
    return op_scale_all<T, D>(g, p.x, dts, true, dbl, tail_then);
}

// mom_step!  src/Flow.jl:153-169
template <class T, int D>
static int flow_mom_step(wl_flow *a, wl_mg *b, double dt, const double *U, const double *gp, const double *gc,
                         double *dt_next, int *n2) {
    const wl_flow_desc &d = a->d;
    const G g = mkG(&d.g);
    T *u = (T *)d.u, *u0 = (T *)d.u0, *f = (T *)d.f, *V = (T *)d.V, *mu0 = (T *)d.mu0, *mu1 = (T *)d.mu1;
    // (z-slab runs: u carries a 2-plane halo for QUICK, f a 1-plane halo for mu_ddn; exchanges are no-ops otherwise)
    // (the x-ghost cells of the interior rows are written by the kernel that produces the row: XBc, wl_set_option(23))
    const XBc<T> xbc{(D == 3 && d.perdir_mask == 0 && ctx().opt[7] && ctx().opt[23]) ? 1 : 0, d.exitBC ? 1 : 0, (T)U[0]};
    bool xd = false;
    // `turns`: BDIM! is finished inside conv_diff! on the body-free rows (wl_set_option(27), CdFin in wl_convdiff.h).  The kernel
    // that forms f cannot overwrite the velocity its neighbours still read, so the two velocity arrays take turns: the predictor
    // reads `u` (which thereby IS u0: no copy) and writes u' into the flow's u0 array; the corrector reads u' there and u0 in
    // `u`, cell by cell, and writes the new velocity over it.  On return `u` holds the new velocity as always and the u0 ARRAY
    // holds u' instead of the old velocity -- the reference overwrites u0 before it reads it (Flow.jl:154), DESIGN.md 7.8.
    // Otherwise (2-D, periodic directions, convective exit, no row flags): `a.u0 .= a.u` rides in the predictor's conv_diff!
    // and BDIM! #2 is a pass of its own, in place.
    bool turns = false;
    if constexpr (D == 3)
        turns = ctx().opt[27] && ctx().opt[3] && a->rowfree && a->busy && d.perdir_mask == 0 && !d.exitBC && conv_diff_tiled<D>(g, 0);
    T *const up = turns ? u0 : u;   // where the predictor's velocity u' lives
    const CdFin<T> fin1{up, a->rowfree, xbc.on, (T)U[0]}, fin2{u, a->rowfree, xbc.on, (T)U[0]};
    (void)fin1; (void)fin2;
    // predictor (:157-161): [u0 .= u (:154)] + conv_diff! + accelerate! + BDIM! #1 in ONE kernel; scale_u!(a,0) is folded into
    // BDIM! #2 (MODE 1: u = ...)
    if (turns) {
        if constexpr (D == 3) {
            WL_TRY((op_conv_diff<T, D, true, false, 1>(g, f, u, d.nu, 0, u, V, dt, gp, gp != nullptr, nullptr, false, &fin1)));
            WL_TRY((op_bdim2_busy<T, 1>(g, up, up, f, V, mu0, mu1, a->busy, a->nbusy, a->nbusy_lo, a->nbusy_hi, xbc, a->segfree())));   // + exchange of f
            xd = xbc.on != 0;
        }
    } else {
        WL_TRY((op_conv_diff<T, D, true, true>(g, f, u, d.nu, d.perdir_mask, nullptr, V, dt, gp, gp != nullptr, u0)));
        // σ's top ghost cells: the flux scratch Φ the reference's conv_diff! leaves there (Flow.jl:157), read by its whole-array
        // z⋅ϵ in the projection that follows (periodic runs only: elsewhere ϵ's ghosts are zero) and by maximum(a.σ) in CFL --
        // which sees the corrector's values, so a non-periodic run skips the predictor's.
        if (d.perdir_mask != 0) WL_TRY((op_sigma_ghosts<T, D>(g, (T *)d.sigma, u, d.nu, d.perdir_mask)));
        WL_TRY((op_bdim2<T, D, 1>(g, u, f, V, mu0, mu1, a->rowfree, a->busy, a->nbusy, true, &xbc, &xd, a->segfree())));   // + exchange of f (overlapped)
    }
    WL_TRY((op_bc_vec<T, D>(g, up, U, d.exitBC, d.perdir_mask, xd)));
    if (d.exitBC) WL_TRY((op_exit_bc<T, D>(g, up, u0, U, dt, a->sc.partials, a->sc.st)));
    // (the predictor's closing `x ./= dt` and the corrector's opening `x .*= 0.5dt` are ONE pass over x: nothing in between reads p)
    const ScaleOp corr_head = project_scale<T>(dt, 0.5);
    const bool chain = ctx().opt[14] != 0;
    WL_TRY((flow_project<T, D>(a, b, dt, 1.0, &n2[0], true, false, chain ? &corr_head : nullptr, &xbc, &xd, up)));   // + 1-plane exchange of u' (overlapped)
    WL_TRY((op_bc_vec<T, D>(g, up, U, d.exitBC, d.perdir_mask, xd)));
    // corrector (:164-167); the 2-plane exchange of u' is issued inside op_conv_diff (overlapped with its inner planes);
    // σ's ghost cells: Φ of the corrector (Flow.jl:164), formed from u'; scale_u!(a,0.5) is folded into BDIM! #2 (MODE 2)
    if (turns) {
        if constexpr (D == 3) {
            WL_TRY((op_conv_diff<T, D, true, false, 2>(g, f, up, d.nu, 0, u, V, dt, gc, gc != nullptr, nullptr, true, &fin2)));
            WL_TRY((op_sigma_ghosts<T, D>(g, (T *)d.sigma, up, d.nu, 0)));
            WL_TRY((op_bdim2_busy<T, 2>(g, u, up, f, V, mu0, mu1, a->busy, a->nbusy, a->nbusy_lo, a->nbusy_hi, xbc, a->segfree())));
            xd = xbc.on != 0;
        }
    } else {
        WL_TRY((op_conv_diff<T, D, true>(g, f, u, d.nu, d.perdir_mask, u0, V, dt, gc, gc != nullptr, nullptr, true)));
        WL_TRY((op_sigma_ghosts<T, D>(g, (T *)d.sigma, u, d.nu, d.perdir_mask)));
        WL_TRY((op_bdim2<T, D, 2>(g, u, f, V, mu0, mu1, a->rowfree, a->busy, a->nbusy, true, &xbc, &xd, a->segfree())));
    }
    WL_TRY((op_bc_vec<T, D>(g, u, U, d.exitBC, d.perdir_mask, xd)));
    WL_TRY((flow_project<T, D>(a, b, dt, 0.5, &n2[1], true, chain, nullptr, &xbc, &xd)));
    WL_TRY((op_bc_vec<T, D>(g, u, U, d.exitBC, d.perdir_mask, xd)));
    // push!(a.dt, CFL(a)) (:168); the end-of-step 2-plane exchange of u is issued inside (overlapped with the kernel)
    WL_TRY((op_cfl<T, D>(g, (T *)d.sigma, u, d.nu, a->sc.partials, a->sc.st, true)));
    WL_TRY(a->sc.fetch());
    *dt_next = a->sc.hst->out[0];
    return 0;
}

template <class T, int D> static int red_L2(const G &g, const T *a, Scratch &S) {
    return op_reduce<T, D>(g, WL_K_DOT, RED_SUM, 0.0, [=] __device__(long I) { const double v = (double)a[I]; return v * v; },
                           S.partials, S.st, 0);
}
template <class T, int D> static int red_dot(const G &g, const T *a, const T *b, Scratch &S) {
    return op_reduce<T, D>(g, WL_K_DOT, RED_SUM, 0.0, [=] __device__(long I) { return (double)a[I] * (double)b[I]; },
                           S.partials, S.st, 0, true);
}
template <class T, int D> static int red_sum(const G &g, const T *a, Scratch &S) {
    return op_reduce<T, D>(g, WL_K_DOT, RED_SUM, 0.0, [=] __device__(long I) { return (double)a[I]; }, S.partials, S.st, 0, true);
}
template <class T, int D> static int red_max(const G &g, const T *a, Scratch &S) {
    return op_reduce<T, D>(g, WL_K_DOT, RED_MAX, -1e300, [=] __device__(long I) { return (double)a[I]; }, S.partials, S.st, 0, true);
}
template <class T, int D>
static int bdim_full(const G &g, T *u, const T *u0, T *f, const T *V, const T *mu0, const T *mu1, double dt) {
    WL_TRY((op_bdim1<T, D>(g, f, u0, V, dt)));
    return op_bdim2<T, D, 0>(g, u, f, V, mu0, mu1);
}
// middle eigenvalue of a symmetric 3x3 matrix (closed form, Smith 1961) -- lambda2 = eigvals(Hermitian(S^2+O^2))[2]
__host__ __device__ inline double sym3_mid_eig(double a00, double a01, double a02, double a11, double a12, double a22) {
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) / 3.0;
    if (p1 == 0.0) {   // diagonal
        double x = a00, y = a11, z = a22, t;
        if (x > y) { t = x; x = y; y = t; }
        if (y > z) { t = y; y = z; z = t; }
        if (x > y) { t = x; x = y; y = t; }
        return y;
    }
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    const double p = sqrt(p2 / 6.0);
    const double c00 = b00 / p, c11 = b11 / p, c22 = b22 / p, c01 = a01 / p, c02 = a02 / p, c12 = a12 / p;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    const double e1 = q + 2.0 * p * cos(phi), e3 = q + 2.0 * p * cos(phi + 2.0943951023931953);
    return 3.0 * q - e1 - e3;
}
template <class T, int D>
static int op_metric(const G &g, int kind, T *out, const T *u, int ipar, const double *par, const double *par2) {
    const G gg = g;
    double p3[3] = {0, 0, 0}, q3[3] = {0, 0, 0};
    if (par) for (int d = 0; d < D; ++d) p3[d] = par[d];
    if (par2) for (int d = 0; d < D; ++d) q3[d] = par2[d];
    const double p0 = p3[0], p1 = p3[1], p2 = p3[2], q0 = q3[0], q1 = q3[1], q2 = q3[2];
    return launch_range(WL_K_MISC, r_inside(g), [=] __device__(int i, int j, int k) {
        const long I = gg.at(i, j, k);
        const long S[3] = {gg.s[0], gg.s[1], gg.s[2]};
        const long SC = gg.sc;
        auto U = [&](int c, long off) -> T { return u[I + off + (long)c * SC]; };
        auto dudx = [&](int a, int b) -> T {   // Metrics.jl:28-31
            if (a == b) return U(a, S[a]) - U(a, 0);
            return (U(a, S[b]) + U(a, S[b] + S[a]) - U(a, -S[b]) - U(a, -S[b] + S[a])) / (T)4;
        };
        T res = 0;
        if (kind == WL_M_KE) {   // Metrics.jl:20-22
            const double UU[3] = {p0, p1, p2};
            double s = 0;   // (Float64 accumulation: exact for the reference's Float64 U, >= its precision for U=0)
            for (int c = 0; c < D; ++c) { const double v = (double)(T)(U(c, 0) + U(c, S[c])) - 2.0 * UU[c]; s += v * v; }
            res = (T)(0.125 * s);
        } else if (kind == WL_M_CURL) {   // Metrics.jl:54: permute((j,k)->d(j,CI(I,k),u), i), backward differences
            const int a = (ipar + 1) % 3, b = (ipar + 2) % 3;
            res = (U(b, 0) - U(b, -S[a])) - (U(a, 0) - U(a, -S[b]));
        } else if (D == 3) {
            T w[3];
            for (int c = 0; c < 3; ++c) { const int a = (c + 1) % 3, b = (c + 2) % 3; w[c] = dudx(b, a) - dudx(a, b); }   // Metrics.jl:60
            if (kind == WL_M_OMAG) {
                res = (T)sqrt((double)(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
            } else if (kind == WL_M_OTHETA) {   // Metrics.jl:72-76
                const double x[3] = {(double)i - 0.5 - q0, (double)j - 0.5 - q1, (double)(k + gg.kz0) - 0.5 - q2};
                const double th[3] = {p1 * x[2] - p2 * x[1], p2 * x[0] - p0 * x[2], p0 * x[1] - p1 * x[0]};
                const double n = sqrt(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
                res = n <= 2.220446049250313e-16 * n ? (T)0 : (T)((th[0] * (double)w[0] + th[1] * (double)w[1] + th[2] * (double)w[2]) / n);
            } else {   // lambda2, Metrics.jl:41-45
                double J[3][3], M[3][3];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) J[a][b] = (double)dudx(a, b);
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
                res = (T)sym3_mid_eig(M[0][0], M[0][1], M[0][2], M[1][1], M[1][2], M[2][2]);
            }
        }
        out[I] = res;
    });
}

// du_i/dx_j at the centre of cell I (src/Metrics.jl:28-31), in T
template <class T> __device__ inline T dudx(const G &g, const T *u, long I, int i, int j) {
    const T *ui = u + (long)i * g.sc;
    if (i == j) return ui[I + g.s[i]] - ui[I];
    return (ui[I + g.s[j]] + ui[I + g.s[j] + g.s[i]] - ui[I - g.s[j]] - ui[I - g.s[j] + g.s[i]]) / (T)4;
}
// Metrics.jl:109-113 viscous_force over the band
template <class T>
__global__ __launch_bounds__(256) void k_vforce(G g, const T *u, const int64_t *idx, const double *nds, int64_t nband, T nu,
                                                double *partials) {
    double acc[3] = {0, 0, 0};
    const int D = g.D;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nband; b += (int64_t)gridDim.x * 256) {
        const long I = idx[b];
        for (int i = 0; i < D; ++i) {
            double s = 0;
            for (int j = 0; j < D; ++j) {
                const T m = -nu * (T)(dudx<T>(g, u, I, i, j) + dudx<T>(g, u, I, j, i));   // (-nu * grad2u)[i,j] in T
                s += (double)m * nds[b * D + j];
            }
            acc[i] += (double)(T)s;                                                        // df[I,:] is a T array
        }
    }
    block_red<3>(acc, RED_SUM);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c) partials[(long)c * gridDim.x + blockIdx.x] = acc[c];
}
// Metrics.jl:130-134 pressure_moment over the band
template <class T>
__global__ __launch_bounds__(256) void k_pmoment(G g, const T *p, const int64_t *idx, const double *nds, int64_t nband,
                                                 double x0, double y0, double z0, double *partials) {
    double acc[3] = {0, 0, 0};
    const int D = g.D;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nband; b += (int64_t)gridDim.x * 256) {
        const long I = idx[b];
        const long k = D > 2 ? I / g.s[2] : 0, rem = D > 2 ? I - k * g.s[2] : I;
        const long j = rem / g.s[1], i = rem - j * g.s[1];
        const double rx = (double)i - 0.5 - x0, ry = (double)j - 0.5 - y0, rz = (double)(k + g.kz0) - 0.5 - z0;  // loc(0,I)-x0
        const double pv = (double)p[I];
        if (D == 3) {
            const double nx = nds[b * 3], ny = nds[b * 3 + 1], nz = nds[b * 3 + 2];
            acc[0] += (double)(T)(pv * (ry * nz - rz * ny));
            acc[1] += (double)(T)(pv * (rz * nx - rx * nz));
            acc[2] += (double)(T)(pv * (rx * ny - ry * nx));
        } else {
            const double m = (double)(T)(pv * (rx * nds[b * 2 + 1] - ry * nds[b * 2]));
            acc[0] += m; acc[1] += m;
        }
    }
    block_red<3>(acc, RED_SUM);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c) partials[(long)c * gridDim.x + blockIdx.x] = acc[c];
}
template <class T>
__global__ __launch_bounds__(256) void k_pforce(const T *p, const int64_t *idx, const double *nds, int64_t nband, int D,
                                                 double *partials) {
    double acc[3] = {0, 0, 0};
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nband; b += (int64_t)gridDim.x * 256) {
        const double pv = (double)p[idx[b]];
        for (int c = 0; c < D; ++c) acc[c] += (double)(T)(pv * nds[b * D + c]);  // df[I,:] is a T array
    }
    block_red<3>(acc, RED_SUM);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c) partials[(long)c * gridDim.x + blockIdx.x] = acc[c];
}
// compact the busy INTERIOR rows of a->rowbuf on the host (n1*n2 bytes; this runs once per measure!, not per step)
static int flow_compact_busy(wl_flow *a, const G &g, int D) {
    a->rowfree = a->rowbuf;
    const size_t nrows = (size_t)g.n[1] * (size_t)(D > 2 ? g.n[2] : 1);
    std::vector<unsigned char> fl(nrows);
    WL_HIP(hipMemcpyAsync(fl.data(), a->rowbuf, nrows, hipMemcpyDeviceToHost, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    const Range R = r_inside(g);
    std::vector<int> list;
    for (int k = R.lo[2]; k <= R.hi[2]; ++k)
        for (int j = 1; j <= g.n[1] - 2; ++j)
            if (!fl[(size_t)j + (size_t)g.n[1] * k]) list.push_back(j + g.n[1] * k);
    if (list.size() > a->busy_cap) {
        if (a->busy) (void)hipFree(a->busy);
        a->busy_cap = list.size() * 2 + 64;
        WL_HIP(wl_dev_alloc((void **)&a->busy, a->busy_cap * sizeof(int)));
    }
    if (!a->busy) { a->busy_cap = 64; WL_HIP(wl_dev_alloc((void **)&a->busy, a->busy_cap * sizeof(int))); }
    if (!list.empty()) WL_HIP(hipMemcpyAsync(a->busy, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    a->nbusy = (int)list.size();
    a->nbusy_lo = a->nbusy_hi = 0;
    if (D > 2 && R.hi[2] > R.lo[2])
        for (int row : list) { a->nbusy_lo += (row / g.n[1] == R.lo[2]); a->nbusy_hi += (row / g.n[1] == R.hi[2]); }
    else
        a->nbusy_lo = a->nbusy;   // a single plane: every row reads both neighbours
    return 0;
}
template <class T, int D> static int flow_update(wl_flow *a) {
    const G g = mkG(&a->d.g);
    WL_TRY((op_rowflags<T, D>(g, (const T *)a->d.V, (const T *)a->d.mu0, (const T *)a->d.mu1, a->rowbuf, a->d.perdir_mask, D == 3 ? a->segbuf : nullptr)));
    a->seg_valid = (D == 3 && a->segbuf != nullptr);
    a->prev_valid = false;   // the arrays were written by someone else: the next native measure! rewrites every row
    return flow_compact_busy(a, g, D);
}
static BodyDev body_dev(const wl_body_desc *bodies) {
    BodyDev o;
    o.n = bodies[0].count > 0 ? bodies[0].count : 1;
    for (int l = 0; l < o.n; ++l) {
        const wl_body_desc *b = bodies + l;
        LeafDev &d = o.leaf[l];
        d.family = b->family; d.ident = b->identity_map != 0; d.op = b->op;
        for (int q = 0; q < 8; ++q) d.p[q] = b->p[q];
        for (int q = 0; q < 9; ++q) { d.A[q] = b->A[q]; d.dA[q] = b->dA[q]; d.Ainv[q] = b->Ainv[q]; }
        for (int q = 0; q < 3; ++q) { d.b[q] = b->b[q]; d.db[q] = b->db[q]; }
    }
    for (int l = o.n; l < WL_BODY_MAXLEAF; ++l) o.leaf[l] = o.leaf[0];
    return o;
}
static int measure_alloc(wl_flow *a, size_t nrows) {
    if (a->rowcount) return 0;
    WL_HIP(wl_dev_alloc((void **)&a->rowcount, nrows * sizeof(int)));
    WL_HIP(wl_dev_alloc((void **)&a->rowoff, (nrows + 1) * sizeof(long)));
    WL_HIP(wl_dev_alloc((void **)&a->touched, nrows));
    WL_HIP(wl_dev_alloc((void **)&a->prev, nrows));
    WL_HIP(wl_dev_alloc((void **)&a->changed, nrows));
    return 0;
}
template <class T, int D> static int measure_rows(wl_flow *a, const wl_body_desc *body, double eps, int64_t *nband) {
    const G g = mkG(&a->d.g);
    const size_t nrows = (size_t)g.n[1] * (size_t)(D > 2 ? g.n[2] : 1);
    WL_TRY(measure_alloc(a, nrows));
    const BodyDev B = body_dev(body);
    const T d2 = (T)((2 + eps) * (2 + eps));
    {
        Prof p(WL_K_MISC, g.cells());
        hipLaunchKernelGGL((k_measure_rows<T, D>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, B, (T *)a->d.sigma, d2,
                           a->rowcount, a->touched);
        WL_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, ctx().stream, (const int *)a->rowcount, a->rowoff, (long)nrows);
        WL_HIP(hipGetLastError());
    }
    long tot = 0;
    WL_HIP(hipMemcpyAsync(&tot, a->rowoff + nrows, sizeof(long), hipMemcpyDeviceToHost, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    a->nband = tot;
    *nband = tot;
    return 0;
}
template <class T, int D> static int measure_fill(wl_flow *a, const wl_body_desc *body, double eps, int64_t *cand) {
    const G g = mkG(&a->d.g);
    const size_t nrows = (size_t)g.n[1] * (size_t)(D > 2 ? g.n[2] : 1);
    const BodyDev B = body_dev(body);
    const T d2 = (T)((2 + eps) * (2 + eps));
    {
        Prof p(WL_K_MISC, g.cells());
        hipLaunchKernelGGL((k_measure_fill<T, D>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, B, (const T *)a->d.sigma, d2,
                           (2 + eps) * (2 + eps), eps, (T *)a->d.mu0, (T *)a->d.mu1, (T *)a->d.V, (const unsigned char *)a->touched,
                           (const unsigned char *)a->prev, !a->prev_valid, (const long *)a->rowoff, (long *)cand);
        WL_HIP(hipGetLastError());
    }
    a->nband = -1;
    // Body.jl:51-52, then what the host does after the reference's measure!: z-slab halos, body-free row flags
    const double zero[3] = {0, 0, 0};
    WL_TRY((op_bc_vec<T, D>(g, (T *)a->d.mu0, zero, 0, a->d.perdir_mask)));
    WL_TRY((op_bc_vec<T, D>(g, (T *)a->d.V, zero, a->d.exitBC, a->d.perdir_mask)));
    WL_TRY((halo_exchange<T>(g, (T *)a->d.mu0, D, 2)));
    WL_TRY((halo_exchange<T>(g, (T *)a->d.V, D, 2)));
    a->seg_valid = false;   // (the native measure! knows touched ROWS only: every segment of a busy row takes the general statement)
    hipLaunchKernelGGL((k_rowflags_touched<D>), dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, ctx().stream, g,
                       (const unsigned char *)a->touched, a->rowbuf, a->d.perdir_mask);
    WL_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_rows_changed, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, ctx().stream, (const unsigned char *)a->touched,
                       (const unsigned char *)a->prev, !a->prev_valid, a->changed_valid && a->changed_pending, a->changed, (long)nrows);
    WL_HIP(hipGetLastError());
    a->changed_valid = true;
    a->changed_pending = true;
    WL_HIP(hipMemcpyAsync(a->prev, a->touched, nrows, hipMemcpyDeviceToDevice, ctx().stream));
    WL_TRY(flow_compact_busy(a, g, D));
    a->prev_valid = true;
    return 0;
}
static int check_body(const wl_body_desc *b, int D) {
    if (!b) return fail(WL_E_ARG, "null body", __FILE__, __LINE__);
    const int n = b[0].count > 0 ? b[0].count : 1;
    if (n > WL_BODY_MAXLEAF) return fail(WL_E_ARG, "more than WL_BODY_MAXLEAF leaves in a composite body", __FILE__, __LINE__);
    for (int l = 0; l < n; ++l) {
        const int f = b[l].family;
        if (f != WL_BODY_SPHERE && f != WL_BODY_TORUS && f != WL_BODY_PLATE && f != WL_BODY_CYLINDER)
            return fail(WL_E_ARG, "unknown body family", __FILE__, __LINE__);
        if (f == WL_BODY_TORUS && D != 3) return fail(WL_E_ARG, "the torus family needs D == 3", __FILE__, __LINE__);
        if (l > 0 && (b[l].op < WL_BODY_OP_UNION || b[l].op > WL_BODY_OP_INTERSECT))
            return fail(WL_E_ARG, "unknown composite operation", __FILE__, __LINE__);
    }
    return 0;
}
// shared driver of the band reductions (pressure_force / viscous_force / pressure_moment)
template <class KERNEL> static int band_reduce(const G &gg, Scratch &S, int64_t nband, int D, double out[3], KERNEL launch) {
    out[0] = out[1] = out[2] = 0;
    int nb = (int)((nband + 255) / 256);
    if (nb > 1024) nb = 1024;
    if (nband <= 0) nb = 0;   // a rank whose slab holds no part of the body still joins the all-reduce
    if (nb > 0) {
        Prof pr(WL_K_PFORCE, nband);
        launch(nb);
        WL_HIP(hipGetLastError());
    }
    State *st = S.st;
    WL_TRY((launch_finalize<3>(gg.dist, S.partials, nb, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
        st->out[0] = v[0]; st->out[1] = v[1]; st->out[2] = v[2]; })));
    WL_TRY(S.fetch());
    for (int c = 0; c < D; ++c) out[c] = S.hst->out[c];
    return 0;
}
// Snapshot staging (VTK write / restart): planes klo..khi of a field <-> a DENSE array-of-tuples buffer on the device,
// dst[((kk*n1 + j)*n0 + i)*nct + c] = a_c[i, j, klo+kk] (c < ncomp; 0 for ncomp <= c < nct): the row padding goes, the components
// interleave (VTK's tuple order, ext/WaterLilyWriteVTKExt.jl:79 components_first).  One wavefront per x-row.
template <class T, bool PACK>
__global__ __launch_bounds__(256) void k_snapshot(G g, T *a, int ncomp, int nct, int klo, long nrows, T *buf) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int j = (int)(row % g.n[1]), kk = (int)(row / g.n[1]);
    const long src = g.at(0, j, klo + kk), dst = row * (long)g.n[0];
    for (int i = threadIdx.x & 63; i < g.n[0]; i += 64)
        for (int c = 0; c < nct; ++c) {
            if (PACK) buf[(dst + i) * nct + c] = c < ncomp ? a[(long)c * g.sc + src + i] : (T)0;
            else if (c < ncomp) a[(long)c * g.sc + src + i] = buf[(dst + i) * nct + c];
        }
}
template <class T> static int snapshot_copy(const G &g, T *a, int ncomp, int nct, int klo, int khi, T *buf, bool pack) {
    const long nrows = (long)g.n[1] * (khi - klo + 1);
    Prof p(WL_K_COPY, nrows * g.n[0] * nct);
    if (pack) hipLaunchKernelGGL((k_snapshot<T, true>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, a, ncomp, nct, klo, nrows, buf);
    else hipLaunchKernelGGL((k_snapshot<T, false>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx().stream, g, a, ncomp, nct, klo, nrows, buf);
    return (int)hipGetLastError();
}
template <class T, int D> static int restrictL_full(const G &A, T *a, const G &B, const T *b, int permask) {
    WL_TRY((op_restrictL<T, D>(A, a, B, b, permask)));
    return coarse_L_finish<T, D>(A, a, B, permask);
}

template <class T, int D> static int conv_diff_phi(const G &g, T *r, const T *u, T *Phi, double nu, int permask) {
    WL_TRY((op_conv_diff<T, D, false>(g, r, u, nu, permask, nullptr, nullptr, 0.0, nullptr, false)));
    return Phi ? op_sigma_ghosts<T, D>(g, Phi, u, nu, permask) : 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

// ---- communicator
int wl_comm_unique_id(void *out128) {
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(WL_E_STATE, ncclGetErrorString(r), __FILE__, __LINE__);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    memcpy(out128, &id, 128);
    return 0;
}
int wl_comm_init_rccl(const void *id128, int rank, int nranks) {
    if (ctx().comm) return fail(WL_E_STATE, "communicator already initialised", __FILE__, __LINE__);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    RcclComm *c = new RcclComm();
    c->rank = rank; c->size = nranks;
    ncclResult_t r = ncclCommInitRank(&c->nc, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return fail(WL_E_STATE, ncclGetErrorString(r), __FILE__, __LINE__); }
    const char *ov = getenv("WL_OVERLAP");
    if (!(ov && ov[0] == '0')) {   // overlapped exchanges get their own communicator (same ranks); on failure they share `nc`
        ncclComm_t h = nullptr;
        if (ncclCommSplit(c->nc, 0, rank, &h, nullptr) == ncclSuccess && h) c->nch = h;
    }
    ctx().comm = c;
    return 0;
}
int wl_comm_init_host(int rank, int nranks, wl_host_sendrecv_fn sr, wl_host_allreduce_fn ar, wl_host_allgather_fn ag,
                      void *user) {
    if (ctx().comm) return fail(WL_E_STATE, "communicator already initialised", __FILE__, __LINE__);
    if (!sr || !ar || !ag) return fail(WL_E_ARG, "null callback", __FILE__, __LINE__);
    HostComm *c = new HostComm();
    c->rank = rank; c->size = nranks; c->sr = sr; c->ar = ar; c->ag = ag; c->user = user;
    ctx().comm = c;
    return 0;
}
int wl_comm_init_loopback(int rank, int nranks) {
    if (ctx().comm) return fail(WL_E_STATE, "communicator already initialised", __FILE__, __LINE__);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(WL_E_ARG, "wl_comm_init_loopback: bad rank", __FILE__, __LINE__);
    LoopComm *c = new LoopComm();
    c->rank = rank; c->size = nranks;
    ctx().comm = c;
    return 0;
}
static void mailbox_release() {
    Mailbox *mb = ctx().mbox;
    if (!mb) return;
    if (mb->host) { (void)hipHostUnregister(mb->host); munmap(mb->host, mb->bytes); }
    if (mb->err_host) (void)hipHostFree(mb->err_host);
    delete mb;
    ctx().mbox = nullptr;
}
// Mailbox all-reduce (wl_common.h: Mailbox).  `shm_name` ("/wlhip-...") names a POSIX shared-memory object every rank of the
// node opens; the rank with create != 0 makes and sizes it first (the host orders the calls: create, barrier, open).
int wl_comm_mailbox(const char *shm_name, int create) {
    Comm *cm = ctx().comm;
    if (!cm) return fail(WL_E_STATE, "wl_comm_mailbox: initialise the communicator first", __FILE__, __LINE__);
    if (!shm_name || shm_name[0] != '/') return fail(WL_E_ARG, "wl_comm_mailbox: name must start with '/'", __FILE__, __LINE__);
    if (cm->size > WL_MBOX_MAXRANKS) return fail(WL_E_ARG, "wl_comm_mailbox: too many ranks", __FILE__, __LINE__);
    if (ctx().mbox) { (void)hipStreamSynchronize(ctx().stream); mailbox_release(); }   // (a replacement: nothing may still be polling the old one)
    const size_t bytes = ((2 * (size_t)cm->size * sizeof(MboxSlot) + 4095) / 4096) * 4096;
    const int fd = create ? shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(shm_name, O_RDWR, 0600);
    if (fd < 0) return fail(WL_E_STATE, "wl_comm_mailbox: shm_open failed", __FILE__, __LINE__);
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return fail(WL_E_STATE, "wl_comm_mailbox: ftruncate failed", __FILE__, __LINE__); }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) { close(fd); return fail(WL_E_STATE, "wl_comm_mailbox: shared object too small", __FILE__, __LINE__); }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // (fresh shared memory reads as zero: seq 0 = nothing posted)
    close(fd);
    if (p == MAP_FAILED) return fail(WL_E_STATE, "wl_comm_mailbox: mmap failed", __FILE__, __LINE__);
    Mailbox *mb = new Mailbox();
    mb->bytes = bytes;
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) { munmap(p, bytes); delete mb; return fail((int)e, "wl_comm_mailbox: hipHostRegister", __FILE__, __LINE__); }
    mb->host = (MboxSlot *)p;
    void *dp = nullptr;
    e = hipHostGetDevicePointer(&dp, p, 0);
    if (e == hipSuccess) e = wl_host_alloc((void **)&mb->err_host, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) { *mb->err_host = 0; e = hipHostGetDevicePointer((void **)&mb->err_dev, mb->err_host, 0); }
    if (e != hipSuccess) {
        (void)hipHostUnregister(p); munmap(p, bytes);
        if (mb->err_host) (void)hipHostFree(mb->err_host);
        delete mb;
        return fail((int)e, "wl_comm_mailbox: device mapping", __FILE__, __LINE__);
    }
    mb->dev = (MboxSlot *)dp;
    if (ctx().wall_khz <= 0.0) {   // ticks per millisecond of wall_clock64() (constant-rate counter: 100 MHz on gfx9)
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
            ctx().wall_khz = (double)khz;
        else ctx().wall_khz = 1e5;
    }
    ctx().mbox = mb;
    return 0;
}
int wl_comm_mailbox_off(void) {
    if (ctx().mbox) {
        (void)hipStreamSynchronize(ctx().stream);
        mailbox_release();
    }
    return 0;
}
int wl_comm_mailbox_active(int *on) {
    if (!on) return fail(WL_E_ARG, "null output", __FILE__, __LINE__);
    *on = ctx().mbox != nullptr;
    return 0;
}
int wl_comm_finalize(void) {
    if (ctx().comm) {
        (void)hipStreamSynchronize(ctx().stream);
        if (ctx().cstream) (void)hipStreamSynchronize(ctx().cstream);
        mailbox_release();
        delete ctx().comm;
        ctx().comm = nullptr;
    }
    return 0;
}
int wl_comm_rank(int *rank, int *nranks) {
    *rank = ctx().comm ? ctx().comm->rank : 0;
    *nranks = ctx().comm ? ctx().comm->size : 1;
    return 0;
}
int wl_halo_exchange(wl_dtype t, const wl_grid *g, void *a, int ncomp, int depth) {
    WL_TRY(check_grid(g));
    const G gg = mkG(g);
    if (t == WL_F32) return halo_exchange<float>(gg, (float *)a, ncomp, depth);
    return halo_exchange<double>(gg, (double *)a, ncomp, depth);
}
int wl_allreduce(double *vals, int n, int op) {
    Comm *cm = ctx().comm;
    if (!cm || cm->size == 1) return 0;
    if (n > 4) return fail(WL_E_ARG, "wl_allreduce: n <= 4", __FILE__, __LINE__);
    int rc = 0;
    Scratch &S = global_scratch(&rc);
    WL_TRY(rc);
    WL_HIP(hipMemcpyAsync(S.st->red, vals, sizeof(double) * n, hipMemcpyHostToDevice, ctx().stream));
    const double init = op == 0 ? 0.0 : -1e300;
    // (np = 1: the "partials" are the values themselves; st->out doubles as the staging so that input and output differ)
    double *in = S.st->red, *out4 = S.st->out;
    int rc2 = 0;
    if (ctx().mbox) {
        WL_HIP(hipMemcpyAsync(out4, in, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx().stream));
        switch (n) {
            case 1: rc2 = reduce_allreduce<1>(out4, 1, op, init, in); break;
            case 2: rc2 = reduce_allreduce<2>(out4, 1, op, init, in); break;
            case 3: rc2 = reduce_allreduce<3>(out4, 1, op, init, in); break;
            default: rc2 = reduce_allreduce<4>(out4, 1, op, init, in); break;
        }
    } else rc2 = cm->allreduce(S.st->red, n, op);
    WL_TRY(rc2);
    WL_HIP(hipMemcpyAsync(vals, S.st->red, sizeof(double) * n, hipMemcpyDeviceToHost, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    if (ctx().mbox && *ctx().mbox->err_host)
        return fail(WL_E_STATE, "mailbox all-reduce: gave up waiting for a peer rank (is every rank still running?)", __FILE__, __LINE__);
    return 0;
}

int wl_abi_version(void) { return WL_ABI_VERSION; }
const char *wl_last_error(void) { return ctx().err.c_str(); }
int wl_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail((int)e, "hipGetDeviceCount", __FILE__, __LINE__); }
    *n = c;
    return 0;
}
int wl_set_device(int dev) { WL_HIP(hipSetDevice(dev)); return 0; }
int wl_set_stream(void *s) { ctx().stream = (hipStream_t)s; return 0; }
int wl_sync(void) { WL_HIP(hipStreamSynchronize(ctx().stream)); return 0; }
int wl_malloc(void **p, size_t bytes) { WL_HIP(wl_dev_alloc(p, bytes)); return 0; }
int wl_free(void *p) { WL_HIP(hipFree(p)); return 0; }
int wl_h2d(void *dst, const void *src, size_t bytes) {
    WL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}
int wl_d2h(void *dst, const void *src, size_t bytes) {
    WL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}
int wl_memset0(void *p, size_t bytes) { WL_HIP(hipMemsetAsync(p, 0, bytes, ctx().stream)); return 0; }
int wl_h2d_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height) {
    if (width > dpitch || width > spitch) return fail(WL_E_ARG, "wl_h2d_2d: row wider than a pitch", __FILE__, __LINE__);
    WL_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}
int wl_d2h_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height) {
    if (width > dpitch || width > spitch) return fail(WL_E_ARG, "wl_d2h_2d: row wider than a pitch", __FILE__, __LINE__);
    WL_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}

#define WL_GS()                                  \
    WL_TRY(check_grid(g));                       \
    int rc__ = 0;                                \
    Scratch &S = global_scratch(&rc__);          \
    (void)S;                                     \
    WL_TRY(rc__);                                \
    const G gg = mkG(g)

int wl_bc_vec(wl_dtype t, const wl_grid *g, void *a, const double A[3], int saveexit, int perdir_mask) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_bc_vec<T, DD>(gg, (T *)a, A, saveexit, perdir_mask)));
}
int wl_bc_per(wl_dtype t, const wl_grid *g, void *a, int perdir_mask) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_bc_per<T, DD>(gg, (T *)a, perdir_mask)));
}
int wl_exit_bc(wl_dtype t, const wl_grid *g, void *u, const void *u0, const double U[3], double dt) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_exit_bc<T, DD>(gg, (T *)u, (const T *)u0, U, dt, S.partials, S.st)));
}

#define WL_RED(CALL)                 \
    do {                             \
        int r1__ = [&]() -> int { WL_DISPATCH(t, g->D, CALL); }(); \
        if (r1__) return r1__;       \
        WL_TRY(S.fetch());           \
        *out = S.hst->out[0];        \
        return 0;                    \
    } while (0)

int wl_L2_inside(wl_dtype t, const wl_grid *g, const void *a, double *out) {
    WL_GS();
    WL_RED((red_L2<T, DD>(gg, (const T *)a, S)));
}
int wl_dot(wl_dtype t, const wl_grid *g, const void *a, const void *b, double *out) {
    WL_GS();
    WL_RED((red_dot<T, DD>(gg, (const T *)a, (const T *)b, S)));
}
int wl_sum(wl_dtype t, const wl_grid *g, const void *a, double *out) {
    WL_GS();
    WL_RED((red_sum<T, DD>(gg, (const T *)a, S)));
}
int wl_max(wl_dtype t, const wl_grid *g, const void *a, double *out) {
    WL_GS();
    WL_RED((red_max<T, DD>(gg, (const T *)a, S)));
}

int wl_conv_diff(wl_dtype t, const wl_grid *g, void *r, const void *u, void *Phi, double nu, int perdir_mask) {
    WL_GS();
    WL_DISPATCH(t, g->D, (conv_diff_phi<T, DD>(gg, (T *)r, (const T *)u, (T *)Phi, nu, perdir_mask)));
}
int wl_accelerate(wl_dtype t, const wl_grid *g, void *r, const double acc[3]) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_accelerate<T, DD>(gg, (T *)r, acc)));
}
int wl_bdim(wl_dtype t, const wl_grid *g, void *u, const void *u0, void *f, const void *V, const void *mu0,
            const void *mu1, double dt) {
    WL_GS();
    WL_DISPATCH(t, g->D, (bdim_full<T, DD>(gg, (T *)u, (const T *)u0, (T *)f, (const T *)V, (const T *)mu0,
                                           (const T *)mu1, dt)));
}
int wl_scale_u(wl_dtype t, const wl_grid *g, void *u, double scale) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_scale_u<T, DD>(gg, (T *)u, scale)));
}
int wl_div(wl_dtype t, const wl_grid *g, void *z, const void *u) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_div<T, DD>(gg, (T *)z, (const T *)u)));
}
int wl_cfl(wl_dtype t, const wl_grid *g, void *sigma, const void *u, double nu, double *out) {
    WL_GS();
    WL_RED((op_cfl<T, DD>(gg, (T *)sigma, (const T *)u, nu, S.partials, S.st)));
}
int wl_set_diag(wl_dtype t, const wl_grid *g, void *Dg, void *iD, const void *L) {
    WL_GS();
    WL_DISPATCH(t, g->D, (op_set_diag<T, DD>(gg, (T *)Dg, (T *)iD, (const T *)L)));
}
int wl_restrictL(wl_dtype t, const wl_grid *ga, void *a, const wl_grid *gb, const void *b, int perdir_mask) {
    WL_TRY(check_grid(ga));
    WL_TRY(check_grid(gb));
    const G A = mkG(ga), B = mkG(gb);
    WL_DISPATCH(t, ga->D, (restrictL_full<T, DD>(A, (T *)a, B, (const T *)b, perdir_mask)));
}
int wl_restrict(wl_dtype t, const wl_grid *ga, void *a, const wl_grid *gb, const void *b) {
    WL_TRY(check_grid(ga));
    WL_TRY(check_grid(gb));
    const G A = mkG(ga), B = mkG(gb);
    WL_DISPATCH(t, ga->D, (op_restrict<T, DD>(A, (T *)a, B, (const T *)b)));
}
int wl_prolongate(wl_dtype t, const wl_grid *ga, void *a, const wl_grid *gb, const void *b) {
    WL_TRY(check_grid(ga));
    WL_TRY(check_grid(gb));
    const G A = mkG(ga), B = mkG(gb);
    WL_DISPATCH(t, ga->D, (op_prolongate<T, DD>(A, (T *)a, B, (const T *)b)));
}

// ---- multigrid handle
int wl_mg_create(wl_mg **out, wl_dtype t, int nlevels, const wl_level_desc *levels, int perdir_mask) {
    if (!out || !levels || nlevels < 1) return fail(WL_E_ARG, "wl_mg_create: bad arguments", __FILE__, __LINE__);
    if (nlevels == 2) return fail(WL_E_LEVELS, "MultiLevelPoisson requires size=a2^n, where n>2", __FILE__, __LINE__);
    for (int l = 0; l < nlevels; ++l) {
        WL_TRY(check_grid(&levels[l].g));
        const wl_level_desc &d = levels[l];
        if (!d.L || !d.D || !d.iD || !d.x || !d.eps || !d.r || !d.z)
            return fail(WL_E_ARG, "wl_mg_create: null level array", __FILE__, __LINE__);
        if (l > 0)
            for (int k = 0; k < d.g.D; ++k) {
                auto gext = [k](const wl_grid &g) { return (k == 2 && g.nzg > 0) ? g.nzg : g.n[k]; };
                if (gext(d.g) != 1 + gext(levels[l - 1].g) / 2)  // restrictML, src/MultiLevelPoisson.jl:20
                    return fail(WL_E_ARG, "wl_mg_create: level extents must be 1+N/2 of the finer level", __FILE__, __LINE__);
            }
    }
    wl_mg *m = new wl_mg();
    m->t = t; m->D = levels[0].g.D; m->nlev = nlevels; m->permask = perdir_mask;
    m->lev.assign(levels, levels + nlevels);
    int rc = m->sc.init();
    if (rc) { delete m; return rc; }
    rc = m->alloc_rowc();
    if (rc) { m->free_scratch(); m->sc.release(); delete m; return rc; }
    rc = wl_mg_update(m);
    if (rc) { m->free_scratch(); m->sc.release(); delete m; return rc; }
    *out = m;
    return 0;
}
int wl_mg_destroy(wl_mg *m) {
    if (!m) return 0;
    (void)hipStreamSynchronize(ctx().stream);
    m->sc.release();
    m->free_scratch();
    delete m;
    return 0;
}
#define WL_MG_DISPATCH(CALL) WL_DISPATCH(m->t, m->D, CALL)
#define WL_LEVEL_OK() \
    if (!m || level < 0 || level >= m->nlev) return fail(WL_E_ARG, "bad level", __FILE__, __LINE__)

int wl_mg_update(wl_mg *m) { WL_MG_DISPATCH((mg_update<T, DD>(m))); }
int wl_mg_update_changed(wl_mg *m, wl_flow *a) {
    if (!m || !a) return fail(WL_E_ARG, "null handle", __FILE__, __LINE__);
    const wl_grid &gm = m->lev[0].g, &gf = a->d.g;
    // a periodic z across slabs (ring): the ghost plane's source row lives on another rank -> full update
    const bool zper = (m->permask >> 2) & 1, yper = (m->permask >> 1) & 1;
    const bool usable = a->changed_valid && a->changed && m->D == 3 && gf.D == 3 && gm.n[0] == gf.n[0] && gm.n[1] == gf.n[1] &&
                        gm.n[2] == gf.n[2] && m->lev[0].L == a->d.mu0 && !m->rowc.empty() && m->rowc[0] && !(zper && gm.nzg > 0);
    a->changed_pending = false;   // consumed (by the partial or by the full update below)
    if (!usable) return wl_mg_update(m);
    const long nrows = (long)gm.n[1] * gm.n[2];
    if (!m->dirty) WL_HIP(wl_dev_alloc((void **)&m->dirty, (size_t)nrows));
    // D, iD and the row constants of row (j,k) read L of the rows (j,k), (j+1,k), (j,k+1)
    hipLaunchKernelGGL(k_rows_dirty, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, ctx().stream, (const unsigned char *)a->changed,
                       m->dirty, gm.n[1], gm.n[2], (int)yper, (int)zper);
    WL_HIP(hipGetLastError());
    const unsigned char *dirty = m->dirty;
    WL_MG_DISPATCH((mg_update<T, DD>(m, dirty)));
}
int wl_mg_mult(wl_mg *m, int level, void *x) {
    WL_LEVEL_OK();
    WL_MG_DISPATCH((op_mult<T, DD>(lvl<T>(m, level), (T *)x, m->permask)));
}
int wl_mg_residual(wl_mg *m, int level) {
    WL_LEVEL_OK();
    WL_MG_DISPATCH((op_residual<T, DD>(lvl<T>(m, level), m->permask, m->sc.partials, m->sc.st)));
}
int wl_mg_increment(wl_mg *m, int level) {
    WL_LEVEL_OK();
    WL_MG_DISPATCH((op_increment<T, DD>(lvl<T>(m, level), m->permask)));
}
int wl_mg_jacobi(wl_mg *m, int level, int it) {
    WL_LEVEL_OK();
    WL_MG_DISPATCH((op_jacobi<T, DD>(lvl<T>(m, level), it, m->permask)));
}
int wl_mg_pcg(wl_mg *m, int level, int it, int *n_updates) {
    WL_LEVEL_OK();
    int rc = [&]() -> int { WL_MG_DISPATCH((op_pcg<T, DD>(lvl<T>(m, level), it, m->permask, m->sc.partials, m->sc.st, false, -1, level == 0))); }();
    if (rc) return rc;
    if (n_updates) {
        WL_TRY(m->sc.fetch());
        *n_updates = m->sc.hst->nupd;
    }
    return 0;
}
int wl_mg_uniform_rows(wl_mg *m, int level, long long *n_uniform, long long *n_rows) {
    WL_LEVEL_OK();
    if (!n_uniform || !n_rows) return fail(WL_E_ARG, "wl_mg_uniform_rows: null output", __FILE__, __LINE__);
    const wl_grid &g = m->lev[level].g;
    *n_uniform = 0;
    *n_rows = (long long)(g.n[1] - 2) * (g.D == 3 ? (g.own_hi > 0 ? g.own_hi - g.own_lo + 1 : g.n[2] - 2) : 1);
    if (g.D != 3 || level >= (int)m->rowc.size() || !m->rowc[level]) return 0;
    const size_t es = m->t == WL_F32 ? 4 : 8, cnt = (size_t)g.n[1] * g.n[2] * RC_N;
    std::vector<char> h(cnt * es);
    WL_HIP(hipMemcpyAsync(h.data(), m->rowc[level], cnt * es, hipMemcpyDeviceToHost, ctx().stream));
    WL_HIP(hipStreamSynchronize(ctx().stream));
    for (size_t r = 0; r < cnt; r += RC_N) {
        const double c = m->t == WL_F32 ? (double)reinterpret_cast<const float *>(h.data())[r] : reinterpret_cast<const double *>(h.data())[r];
        if (c == c) ++*n_uniform;
    }
    return 0;
}
int wl_mg_L2(wl_mg *m, int level, double *out) {
    WL_LEVEL_OK();
    int rc = [&]() -> int { WL_MG_DISPATCH((op_L2<T, DD>(lvl<T>(m, level), m->sc.partials, m->sc.st))); }();
    if (rc) return rc;
    WL_TRY(m->sc.fetch());
    *out = m->sc.hst->r2;
    return 0;
}
int wl_mg_Linf(wl_mg *m, int level, double *out) {
    WL_LEVEL_OK();
    if (!out) return fail(WL_E_ARG, "wl_mg_Linf: null output", __FILE__, __LINE__);
    int rc = [&]() -> int { WL_MG_DISPATCH((mg_Linf<T, DD>(m, level))); }();
    if (rc) return rc;
    WL_TRY(m->sc.fetch());
    *out = m->sc.hst->out[1];
    return 0;
}
int wl_mg_log(wl_mg *m, int on) {
    if (!m) return fail(WL_E_ARG, "null handle", __FILE__, __LINE__);
    m->log_on = on != 0;
    if (!on) m->log.clear();
    return 0;
}
int wl_mg_log_read(wl_mg *m, double *rows, int cap, int *n) {
    if (!m || !n || (cap > 0 && !rows)) return fail(WL_E_ARG, "wl_mg_log_read: null argument", __FILE__, __LINE__);
    const int have = (int)(m->log.size() / 3);
    *n = have;
    if (cap <= 0) return 0;                        // a query: nothing is consumed
    const int take = have < cap ? have : cap;
    for (int q = 0; q < 3 * take; ++q) rows[q] = m->log[q];
    m->log.erase(m->log.begin(), m->log.begin() + 3 * take);   // what did not fit stays for the next read
    return 0;
}
int wl_mg_vcycle(wl_mg *m, int level) {
    if (!m || level < 0 || level + 1 >= m->nlev) return fail(WL_E_ARG, "bad level", __FILE__, __LINE__);
    WL_MG_DISPATCH((mg_vcycle<T, DD>(m, level)));
}
int wl_mg_solve(wl_mg *m, double tol, int itmx, int *n_iter) {
    if (!m) return fail(WL_E_ARG, "null handle", __FILE__, __LINE__);
    WL_MG_DISPATCH((mg_solve<T, DD>(m, tol, itmx, n_iter)));
}

// ---- flow handle
int wl_flow_create(wl_flow **out, wl_dtype t, const wl_flow_desc *d) {
    if (!out || !d) return fail(WL_E_ARG, "wl_flow_create: bad arguments", __FILE__, __LINE__);
    WL_TRY(check_grid(&d->g));
    if (!d->u || !d->u0 || !d->f || !d->p || !d->sigma || !d->V || !d->mu0 || !d->mu1)
        return fail(WL_E_ARG, "wl_flow_create: null field", __FILE__, __LINE__);
    wl_flow *a = new wl_flow();
    a->t = t; a->d = *d;
    int rc = a->sc.init();
    if (rc) { delete a; return rc; }
    const size_t nrows = (size_t)d->g.n[1] * (size_t)(d->g.D > 2 ? d->g.n[2] : 1);
    if (wl_dev_alloc((void **)&a->rowbuf, nrows) != hipSuccess) { a->sc.release(); delete a; return fail(WL_E_STATE, "wl_dev_alloc(row flags)", __FILE__, __LINE__); }
    if (d->g.D > 2) {
        const size_t ntx = (size_t)(d->g.n[0] - 2 + 63) / 64;
        if (wl_dev_alloc((void **)&a->segbuf, nrows * (ntx ? ntx : 1)) != hipSuccess) {
            (void)hipFree(a->rowbuf); a->sc.release(); delete a;
            return fail(WL_E_STATE, "wl_dev_alloc(segment flags)", __FILE__, __LINE__);
        }
    }
    *out = a;
    return 0;
}
int wl_flow_destroy(wl_flow *a) {
    if (!a) return 0;
    (void)hipStreamSynchronize(ctx().stream);
    a->sc.release();
    if (a->rowbuf) (void)hipFree(a->rowbuf);
    if (a->segbuf) (void)hipFree(a->segbuf);
    if (a->busy) (void)hipFree(a->busy);
    if (a->rowcount) (void)hipFree(a->rowcount);
    if (a->rowoff) (void)hipFree(a->rowoff);
    if (a->touched) (void)hipFree(a->touched);
    if (a->prev) (void)hipFree(a->prev);
    if (a->changed) (void)hipFree(a->changed);
    delete a;
    return 0;
}
int wl_flow_update(wl_flow *a) {
    if (!a) return fail(WL_E_ARG, "null handle", __FILE__, __LINE__);
    a->changed_valid = false;   // the arrays were rewritten by the caller: nothing is known about which rows changed
    a->changed_pending = false;
    WL_DISPATCH(a->t, a->d.g.D, (flow_update<T, DD>(a)));
}
int wl_measure_rows(wl_flow *a, const wl_body_desc *body, double eps, int64_t *nband) {
    if (!a || !nband) return fail(WL_E_ARG, "wl_measure_rows: null argument", __FILE__, __LINE__);
    WL_TRY(check_body(body, a->d.g.D));
    if (!(eps > 0)) return fail(WL_E_ARG, "wl_measure_rows: eps must be positive", __FILE__, __LINE__);
    WL_DISPATCH(a->t, a->d.g.D, (measure_rows<T, DD>(a, body, eps, nband)));
}
int wl_measure_fill(wl_flow *a, const wl_body_desc *body, double eps, int64_t *cand_dev) {
    if (!a) return fail(WL_E_ARG, "null handle", __FILE__, __LINE__);
    WL_TRY(check_body(body, a->d.g.D));
    if (a->nband < 0) return fail(WL_E_STATE, "wl_measure_fill: call wl_measure_rows first", __FILE__, __LINE__);
    if (a->nband > 0 && !cand_dev) return fail(WL_E_ARG, "wl_measure_fill: null candidate buffer", __FILE__, __LINE__);
    WL_DISPATCH(a->t, a->d.g.D, (measure_fill<T, DD>(a, body, eps, cand_dev)));
}
int wl_body_nds(const wl_grid *g, const wl_body_desc *body, const int64_t *cand_dev, int64_t n, double *nds_dev) {
    WL_TRY(check_grid(g));
    WL_TRY(check_body(body, g->D));
    if (n <= 0) return 0;
    if (!cand_dev || !nds_dev) return fail(WL_E_ARG, "wl_body_nds: null buffer", __FILE__, __LINE__);
    const G gg = mkG(g);
    const BodyDev B = body_dev(body);
    Prof p(WL_K_PFORCE, n);
    if (g->D == 3) hipLaunchKernelGGL((k_body_nds<3>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, gg, B, (const long *)cand_dev, (long)n, nds_dev);
    else hipLaunchKernelGGL((k_body_nds<2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, gg, B, (const long *)cand_dev, (long)n, nds_dev);
    return (int)hipGetLastError();
}
static int check_pair(const wl_flow *a, const wl_mg *b) {
    if (!a || !b) return fail(WL_E_ARG, "null handle", __FILE__, __LINE__);
    if (a->t != b->t || a->d.g.D != b->D) return fail(WL_E_ARG, "flow/poisson type mismatch", __FILE__, __LINE__);
    const wl_level_desc &l0 = b->lev[0];
    // src/WaterLily.jl:77: pois.x === flow.p, pois.L === flow.mu0, pois.z === flow.sigma
    if (l0.x != a->d.p || l0.L != a->d.mu0 || l0.z != a->d.sigma)
        return fail(WL_E_ARG, "level-1 (x,L,z) must alias flow (p,mu0,sigma)", __FILE__, __LINE__);
    for (int k = 0; k < 3; ++k)
        if (l0.g.n[k] != a->d.g.n[k] || l0.g.s[k] != a->d.g.s[k]) return fail(WL_E_ARG, "flow/poisson grid mismatch", __FILE__, __LINE__);
    return 0;
}
int wl_project(wl_flow *a, wl_mg *b, double dt, double w, int *n_iter) {
    WL_TRY(check_pair(a, b));
    WL_DISPATCH(a->t, a->d.g.D, (flow_project<T, DD>(a, b, dt, w, n_iter)));
}
int wl_mom_step(wl_flow *a, wl_mg *b, double dt, const double U[3], const double *acc_pred, const double *acc_corr,
                double *dt_next, int n_iter[2]) {
    WL_TRY(check_pair(a, b));
    if (!U || !dt_next || !n_iter) return fail(WL_E_ARG, "wl_mom_step: null output", __FILE__, __LINE__);
    WL_DISPATCH(a->t, a->d.g.D, (flow_mom_step<T, DD>(a, b, dt, U, acc_pred, acc_corr, dt_next, n_iter)));
}

// ---- Metrics.jl:94-100
int wl_vforce(wl_dtype t, const wl_grid *g, const void *u, const int64_t *idx, const double *nds, int64_t nband, double nu,
              double out[3]) {
    WL_GS();
    return band_reduce(gg, S, nband, g->D, out, [&](int nb) {
        if (t == WL_F32) hipLaunchKernelGGL(k_vforce<float>, dim3(nb), dim3(256), 0, ctx().stream, gg, (const float *)u, idx, nds, nband, (float)nu, S.partials);
        else hipLaunchKernelGGL(k_vforce<double>, dim3(nb), dim3(256), 0, ctx().stream, gg, (const double *)u, idx, nds, nband, nu, S.partials);
    });
}
int wl_pmoment(wl_dtype t, const wl_grid *g, const void *p, const int64_t *idx, const double *nds, int64_t nband,
               const double x0[3], double out[3]) {
    WL_GS();
    const double a = x0[0], b = x0[1], c = g->D > 2 ? x0[2] : 0.0;
    return band_reduce(gg, S, nband, g->D, out, [&](int nb) {
        if (t == WL_F32) hipLaunchKernelGGL(k_pmoment<float>, dim3(nb), dim3(256), 0, ctx().stream, gg, (const float *)p, idx, nds, nband, a, b, c, S.partials);
        else hipLaunchKernelGGL(k_pmoment<double>, dim3(nb), dim3(256), 0, ctx().stream, gg, (const double *)p, idx, nds, nband, a, b, c, S.partials);
    });
}
int wl_metric(wl_dtype t, const wl_grid *g, int kind, void *out, const void *u, int ipar, const double par[3],
              const double par2[3]) {
    WL_GS();
    if (kind < 0 || kind > WL_M_LAMBDA2 || (kind >= WL_M_OMAG && g->D != 3) || (kind == WL_M_CURL && (ipar < 0 || ipar > 2)))
        return fail(WL_E_ARG, "wl_metric: bad kind/component for this dimension", __FILE__, __LINE__);
    WL_DISPATCH(t, g->D, (op_metric<T, DD>(gg, kind, (T *)out, (const T *)u, ipar, par, par2)));
}
int wl_pforce(wl_dtype t, const wl_grid *g, const void *p, const int64_t *idx, const double *nds, int64_t nband,
              double out[3]) {
    WL_GS();
    (void)gg;
    out[0] = out[1] = out[2] = 0;
    int nb = (int)((nband + 255) / 256);
    if (nb > 1024) nb = 1024;
    if (nband <= 0) nb = 0;   // a rank whose slab holds no part of the body still joins the all-reduce
    if (nb > 0) {
        Prof pr(WL_K_PFORCE, nband);
        if (t == WL_F32)
            hipLaunchKernelGGL(k_pforce<float>, dim3(nb), dim3(256), 0, ctx().stream, (const float *)p, idx, nds, nband, g->D, S.partials);
        else
            hipLaunchKernelGGL(k_pforce<double>, dim3(nb), dim3(256), 0, ctx().stream, (const double *)p, idx, nds, nband, g->D, S.partials);
        WL_HIP(hipGetLastError());
    }
    State *st = S.st;
    WL_TRY((launch_finalize<3>(gg.dist, S.partials, nb, RED_SUM, 0.0, st->red, [=] __device__(const double *v) {
        st->out[0] = v[0]; st->out[1] = v[1]; st->out[2] = v[2]; })));
    WL_TRY(S.fetch());
    for (int c = 0; c < g->D; ++c) out[c] = S.hst->out[c];
    return 0;
}

static int snapshot_args(const wl_grid *g, const void *a, int ncomp, int nct, int klo, int khi, const void *buf) {
    WL_TRY(check_grid(g));
    if (!a || !buf) return fail(WL_E_ARG, "wl_snapshot: null buffer", __FILE__, __LINE__);
    if (ncomp < 1 || nct < ncomp || nct > 9) return fail(WL_E_ARG, "wl_snapshot: need 1 <= ncomp <= ntuple <= 9", __FILE__, __LINE__);
    const int n2 = g->D == 3 ? g->n[2] : 1;
    if (klo < 0 || khi >= n2 || klo > khi) return fail(WL_E_ARG, "wl_snapshot: plane range outside the local array", __FILE__, __LINE__);
    return 0;
}
int wl_snapshot_pack(wl_dtype t, const wl_grid *g, const void *a, int ncomp, int ntuple, int klo, int khi, void *dst) {
    WL_TRY(snapshot_args(g, a, ncomp, ntuple, klo, khi, dst));
    const G gg = mkG(g);
    if (t == WL_F32) return snapshot_copy<float>(gg, (float *)const_cast<void *>(a), ncomp, ntuple, klo, khi, (float *)dst, true);
    return snapshot_copy<double>(gg, (double *)const_cast<void *>(a), ncomp, ntuple, klo, khi, (double *)dst, true);
}
int wl_snapshot_unpack(wl_dtype t, const wl_grid *g, void *a, int ncomp, int ntuple, int klo, int khi, const void *src) {
    WL_TRY(snapshot_args(g, a, ncomp, ntuple, klo, khi, src));
    const G gg = mkG(g);
    if (t == WL_F32) return snapshot_copy<float>(gg, (float *)a, ncomp, ntuple, klo, khi, (float *)const_cast<void *>(src), false);
    return snapshot_copy<double>(gg, (double *)a, ncomp, ntuple, klo, khi, (double *)const_cast<void *>(src), false);
}

int wl_set_option(int key, int value) {
    if (key < 0 || key >= 32 || !((WL_OPT_LIVE >> key) & 1u)) return fail(WL_E_ARG, "wl_set_option: no such option", __FILE__, __LINE__);
    ctx().opt[key] = value;
    return 0;
}
int wl_get_option(int key, int *value) {
    if (!value || key < 0 || key >= 32 || !((WL_OPT_LIVE >> key) & 1u)) return fail(WL_E_ARG, "wl_get_option: no such option", __FILE__, __LINE__);
    *value = ctx().opt[key];
    return 0;
}

// ---- measurement support
static const char *KNAMES[WL_K_COUNT] = {
    "conv_diff", "bdim", "bc", "div", "correct", "cfl", "scale", "residual", "jacobi", "increment", "smooth",
    "restrict", "prolongate", "pcg_init", "pcg_mult_dot", "pcg_update", "pcg_direction", "dot", "scalar", "set_diag",
    "restrictL", "copy", "pforce", "misc"};
const char *wl_kernel_name(int k) { return (k >= 0 && k < WL_K_COUNT) ? KNAMES[k] : "?"; }
int wl_prof_select(int kclass, int64_t min_cells) {
    ctx().prof_class = kclass;
    ctx().prof_min_cells = min_cells;
    return 0;
}
int wl_prof_reset(void) {
    Ctx &c = ctx();
    for (int k = 0; k < WL_K_COUNT; ++k) { c.launches[k] = 0; c.cells[k] = 0; }
    for (auto &e : c.evts) { c.pool.push_back(e.a); c.pool.push_back(e.b); }
    c.evts.clear();
    if (c.comm) for (int q = 0; q < 6; ++q) c.comm->cnt[q] = 0;
    return 0;
}
// back-to-back all-reduces of one double on the library's stream, as the solver issues them (local reduction of a few
// partials + exchange), timed with two events: microseconds per all-reduce.  Every rank calls it with the same `reps`.
int wl_prof_allreduce_us(int reps, double *us_per_op) {
    Comm *cm = ctx().comm;
    if (!us_per_op || reps < 1) return fail(WL_E_ARG, "wl_prof_allreduce_us: bad arguments", __FILE__, __LINE__);
    *us_per_op = 0.0;
    if (!cm || cm->size == 1) return 0;
    int rc = 0;
    Scratch &S = global_scratch(&rc);
    WL_TRY(rc);
    WL_HIP(hipMemsetAsync(S.partials, 0, sizeof(double) * 64, ctx().stream));
    hipEvent_t a, b;
    WL_HIP(hipEventCreate(&a));
    WL_HIP(hipEventCreate(&b));
    for (int w = 0; w < 3 && !rc; ++w) rc = reduce_allreduce<1>(S.partials, 64, RED_SUM, 0.0, S.st->red);   // warm-up
    if (!rc) rc = (int)hipEventRecord(a, ctx().stream);
    for (int q = 0; q < reps && !rc; ++q) rc = reduce_allreduce<1>(S.partials, 64, RED_SUM, 0.0, S.st->red);
    if (!rc) rc = (int)hipEventRecord(b, ctx().stream);
    if (!rc) rc = (int)hipEventSynchronize(b);
    float ms = 0;
    if (!rc) rc = (int)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (rc) return rc < 10000 ? fail(rc, "wl_prof_allreduce_us", __FILE__, __LINE__) : rc;
    if (ctx().mbox && *ctx().mbox->err_host) return fail(WL_E_STATE, "mailbox all-reduce: gave up waiting for a peer rank", __FILE__, __LINE__);
    *us_per_op = (double)ms * 1e3 / reps;
    return 0;
}
int wl_prof_reset_comm(void) {
    if (ctx().comm) for (int q = 0; q < 6; ++q) ctx().comm->cnt[q] = 0;
    return 0;
}
int wl_prof_comm(int64_t out[6]) {
    if (!out) return fail(WL_E_ARG, "wl_prof_comm: null output", __FILE__, __LINE__);
    for (int q = 0; q < 6; ++q) out[q] = ctx().comm ? ctx().comm->cnt[q] : 0;
    return 0;
}
int wl_prof_allocs(int64_t *count, int64_t *bytes) {
    if (!count || !bytes) return fail(WL_E_ARG, "wl_prof_allocs: null output", __FILE__, __LINE__);
    *count = ctx().n_alloc;
    *bytes = ctx().alloc_bytes;
    return 0;
}
int wl_prof_counts(int kclass, int64_t *launches, int64_t *cells) {
    if (kclass < 0 || kclass >= WL_K_COUNT) return fail(WL_E_ARG, "bad kernel class", __FILE__, __LINE__);
    *launches = ctx().launches[kclass];
    *cells = ctx().cells[kclass];
    return 0;
}
int wl_prof_overlapped(int64_t *count) {
    if (!count) return fail(WL_E_ARG, "wl_prof_overlapped: null output", __FILE__, __LINE__);
    *count = ctx().n_overlapped;
    return 0;
}
int wl_prof_timed(int64_t *launches, int64_t *cells, double *ms) {
    Ctx &c = ctx();
    WL_HIP(hipStreamSynchronize(c.stream));
    int64_t n = 0, nc = 0;
    double tot = 0;
    for (auto &e : c.evts) {
        float t = 0;
        WL_HIP(hipEventElapsedTime(&t, e.a, e.b));
        tot += t; ++n; nc += e.cells;
    }
    *launches = n; *cells = nc; *ms = tot;
    return 0;
}

}  // extern "C"

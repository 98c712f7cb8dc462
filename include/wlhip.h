/* wlhip.h -- C ABI of libwlhip.so, the MI355X-native backend for WaterLily's `sim_step!` hot path.
 *
 * This header is the drop-in boundary.  Every entry point replaces one method of the reference
 * (/root/reference, file:line cited per function) that a `mem=<device array>` backend overrides at
 * function granularity (precedent: ext/WaterLilyAMDGPUExt.jl:24).
 * The reference-side binding (a Julia package extension made of `ccall`s) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, doubles; no C++/torch types; every function returns
 *     0 on success or a non-zero code (hipError_t, or WL_E_* below); wl_last_error() gives the text.
 *   - arrays follow the reference layout (src/Flow.jl:112-118): column-major, ONE ghost layer per side,
 *     vector fields as separate component blocks (SoA).  Strides are explicit (wl_grid), so both the dense
 *     Julia layout (s = {1, n0, n0*n1}, sc = n0*n1*n2) and a padded/aligned allocation are accepted.
 *   - directions/components are 0-based; `perdir_mask` bit j set = direction j periodic
 *     (reference: 1-based tuple `perdir`).
 *   - scalars cross the ABI as double and are rounded to the field type T where the reference holds a T.
 *   - all work is enqueued on one HIP stream (wl_set_stream; default: the null stream).  Functions that
 *     return a scalar synchronise that stream; all others are asynchronous.
 *   - the caller owns every field array; handles own only internal scratch (reduction partials, solver
 *     scalars).  Aliasing required by the reference is honoured: pois.x === flow.p, pois.L === flow.mu0,
 *     pois.z === flow.sigma (src/WaterLily.jl:77) -- pass the same pointers.
 */
#ifndef WLHIP_H
#define WLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WL_ABI_VERSION 6

typedef enum wl_dtype { WL_F32 = 0, WL_F64 = 1 } wl_dtype;

enum { WL_OK = 0, WL_E_ARG = 10001, WL_E_LEVELS = 10002, WL_E_NOGPU = 10003, WL_E_STATE = 10004 };

/* One grid (= one multigrid level).  n[] INCLUDES the ghost layer (src/Flow.jl:112 `Ng = N .+ 2`). */
typedef struct wl_grid {
    int32_t D;     /* 2 or 3 */
    int32_t n[3];  /* LOCAL extents incl. ghost/halo planes; n[2] = 1 when D == 2 */
    int64_t s[3];  /* element strides; s[0] must be 1 */
    int64_t sc;    /* element stride between the components of a vector field on this grid */
    /* z-slab decomposition (multi-GPU, D == 3).  nzg == 0 means "not decomposed" (the other three are ignored).
     * nzg  : extent along z of the UNDECOMPOSED array, ghosts included (the reference's N[3]+2);
     * kz0  : global z index of local plane 0 (may be negative: unused padding plane below the domain);
     * own_lo, own_hi : LOCAL plane range this rank owns (writes); planes outside are halo copies of a
     *        neighbour's owned planes (filled by wl_halo_exchange) or unused padding. */
    int32_t nzg, kz0, own_lo, own_hi;
    /* zring != 0: z is periodic ACROSS the slabs -- the ranks form a ring, the two z ghost planes are owned by nobody
     * and are filled, like every halo plane, by the neighbour exchange (rank 0 <-> rank P-1 included). */
    int32_t zring;
} wl_grid;

/* ------------------------------------------------------------------ runtime */
int wl_abi_version(void);
const char *wl_last_error(void);
int wl_device_count(int *n);
int wl_set_device(int dev);
int wl_set_stream(void *hip_stream); /* hipStream_t; NULL = null stream */
int wl_sync(void);
/* device memory for hosts without their own allocator (the Julia shim); torch hosts pass data_ptr() */
int wl_malloc(void **p, size_t bytes);
int wl_free(void *p);
int wl_h2d(void *dst, const void *src, size_t bytes);
int wl_d2h(void *dst, const void *src, size_t bytes);
int wl_memset0(void *p, size_t bytes);
/* the same copies between a dense host array and a PITCHED device array (rows of `width` bytes, `height` of them: every x-row of
 * every plane and component): how a host keeps the reference's dense arrays on its side (`Array(a)`, `copyto!`) while the
 * device rows sit on 128-byte boundaries -- the layout the kernels are 5 % faster on */
int wl_h2d_2d(void *dst_dev, size_t dpitch, const void *src_host, size_t spitch, size_t width, size_t height);
int wl_d2h_2d(void *dst_host, size_t dpitch, const void *src_dev, size_t spitch, size_t width, size_t height);

/* ------------------------------------------------------------------ multi-GPU communicator (one per process)
 * z-slab decomposition, one process per GPU.  Production: RCCL over xGMI -- rank 0 calls
 * wl_comm_unique_id, the host broadcasts the 128 bytes (torch.distributed / MPI / files), every rank calls
 * wl_comm_init_rccl.  Testing: wl_comm_init_host routes every collective through host callbacks (device
 * buffers are staged through pinned memory), so any transport (torch.distributed gloo) can carry it. */
int wl_comm_unique_id(void *out128);
int wl_comm_init_rccl(const void *id128, int rank, int nranks);
/* sendrecv: exchange `bytes` with both z-neighbours (host pointers; NULL where there is no neighbour).
 * allreduce: in-place on n doubles, op 0 = sum, 1 = max.  allgather: every rank contributes `bytes` at
 * offset rank*bytes of `buf`. */
typedef int (*wl_host_sendrecv_fn)(void *user, const void *send_lo, void *recv_lo, const void *send_hi, void *recv_hi,
                                   int64_t bytes, int peer_lo, int peer_hi);
typedef int (*wl_host_allreduce_fn)(void *user, double *vals, int n, int op);
typedef int (*wl_host_allgather_fn)(void *user, void *buf, int64_t bytes);
int wl_comm_init_host(int rank, int nranks, wl_host_sendrecv_fn sr, wl_host_allreduce_fn ar, wl_host_allgather_fn ag,
                      void *user);
/* Measurement: ONE process plays rank `rank` of an `nranks`-way z-slab run on one GPU.  Its neighbours are taken to be copies of
 * itself (the planes it would send up arrive from below and vice versa, as device copies), a sum over the ranks is nranks times
 * the local value, an all-gather repeats the local segment: the compute and launch time of a rank, without a wire. */
int wl_comm_init_loopback(int rank, int nranks);
/* Mailbox all-reduce for the run's scalars (dot products, CFL maximum, force sums): after the communicator exists every rank
 * of the NODE opens the same POSIX shared-memory object `shm_name` ("/name"); the one rank that passes create != 0 must
 * have returned before the others call (the host orders it: create on rank 0, barrier, open elsewhere, barrier, then rank 0
 * may shm_unlink the name).  From then on the library sums its scalars through that block of pinned host memory -- one
 * system-scope store + one poll per peer, combined in rank order (bit-identical on every rank) -- instead of one RCCL
 * all-reduce per value; halo planes and the coarse-level all-gather stay on RCCL.  Waits are bounded in wall-clock time
 * (wl_set_option(26), seconds; 0 = unbounded): a rank that gives up reports WL_E_STATE at the caller's next synchronising
 * call.  Optional: without it scalars use ncclAllReduce. */
int wl_comm_mailbox(const char *shm_name, int create);
int wl_comm_mailbox_off(void);          /* back to the communicator's all-reduce (every rank must call it at the same point) */
int wl_comm_mailbox_active(int *on);
int wl_comm_finalize(void);
int wl_comm_rank(int *rank, int *nranks);
/* fill the halo planes of a (vector) field from the z-neighbours: `depth` planes each side, `ncomp` components */
int wl_halo_exchange(wl_dtype t, const wl_grid *g, void *a, int ncomp, int depth);
/* all-reduce n host doubles over the ranks (op 0 = sum, 1 = max); identity when there is no communicator */
int wl_allreduce(double *vals_host, int n, int op);

/* ------------------------------------------------------------------ util.jl operators */
/* BC!(a,A,saveexit,perdir)            src/util.jl:192-210 */
int wl_bc_vec(wl_dtype t, const wl_grid *g, void *a, const double A[3], int saveexit, int perdir_mask);
/* perBC!(a,perdir)                    src/util.jl:227-231 */
int wl_bc_per(wl_dtype t, const wl_grid *g, void *a, int perdir_mask);
/* exitBC!(u,u0,U,dt)                  src/util.jl:216-222 */
int wl_exit_bc(wl_dtype t, const wl_grid *g, void *u, const void *u0, const double U[3], double dt);
/* L2(a) = sum(abs2, inside(a))        src/util.jl:68 (ext/WaterLilyAMDGPUExt.jl:24) */
int wl_L2_inside(wl_dtype t, const wl_grid *g, const void *a, double *out);
/* dot / sum / maximum over the WHOLE array, ghost cells included: LinearAlgebra.dot, Base.sum, Base.maximum as used at
 * src/Poisson.jl:94,126,131,137,146 and src/Flow.jl:174 (z-slab runs: over the planes the ranks own, all-reduced) */
int wl_dot(wl_dtype t, const wl_grid *g, const void *a, const void *b, double *out);
int wl_sum(wl_dtype t, const wl_grid *g, const void *a, double *out);
int wl_max(wl_dtype t, const wl_grid *g, const void *a, double *out);

/* ------------------------------------------------------------------ Flow.jl operators */
/* conv_diff!(r,u,Phi;nu,perdir)       src/Flow.jl:36-60.  The kernels are in gather form and need no scratch; what the
 * reference's scatter form LEAVES in Phi where a later whole-array reduction reads it -- the top ghost cells (its loops run
 * over inside_u, util.jl:55-57: "top ghost included"; Phi is flow.sigma inside mom_step!, src/Flow.jl:157,164) -- is written
 * to Phi when Phi != NULL: each such cell gets the flux of the last (i,j) loop pair whose range holds it.  Interior cells of
 * Phi (scratch the reference overwrites before reading) are not touched. */
int wl_conv_diff(wl_dtype t, const wl_grid *g, void *r, const void *u, void *Phi, double nu, int perdir_mask);
/* accelerate!(r,dt,g,U)               src/Flow.jl:68-73  (host evaluates g(i,t)+dU_i/dt) */
int wl_accelerate(wl_dtype t, const wl_grid *g, void *r, const double acc[3]);
/* BDIM!(a)                            src/Flow.jl:131-135 */
int wl_bdim(wl_dtype t, const wl_grid *g, void *u, const void *u0, void *f, const void *V, const void *mu0,
            const void *mu1, double dt);
/* scale_u!(a,scale)                   src/Flow.jl:170 */
int wl_scale_u(wl_dtype t, const wl_grid *g, void *u, double scale);
/* @inside z[I] = div(I,u)             src/Flow.jl:139 (div: :11-17) */
int wl_div(wl_dtype t, const wl_grid *g, void *z, const void *u);
/* CFL(a)                              src/Flow.jl:172-182 */
int wl_cfl(wl_dtype t, const wl_grid *g, void *sigma, const void *u, double nu, double *dt_out);

/* ------------------------------------------------------------------ Poisson.jl / MultiLevelPoisson.jl */
/* set_diag!(D,iD,L)                   src/Poisson.jl:42-54 */
int wl_set_diag(wl_dtype t, const wl_grid *g, void *D, void *iD, const void *L);
/* restrictL!(a,b;perdir)              src/MultiLevelPoisson.jl:26-32 */
int wl_restrictL(wl_dtype t, const wl_grid *ga, void *a, const wl_grid *gb, const void *b, int perdir_mask);
/* restrict!(a,b), prolongate!(a,b)    src/MultiLevelPoisson.jl:33-34 */
int wl_restrict(wl_dtype t, const wl_grid *ga, void *a, const wl_grid *gb, const void *b);
int wl_prolongate(wl_dtype t, const wl_grid *ga, void *a, const wl_grid *gb, const void *b);

/* One level of the hierarchy = the fields of `Poisson` (src/Poisson.jl:21-30); caller-owned arrays. */
typedef struct wl_level_desc {
    wl_grid g;
    void *L, *D, *iD, *x, *eps, *r, *z;
} wl_level_desc;

typedef struct wl_mg wl_mg; /* opaque: Poisson (nlevels==1) or MultiLevelPoisson */

/* Poisson(x,L,z) src/Poisson.jl:31-37 / MultiLevelPoisson(x,L,z) src/MultiLevelPoisson.jl:51-59.
 * The host builds the level shapes (`restrictML`, :18-25) and allocates; create() validates them, fills
 * the coarse L by restrictL! and D,iD by set_diag!.  nlevels==2 is rejected with WL_E_LEVELS
 * ("MultiLevelPoisson requires size=a2^n, where n>2"). */
int wl_mg_create(wl_mg **out, wl_dtype t, int nlevels, const wl_level_desc *levels, int perdir_mask);
int wl_mg_destroy(wl_mg *m);
/* update!(ml)                         src/MultiLevelPoisson.jl:62-68 (src/Poisson.jl:46 for one level) */
int wl_mg_update(wl_mg *m);
/* Introspection: how many interior x-rows of `level` carry one coefficient value on all their faces (the 7-point kernels
 * skip the loads of L there, see wl_set_option key 9) out of how many owned interior rows.  0 for D==2. */
int wl_mg_uniform_rows(wl_mg *m, int level, long long *n_uniform, long long *n_rows);
/* mult!(p,x): p.z = A x               src/Poisson.jl:62-68 */
int wl_mg_mult(wl_mg *m, int level, void *x);
/* residual!(p)                        src/Poisson.jl:91-97 */
int wl_mg_residual(wl_mg *m, int level);
/* increment!(p)                       src/Poisson.jl:99-103 */
int wl_mg_increment(wl_mg *m, int level);
/* Jacobi!(p;it)                       src/Poisson.jl:110-113 */
int wl_mg_jacobi(wl_mg *m, int level, int it);
/* pcg!(p;it)                          src/Poisson.jl:123-143; n_updates = number of (x,r) updates done */
int wl_mg_pcg(wl_mg *m, int level, int it, int *n_updates);
/* L2(p) = r.r                         src/Poisson.jl:146 */
int wl_mg_L2(wl_mg *m, int level, double *out);
/* L∞(p) = maximum(abs, p.r)            src/Poisson.jl:147 (over inside(r): its ghost entries are zero) */
int wl_mg_Linf(wl_mg *m, int level, double *out);
/* Vcycle!(ml;l)                       src/MultiLevelPoisson.jl:70-82 */
int wl_mg_vcycle(wl_mg *m, int level);
/* solver!(p;tol,itmx)                 src/MultiLevelPoisson.jl:87-99 (src/Poisson.jl:162-172 for one
 * level).  n_iter receives the value the reference pushes onto `p.n`. */
int wl_mg_solve(wl_mg *m, double tol, int itmx, int *n_iter);
/* The pressure-solver log of the reference (`@log ", $n, $(L∞(p)), $r₂\n"`, src/Poisson.jl:164,167,
 * src/MultiLevelPoisson.jl:90,94; macro and file format: src/util.jl:4-24).  wl_mg_log(m, 1) makes every later
 * wl_mg_solve / wl_project / wl_mom_step on this hierarchy record one row {n, L∞(p), L₂(p)} for the initial residual
 * (n = 0) and for each iteration -- two extra reductions and a host synchronisation per row, so it is off by default.
 * wl_mg_log_read copies up to `cap` rows (3 doubles each, oldest first) into `rows` and removes
 * them from the record; *n receives the number of rows that were waiting (rows beyond cap stay for the next read; cap = 0 is a
 * query that consumes nothing).  The host prints the "p" / "c"
 * prefixes of src/Flow.jl:158,165 itself. */
int wl_mg_log(wl_mg *m, int on);
int wl_mg_log_read(wl_mg *m, double *rows, int cap, int *n);

/* ------------------------------------------------------------------ Flow (src/Flow.jl:92-122) */
typedef struct wl_flow_desc {
    wl_grid g;
    void *u, *u0, *f, *p, *sigma, *V, *mu0, *mu1; /* mu1[I,i,j] = component i + D*j */
    double nu;
    int32_t exitBC;
    int32_t perdir_mask;
} wl_flow_desc;

typedef struct wl_flow wl_flow; /* opaque */

int wl_flow_create(wl_flow **out, wl_dtype t, const wl_flow_desc *desc);
int wl_flow_destroy(wl_flow *a);
/* Must be called after the coefficient fields mu0, mu1, V were (re)written -- i.e. at the end of measure!(flow,body)
 * (src/Body.jl:31-53), next to update!(pois).  It rebuilds the per-row "body-free" flags (mu1 == 0, V == 0,
 * mu0 == 1 on a whole x-row) that let BDIM! skip those 15 coefficient reads where they are known constants; results
 * are identical with or without the flags.  Until the first call every row takes the general path. */
int wl_flow_update(wl_flow *a);
/* update!(pois) after a native measure! (wl_measure_fill) of the flow whose mu0 is this hierarchy's L: on level 0 only the
 * x-rows that measure! rewrote (and their lower y / z neighbours, whose diagonal reads them) get D, iD and row constants
 * recomputed -- every other row holds the same values already; levels >= 1 are rebuilt in full.  Same results as
 * wl_mg_update (to which it falls back when the flow's last change was not a native measure!, or in 2-D).
 * Periodic y / z: the last interior row also follows the first one (its upper ghost row is that row's periodic copy); a
 * z-periodic ring of slabs takes the full update.  Consumes the flow's changed-row record: flags of several
 * wl_measure_fill calls accumulate until this call has used them -- so ONE hierarchy per flow may be updated this way (the
 * reference builds exactly one, src/WaterLily.jl:77); a second hierarchy on the same mu0 must take wl_mg_update. */
int wl_mg_update_changed(wl_mg *m, wl_flow *a);
/* measure!(flow, body; t, eps)        src/Body.jl:31-53 for a PARAMETRIC body: an sdf family with closed-form gradient
 * (what ForwardDiff.gradient returns, src/AutoBody.jl:119) composed with an affine map xi = A x + b evaluated by the
 * host at the measured time together with its time derivative and inverse (AutoBody.jl:128-130: V = -J \ d(map)/dt).
 * Bodies defined by arbitrary closures stay on the host side of the ABI (waterlily_amd.body: torch).
 *   family WL_BODY_SPHERE: p = {c0, c1, c2, radius}  sdf = sqrt(sum(abs2, xi - c)) - radius   (circle when D == 2)
 *   family WL_BODY_TORUS : p = {c0, c1, c2, R, r}    sdf = norm((xi0-c0, norm((xi1-c1, xi2-c2)) - R)) - r   (D == 3)
 *   family WL_BODY_PLATE : p = {a, thk}              sdf = norm(xi - (clamp(xi0,-a,a), 0[, 0])) - thk   (the reference's
 *                                                    test plate, test/maintests.jl:375: a stadium / capsule about the xi0 axis)
 * Matrices are row-major 3x3 (the upper-left 2x2 block when D == 2). */
enum { WL_BODY_SPHERE = 0, WL_BODY_TORUS = 1, WL_BODY_PLATE = 2, WL_BODY_CYLINDER = 3 };
/*   family WL_BODY_CYLINDER: p = {c0, c1, c2, radius, m0, m1, m2}  sdf = sqrt(sum_a m_a != 0 (xi_a - c_a)^2) - radius: a circle
 *                                                    extruded along the axes whose m_a is 0 (the 3-D cylinder of the reference's
 *                                                    examples/ThreeD_cylinder*.jl)
 * A body may be a COMPOSITE: an array of up to WL_BODY_MAXLEAF descriptors combined left to right like the reference's
 * `Bodies(bodies, ops)` (src/AutoBody.jl:40-110; the AutoBody operators +, ∪, ∩, - of :22-34 build the same thing):
 * element 0 carries `count`, element l > 0 its operation `op` against the composite of the elements before it.  The
 * distance is the min / max of the leaves' distances; normal, map and velocity are those of the ACTIVE leaf (:73-93). */
enum { WL_BODY_OP_UNION = 0, WL_BODY_OP_MINUS = 1, WL_BODY_OP_INTERSECT = 2 };
#define WL_BODY_MAXLEAF 6
typedef struct wl_body_desc {
    int32_t family;
    int32_t identity_map;   /* != 0: xi = x (A, b, dA, db, Ainv are ignored; V = 0) */
    double p[8];
    double A[9], b[3], dA[9], db[3], Ainv[9];
    int32_t op;             /* elements 1.. of a composite: WL_BODY_OP_* */
    int32_t count;          /* element 0: number of descriptors in the array (0 is read as 1) */
} wl_body_desc;
/* Part 1: sigma = sdf at every interior cell centre (Body.jl:34) and the number of band cells d^2 < (2+eps)^2 (:35)
 * this rank will report (synchronises).  Part 2 (same body / eps): mu0, mu1, V (:36-48), then BC!(mu0,0) and
 * BC!(V,0,exitBC) (:51-52), the z-slab halo exchange, and the body-free row flags of wl_flow_update; cand_dev (device,
 * nband entries) receives the band cells as LOCAL dense column-major indices i + n0*(j + n1*k), ascending.
 * Only x-rows that hold a band / inside cell now, or held one at the previous wl_measure_fill, are rewritten. */
int wl_measure_rows(wl_flow *a, const wl_body_desc *body, double eps, int64_t *nband);
int wl_measure_fill(wl_flow *a, const wl_body_desc *body, double eps, int64_t *cand_dev);
/* nds(body, loc(0,I), t) = n * kern(clamp(d,-1,1))  (src/Metrics.jl:84-87, Float64) for n listed cells (indices as
 * written by wl_measure_fill): nds_dev[b*D + c] */
int wl_body_nds(const wl_grid *g, const wl_body_desc *body, const int64_t *cand_dev, int64_t n, double *nds_dev);
/* project!(a,b,w)                     src/Flow.jl:137-145 */
int wl_project(wl_flow *a, wl_mg *b, double dt, double w, int *n_iter);
/* mom_step!(a,b)                      src/Flow.jl:153-169.  dt = a.dt[end]; U = BCTuple(a.U,a.dt,N);
 * acc_pred/acc_corr = g(i,t)+dU_i/dt at t=sum(dt[1:end-1]) and t=sum(dt) (NULL when accelerate! is a
 * no-op, :73).  dt_next receives CFL(a); n_iter[2] the two entries pushed onto pois.n.
 * The u0 ARRAY is scratch across the call, as in the reference (it is overwritten by `a.u⁰ .= a.u` before anything reads
 * it, :154): on return it holds either the velocity the step started from (the reference's copy) or -- 3-D, no periodic
 * direction, no convective exit, wl_set_option(27) -- the predictor's velocity u', because there the two velocity arrays
 * take turns instead of being copied (the predictor reads u and writes u' into u0, the corrector writes the new velocity
 * back into u).  u, p, f, sigma, the time step and the V-cycle counts are the same bits either way. */
int wl_mom_step(wl_flow *a, wl_mg *b, double dt, const double U[3], const double *acc_pred,
                const double *acc_corr, double *dt_next, int n_iter[2]);

/* ------------------------------------------------------------------ Metrics.jl */
/* pressure_force(p,df,body,t)         src/Metrics.jl:94-100.  nds(body,x,t) (:84-87) runs user closures,
 * so the host evaluates it once per measure! and hands over the compact band of non-zero entries:
 * idx[b] = linear element offset of the cell in p (using g->s), nds[b*D + c] = n_c * kern(d), Float64.
 * out[c] = sum_b Float64( T( p[idx[b]] * nds[b,c] ) ). */
int wl_pforce(wl_dtype t, const wl_grid *g, const void *p, const int64_t *idx_dev, const double *nds_dev,
              int64_t nband, double out[3]);

/* viscous_force(u,nu,df,body,t)       src/Metrics.jl:109-113: out = sum_band Float64( T( -nu * (du_i/dx_j + du_j/dx_i) * nds ) ),
 * same band hand-over as wl_pforce (idx are element offsets into one component of u). */
int wl_vforce(wl_dtype t, const wl_grid *g, const void *u, const int64_t *idx_dev, const double *nds_dev, int64_t nband,
              double nu, double out[3]);
/* pressure_moment(x0,p,df,body,t)    src/Metrics.jl:130-134: out = sum_band Float64( T( p * cross(loc(0,I)-x0, nds) ) );
 * D == 2: the scalar cross product is returned in every component, like the reference's broadcast. */
int wl_pmoment(wl_dtype t, const wl_grid *g, const void *p, const int64_t *idx_dev, const double *nds_dev, int64_t nband,
               const double x0[3], double out[3]);

/* Field metrics over inside(out) (src/Metrics.jl:14-77), `@inside out[I] = metric(I,u)`:
 *   WL_M_KE      ke(I,u,U)            0.125*sum_i (u[I,i]+u[I+d_i,i]-2U_i)^2          (par = U)
 *   WL_M_CURL    curl(i,I,u)          component i=ipar of curl u at the cell EDGE      (D==2: i=3 -> ipar=2)
 *   WL_M_OMAG    omega_mag(I,u)       |curl u| at the cell centre                      (D==3)
 *   WL_M_OTHETA  omega_theta(I,z,c,u) omega . theta, theta = z x (loc(0,I)-c)          (par = z[3], par2 = c[3]; D==3)
 *   WL_M_LAMBDA2 lambda2(I,u)         middle eigenvalue of S^2+Omega^2                 (D==3) */
enum { WL_M_KE = 0, WL_M_CURL = 1, WL_M_OMAG = 2, WL_M_OTHETA = 3, WL_M_LAMBDA2 = 4 };
int wl_metric(wl_dtype t, const wl_grid *g, int kind, void *out, const void *u, int ipar, const double par[3],
              const double par2[3]);

/* ------------------------------------------------------------------ snapshots (VTK write / restart, ext/WaterLilyWriteVTKExt.jl:57-66,
 * ext/WaterLilyReadVTKExt.jl:28-45).  The reference copies whole fields to the host (`a.flow.u |> Array`) and permutes the vector
 * components to the front there (components_first, :79).  Here the field's LOCAL planes klo..khi are packed on the device into
 * a dense array-of-tuples staging buffer -- dst[((kk*n1 + j)*n0 + i)*ntuple + c] = a_c[i, j, klo+kk] for c < ncomp, 0 for the
 * padding components ncomp <= c < ntuple (VTK vectors carry 3) -- which the host then moves with ONE asynchronous copy on a
 * side stream while the next time step runs; unpack is the inverse (restart).  Enqueued on the library's stream. */
int wl_snapshot_pack(wl_dtype t, const wl_grid *g, const void *a, int ncomp, int ntuple, int klo, int khi, void *dst);
int wl_snapshot_unpack(wl_dtype t, const wl_grid *g, void *a, int ncomp, int ntuple, int klo, int khi, const void *src);

/* ------------------------------------------------------------------ switches
 * Every key selects between the form of an operator the reference writes and a traffic-saving form of it that produces the
 * same bits (tests flip them one by one, and all at once); a few are tuning values.  The defaults are the measured winners;
 * round 4 retired the keys whose alternative was a recorded loss (11, 12, 20, 21, 24, 25, 28: DESIGN.md, measured dead ends).
 * key 0: 1 = use the 16-B-vectorised z-marching 7-point kernel where it applies (default), 0 = generic range kernel
 * key 1: 1 = fused V-cycle smoothers (default), 0 = the reference's two-pass Jacobi!/increment!/prolongate!
 * key 2: 1 = LDS-tiled marching conv_diff kernel (default), 0 = generic gather kernel
 * key 3: 1 = BDIM! uses the body-free row flags (default), 0 = general path everywhere
 * key 4: rows per thread of the vectorised 7-point kernel (a 256-thread workgroup covers 4x that many rows): 1, 2, or
 *        0 (default) = 2 on levels of >= 2^26 interior cells with an even y extent, else 1.  Same values either way.
 * key 5: != 0 = 16-B vectorised streaming pcg kernels (default), 0 = scalar range kernels
 * key 6: 1 = multigrid levels <= 4096 cells run as one single-workgroup launch per V-cycle (default), 0 = per-op launches
 * key 7: 1 = BC! as one closed-form launch (default), 0 = the reference's sequence of plane loops
 * key 8: 1 = pcg! applies x += alpha*eps in the direction kernel instead of the update kernel (default; one array
 *        pass less per iteration, identical values), 0 = in the update kernel as the reference orders it
 * key 9: 1 = the 7-point kernels skip the loads of L in rows whose face coefficients are all one number (rows clear
 *        of the body and the domain faces; constants recorded by wl_mg_update) (default), 0 = always load L
 * key 10: 1 = inside solver! the start of pcg! (eps = r*iD, rho) is evaluated by the prolongate!+increment! kernel that
 *         has just produced r (default), 0 = by pcg!'s own first kernel
 * key 13: 1 = pcg! does not store z' = r*iD, the direction kernel recomputes it (default), 0 = stored as in the reference
 * key 14: 1 = inside mom_step! the predictor's closing `x ./= dt` and the corrector's opening `x .*= 0.5dt`
 *         (Flow.jl:144,139) are one pass over x, each rounding kept (default), 0 = two passes
 * key 15: 1 = on levels of at most 2^25 cells pcg!'s dot products are finished by the kernel that follows (no
 *         one-workgroup finalize launches inside a pcg! call; single rank) (default), 2 = on every level, 0 = separate
 *         finalize launch after every dot product
 * keys 16, 17: grid size of the 7-point / streaming vector kernels in units of 1024 workgroups (defaults 4 / 16: measured
 *         at 512^3, the streaming kernels gain 3-6 % from shorter z-chunks, the 7-point kernels do not)
 * key 18: 1 = conv_diff! evaluates each interior face flux once and shares it between the two cells (shared-flux LDS kernel
 *         on the tiles / planes whose y and z faces are all interior; 64x8 tiles in Float32, 64x4 in Float64) (default),
 *         0 = every cell gathers its six fluxes
 * key 19: 1 = on levels of 2^22 .. 2^26 cells pcg! does not store z = A*eps: its update kernel is a second 7-point kernel over
 *         eps that forms the same A*eps again and applies r -= alpha*(A*eps) (default; 3-D vector kernels), 3 = on every
 *         level below 2^26 cells, 2 = on every level, 0 = the mult kernel always stores z
 * key 22: 1 = inside wl_mom_step / wl_project (3-D, one device) z = div(u) is formed by the residual! kernel itself, the
 *         z array is neither written nor read (default), 0 = separate div pass
 * key 23: 1 = inside wl_mom_step (3-D, x not periodic) the x-ghost cells of the interior rows that BC!(u,U) sets are
 *         written by the kernel that has just produced the row (BDIM!, the velocity correction); the BC launch that
 *         follows covers the y and z planes only (default), 0 = BC! writes all six planes
 * key 26: bound of a mailbox all-reduce's wait for a peer in SECONDS of the device's wall clock (default 600; 0 = unbounded,
 *         like a collective)
 * key 27: 1 = inside wl_mom_step (3-D, no periodic direction, no convective exit) the conv_diff! kernels finish BDIM!
 *         (Flow.jl:134, scale_u! :166) on the body-free x-rows themselves: the row's new velocity is stored from the registers
 *         that hold f, V is not read there, and no separate pass over those rows runs (6T + 9T per cell and step less);
 *         the u0 array holds u' on return (see wl_mom_step) (default), 0 = separate BDIM! pass, u0 = the copy of u
 * key 30: 1 = consecutive marching kernels sweep their tiles in opposite directions, each XCD starting on the lines the
 *         kernel before it touched last (L2 / Infinity Cache) (default), 0 = always ascending.  Same bits either way.
 * key 31: 1 = inside the one-workgroup bottom of the V-cycle (levels of <= 4096 cells) pcg! keeps its level in registers and
 *         LDS for the whole call (default), 0 = every phase goes through global memory.  Same bits either way.
 * Any other key: WL_E_ARG. */
int wl_set_option(int key, int value);
int wl_get_option(int key, int *value);

/* ------------------------------------------------------------------ measurement support */
/* Kernel classes for launch counting and HIP-event timing (bench.py roofline leg). */
enum {
    WL_K_CONVDIFF = 0, WL_K_BDIM = 1, WL_K_BC = 2, WL_K_DIV = 3, WL_K_CORRECT = 4, WL_K_CFL = 5,
    WL_K_SCALE = 6, WL_K_RESIDUAL = 7, WL_K_JACOBI = 8, WL_K_INCREMENT = 9, WL_K_SMOOTH = 10,
    WL_K_RESTRICT = 11, WL_K_PROLONG = 12, WL_K_PCG_INIT = 13, WL_K_PCG_MULT = 14, WL_K_PCG_UPDATE = 15,
    WL_K_PCG_DIR = 16, WL_K_DOT = 17, WL_K_SCALAR = 18, WL_K_SETDIAG = 19, WL_K_RESTRICTL = 20,
    WL_K_COPY = 21, WL_K_PFORCE = 22, WL_K_MISC = 23, WL_K_COUNT = 24
};
const char *wl_kernel_name(int kclass);
/* time launches of `kclass` (-1 = none) issued for grids with >= min_cells cells, with hipEvents */
int wl_prof_select(int kclass, int64_t min_cells);
int wl_prof_reset(void);
/* launches / cells processed per class since the last reset (all classes, all levels) */
int wl_prof_counts(int kclass, int64_t *launches, int64_t *cells);
/* device and pinned-host allocations the library itself has made since it was loaded (count, bytes): a steady wl_mom_step makes
 * none (the reference bounds mom_step!'s allocations the same way, test/alloctest.jl:17-27) */
int wl_prof_allocs(int64_t *count, int64_t *bytes);
/* number of stencil launches since start-up that were split to overlap a z-slab halo exchange (comm stream) */
int wl_prof_overlapped(int64_t *count);
/* collectives issued by this rank since the last wl_prof_reset (z-slab runs; all zero without a communicator):
 * out[0] all-reduces, out[1] halo exchanges (one grouped send/recv batch each), out[2] send/recv pairs inside them (one per
 * component and neighbour side), out[3] all-gathers, out[4] bytes this rank sent in halo exchanges, out[5] bytes it
 * contributed to all-gathers */
int wl_prof_comm(int64_t out[6]);
int wl_prof_reset_comm(void);
/* `reps` back-to-back all-reduces of one double, issued the way the solver issues them (mailbox or communicator): microseconds
 * per all-reduce, timed on the device.  Every rank calls it with the same reps; 0 without a communicator. */
int wl_prof_allreduce_us(int reps, double *us_per_op);           /* zero these six counters only (wl_prof_reset zeroes them too) */
/* for the selected class: timed launches, their summed cells, summed milliseconds (synchronises) */
int wl_prof_timed(int64_t *launches, int64_t *cells, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* WLHIP_H */

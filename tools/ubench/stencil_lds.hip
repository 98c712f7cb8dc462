// stencil_lds.hip -- micro-benchmark (NOT part of the product): how fast can gfx950 sweep a 512^3 Float32 array with a
// 7-point access pattern when the operand planes are staged in an LDS ring filled by global_load_lds (no destination
// registers for the loads in flight), against (a) a flat 16-B read stream and (b) a z-marching register-window read?
//   build: hipcc -O3 --offload-arch=gfx950 -o stencil_lds stencil_lds.hip      run: ./stencil_lds [n=512] [reps=20]
// Every kernel reduces to one checksum per workgroup so that no load can be optimised away; checksums of the three
// stencil forms must agree exactly (same per-thread summation order is NOT required: printed as double sums to 1e-6).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline void block_out(double acc, double *out) {
    __shared__ double sm[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    acc = wave_sum(acc);
    if (lane == 0) sm[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sm[i]; out[blockIdx.x] = s; }
}

// (a) flat read stream, UNROLL independent 16-B loads in flight per thread
template <int UNROLL> __global__ __launch_bounds__(256) void k_flat(const f4 *a, long nv, double *out) {
    double acc = 0;
    const long t0 = (long)blockIdx.x * 256 + threadIdx.x, nt = (long)gridDim.x * 256;
    for (long q = t0; q < nv; q += nt * UNROLL) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (q + u * nt < nv) ? a[q + u * nt] : f4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (double)v[u].x + (double)v[u].y + (double)v[u].z + (double)v[u].w;
    }
    block_out(acc, out);
}

// geometry shared by the stencil kernels: interior n^3, pitch sy (floats), plane sz; tiles of 256 x TY cells marching in z
struct Geo { int n; long sy, sz; int ntx, nty, clen; };

// (b) register-window 7-point sweep: what the product's kernel does (1 plane ahead, R = 2 rows per thread, halo rows and
// edge cells loaded from global), coefficient 1 everywhere: Ae = -6 e + sum of 6 neighbours; acc += Ae * e
template <bool STORE, bool RC, int ARITH = 0, bool XCD = false>
__global__ __launch_bounds__(256) void k_regwin(const float *e, Geo g, double *out, float *z, const float *rowc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tpp = g.ntx * g.nty;
    const int nb_ = gridDim.x, b_ = blockIdx.x;
    const int lb_ = XCD ? ((b_ & 7) * (nb_ >> 3) + (b_ >> 3)) : b_;   // XCD: every XCD gets a CONTIGUOUS range of tiles (the product's mapping)
    const int ch = lb_ / tpp, pt = lb_ - ch * tpp;
    const int i = 1 + (pt % g.ntx) * 256 + lane * 4, jb = 1 + (pt / g.ntx) * 8 + wv * 2;
    const int k0 = 1 + ch * g.clen, k1 = min(g.n + 1, k0 + g.clen);
    double acc = 0;
    const long col = (long)i + g.sy * jb;
    auto LD = [&](long o) { return *reinterpret_cast<const f4 *>(e + o); };
    f4 em[2], ec[2];
    for (int q = 0; q < 2; ++q) { em[q] = LD(col + q * g.sy + g.sz * (k0 - 1)); ec[q] = LD(col + q * g.sy + g.sz * k0); }
    f4 own[2] = {LD(col + g.sz * (k0 + 1)), LD(col + g.sy + g.sz * (k0 + 1))};
    f4 hlo = LD(col - g.sy + g.sz * k0), hhi = LD(col + 2 * g.sy + g.sz * k0);
    float lf[2] = {0, 0}, rg[2] = {0, 0};
    for (int q = 0; q < 2; ++q) { if (lane == 0) lf[q] = e[col + q * g.sy + g.sz * k0 - 1]; if (lane == 63) rg[q] = e[col + q * g.sy + g.sz * k0 + 4]; }
    typedef float rc8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) rc8 *cptr;
    const int jbu = __builtin_amdgcn_readfirstlane(jb);
    rc8 rk[2] = {rc8{1, 0, 0, 0, 0, 0, 0, 0}, rc8{1, 0, 0, 0, 0, 0, 0, 0}};
    if (RC) for (int q = 0; q < 2; ++q) rk[q] = *(cptr)(rowc + 8 * ((long)(jbu + q) + (long)(g.n + 2) * k0));
    for (int k = k0; k < k1; ++k) {
        const int kn = min(k + 1, k1 - 1);
        rc8 rkn[2] = {rk[0], rk[1]};
        if (RC) for (int q = 0; q < 2; ++q) rkn[q] = *(cptr)(rowc + 8 * ((long)(jbu + q) + (long)(g.n + 2) * kn));
        f4 nown[2] = {LD(col + g.sz * (kn + 1)), LD(col + g.sy + g.sz * (kn + 1))};
        f4 nlo = LD(col - g.sy + g.sz * kn), nhi = LD(col + 2 * g.sy + g.sz * kn);
        float nlf[2] = {0, 0}, nrg[2] = {0, 0};
        for (int q = 0; q < 2; ++q) { if (lane == 0) nlf[q] = e[col + q * g.sy + g.sz * kn - 1]; if (lane == 63) nrg[q] = e[col + q * g.sy + g.sz * kn + 4]; }
        for (int q = 0; q < 2; ++q) {
            const f4 c = ec[q], ym = q == 0 ? hlo : ec[0], yp = q == 1 ? hhi : ec[1], zm = em[q], zp = own[q];
            float left = __shfl_up(c.w, 1, 64), right = __shfl_down(c.x, 1, 64);
            if (lane == 0) left = lf[q];
            if (lane == 63) right = rg[q];
            const float xm[4] = {left, c.x, c.y, c.z}, xp[4] = {c.y, c.z, c.w, right};
            const float cc[4] = {c.x, c.y, c.z, c.w}, a1[4] = {ym.x, ym.y, ym.z, ym.w}, a2[4] = {yp.x, yp.y, yp.z, yp.w};
            const float a3[4] = {zm.x, zm.y, zm.z, zm.w}, a4[4] = {zp.x, zp.y, zp.z, zp.w};
            float cf = 1.f;
            if (RC) {
                cf = rk[q][0];
                if (!(cf == cf)) cf = e[col + q * g.sy + g.sz * k];   // "not uniform": a load behind a rarely taken branch (never here)
            }
            f4 aev;
            float *ap = reinterpret_cast<float *>(&aev);
            for (int v = 0; v < 4; ++v) {
                float ae;
                if (ARITH == 1) {        // the product's sequence: diagonal from the six faces, seven products, separately rounded
                    const float c = cf;
                    float dg = 0;
                    dg -= (c + c); dg -= (c + c); dg -= (c + c);
                    float s_ = cc[v] * dg;
                    s_ += xm[v] * c + xp[v] * c;
                    s_ += a1[v] * c + a2[v] * c;
                    s_ += a3[v] * c + a4[v] * c;
                    ae = s_;
                } else if (ARITH == 2) { // unit coefficients: the same values with the products by 1 dropped
                    float s_ = cc[v] * -6.f;
                    s_ += xm[v] + xp[v];
                    s_ += a1[v] + a2[v];
                    s_ += a3[v] + a4[v];
                    ae = s_;
                } else ae = cf * (-6.f * cc[v] + xm[v] + xp[v] + a1[v] + a2[v] + a3[v] + a4[v]);
                ap[v] = ae;
                acc += (double)ae * (double)cc[v];
            }
            if (STORE) *reinterpret_cast<f4 *>(z + col + q * g.sy + g.sz * k) = aev;
        }
        for (int q = 0; q < 2; ++q) { em[q] = ec[q]; ec[q] = own[q]; own[q] = nown[q]; lf[q] = nlf[q]; rg[q] = nrg[q]; rk[q] = rkn[q]; }
        hlo = nlo; hhi = nhi;
    }
    block_out(acc, out);
}

// (c) the same sweep through an LDS ring of NS planes x (8+2) rows x 256 cells filled by global_load_lds_dwordx4: NS-2 planes
// in flight while one is worked on, no destination registers; y-halo rows are shared by the four wavefronts through LDS;
// the 20 x-edge cells of a plane arrive by ONE global_load_lds_dword (per-lane source addresses, contiguous LDS image)
template <int NS> __global__ __launch_bounds__(256) void k_ldsring(const float *e, Geo g, double *out) {
    constexpr int TY = 8, ROWS = TY + 2;
    // ONE __shared__ object for everything (cdna_hip_programming.md: a second object beside a glds staging array can make hipcc
    // drain the DMA queue -- s_waitcnt vmcnt(0) -- before every ds_read): [NS][ROWS][256] ring, [NS][64] edge cells, 16 doubles
    __shared__ __attribute__((aligned(16))) float smem[NS * ROWS * 256 + NS * 64 + 32];
    auto ring = [&](int s_, int r_) -> float * { return smem + ((long)s_ * ROWS + r_) * 256; };
    float *edge0 = smem + NS * ROWS * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tpp = g.ntx * g.nty;
    const int ch = blockIdx.x / tpp, pt = blockIdx.x - ch * tpp;
    const int i0 = 1 + (pt % g.ntx) * 256, j0 = 1 + (pt / g.ntx) * TY;
    const int k0 = 1 + ch * g.clen, k1 = min(g.n + 1, k0 + g.clen);
    const long base = (long)i0 + lane * 4 + g.sy * (j0 - 1);          // row 0 of the tile (the lower halo row), this lane's cells
    // issue the loads of plane p into its slot: rows wv, wv+4, wv+8 (< ROWS) by this wavefront; wavefront 3 also the edge cells
    auto issue = [&](int p) {
        const int s = p % NS;
        const int pc = min(max(p, 0), g.n + 1);
        for (int r = wv; r < ROWS; r += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(e + base + g.sy * r + g.sz * pc),
                                             (__attribute__((address_space(3))) void *)ring(s, r), 16, 0, 0);
        if (wv == 3) {   // lane l < 20: row l>>1, side l&1
            const int r = min(lane >> 1, ROWS - 1), side = lane & 1;
            const long o = (long)i0 + (side ? 256 : -1) + g.sy * (j0 - 1 + r) + g.sz * pc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(e + o),
                                             (__attribute__((address_space(3))) void *)(edge0 + s * 64), 4, 0, 0);
        }
    };
    // per-wavefront number of load instructions per plane: rows {3,3,2,2} + edge on wavefront 3 -> {3,3,2,3}
    double acc = 0;
    // prologue: planes k0-1 .. k0+NS-3 in flight, wait for k0-1 and k0 and k0+1
    for (int p = k0 - 1; p <= k0 + NS - 3; ++p) issue(p);
    // wait until at most (NS-3) planes' worth of this wavefront's loads are outstanding: planes k0-1, k0 landed
    auto wait_keep = [&](int planes) {   // planes in {0,1,2,3}
        const int per = (wv == 2) ? 2 : 3;
        const int n = per * planes;
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    wait_keep(NS - 3);           // outstanding may be: planes k0+1 .. k0+NS-3  (NS-3 planes)
    __builtin_amdgcn_s_barrier();
    auto RD = [&](int s_, int r_) { return *reinterpret_cast<const f4 *>(ring(s_, r_) + lane * 4); };
    const int r0 = 1 + 2 * wv;   // first own row of this wavefront in the tile
    f4 em[2] = {RD((k0 - 1) % NS, r0), RD((k0 - 1) % NS, r0 + 1)}, ec[2] = {RD(k0 % NS, r0), RD(k0 % NS, r0 + 1)};
    for (int k = k0; k < k1; ++k) {
        // slot (k-2)%NS is free since the barrier of iteration k-1: plane k+NS-2 goes there, BEFORE the wait (longer flight).
        // In flight now: planes k+1 .. k+NS-2; plane k+1 must have landed -> keep NS-3 planes' loads outstanding.
        issue(k + NS - 2);
        wait_keep(NS - 3);
        __builtin_amdgcn_s_barrier();            // every wavefront's rows of plane k+1 are in LDS
        const int sc = k % NS, sp = (k + 1) % NS;
        const f4 own0 = RD(sp, r0), own1 = RD(sp, r0 + 1), hlo = RD(sc, r0 - 1), hhi = RD(sc, r0 + 2);
        const float *ed = edge0 + sc * 64;
        const float el0 = ed[2 * r0], er0 = ed[2 * r0 + 1], el1 = ed[2 * r0 + 2], er1 = ed[2 * r0 + 3];
        const f4 own[2] = {own0, own1};
        const float lf[2] = {el0, el1}, rg[2] = {er0, er1};
        for (int q = 0; q < 2; ++q) {
            const f4 c = ec[q], ym = q == 0 ? hlo : ec[0], yp = q == 1 ? hhi : ec[1], zm = em[q], zp = own[q];
            float left = __shfl_up(c.w, 1, 64), right = __shfl_down(c.x, 1, 64);
            if (lane == 0) left = lf[q];
            if (lane == 63) right = rg[q];
            const float xm[4] = {left, c.x, c.y, c.z}, xp[4] = {c.y, c.z, c.w, right};
            const float cc[4] = {c.x, c.y, c.z, c.w}, a1[4] = {ym.x, ym.y, ym.z, ym.w}, a2[4] = {yp.x, yp.y, yp.z, yp.w};
            const float a3[4] = {zm.x, zm.y, zm.z, zm.w}, a4[4] = {zp.x, zp.y, zp.z, zp.w};
            for (int v = 0; v < 4; ++v) { const float ae = -6.f * cc[v] + xm[v] + xp[v] + a1[v] + a2[v] + a3[v] + a4[v]; acc += (double)ae * (double)cc[v]; }
        }
        for (int q = 0; q < 2; ++q) { em[q] = ec[q]; ec[q] = own[q]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {   // block reduction in the same shared object
        double *sm = reinterpret_cast<double *>(smem + NS * ROWS * 256 + NS * 64);
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) sm[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0) out[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 20;
    const long sy = ((n + 2 + 31) / 32) * 32, sz = sy * (n + 2);
    const long lead = 31, total = lead + sz * (n + 2) + 64;
    float *buf;
    CK(hipMalloc(&buf, total * sizeof(float)));
    std::vector<float> h(total);
    unsigned s = 12345;
    for (long q = 0; q < total; ++q) { s = s * 1664525u + 1013904223u; h[q] = (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    CK(hipMemcpy(buf, h.data(), total * sizeof(float), hipMemcpyHostToDevice));
    float *e = buf + lead;   // element [1] of every row is 128-B aligned (like the product's padded layout)
    Geo g{n, sy, sz, n / 256, n / 8, 16};
    const int nchunk = (n + g.clen - 1) / g.clen, nblk = g.ntx * g.nty * nchunk;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 65536));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // FLUSH=1 in the environment: a 2 GB fill of another buffer runs before every timed launch (what a kernel meets inside a
    // time step: nothing of its operands left in the 256 MB Infinity Cache by the launch before it); each launch timed alone
    const bool flush = getenv("FLUSH") && getenv("FLUSH")[0] == '1';
    float *scratch = nullptr;
    const size_t nscr = 512ull << 20;   // floats = 2 GB
    if (flush) CK(hipMalloc(&scratch, nscr * sizeof(float)));
    auto run = [&](const char *name, auto launch, double bytes) {
        launch();
        CK(hipDeviceSynchronize());
        float ms = 0;
        if (flush) {
            for (int r = 0; r < reps; ++r) {
                CK(hipMemsetAsync(scratch, r & 0xff, nscr * sizeof(float), 0));
                CK(hipEventRecord(a));
                launch();
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float t;
                CK(hipEventElapsedTime(&t, a, b));
                ms += t;
            }
        } else {
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
        }
        std::vector<double> ho(65536);
        CK(hipMemcpy(ho.data(), out, sizeof(double) * 65536, hipMemcpyDeviceToHost));
        double sum = 0;
        for (double v : ho) sum += v;
        printf("%-34s %8.1f us  %6.2f TB/s of %.3f GB   checksum %.6e\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, bytes / 1e9, sum);
    };
    const double arr = (double)n * n * n * 4;
    const long nv = (sz * (n + 2)) / 4;
    CK(hipMemset(out, 0, sizeof(double) * 65536));
    run("flat read, 1 load in flight", [&] { hipLaunchKernelGGL(k_flat<1>, dim3(8192), dim3(256), 0, 0, (const f4 *)(e + 1), nv - 8, out); }, (double)nv * 16);
    run("flat read, 2 loads in flight", [&] { hipLaunchKernelGGL(k_flat<2>, dim3(8192), dim3(256), 0, 0, (const f4 *)(e + 1), nv - 8, out); }, (double)nv * 16);
    run("flat read, 4 loads in flight", [&] { hipLaunchKernelGGL(k_flat<4>, dim3(8192), dim3(256), 0, 0, (const f4 *)(e + 1), nv - 8, out); }, (double)nv * 16);
    CK(hipMemset(out, 0, sizeof(double) * 65536));
    float *zb, *rowc;
    CK(hipMalloc(&zb, total * sizeof(float)));
    CK(hipMalloc(&rowc, sizeof(float) * 8 * (n + 2) * (n + 2)));
    { std::vector<float> hr((size_t)8 * (n + 2) * (n + 2), 1.0f); CK(hipMemcpy(rowc, hr.data(), hr.size() * 4, hipMemcpyHostToDevice)); }
    float *z = zb + lead;
    run("7-point, register window (R=2)", [&] { hipLaunchKernelGGL((k_regwin<false, false>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, arr);
    run("  + row constants (s_load x8)", [&] { hipLaunchKernelGGL((k_regwin<false, true>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, arr);
    run("  + store z", [&] { hipLaunchKernelGGL((k_regwin<true, false>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, 2 * arr);
    run("  + store z, XCD-contiguous tile map", [&] { hipLaunchKernelGGL((k_regwin<true, false, 0, true>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, 2 * arr);
    run("  no store,  XCD-contiguous tile map", [&] { hipLaunchKernelGGL((k_regwin<false, false, 0, true>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, arr);
    run("  + store z, product arithmetic", [&] { hipLaunchKernelGGL((k_regwin<true, false, 1>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, 2 * arr);
    run("  + store z, unit-coefficient form", [&] { hipLaunchKernelGGL((k_regwin<true, false, 2>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, 2 * arr);
    run("  + store z + row constants", [&] { hipLaunchKernelGGL((k_regwin<true, true>), dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out, z, (const float *)rowc); }, 2 * arr);
    for (int cl : {8, 32, 64}) {
        Geo g2 = g; g2.clen = cl;
        const int nb2 = g.ntx * g.nty * ((n + cl - 1) / cl);
        char nm[64]; snprintf(nm, sizeof nm, "  + store z, chunks of %d planes", cl);
        run(nm, [&] { hipLaunchKernelGGL((k_regwin<true, false>), dim3(nb2), dim3(256), 0, 0, (const float *)e, g2, out, z, (const float *)rowc); }, 2 * arr);
    }
    CK(hipMemset(out, 0, sizeof(double) * 65536));
    run("7-point, LDS ring NS=4 (glds)", [&] { hipLaunchKernelGGL(k_ldsring<4>, dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out); }, arr);
    CK(hipMemset(out, 0, sizeof(double) * 65536));
    run("7-point, LDS ring NS=5 (glds)", [&] { hipLaunchKernelGGL(k_ldsring<5>, dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out); }, arr);
    CK(hipMemset(out, 0, sizeof(double) * 65536));
    run("7-point, LDS ring NS=6 (glds)", [&] { hipLaunchKernelGGL(k_ldsring<6>, dim3(nblk), dim3(256), 0, 0, (const float *)e, g, out); }, arr);
    return 0;
}

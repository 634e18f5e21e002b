// seq_kernels.hip -- gr and Jacobi/barycentric modify_orbits_forces for arbitrary mass layouts in
// the reference's own ORDER of operations (reference-order arithmetic, REBXAMD_MODE_EXACT, at most
// REBXAMD_SEQ_MAX_BODIES bodies).
//
// The reference accumulates these forces body by body:
//   * gr (src/gr.c:80-167): Newtonian pair sums in ascending partner index, REBOUND's
//     inertial -> Jacobi recurrence (running mass eta and running sums m q, forwards), the
//     post-Newtonian term per Jacobi body, and the Jacobi -> inertial recurrence backwards;
//   * modify_orbits_forces through rebx_com_force (src/rebxtools.c:45-126): the centre of mass
//     folded forwards over all bodies, then down-dated body by body backwards (Jacobi), every body
//     j <= i losing massratio_i a_i in the order i = N-1 .. j.
// A parallel prefix sum adds the same numbers in another order and lands a few ulp away (the passes
// of scan_kernels.hip).  Here the recurrences that are chains are walked as chains -- by a few lanes
// of ONE workgroup, one lane per component, a tile of 256 bodies staged in LDS at a time -- and
// everything that does not depend on the order (the per-body terms, the Newtonian sums, the
// application of the reactions, whose order is fixed per body) runs in parallel around them.  The
// results carry the bits of the sequential evaluation.  Cost: about 50-150 clock cycles per body in
// the chains (a division sits inside the centre-of-mass chain), i.e. milliseconds at 65 536 bodies.
// Massless bodies behind the last massive one (the caller's promise view->tail_massless) need no chain:
// they add zeros to every running sum, so each of them sees the sums as the last massive body left them
// and hands nothing back; the chains then cover the bodies up to the last massive one only, however
// many massless bodies follow (they are evaluated in parallel against those final sums).
#include "internal.h"
#include "effects.h"

namespace rebxamd {

static constexpr int QT = 256;          // bodies per tile; threads of every workgroup here

// ---- massive bodies in index order: x, y, z, m, index  (one workgroup walks the tiles)
__global__ __launch_bounds__(QT) void seq_compact(View v, double* cx, double* cy, double* cz, double* cm, long long* cidx, long long* count) {
    __shared__ int wcount[QT / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    long long base = 0;
    for (long long t0 = 0; t0 < v.n; t0 += QT) {
        const long long i = t0 + tid;
        const bool massive = i < v.n && v.m[i] != 0.;
        const unsigned long long mask = __ballot(massive);
        if (lane == 0) wcount[wave] = __popcll(mask);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < QT / 64; w++) { all += wcount[w]; if (w < wave) before += wcount[w]; }
        if (massive) {
            const long long pos = base + before + __popcll(mask & ((1ull << lane) - 1ull));
            cx[pos] = v.x[i]; cy[pos] = v.y[i]; cz[pos] = v.z[i]; cm[pos] = v.m[i]; cidx[pos] = i;
        }
        base += all;
        __syncthreads();
    }
    if (tid == 0) count[0] = base;
}

// ---- Newtonian acceleration of every body from the massive ones, partners in ascending index and
// every pair in the orientation the reference's pair loop gives it (src/gr.c:86-103)
__global__ __launch_bounds__(QT) void seq_newton(View v, double G, const double* cx, const double* cy, const double* cz, const double* cm,
                                                 const long long* cidx, const long long* count, double* nax, double* nay, double* naz) {
    __shared__ double tx[QT], ty[QT], tz[QT], tm[QT];
    __shared__ long long ti[QT];
    const long long i = (long long)blockIdx.x * QT + threadIdx.x;
    const bool live = i < v.n;
    const long long M = count[0];
    double xi = 0., yi = 0., zi = 0.;
    if (live) { xi = v.x[i]; yi = v.y[i]; zi = v.z[i]; }
    double ax = 0., ay = 0., az = 0.;
    for (long long j0 = 0; j0 < M; j0 += QT) {
        const long long j = j0 + threadIdx.x;
        __syncthreads();
        if (j < M) { tx[threadIdx.x] = cx[j]; ty[threadIdx.x] = cy[j]; tz[threadIdx.x] = cz[j]; tm[threadIdx.x] = cm[j]; ti[threadIdx.x] = cidx[j]; }
        __syncthreads();
        const int cnt = (int)((M - j0) < QT ? (M - j0) : QT);
        for (int k = 0; k < cnt; k++) {
            if (ti[k] == i) continue;
            newton_pair_exact(xi, yi, zi, tx[k], ty[k], tz[k], tm[k], ti[k] < i, G, ax, ay, az);
        }
    }
    if (live) { nax[i] = ax; nay[i] = ay; naz[i] = az; }
}

// ---- gr, forwards: the running mass and the nine running sums m q (q = position, velocity, Newtonian
// acceleration), one lane per sum; then, for the bodies of the tile in parallel, Jacobi state and
// post-Newtonian term.  aj <- the term (zero for body 0); mtot[0] <- the total mass as the chain leaves it.
__global__ __launch_bounds__(QT) void seq_gr_forward(View v, DevOp op, double G, const double* nax, const double* nay, const double* naz,
                                                     double* ajx, double* ajy, double* ajz, double* mtot, unsigned* status, double* sums10) {
#pragma clang fp contract(off)
    __shared__ double q[10][QT];        // m, x, y, z, vx, vy, vz, newtonian ax, ay, az of the tile
    __shared__ double pre[10][QT];      // the sums over every body before each body of the tile
    const int tid = threadIdx.x;
    const double mu = G * v.m[0];
    double s = 0.;                      // lanes 0..9: the running sum of their component
    for (long long t0 = 0; t0 < v.n; t0 += QT) {
        const long long i = t0 + tid;
        const bool live = i < v.n;
        q[0][tid] = live ? v.m[i] : 0.;
        q[1][tid] = live ? v.x[i] : 0.;  q[2][tid] = live ? v.y[i] : 0.;  q[3][tid] = live ? v.z[i] : 0.;
        q[4][tid] = live ? v.vx[i] : 0.; q[5][tid] = live ? v.vy[i] : 0.; q[6][tid] = live ? v.vz[i] : 0.;
        q[7][tid] = live ? nax[i] : 0.;  q[8][tid] = live ? nay[i] : 0.;  q[9][tid] = live ? naz[i] : 0.;
        __syncthreads();
        if (tid < 10) {
            const int cnt = (int)((v.n - t0) < QT ? (v.n - t0) : QT);
#pragma unroll 8
            for (int k = 0; k < cnt; k++) {
                const double mk = q[0][k];
                const double term = tid == 0 ? mk : mk * q[tid][k];
                pre[tid][k] = s;
                s = (t0 + k == 0) ? term : s + term;       // (body 0 starts the sums: eta = m_0, s = eta q_0)
            }
        }
        __syncthreads();
        if (live) {
            V3 a = {0., 0., 0.};
            if (i >= 1) {
                const double ei = 1. / pre[0][tid];
                Jac j;
                j.x = q[1][tid] - pre[1][tid] * ei;  j.y = q[2][tid] - pre[2][tid] * ei;  j.z = q[3][tid] - pre[3][tid] * ei;
                j.vx = q[4][tid] - pre[4][tid] * ei; j.vy = q[5][tid] - pre[5][tid] * ei; j.vz = q[6][tid] - pre[6][tid] * ei;
                j.ax = q[7][tid] - pre[7][tid] * ei; j.ay = q[8][tid] - pre[8][tid] * ei; j.az = q[9][tid] - pre[9][tid] * ei;
                if (gr_term_exact(j, mu, op.c, op.iarg, a) == 10) atomicOr(status, WARN_GR_ITER);
            }
            ajx[i] = a.x; ajy[i] = a.y; ajz[i] = a.z;
        }
        __syncthreads();
    }
    if (tid == 0) mtot[0] = s;
    if (tid < 10) sums10[tid] = s;       // what every massless body behind these bodies is measured from
}

// ---- gr for the massless bodies [first, n) behind the last massive one: their Jacobi state is taken from the final running
// sums (they add zeros to them), their post-Newtonian term comes back unchanged (the reverse recurrence carries zero past them)
__global__ __launch_bounds__(QT) void seq_gr_tail(View v, DevOp op, double G, long long first, const double* nax, const double* nay, const double* naz,
                                                  const double* sums10, unsigned* status) {
#pragma clang fp contract(off)
    const long long i = first + (long long)blockIdx.x * QT + threadIdx.x;
    if (i >= v.n) return;
    const double ei = 1. / sums10[0];
    Jac j;
    j.x = v.x[i] - sums10[1] * ei;  j.y = v.y[i] - sums10[2] * ei;  j.z = v.z[i] - sums10[3] * ei;
    j.vx = v.vx[i] - sums10[4] * ei; j.vy = v.vy[i] - sums10[5] * ei; j.vz = v.vz[i] - sums10[6] * ei;
    j.ax = nax[i] - sums10[7] * ei; j.ay = nay[i] - sums10[8] * ei; j.az = naz[i] - sums10[9] * ei;
    V3 a;
    if (gr_term_exact(j, G * v.m[0], op.c, op.iarg, a) == 10) atomicOr(status, WARN_GR_ITER);
    // (the reference adds the recurrence's running value, a zero here, before it adds the term to the particle)
    v.ax[i] += a.x + 0.; v.ay[i] += a.y + 0.; v.az[i] += a.z + 0.;
}

// ---- gr, backwards: Jacobi -> inertial accelerations (REBOUND's reverse recurrence), added to the
// acceleration columns.  The masses run down their own chain (one lane); the reciprocals of a tile are
// formed in parallel; three lanes walk the three components.
__global__ __launch_bounds__(QT) void seq_gr_reverse(View v, const double* ajx, const double* ajy, const double* ajz, const double* mtot) {
#pragma clang fp contract(off)
    __shared__ double mk[QT], aj[3][QT], eb[QT], ea[QT], ei[QT], out[3][QT];
    const int tid = threadIdx.x;
    double eta = mtot[0];               // lane 0: the running mass
    double s = 0. * eta;                // lanes 0..2: Jacobi slot 0 carries zero acceleration
    for (long long t0 = ((v.n - 1) / QT) * QT; t0 >= 0; t0 -= QT) {
        const long long i = t0 + tid;
        const bool live = i < v.n;
        const int cnt = (int)((v.n - t0) < QT ? (v.n - t0) : QT);
        mk[tid] = live ? v.m[i] : 0.;
        aj[0][tid] = live ? ajx[i] : 0.; aj[1][tid] = live ? ajy[i] : 0.; aj[2][tid] = live ? ajz[i] : 0.;
        __syncthreads();
        if (tid == 0) {
            for (int k = cnt - 1; k >= 0; k--) {
                if (t0 + k == 0) { eb[0] = eta; break; }
                eb[k] = eta;
                eta -= mk[k];
                ea[k] = eta;
            }
        }
        __syncthreads();
        ei[tid] = 1. / eb[tid];
        __syncthreads();
        if (tid < 3) {
#pragma unroll 8
            for (int k = cnt - 1; k >= 0; k--) {
                if (t0 + k == 0) { out[tid][0] = s * ei[0]; break; }      // body 0: s / (what is left of the mass)
                const double a = aj[tid][k];
                s = (s - mk[k] * a) * ei[k];
                out[tid][k] = a + s;
                s *= ea[k];
            }
        }
        __syncthreads();
        if (live) { v.ax[i] += out[0][tid]; v.ay[i] += out[1][tid]; v.az[i] += out[2][tid]; }
        __syncthreads();
    }
}

// ---- modify_orbits_forces: the centre of mass folded over all bodies (reb_get_com), then -- Jacobi
// coordinates -- down-dated body by body from the last (reb_get_com_without_particle) with the force of
// every body of the tile evaluated against the centre of mass of the bodies before it.  Six lanes walk
// the six components; each carries the mass along.
// Jacobi: a <- force per body, r <- massratio * force, mr <- massratio (zeros for body 0).
// com7 <- the centre of mass of all bodies (x y z vx vy vz m): barycentric coordinates measure every body from it
// (seq_mof_terms_about), Jacobi coordinates the massless bodies behind `v`.
// ntail: massless bodies that follow the bodies of `v` in the caller's array.  The reference folds them in as well and takes them
// out again one by one; with zero mass each such step only re-rounds the centre of mass, x <- (x m) / m, a monotone map that
// settles after a step or two: it is applied until it does (at most as often as the reference applies it), com7 <- the result.
__global__ __launch_bounds__(QT) void seq_mof_com(View v, int jacobi, long long ntail, double* com7, double* ax_, double* ay_, double* az_,
                                                  double* rx, double* ry, double* rz, double* mr) {
#pragma clang fp contract(off)
    __shared__ double q[7][QT];         // x y z vx vy vz m of the tile
    __shared__ double cm[7][QT];        // centre of mass each body of the tile is measured from
    const int tid = threadIdx.x;
    double x = 0., mm = 0.;
    auto stage = [&](long long t0) {
        const long long i = t0 + tid;
        const bool live = i < v.n;
        q[0][tid] = live ? v.x[i] : 0.;  q[1][tid] = live ? v.y[i] : 0.;  q[2][tid] = live ? v.z[i] : 0.;
        q[3][tid] = live ? v.vx[i] : 0.; q[4][tid] = live ? v.vy[i] : 0.; q[5][tid] = live ? v.vz[i] : 0.;
        q[6][tid] = live ? v.m[i] : 0.;
    };
    for (long long t0 = 0; t0 < v.n; t0 += QT) {
        stage(t0);
        __syncthreads();
        if (tid < 6) {
            const int cnt = (int)((v.n - t0) < QT ? (v.n - t0) : QT);
            for (int k = 0; k < cnt; k++) {
                const double pm = q[6][k];
                x = x * mm + q[tid][k] * pm;
                mm += pm;
                if (mm > 0.) x /= mm;
            }
        }
        __syncthreads();
    }
    if (tid < 6 && mm > 0.) {
        const long long most = 2 * ntail < 128 ? 2 * ntail : 128;
        for (long long k = 0; k < most; k++) {
            const double xn = (x * mm + 0.) / mm;
            if (xn == x) break;
            x = xn;
        }
    }
    if (tid < 6) com7[tid] = x;
    if (tid == 0) com7[6] = mm;
    if (!jacobi) return;
    for (long long t0 = ((v.n - 1) / QT) * QT; t0 >= 0; t0 -= QT) {
        stage(t0);
        __syncthreads();
        if (tid < 6) {
            const int cnt = (int)((v.n - t0) < QT ? (v.n - t0) : QT);
            for (int k = cnt - 1; k >= 0 && t0 + k >= 1; k--) {
                const double pm = q[6][k];
                x = x * mm - q[tid][k] * pm;
                mm -= pm;
                if (mm > 0.) x /= mm;
                cm[tid][k] = x;
                if (tid == 0) cm[6][k] = mm;
            }
        }
        __syncthreads();
        const long long i = t0 + tid;
        if (i < v.n) {
            V3 a = {0., 0., 0.};
            double w = 0.;
            if (i >= 1) {
                Part p;
                p.x = q[0][tid]; p.y = q[1][tid]; p.z = q[2][tid]; p.vx = q[3][tid]; p.vy = q[4][tid]; p.vz = q[5][tid]; p.m = q[6][tid];
                p.beta = p.Rt = p.k1 = unset_value();
                p.ta = v.ta ? v.ta[i] : unset_value(); p.te = v.te ? v.te[i] : unset_value(); p.ti = v.ti ? v.ti[i] : unset_value();
                Src ref;
                ref.x = cm[0][tid]; ref.y = cm[1][tid]; ref.z = cm[2][tid]; ref.vx = cm[3][tid]; ref.vy = cm[4][tid]; ref.vz = cm[5][tid]; ref.m = cm[6][tid];
                mof_exact(p, ref, a);
                w = p.m / (ref.m + p.m);
            }
            ax_[i] = a.x; ay_[i] = a.y; az_[i] = a.z;
            rx[i] = w * a.x; ry[i] = w * a.y; rz[i] = w * a.z;
            mr[i] = w;
        }
        __syncthreads();
    }
}

// bodies [first, n) against the one centre of mass: every body in barycentric coordinates; in Jacobi coordinates the massless
// bodies behind the last massive one (their mass ratio, and with it their reaction, is zero either way)
__global__ __launch_bounds__(QT) void seq_mof_terms_about(View v, long long first, const double* com7, double* ax_, double* ay_, double* az_,
                                                          double* rx, double* ry, double* rz, double* mr) {
#pragma clang fp contract(off)
    const long long i = first + (long long)blockIdx.x * QT + threadIdx.x;
    if (i >= v.n) return;
    Part p;
    p.x = v.x[i]; p.y = v.y[i]; p.z = v.z[i]; p.vx = v.vx[i]; p.vy = v.vy[i]; p.vz = v.vz[i]; p.m = v.m[i];
    p.beta = p.Rt = p.k1 = unset_value();
    p.ta = v.ta ? v.ta[i] : unset_value(); p.te = v.te ? v.te[i] : unset_value(); p.ti = v.ti ? v.ti[i] : unset_value();
    Src ref;
    ref.x = com7[0]; ref.y = com7[1]; ref.z = com7[2]; ref.vx = com7[3]; ref.vy = com7[4]; ref.vz = com7[5]; ref.m = com7[6];
    V3 a;
    mof_exact(p, ref, a);
    const double w = p.m / ref.m;
    ax_[i] = a.x; ay_[i] = a.y; az_[i] = a.z;
    rx[i] = w * a.x; ry[i] = w * a.y; rz[i] = w * a.z;
    mr[i] = w;
}

// ---- the reference's order of additions into a_j (src/rebxtools.c:71-107): i runs from the last body
// down; at i == j the body gains its own force, and it loses massratio_i a_i for every i >= j (Jacobi:
// i >= 1) or for every i (barycentric); a zero mass ratio subtracts nothing.  One thread per body j;
// the reactions of a tile of bodies i are staged in LDS and read by broadcast.
// nsrc: only bodies [0, nsrc) react (those behind are massless); a body at or behind nsrc gains its own force first -- the
// reference reaches it before any body that reacts.
template <bool JACOBI>
__global__ __launch_bounds__(QT) void seq_mof_apply(View v, long long nsrc, const double* ax_, const double* ay_, const double* az_,
                                                    const double* rx, const double* ry, const double* rz, const double* mr) {
#pragma clang fp contract(off)
    __shared__ double r[4][QT];
    const int tid = threadIdx.x;
    const long long j = (long long)blockIdx.x * QT + tid;
    const bool live = j < v.n;
    double acx = 0., acy = 0., acz = 0., ox = 0., oy = 0., oz = 0.;
    if (live) { acx = v.ax[j]; acy = v.ay[j]; acz = v.az[j]; ox = ax_[j]; oy = ay_[j]; oz = az_[j]; }
    const long long lowest = JACOBI ? (j > 1 ? j : 1) : 0;
    const long long last_tile = (nsrc - 1) / QT, first_tile = JACOBI ? (long long)blockIdx.x : 0;
    if (live && j >= nsrc) { acx += ox; acy += oy; acz += oz; }
    for (long long t = last_tile; t >= first_tile; t--) {
        const long long i0 = t * QT, i = i0 + tid;
        const bool have = i < nsrc;
        r[0][tid] = have ? rx[i] : 0.; r[1][tid] = have ? ry[i] : 0.; r[2][tid] = have ? rz[i] : 0.; r[3][tid] = have ? mr[i] : 0.;
        __syncthreads();
        const int cnt = (int)((nsrc - i0) < QT ? (nsrc - i0) : QT);
        if (t > (long long)blockIdx.x) {
            // every body of the tile lies behind every body of this workgroup
            for (int k = cnt - 1; k >= 0; k--)
                if (r[3][k] != 0.) { acx -= r[0][k]; acy -= r[1][k]; acz -= r[2][k]; }
        } else {
            for (int k = cnt - 1; k >= 0; k--) {
                const long long ii = i0 + k;
                if (ii < lowest) break;
                if (ii == j) { acx += ox; acy += oy; acz += oz; }
                if (r[3][k] != 0.) { acx -= r[0][k]; acy -= r[1][k]; acz -= r[2][k]; }
            }
        }
        __syncthreads();
    }
    if (live) { v.ax[j] = acx; v.ay[j] = acy; v.az[j] = acz; }
}

// ------------------------------------------------------------------ host driver
// ws: the workspace of the prefix-sum passes (scan_workspace_doubles(v.n) doubles).  nseq: the bodies the chains walk -- v.n, or the
// index behind the last massive body when massless bodies follow it; nseq <= REBXAMD_SEQ_MAX_BODIES.
hipError_t launch_seq_op(const View& v, const DevOp& op, double G, double* ws, unsigned* status, hipStream_t st, long long nseq) {
    const long long n = v.n;
    if (n <= 0) return hipSuccess;
    if (nseq > n || nseq < 1) nseq = n;
    const long long ns = (n + 1) & ~1LL;
    double* c0 = ws; double* c1 = c0 + ns; double* c2 = c1 + ns;
    double* c3 = c2 + ns; double* c4 = c3 + ns; double* c5 = c4 + ns;
    double* c6 = c5 + ns; double* c7 = c6 + ns; double* c8 = c7 + ns; double* c9 = c8 + ns;
    long long* cidx = reinterpret_cast<long long*>(c9 + ns);
    double* small = reinterpret_cast<double*>(cidx + ns);       // 32 doubles of results of the one-workgroup kernels
    View head = v;
    head.n = nseq;
    const long long ntail = n - nseq;
    const dim3 g((unsigned)((n + QT - 1) / QT)), gh((unsigned)((nseq + QT - 1) / QT)), gt((unsigned)((ntail + QT - 1) / QT)), one(1), b(QT);
    if (op.kind == REBXAMD_OP_GR) {
        long long* count = reinterpret_cast<long long*>(small);
        double* mtot = small + 1;
        double* sums10 = small + 2;
        hipLaunchKernelGGL(seq_compact, one, b, 0, st, head, c6, c7, c8, c9, cidx, count);
        hipLaunchKernelGGL(seq_newton, g, b, 0, st, v, G, (const double*)c6, (const double*)c7, (const double*)c8, (const double*)c9,
                           (const long long*)cidx, (const long long*)count, c0, c1, c2);
        hipLaunchKernelGGL(seq_gr_forward, one, b, 0, st, head, op, G, (const double*)c0, (const double*)c1, (const double*)c2, c3, c4, c5, mtot, status, sums10);
        if (ntail > 0) hipLaunchKernelGGL(seq_gr_tail, gt, b, 0, st, v, op, G, nseq, (const double*)c0, (const double*)c1, (const double*)c2, (const double*)sums10, status);
        hipLaunchKernelGGL(seq_gr_reverse, one, b, 0, st, head, (const double*)c3, (const double*)c4, (const double*)c5, (const double*)mtot);
        return hipGetLastError();
    }
    const bool jacobi = op.iarg == REBX_COORDINATES_JACOBI;
    hipLaunchKernelGGL(seq_mof_com, one, b, 0, st, head, jacobi ? 1 : 0, ntail, small, c0, c1, c2, c3, c4, c5, c6);
    if (jacobi) {
        if (ntail > 0) hipLaunchKernelGGL(seq_mof_terms_about, gt, b, 0, st, v, nseq, (const double*)small, c0, c1, c2, c3, c4, c5, c6);
        hipLaunchKernelGGL(seq_mof_apply<true>, g, b, 0, st, v, nseq, (const double*)c0, (const double*)c1, (const double*)c2,
                           (const double*)c3, (const double*)c4, (const double*)c5, (const double*)c6);
    } else {
        hipLaunchKernelGGL(seq_mof_terms_about, g, b, 0, st, v, 0LL, (const double*)small, c0, c1, c2, c3, c4, c5, c6);
        hipLaunchKernelGGL(seq_mof_apply<false>, g, b, 0, st, v, nseq, (const double*)c0, (const double*)c1, (const double*)c2,
                           (const double*)c3, (const double*)c4, (const double*)c5, (const double*)c6);
    }
    return hipGetLastError();
}

}  // namespace rebxamd

// grfull_kernels.hip -- gr_full (1PN Einstein-Infeld-Hoffmann, all pairs) for gfx950.
//
// Reference: src/gr_full.c:61-231.  The reference recomputes two potential sums inside its
// pair loop (O(N^3)) and keeps N x N separations on the stack; here every pass is one
// O(N^2) all-pairs sweep over fp64 columns:
//   pot   : pot_i   = sum_{k!=i} (1/c^2) G m_k / r_ik          (a2 of the reference; a1 = 4 pot_i)
//   const : ac_i    = sum_{j!=i} [ G m_j d_ij/r^3 (a1+..+a6) + G m_j (d_ij.(4v_i-3v_j)) (v_i-v_j)/r^3/c^2 ]
//   sweep : anew_i  = ac_i + sum_{j!=i} [ G m_j d_ij/r^3 (d_ij.t_j)/(2c^2) + 7/(2c^2) G m_j t_j / r ],
//           t_j = a_newton_j + a_old_j
// Not a dense contraction (every pair needs its own reciprocal square root), so this is fp64
// vector-ALU work, tiled through LDS: a workgroup owns 256 rows i (one per thread), stages
// 256-body tiles of j in LDS and every lane reads the same j by broadcast.  The j range is
// split over blockIdx.y so that small row shards still fill 256 CUs; the split partial sums are
// combined in fixed order by a second tiny kernel (deterministic, no atomics), which also
// applies the reference's stopping rule with a wavefront ballot.
//
// The *exact* variants walk j sequentially in one workgroup column with the reference's
// expression order and no FMA contraction; they exist for bit-level parity checks at small N.
#include "internal.h"
#include <cfloat>

namespace rebxamd {

static constexpr int GB = 256;      // rows per workgroup == bodies per LDS tile

enum { PASS_POT = 0, PASS_CONST = 1, PASS_SWEEP = 2, PASS_SWEEP_POT = 3 };    // the last: the first sweep (a_old = 0) and the potential sums in one pass

struct GrFullArgs {
    long long n, row0, row1;
    long long jchunk;                   // bodies per blockIdx.y
    double G, C2, invC2;
    const double *x, *y, *z, *vx, *vy, *vz, *m;
    const double *pot;                  // const pass
    const double *anx, *any_, *anz, *aox, *aoy, *aoz;   // sweep pass
    double* part;                       // [nchunks][nrows][ncomp]
    const unsigned* stop;               // sweeps queued ahead of the host: the pass returns at once when *stop is set (nullptr: never)
};

// ------------------------------------------------------------------ fast passes
struct __align__(16) TileA { double x, y, z, gm; };     // position and G*m of body j
struct __align__(16) TileB { double a, b, c, d; };      // const: v_j, phi_j - 2 v_j^2 ; sweep: B_j = 3.5 G m_j (a_newton_j + a_old_j), and B_j.x / 7
struct __align__(16) TileC { double b, c; };            // sweep: B_j.y / 7, B_j.z / 7

__device__ __forceinline__ double rsq_refined(double x) {
    // hardware estimate (~2^-23) + one Newton step: y (1 + e/2) with e = 1 - x y^2 leaves a relative
    // error of 3/8 e^2 ~ 5e-15, four orders below this path's 1e-10 bar (the per-particle forces,
    // whose bar is 1e-12, use the third-order step of effects.h); one instruction less per pair in
    // kernels that are bound by instruction issue
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(0.5 * y, e, y);
}

static constexpr double K_V = 1.224744871391589;       // sqrt(1.5): the const pass stages velocities scaled by it
static constexpr double INV_K = 0.816496580927726;     // 1 / sqrt(1.5)

// State of one row i while it sweeps over the bodies j.
struct Row {
    double xi, yi, zi;
    double ax, ay, az, bx, by, bz, p4;  // const pass: (4/k) v_i, 4 v_i, 4 phi_i c^2 - v_i^2   (k = sqrt(1.5), see grfull_tile)
    double s0, s1, s2, sp;              // running sums (sp: the potential's; const pass: sum of w2)
    double t0, t1, t2;                  // const pass: sum of w2 * (k v_j)
    int self;           // index of the row inside the current tile, -1 if it is not in it
};

// One 256-body tile against this thread's R rows: every body fetched from LDS is used R times.
// DIAG: the tile contains rows of this workgroup, so the self pair has to be masked (only the
// tiles that overlap the workgroup's own rows pay for the compare).
// sweep: with B_j = 3.5 G m_j (a_newton_j + a_old_j) the pair's term is  (1/r) ( (d.B_j/7) d / r^2 + B_j ):
// 22 instructions per pair (3 differences, 3 for r^2, 5 for the refined reciprocal square root, 1 for
// 1/r^2, 3 for the dot product with B_j/7, 1 product, 2 x 3 fused multiply-adds), the mass already inside
// B_j and both B_j and B_j/7 staged, so that no per-pair scaling is left.
// SHORT: only the first `cnt` bodies of the tile (the one-launch evaluation of a small system: the bodies behind them are
// padding, whose terms are zeros).
template <int PASS, bool DIAG, int R, bool SHORT = false>
__device__ __forceinline__ void grfull_tile(const TileA* __restrict__ ta, const TileB* __restrict__ tb, const TileC* __restrict__ tc, Row (&row)[R], const int cnt = GB) {
    constexpr bool SWEEP = PASS == PASS_SWEEP || PASS == PASS_SWEEP_POT;
    const int lim = SHORT ? cnt : GB;
#pragma unroll 4
    for (int jj = 0; jj < lim; jj++) {
        const TileA A = ta[jj];
        TileB B;
        TileC C;
        if (PASS != PASS_POT) B = tb[jj];
        if (SWEEP) C = tc[jj];
#pragma unroll
        for (int q = 0; q < R; q++) {
            Row& w = row[q];
            const double dx = w.xi - A.x, dy = w.yi - A.y, dz = w.zi - A.z;
            double r2 = dx * dx + dy * dy + dz * dz;
            if (SWEEP) {
                const bool self = DIAG && jj == w.self;
                if (self) r2 = 1.;
                double ri = rsq_refined(r2);
                if (self) ri = 0.;
                const double h = (ri * ri) * (dx * B.d + dy * C.b + dz * C.c);
                w.s0 = fma(ri, fma(h, dx, B.a), w.s0);
                w.s1 = fma(ri, fma(h, dy, B.b), w.s1);
                w.s2 = fma(ri, fma(h, dz, B.c), w.s2);
                if (PASS == PASS_SWEEP_POT) w.sp = fma(A.gm, ri, w.sp);
                continue;
            }
            double gm = A.gm;
            if (DIAG && jj == w.self) { gm = 0.; r2 = 1.; }
            const double ri = rsq_refined(r2);
            if (PASS == PASS_POT) {
                w.s0 = fma(gm, ri, w.s0);
            } else {
                // The pair's term is  g3 ( f1 d + f2 (v_i - v_j) ),  g3 = G m_j / r^3,
                //   f1 = 4 phi_i + phi_j - v_i^2 - 2 v_j^2 + 4 v_i.v_j + 1.5 (d.v_j)^2 / r^2,   f2 = d.(4 v_i - 3 v_j)     (units of c^2).
                // Every use of v_j is linear or a square, so the tile holds k v_j with k = sqrt(1.5) (B.a..c; B.d = phi_j c^2 - 2 v_j^2)
                // and the row the matching multiples of v_i: 1.5 (d.v_j)^2 = (d.k v_j)^2, 4 v_i.v_j = ((4/k) v_i).(k v_j),
                // 3 d.v_j = (3/k) d.(k v_j).  The part  g3 f2 (v_i - v_j)  is summed as  v_i sum(w2) - (1/k) sum(w2 k v_j)  with
                // w2 = g3 f2, the factors outside the sums applied once per row: 36 instructions per pair instead of 42.
                const double ri2 = ri * ri;
                const double g3 = gm * ri2 * ri;
                const double dvj = dx * B.a + dy * B.b + dz * B.c;
                const double dvi = dx * w.bx + dy * w.by + dz * w.bz;
                const double vv = fma(w.ax, B.a, fma(w.ay, B.b, fma(w.az, B.c, w.p4 + B.d)));
                const double f1 = fma(ri2 * dvj, dvj, vv);
                const double f2 = fma(-2.449489742783178, dvj, dvi);        // 3 / sqrt(1.5) = sqrt(6)
                const double u = g3 * f1, w2 = g3 * f2;
                w.s0 = fma(u, dx, w.s0); w.s1 = fma(u, dy, w.s1); w.s2 = fma(u, dz, w.s2);
                w.t0 = fma(w2, B.a, w.t0); w.t1 = fma(w2, B.b, w.t1); w.t2 = fma(w2, B.c, w.t2);
                w.sp += w2;
            }
        }
    }
}

// what a row carries into a pass (i: its body)
template <int PASS>
__device__ __forceinline__ void grfull_row_init(const GrFullArgs& a, long long i, bool live, Row& w) {
    w.xi = w.yi = w.zi = w.ax = w.ay = w.az = w.bx = w.by = w.bz = w.p4 = 0.;
    w.s0 = w.s1 = w.s2 = w.sp = w.t0 = w.t1 = w.t2 = 0.;
    w.self = -1;
    if (live) {
        w.xi = a.x[i]; w.yi = a.y[i]; w.zi = a.z[i];
        if (PASS == PASS_CONST) {
            const double vxi = a.vx[i], vyi = a.vy[i], vzi = a.vz[i];
            const double vi2 = vxi * vxi + vyi * vyi + vzi * vzi;
            w.p4 = 4. * a.pot[i] * a.C2 - vi2;      // work in units where 1/c^2 is applied once at the end
            w.bx = 4. * vxi; w.by = 4. * vyi; w.bz = 4. * vzi;
            w.ax = w.bx * INV_K; w.ay = w.by * INV_K; w.az = w.bz * INV_K;
        }
    }
}

// body j (behind j1: padding) into slot `slot` of the tile
template <int PASS>
__device__ __forceinline__ void grfull_stage(const GrFullArgs& a, long long j, long long j1, int slot, TileA* ta, TileB* tb, TileC* tc) {
    constexpr bool SWEEP = PASS == PASS_SWEEP || PASS == PASS_SWEEP_POT;
    if (j < j1) {
        TileA A; A.x = a.x[j]; A.y = a.y[j]; A.z = a.z[j]; A.gm = a.G * a.m[j];
        ta[slot] = A;
        if (PASS == PASS_CONST) {
            const double vxj = a.vx[j], vyj = a.vy[j], vzj = a.vz[j];
            TileB B; B.a = K_V * vxj; B.b = K_V * vyj; B.c = K_V * vzj;
            B.d = a.pot[j] * a.C2 - 2. * (vxj * vxj + vyj * vyj + vzj * vzj);
            tb[slot] = B;
        }
        if (SWEEP) {
            // (the first sweep starts from a_old = 0 and does not read it)
            const double tx = PASS == PASS_SWEEP ? a.anx[j] + a.aox[j] : a.anx[j];
            const double ty = PASS == PASS_SWEEP ? a.any_[j] + a.aoy[j] : a.any_[j];
            const double tz = PASS == PASS_SWEEP ? a.anz[j] + a.aoz[j] : a.anz[j];
            const double g35 = 3.5 * A.gm, g05 = 0.5 * A.gm;
            TileB B; B.a = g35 * tx; B.b = g35 * ty; B.c = g35 * tz; B.d = g05 * tx;
            TileC C; C.b = g05 * ty; C.c = g05 * tz;
            tb[slot] = B; tc[slot] = C;
        }
    } else {
        // padding bodies: zero mass far away contributes exactly nothing
        TileA A; A.x = 1.e100; A.y = 1.e100; A.z = 1.e100; A.gm = 0.;
        TileB B; B.a = B.b = B.c = B.d = 0.;
        ta[slot] = A; tb[slot] = B;
        if (SWEEP) { TileC C; C.b = C.c = 0.; tc[slot] = C; }
    }
}

// what the constant pass leaves per row:  + v_i sum(w2) - (1/k) sum(w2 k v_j);  row.b = 4 v_i
__device__ __forceinline__ void grfull_const_result(const Row& w, double scale, double& o0, double& o1, double& o2) {
    o0 = fma(0.25 * w.bx, w.sp, fma(-INV_K, w.t0, w.s0)) * scale;
    o1 = fma(0.25 * w.by, w.sp, fma(-INV_K, w.t1, w.s1)) * scale;
    o2 = fma(0.25 * w.bz, w.sp, fma(-INV_K, w.t2, w.s2)) * scale;
}

// the reference's stopping rule for one row (gr_full.c:206-220): signed ratio, NaN never counts, threshold 1e-30
__device__ __forceinline__ bool grfull_row_unconverged(double s0, double s1, double s2, double o0, double o1, double o2) {
    const double d0 = (fabs(s0) < 1.e-30) ? 0. : fabs(s0 - o0) / s0;
    const double d1 = (fabs(s1) < 1.e-30) ? 0. : fabs(s1 - o1) / s1;
    const double d2 = (fabs(s2) < 1.e-30) ? 0. : fabs(s2 - o2) / s2;
    return (d0 >= 1.e-30) || (d1 >= 1.e-30) || (d2 >= 1.e-30);
}

// Workgroup = 256 threads, GB*R rows: thread t owns rows t, t + 256, ... of the workgroup's block.
// SHORT: the j range of a workgroup is shorter than a tile (systems of a few hundred to a few thousand bodies, split finely
// so that the pairs spread over many compute units): the tile loop stops at the bodies staged.
template <int PASS, int R, bool SHORT = false>
__global__ __launch_bounds__(GB) void grfull_fast_kernel(GrFullArgs a) {
    constexpr bool SWEEP = PASS == PASS_SWEEP || PASS == PASS_SWEEP_POT;
    __shared__ TileA ta[GB];
    __shared__ TileB tb[GB];
    __shared__ TileC tc[SWEEP ? GB : 1];
    const int tid = threadIdx.x;
    if (PASS == PASS_SWEEP && a.stop != nullptr && *a.stop != 0u) return;       // a sweep queued past convergence
    const long long nrows = a.row1 - a.row0;
    const long long j0 = (long long)blockIdx.y * a.jchunk;
    long long j1 = j0 + a.jchunk; if (j1 > a.n) j1 = a.n;
    const long long blk_lo = a.row0 + (long long)blockIdx.x * (GB * R), blk_hi = blk_lo + GB * R;   // rows of this workgroup

    Row row[R];
    long long ri_[R];
    bool live[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const long long r = (long long)blockIdx.x * (GB * R) + q * GB + tid;
        ri_[q] = r;
        live[q] = r < nrows;
        grfull_row_init<PASS>(a, a.row0 + r, live[q], row[q]);
    }

    for (long long jt = j0; jt < j1; jt += GB) {
        const long long j = jt + tid;
        __syncthreads();
        grfull_stage<PASS>(a, j, j1, tid, ta, tb, tc);
        __syncthreads();
        if (jt < blk_hi && jt + GB > blk_lo) {      // workgroup-uniform: this tile overlaps our own rows
#pragma unroll
            for (int q = 0; q < R; q++) {
                const long long i = a.row0 + ri_[q];
                row[q].self = (live[q] && i >= jt && i < jt + GB) ? (int)(i - jt) : -1;
            }
            grfull_tile<PASS, true, R, SHORT>(ta, tb, tc, row, (int)((j1 - jt) < GB ? (j1 - jt) : GB));
        } else {
            grfull_tile<PASS, false, R, SHORT>(ta, tb, tc, row, (int)((j1 - jt) < GB ? (j1 - jt) : GB));
        }
    }
    const double scale = a.invC2;
#pragma unroll
    for (int q = 0; q < R; q++) {
        if (!live[q]) continue;
        if (PASS == PASS_POT) {
            a.part[(long long)blockIdx.y * nrows + ri_[q]] = row[q].s0 * scale;
        } else if (PASS == PASS_SWEEP_POT) {
            double* o = a.part + ((long long)blockIdx.y * nrows + ri_[q]) * 4;
            o[0] = row[q].s0 * scale; o[1] = row[q].s1 * scale; o[2] = row[q].s2 * scale; o[3] = row[q].sp * scale;
        } else if (PASS == PASS_CONST) {
            double* o = a.part + ((long long)blockIdx.y * nrows + ri_[q]) * 3;
            grfull_const_result(row[q], scale, o[0], o[1], o[2]);
        } else {
            double* o = a.part + ((long long)blockIdx.y * nrows + ri_[q]) * 3;
            o[0] = row[q].s0 * scale; o[1] = row[q].s1 * scale; o[2] = row[q].s2 * scale;
        }
    }
}

// combine the j-chunk partial sums (fixed order); sweep: add a_const and test the stopping rule
template <int PASS>
__global__ __launch_bounds__(256) void grfull_combine_kernel(const double* part, int nchunks, long long row0, long long row1,
                                                             const double* acx, const double* acy, const double* acz,
                                                             const double* aox, const double* aoy, const double* aoz,
                                                             double* o0, double* o1, double* o2, unsigned* flag, const unsigned* stop, int stride) {
    // stride: doubles per (chunk, row) entry of `part` (1 for the potential pass, 3 for the others; 4 when the
    // first sweep and the potential sums shared a pass: the sweep's three sums, then the potential's)
    if (PASS == PASS_SWEEP && stop != nullptr && *stop != 0u) return;
    const long long nrows = row1 - row0;
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    bool unconverged = false;
    if (r < nrows) {
        const long long i = row0 + r;
        if (PASS == PASS_POT) {
            double s = 0.;
            for (int c = 0; c < nchunks; c++) s += part[((long long)c * nrows + r) * stride + (stride - 1)];
            o0[i] = s;
        } else {
            double s0 = 0., s1 = 0., s2 = 0.;
            for (int c = 0; c < nchunks; c++) {
                const double* q = part + ((long long)c * nrows + r) * stride;
                s0 += q[0]; s1 += q[1]; s2 += q[2];
            }
            if (PASS == PASS_SWEEP) {
                s0 += acx[i]; s1 += acy[i]; s2 += acz[i];
                unconverged = grfull_row_unconverged(s0, s1, s2, aox[i], aoy[i], aoz[i]);
            }
            o0[i] = s0; o1[i] = s1; o2[i] = s2;
        }
    }
    if (PASS == PASS_SWEEP) {
        const unsigned long long any = __ballot(unconverged);      // one vote per wavefront
        if (any != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
    }
}

// ------------------------------------------------------------------ exact passes (sequential j)
// one row of a pass; returns whether the row fails the stopping rule (sweeps)
template <int PASS>
__device__ __forceinline__ bool grfull_exact_row(const GrFullArgs& a, long long i, const double* acx, const double* acy, const double* acz,
                                                 double* o0, double* o1, double* o2) {
#pragma clang fp contract(off)
    bool unconverged = false;
    {
        const double G = a.G, C2 = a.C2;
        const double xi = a.x[i], yi = a.y[i], zi = a.z[i];
        if (PASS == PASS_POT) {
            double s2 = 0.;
            for (long long k = 0; k < a.n; k++) {
                if (k == i) continue;
                const double dx = xi - a.x[k], dy = yi - a.y[k], dz = zi - a.z[k];
                const double rr = sqrt(dx * dx + dy * dy + dz * dz);
                s2 += (1. / (C2)) * G * a.m[k] / rr;
            }
            o0[i] = s2;
        } else if (PASS == PASS_CONST) {
            const double vxi = a.vx[i], vyi = a.vy[i], vzi = a.vz[i];
            const double vi2 = vxi * vxi + vyi * vyi + vzi * vzi;
            const double a1 = 4. * a.pot[i];        // exactly the reference's own 4/c^2 sum (power-of-two scaling)
            double cx = 0., cy = 0., cz = 0.;
            for (long long j = 0; j < a.n; j++) {
                if (j == i) continue;
                const double dx = xi - a.x[j], dy = yi - a.y[j], dz = zi - a.z[j];
                const double rr = sqrt(dx * dx + dy * dy + dz * dz);
                const double rij2 = rr * rr;
                const double rij3 = rij2 * rr;
                const double vxj = a.vx[j], vyj = a.vy[j], vzj = a.vz[j], mj = a.m[j];
                const double a3 = -vi2 / (C2);
                const double vj2 = vxj * vxj + vyj * vyj + vzj * vzj;
                const double a4 = -2. * vj2 / (C2);
                const double a5 = (4. / (C2)) * (vxi * vxj + vyi * vyj + vzi * vzj);
                const double a6_0 = dx * vxj + dy * vyj + dz * vzj;
                const double a6 = (3. / (2. * C2)) * a6_0 * a6_0 / rij2;
                const double factor1 = a1 + a.pot[j] + a3 + a4 + a5 + a6;
                cx += G * mj * dx * factor1 / rij3;
                cy += G * mj * dy * factor1 / rij3;
                cz += G * mj * dz * factor1 / rij3;
                const double dvx = vxi - vxj, dvy = vyi - vyj, dvz = vzi - vzj;
                const double factor2 = dx * (4. * vxi - 3. * vxj) + dy * (4. * vyi - 3. * vyj) + dz * (4. * vzi - 3. * vzj);
                cx += G * mj * factor2 * dvx / rij3 / (C2);
                cy += G * mj * factor2 * dvy / rij3 / (C2);
                cz += G * mj * factor2 * dvz / rij3 / (C2);
            }
            o0[i] = cx; o1[i] = cy; o2[i] = cz;
        } else {
            double nx = 0., ny = 0., nz = 0.;
            for (long long j = 0; j < a.n; j++) {
                if (j == i) continue;
                const double dx = xi - a.x[j], dy = yi - a.y[j], dz = zi - a.z[j];
                const double rr = sqrt(dx * dx + dy * dy + dz * dz);
                const double rij3 = rr * rr * rr;
                const double mj = a.m[j];
                const double tx = a.anx[j] + a.aox[j], ty = a.any_[j] + a.aoy[j], tz = a.anz[j] + a.aoz[j];
                nx += (G * mj * dx / rij3) * (dx * tx + dy * ty + dz * tz) / (2. * C2) + (7. / (2. * C2)) * G * mj * tx / rr;
                ny += (G * mj * dy / rij3) * (dx * tx + dy * ty + dz * tz) / (2. * C2) + (7. / (2. * C2)) * G * mj * ty / rr;
                nz += (G * mj * dz / rij3) * (dx * tx + dy * ty + dz * tz) / (2. * C2) + (7. / (2. * C2)) * G * mj * tz / rr;
            }
            const double s0 = acx[i] + nx, s1 = acy[i] + ny, s2 = acz[i] + nz;
            unconverged = grfull_row_unconverged(s0, s1, s2, a.aox[i], a.aoy[i], a.aoz[i]);
            o0[i] = s0; o1[i] = s1; o2[i] = s2;
        }
    }
    return unconverged;
}

template <int PASS>
__global__ __launch_bounds__(GB) void grfull_exact_kernel(GrFullArgs a, const double* acx, const double* acy, const double* acz,
                                                          double* o0, double* o1, double* o2, unsigned* flag) {
    if (PASS == PASS_SWEEP && a.stop != nullptr && *a.stop != 0u) return;
    const long long r = (long long)blockIdx.x * GB + threadIdx.x;
    const long long i = a.row0 + r;
    bool unconverged = false;
    if (i < a.row1) unconverged = grfull_exact_row<PASS>(a, i, acx, acy, acz, o0, o1, o2);
    if (PASS == PASS_SWEEP) {
        const unsigned long long any = __ballot(unconverged);
        if (any != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
    }
}

// a_newton = incoming acceleration (+ the 0-1 pair when REBOUND's gravity left it out,
// reference gr_full.c:92-102); a_old = 0
__device__ __forceinline__ void grfull_prepare_row(const View& v, double G, int ignore10, long long i, double* anx, double* any_, double* anz,
                                                   double* a0x, double* a0y, double* a0z) {
#pragma clang fp contract(off)
    double ax = v.ax[i], ay = v.ay[i], az = v.az[i];
    if (ignore10 && i < 2 && v.n > 1) {
        const double dx = v.x[0] - v.x[1], dy = v.y[0] - v.y[1], dz = v.z[0] - v.z[1];
        const double rr = sqrt(dx * dx + dy * dy + dz * dz);
        const double prefact = -G / (rr * rr * rr);
        if (i == 0) { const double p1 = prefact * v.m[1]; ax += p1 * dx; ay += p1 * dy; az += p1 * dz; }
        else        { const double p0 = prefact * v.m[0]; ax -= p0 * dx; ay -= p0 * dy; az -= p0 * dz; }
    }
    anx[i] = ax; any_[i] = ay; anz[i] = az;
    a0x[i] = 0.; a0y[i] = 0.; a0z[i] = 0.;
}
__global__ __launch_bounds__(256) void grfull_prepare_kernel(View v, double G, int ignore10, double* anx, double* any_, double* anz,
                                                             double* a0x, double* a0y, double* a0z) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v.n; i += (long long)gridDim.x * 256)
        grfull_prepare_row(v, G, ignore10, i, anx, any_, anz, a0x, a0y, a0z);
}

__global__ __launch_bounds__(256) void grfull_finish_kernel(View v, const double* ax, const double* ay, const double* az) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v.n; i += (long long)gridDim.x * 256) {
        v.ax[i] += ax[i]; v.ay[i] += ay[i]; v.az[i] += az[i];
    }
}

// Sweeps queued ahead of the host (rebxamd_grfull): ctl[0] = stop (the stopping rule has been met), ctl[1] = sweeps
// run, ctl[2] = the vote of the sweep that has just run (raised by any row that fails the rule).
__global__ void grfull_vote_kernel(unsigned* ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || ctl[0] != 0u) return;
    ctl[1] += 1u;
    if (ctl[2] == 0u) ctl[0] = 1u;
    ctl[2] = 0u;
}
// v.a += the newest iterate: sweep k reads buffer k & 1 and writes the other one, so after ctl[1] sweeps it is
// buffer ctl[1] & 1.  All ten sweeps run and the rule still not met: the reference's warning (gr_full.c:221-223).
__global__ __launch_bounds__(256) void grfull_finish_select_kernel(View v, const double* b0x, const double* b0y, const double* b0z,
                                                                   const double* b1x, const double* b1y, const double* b1z,
                                                                   const unsigned* ctl, unsigned* status) {
    const bool odd = (ctl[1] & 1u) != 0u;
    const double* ax = odd ? b1x : b0x; const double* ay = odd ? b1y : b0y; const double* az = odd ? b1z : b0z;
    if (blockIdx.x == 0 && threadIdx.x == 0 && ctl[0] == 0u) atomicOr(status, WARN_GRFULL_SWEEPS);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < v.n; i += (long long)gridDim.x * 256) {
        v.ax[i] += ax[i]; v.ay[i] += ay[i]; v.az[i] += az[i];
    }
}

// ------------------------------------------------------------------ small systems: one launch per evaluation
// Up to 256 bodies -- what the reference's own O(N^3) loops and N x N stack arrays are used for -- one workgroup holds a row per
// thread, and the whole evaluation (a_newton, potential sums shared with the first sweep, constant part, up to ten sweeps with
// the stopping rule, the addition to the particles) runs in ONE launch, the passes separated by workgroup barriers, instead of
// the three dozen launches of the queued passes (each a few microseconds of front-end time, whatever the size).  The rows go
// through the same functions as in those passes, in the same order, with one j chunk: the results are the same bits.
// w: 13 columns of n doubles (a_newton 3, pot 1, a_const 3, two iterates 3 + 3).  ctl[0] <- 1 when the rule was met,
// ctl[1] <- sweeps run.
template <bool EXACT>
__global__ __launch_bounds__(GB) void grfull_small_kernel(View v, GrFullArgs a, int ignore10, double* w, unsigned* ctl, unsigned* status) {
    __shared__ TileA ta[GB];
    __shared__ TileB tb[GB];
    __shared__ TileC tc[GB];
    const int tid = threadIdx.x;
    const long long n = a.n, i = tid;
    const bool live = i < n;
    double *anx = w, *any_ = w + n, *anz = w + 2 * n, *pot = w + 3 * n;
    double *acx = w + 4 * n, *acy = w + 5 * n, *acz = w + 6 * n;
    double* const b0[3] = {w + 7 * n, w + 8 * n, w + 9 * n};
    double* const b1[3] = {w + 10 * n, w + 11 * n, w + 12 * n};
    a.anx = anx; a.any_ = any_; a.anz = anz; a.pot = pot;
    if (live) grfull_prepare_row(v, a.G, ignore10, i, anx, any_, anz, b0[0], b0[1], b0[2]);
    __syncthreads();
    const double scale = a.invC2;
    double p0 = 0., p1 = 0., p2 = 0.;           // default arithmetic: the pair sums of the first sweep, formed with the potential sums
    Row row;
    if (EXACT) {
        if (live) grfull_exact_row<PASS_POT>(a, i, nullptr, nullptr, nullptr, pot, nullptr, nullptr);
    } else {
        grfull_row_init<PASS_SWEEP_POT>(a, i, live, row);
        grfull_stage<PASS_SWEEP_POT>(a, tid, n, tid, ta, tb, tc);
        __syncthreads();
        row.self = live ? tid : -1;
        grfull_tile<PASS_SWEEP_POT, true, 1, true>(ta, tb, tc, reinterpret_cast<Row(&)[1]>(row), (int)n);
        p0 = row.s0 * scale; p1 = row.s1 * scale; p2 = row.s2 * scale;
        if (live) pot[i] = 0. + row.sp * scale;
    }
    __syncthreads();
    if (EXACT) {
        if (live) grfull_exact_row<PASS_CONST>(a, i, nullptr, nullptr, nullptr, acx, acy, acz);
    } else {
        grfull_row_init<PASS_CONST>(a, i, live, row);
        grfull_stage<PASS_CONST>(a, tid, n, tid, ta, tb, tc);
        __syncthreads();
        row.self = live ? tid : -1;
        grfull_tile<PASS_CONST, true, 1, true>(ta, tb, tc, reinterpret_cast<Row(&)[1]>(row), (int)n);
        if (live) {
            double c0, c1, c2;
            grfull_const_result(row, scale, c0, c1, c2);
            acx[i] = 0. + c0; acy[i] = 0. + c1; acz[i] = 0. + c2;
        }
    }
    __syncthreads();
    int sweeps = 0;
    bool met = false;
    for (int k = 0; k < 10 && !met; k++) {          // the reference's loop bound (gr_full.c:173)
        double* const* src = (k & 1) ? b1 : b0;
        double* const* dst = (k & 1) ? b0 : b1;
        a.aox = src[0]; a.aoy = src[1]; a.aoz = src[2];
        bool unconverged = false;
        if (EXACT) {
            if (live) unconverged = grfull_exact_row<PASS_SWEEP>(a, i, acx, acy, acz, dst[0], dst[1], dst[2]);
        } else {
            double s0 = p0, s1 = p1, s2 = p2;
            if (k > 0) {
                grfull_row_init<PASS_SWEEP>(a, i, live, row);
                grfull_stage<PASS_SWEEP>(a, tid, n, tid, ta, tb, tc);
                __syncthreads();
                row.self = live ? tid : -1;
                grfull_tile<PASS_SWEEP, true, 1, true>(ta, tb, tc, reinterpret_cast<Row(&)[1]>(row), (int)n);
                s0 = row.s0 * scale; s1 = row.s1 * scale; s2 = row.s2 * scale;
            }
            if (live) {
                s0 = (0. + s0) + acx[i]; s1 = (0. + s1) + acy[i]; s2 = (0. + s2) + acz[i];
                unconverged = grfull_row_unconverged(s0, s1, s2, src[0][i], src[1][i], src[2][i]);
                dst[0][i] = s0; dst[1][i] = s1; dst[2][i] = s2;
            }
        }
        sweeps = k + 1;
        met = __syncthreads_or(unconverged ? 1 : 0) == 0;      // (also the barrier between this sweep's rows and the next one's tile)
    }
    double* const* newest = (sweeps & 1) ? b1 : b0;
    if (live) { v.ax[i] += newest[0][i]; v.ay[i] += newest[1][i]; v.az[i] += newest[2][i]; }
    if (tid == 0) {
        ctl[0] = met ? 1u : 0u; ctl[1] = (unsigned)sweeps; ctl[2] = 0u;
        if (!met) atomicOr(status, WARN_GRFULL_SWEEPS);
    }
}

hipError_t launch_grfull_small(const View& v, double G, double c, int gravity_ignore_10, double* w, unsigned* ctl, unsigned* status, bool exact, hipStream_t st) {
    if (v.n <= 0 || v.n > GB) return hipErrorInvalidValue;
    GrFullArgs a;
    a.n = v.n; a.row0 = 0; a.row1 = v.n; a.jchunk = v.n;
    a.G = G; a.C2 = c * c; a.invC2 = 1. / (c * c);
    a.x = v.x; a.y = v.y; a.z = v.z; a.vx = v.vx; a.vy = v.vy; a.vz = v.vz; a.m = v.m;
    a.pot = nullptr; a.anx = a.any_ = a.anz = a.aox = a.aoy = a.aoz = nullptr; a.part = nullptr; a.stop = nullptr;
    if (exact) hipLaunchKernelGGL(grfull_small_kernel<true>, dim3(1), dim3(GB), 0, st, v, a, gravity_ignore_10, w, ctl, status);
    else hipLaunchKernelGGL(grfull_small_kernel<false>, dim3(1), dim3(GB), 0, st, v, a, gravity_ignore_10, w, ctl, status);
    return hipGetLastError();
}
int grfull_small_limit() { return GB; }

// ------------------------------------------------------------------ Hamiltonian (gr_full.c:251-377)
// Canonical velocities vtilde by fixed-point sweeps, then kinetic + Newtonian + post-Newtonian
// energy.  The reference sweeps in place (Gauss-Seidel: body i sees the new values of bodies
// before it); here every sweep reads the previous iterate (Jacobi), which parallelises over i.
// Both contract with ratio ~ (v/c)^2 per sweep to the same fixed point, so after the reference's
// ten sweeps they agree to rounding unless the system is strongly relativistic.
struct __align__(16) TileH { double x, y, z, m, vx, vy, vz, pad; };

// (blk: the workgroup's index among those of the sweep; t: its tile in LDS)
__device__ __forceinline__ void grfull_h_sweep_body(const View& v, double G, double C2, const double* vtx, const double* vty, const double* vtz,
                                                    double* ox, double* oy, double* oz, TileH* t, long long blk) {
#pragma clang fp contract(off)
    const long long i = blk * GB + threadIdx.x;
    const bool live = i < v.n;
    double xi = 0., yi = 0., zi = 0., vix = 0., viy = 0., viz = 0.;
    if (live) { xi = v.x[i]; yi = v.y[i]; zi = v.z[i]; vix = vtx[i]; viy = vty[i]; viz = vtz[i]; }
    double dx = 0., dy = 0., dz = 0.;
    for (long long jt = 0; jt < v.n; jt += GB) {
        const long long j = jt + threadIdx.x;
        __syncthreads();
        TileH h;
        if (j < v.n) { h.x = v.x[j]; h.y = v.y[j]; h.z = v.z[j]; h.m = v.m[j]; h.vx = vtx[j]; h.vy = vty[j]; h.vz = vtz[j]; }
        else { h.x = h.y = h.z = 1.e100; h.m = 0.; h.vx = h.vy = h.vz = 0.; }
        h.pad = 0.;
        t[threadIdx.x] = h;
        __syncthreads();
        const int cnt = (int)((v.n - jt) < GB ? (v.n - jt) : GB);
        for (int k = 0; k < cnt; k++) {
            if (jt + k == i) continue;
            const TileH b = t[k];
            const double xij = b.x - xi, yij = b.y - yi, zij = b.z - zi;
            const double rij2 = xij * xij + yij * yij + zij * zij;
            const double rij = sqrt(rij2);
            const double rijdotvj = b.vx * xij + b.vy * yij + b.vz * zij;
            const double pfac = b.m / rij;
            dx += pfac * (6. * vix - 7. * b.vx - rijdotvj * xij / rij2);
            dy += pfac * (6. * viy - 7. * b.vy - rijdotvj * yij / rij2);
            dz += pfac * (6. * viz - 7. * b.vz - rijdotvj * zij / rij2);
        }
    }
    if (live) {
        const double A = (1. - 0.5 * (vix * vix + viy * viy + viz * viz) / C2);
        dx *= G / (2. * C2); dy *= G / (2. * C2); dz *= G / (2. * C2);
        ox[i] = (v.vx[i] + dx) / A; oy[i] = (v.vy[i] + dy) / A; oz[i] = (v.vz[i] + dz) / A;
    }
}
__global__ __launch_bounds__(GB) void grfull_h_sweep_kernel(View v, double G, double C2, const double* vtx, const double* vty, const double* vtz,
                                                          double* ox, double* oy, double* oz) {
    __shared__ TileH t[GB];
    grfull_h_sweep_body(v, G, C2, vtx, vty, vtz, ox, oy, oz, t, (long long)blockIdx.x);
}

// per-workgroup sums of (kinetic, Newtonian potential, post-Newtonian) -> partials[3*block + c]
__device__ __forceinline__ void grfull_h_energy_body(const View& v, double G, double C2, const double* vtx, const double* vty, const double* vtz, double* partials,
                                                     TileH* t, double (*red)[3], long long blk) {
#pragma clang fp contract(off)
    const long long i = blk * GB + threadIdx.x;
    const bool live = i < v.n;
    double xi = 0., yi = 0., zi = 0., mi = 0., vix = 0., viy = 0., viz = 0.;
    if (live) { xi = v.x[i]; yi = v.y[i]; zi = v.z[i]; mi = v.m[i]; vix = vtx[i]; viy = vty[i]; viz = vtz[i]; }
    const double vt2 = vix * vix + viy * viy + viz * viz;
    double sumk = 0., e_pot = 0., s_a = 0., s_b = 0.;
    // e_pn_i = -G/(4 C2) m_i sum_j m_j/r (6 vt2 - 7 vi.vj - (r.vi)(r.vj)/r^2 + sumk): the sumk part factors out
    for (long long jt = 0; jt < v.n; jt += GB) {
        const long long j = jt + threadIdx.x;
        __syncthreads();
        TileH h;
        if (j < v.n) { h.x = v.x[j]; h.y = v.y[j]; h.z = v.z[j]; h.m = v.m[j]; h.vx = vtx[j]; h.vy = vty[j]; h.vz = vtz[j]; }
        else { h.x = h.y = h.z = 1.e100; h.m = 0.; h.vx = h.vy = h.vz = 0.; }
        h.pad = 0.;
        t[threadIdx.x] = h;
        __syncthreads();
        const int cnt = (int)((v.n - jt) < GB ? (v.n - jt) : GB);
        for (int k = 0; k < cnt; k++) {
            if (jt + k == i) continue;
            const TileH b = t[k];
            const double xij = b.x - xi, yij = b.y - yi, zij = b.z - zi;
            const double rij2 = xij * xij + yij * yij + zij * zij;
            const double rij = sqrt(rij2);
            const double w = b.m / rij;
            sumk -= 2. * G * w;
            const double rijdotvj = b.vx * xij + b.vy * yij + b.vz * zij;
            const double rijdotvi = vix * xij + viy * yij + viz * zij;
            const double vidotvj = vix * b.vx + viy * b.vy + viz * b.vz;
            s_a += w * (6. * vt2 - 7 * vidotvj - rijdotvi * rijdotvj / rij2);
            s_b += w;
            if (jt + k > i) e_pot -= G * mi * w;
        }
    }
    double e[3] = {0., 0., 0.};
    if (live) {
        e[0] = 0.5 * mi * vt2;
        e[1] = e_pot;
        e[2] = -(G / (4. * C2)) * mi * (s_a + sumk * s_b) - mi / (8. * C2) * vt2 * vt2;
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e[c] += __shfl_down(e[c], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) for (int c = 0; c < 3; c++) red[wave][c] = e[c];
    __syncthreads();
    if (threadIdx.x < 3) {
        double ssum = 0.;
        for (int w = 0; w < GB / 64; w++) ssum += red[w][threadIdx.x];
        partials[3 * blk + threadIdx.x] = ssum;
    }
}
__global__ __launch_bounds__(GB) void grfull_h_energy_kernel(View v, double G, double C2, const double* vtx, const double* vty, const double* vtz, double* partials) {
    __shared__ TileH t[GB];
    __shared__ double red[GB / 64][3];
    grfull_h_energy_body(v, G, C2, vtx, vty, vtz, partials, t, red, (long long)blockIdx.x);
}

// Up to 256 bodies: the ten sweeps, the energies and their sums in one launch of one workgroup (the bodies of the kernels
// above, separated by workgroup barriers; with one workgroup the final sums are the workgroup's own).
__global__ __launch_bounds__(GB) void grfull_h_small_kernel(View v, double G, double C2, double* ws, double* out) {
    __shared__ TileH t[GB];
    __shared__ double red[GB / 64][3];
    const long long n = v.n;
    double* a[3] = {ws, ws + n, ws + 2 * n};
    double* b[3] = {ws + 3 * n, ws + 4 * n, ws + 5 * n};
    double* part = ws + 6 * n;
    if (threadIdx.x < n) { a[0][threadIdx.x] = v.vx[threadIdx.x]; a[1][threadIdx.x] = v.vy[threadIdx.x]; a[2][threadIdx.x] = v.vz[threadIdx.x]; }
    for (int q = 0; q < 10; q++) {                  // the reference's loop bound (gr_full.c:279)
        __syncthreads();
        grfull_h_sweep_body(v, G, C2, a[0], a[1], a[2], b[0], b[1], b[2], t, 0);
        for (int k = 0; k < 3; k++) { double* tmp = a[k]; a[k] = b[k]; b[k] = tmp; }
    }
    __syncthreads();
    grfull_h_energy_body(v, G, C2, a[0], a[1], a[2], part, t, red, 0);
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = 0. + part[threadIdx.x];
}

__global__ __launch_bounds__(1024) void grfull_h_sum_kernel(const double* partials, long long nb, double* out) {
    __shared__ double red[16][3];
    double q[3] = {0., 0., 0.};
    for (long long k = threadIdx.x; k < nb; k += 1024) for (int c = 0; c < 3; c++) q[c] += partials[3 * k + c];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q[c] += __shfl_down(q[c], off, 64);
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; c++) red[threadIdx.x >> 6][c] = q[c];
    __syncthreads();
    if (threadIdx.x < 3) {
        double ssum = 0.;
        for (int w = 0; w < 16; w++) ssum += red[w][threadIdx.x];
        out[threadIdx.x] = ssum;
    }
}

size_t grfull_hamiltonian_doubles(long long n) { return (size_t)(6 * n + 3 * ((n + GB - 1) / GB) + 8); }

// ws: grfull_hamiltonian_doubles(n); out = ws + (that) - 8: kinetic, Newtonian, post-Newtonian
hipError_t launch_grfull_hamiltonian(const View& v, double G, double c, double* ws, double** out, hipStream_t st) {
    const long long n = v.n, nb = (n + GB - 1) / GB;
    double* a[3] = {ws, ws + n, ws + 2 * n};
    double* b[3] = {ws + 3 * n, ws + 4 * n, ws + 5 * n};
    double* part = ws + 6 * n;
    *out = part + 3 * nb;
    if (n <= GB) {
        hipLaunchKernelGGL(grfull_h_small_kernel, dim3(1), dim3(GB), 0, st, v, G, c * c, ws, *out);
        return hipGetLastError();
    }
    const size_t bytes = (size_t)n * sizeof(double);
    const double* src[3] = {v.vx, v.vy, v.vz};
    for (int k = 0; k < 3; k++) {
        const hipError_t e = hipMemcpyAsync(a[k], src[k], bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    for (int q = 0; q < 10; q++) {                  // the reference's loop bound (gr_full.c:279)
        hipLaunchKernelGGL(grfull_h_sweep_kernel, dim3((unsigned)nb), dim3(GB), 0, st, v, G, c * c, a[0], a[1], a[2], b[0], b[1], b[2]);
        for (int k = 0; k < 3; k++) { double* tmp = a[k]; a[k] = b[k]; b[k] = tmp; }
    }
    hipLaunchKernelGGL(grfull_h_energy_kernel, dim3((unsigned)nb), dim3(GB), 0, st, v, G, c * c, a[0], a[1], a[2], part);
    hipLaunchKernelGGL(grfull_h_sum_kernel, dim3(1), dim3(1024), 0, st, part, nb, *out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launchers
// scratch for the j-chunk partials is owned by device_api.hip and handed in through `part`
struct GrFullPlan { int rowblocks; int nchunks; long long jchunk; int rows_per_thread; };

static GrFullPlan plan(long long n, long long nrows) {
    GrFullPlan p;
    // two rows per thread halve the LDS traffic per pair; worth it once there are enough rows to
    // keep every CU busy with half as many workgroups
    p.rows_per_thread = nrows >= 8192 ? 2 : 1;
    p.rowblocks = (int)((nrows + GB * p.rows_per_thread - 1) / (GB * p.rows_per_thread));
    long long want = (2048 + p.rowblocks - 1) / p.rowblocks;     // aim at >= 2048 workgroups (8 per CU)
    const long long tiles = (n + GB - 1) / GB;
    if (want > tiles) want = tiles;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    const long long tiles_per_chunk = (tiles + want - 1) / want;
    p.jchunk = tiles_per_chunk * GB;
    p.nchunks = (int)((n + p.jchunk - 1) / p.jchunk);
    // A few hundred to a few thousand bodies: whole tiles leave most compute units idle (1000 bodies: 4 x 4 workgroups), and a
    // workgroup takes ~10 us per tile whatever else happens.  There the j range is cut into pieces of 64 or 128 bodies (at most
    // 64 pieces), which the tile loop of the SHORT kernels stops at.
    if (n > GB && p.rows_per_thread == 1 && (long long)p.rowblocks * p.nchunks < 512) {
        for (long long piece = 64; piece < GB; piece *= 2) {
            const long long pieces = (n + piece - 1) / piece;
            if (pieces <= 64) { p.jchunk = piece; p.nchunks = (int)pieces; break; }
        }
    }
    return p;
}

size_t grfull_partial_bytes(long long n, long long nrows) {
    // (four doubles per row and chunk: the widest entry any pass writes)
    const GrFullPlan p = plan(n, nrows);
    return (size_t)p.nchunks * (size_t)nrows * 4 * sizeof(double);
}

static GrFullArgs base_args(const View& v, double G, double c, long long row0, long long row1) {
    GrFullArgs a;
    a.n = v.n; a.row0 = row0; a.row1 = row1; a.jchunk = v.n;
    a.G = G; a.C2 = c * c; a.invC2 = 1. / (c * c);
    a.x = v.x; a.y = v.y; a.z = v.z; a.vx = v.vx; a.vy = v.vy; a.vz = v.vz; a.m = v.m;
    a.pot = nullptr; a.anx = a.any_ = a.anz = a.aox = a.aoy = a.aoz = nullptr; a.part = nullptr; a.stop = nullptr;
    return a;
}

hipError_t launch_grfull_pass(int pass, const View& v, double G, double c, long long row0, long long row1, const double* pot,
                              const double* anx, const double* any_, const double* anz,
                              const double* aox, const double* aoy, const double* aoz,
                              const double* acx, const double* acy, const double* acz,
                              double* o0, double* o1, double* o2, unsigned* flag, double* part, bool exact, hipStream_t st, const unsigned* stop) {
    const long long nrows = row1 - row0;
    if (nrows <= 0) return hipSuccess;
    GrFullArgs a = base_args(v, G, c, row0, row1);
    a.stop = stop;
    a.pot = pot; a.anx = anx; a.any_ = any_; a.anz = anz; a.aox = aox; a.aoy = aoy; a.aoz = aoz;
    if (exact) {
        const dim3 g((unsigned)((nrows + GB - 1) / GB)), b(GB);
        if (pass == PASS_POT) hipLaunchKernelGGL(grfull_exact_kernel<PASS_POT>, g, b, 0, st, a, acx, acy, acz, o0, o1, o2, flag);
        else if (pass == PASS_CONST) hipLaunchKernelGGL(grfull_exact_kernel<PASS_CONST>, g, b, 0, st, a, acx, acy, acz, o0, o1, o2, flag);
        else hipLaunchKernelGGL(grfull_exact_kernel<PASS_SWEEP>, g, b, 0, st, a, acx, acy, acz, o0, o1, o2, flag);
        return hipGetLastError();
    }
    const GrFullPlan p = plan(v.n, nrows);
    a.jchunk = p.jchunk; a.part = part;
    const dim3 g((unsigned)p.rowblocks, (unsigned)p.nchunks), b(GB);
    const dim3 g2((unsigned)((nrows + 255) / 256)), b2(256);
    const bool two = p.rows_per_thread == 2;
    const bool pieces = p.jchunk < GB;
    if (pass == PASS_POT) {
        if (two) hipLaunchKernelGGL((grfull_fast_kernel<PASS_POT, 2>), g, b, 0, st, a);
        else if (pieces) hipLaunchKernelGGL((grfull_fast_kernel<PASS_POT, 1, true>), g, b, 0, st, a);
        else hipLaunchKernelGGL((grfull_fast_kernel<PASS_POT, 1>), g, b, 0, st, a);
        hipLaunchKernelGGL(grfull_combine_kernel<PASS_POT>, g2, b2, 0, st, part, p.nchunks, row0, row1, acx, acy, acz, aox, aoy, aoz, o0, o1, o2, flag, stop, 1);
    } else if (pass == PASS_CONST) {
        if (two) hipLaunchKernelGGL((grfull_fast_kernel<PASS_CONST, 2>), g, b, 0, st, a);
        else if (pieces) hipLaunchKernelGGL((grfull_fast_kernel<PASS_CONST, 1, true>), g, b, 0, st, a);
        else hipLaunchKernelGGL((grfull_fast_kernel<PASS_CONST, 1>), g, b, 0, st, a);
        hipLaunchKernelGGL(grfull_combine_kernel<PASS_CONST>, g2, b2, 0, st, part, p.nchunks, row0, row1, acx, acy, acz, aox, aoy, aoz, o0, o1, o2, flag, stop, 3);
    } else {
        if (two) hipLaunchKernelGGL((grfull_fast_kernel<PASS_SWEEP, 2>), g, b, 0, st, a);
        else if (pieces) hipLaunchKernelGGL((grfull_fast_kernel<PASS_SWEEP, 1, true>), g, b, 0, st, a);
        else hipLaunchKernelGGL((grfull_fast_kernel<PASS_SWEEP, 1>), g, b, 0, st, a);
        hipLaunchKernelGGL(grfull_combine_kernel<PASS_SWEEP>, g2, b2, 0, st, part, p.nchunks, row0, row1, acx, acy, acz, aox, aoy, aoz, o0, o1, o2, flag, stop, 3);
    }
    return hipGetLastError();
}

// The first sweep (a_old = 0, so it needs neither the potential sums nor a_const yet) and the potential sums in ONE pass over
// the pairs: they share the separations and the reciprocal distances.  Leaves the j-chunk partial sums in `part4` (4 per row);
// combine_pot adds up the potential sums, combine_sweep -- once a_const exists -- the sweep's (with the stopping rule's vote).
hipError_t launch_grfull_sweep0_pot(const View& v, double G, double c, long long row0, long long row1,
                                    const double* anx, const double* any_, const double* anz, double* part4, hipStream_t st) {
    const long long nrows = row1 - row0;
    if (nrows <= 0) return hipSuccess;
    GrFullArgs a = base_args(v, G, c, row0, row1);
    a.anx = anx; a.any_ = any_; a.anz = anz;
    const GrFullPlan p = plan(v.n, nrows);
    a.jchunk = p.jchunk; a.part = part4;
    const dim3 g((unsigned)p.rowblocks, (unsigned)p.nchunks), b(GB);
    if (p.rows_per_thread == 2) hipLaunchKernelGGL((grfull_fast_kernel<PASS_SWEEP_POT, 2>), g, b, 0, st, a);
    else if (p.jchunk < GB) hipLaunchKernelGGL((grfull_fast_kernel<PASS_SWEEP_POT, 1, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((grfull_fast_kernel<PASS_SWEEP_POT, 1>), g, b, 0, st, a);
    return hipGetLastError();
}
hipError_t launch_grfull_sweep0_pot_combine(bool sweep, const View& v, long long row0, long long row1, const double* part4,
                                            const double* acx, const double* acy, const double* acz, const double* aox, const double* aoy, const double* aoz,
                                            double* o0, double* o1, double* o2, unsigned* flag, hipStream_t st) {
    const long long nrows = row1 - row0;
    if (nrows <= 0) return hipSuccess;
    const GrFullPlan p = plan(v.n, nrows);
    const dim3 g2((unsigned)((nrows + 255) / 256)), b2(256);
    if (sweep) hipLaunchKernelGGL(grfull_combine_kernel<PASS_SWEEP>, g2, b2, 0, st, part4, p.nchunks, row0, row1, acx, acy, acz, aox, aoy, aoz, o0, o1, o2, flag, (const unsigned*)nullptr, 4);
    else hipLaunchKernelGGL(grfull_combine_kernel<PASS_POT>, g2, b2, 0, st, part4, p.nchunks, row0, row1, acx, acy, acz, aox, aoy, aoz, o0, o1, o2, flag, (const unsigned*)nullptr, 4);
    return hipGetLastError();
}

hipError_t launch_grfull_prepare(const View& v, double G, int gravity_ignore_10, double* anx, double* any_, double* anz,
                                 double* a0x, double* a0y, double* a0z, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    long long nb = (v.n + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(grfull_prepare_kernel, dim3((unsigned)nb), dim3(256), 0, st, v, G, gravity_ignore_10, anx, any_, anz, a0x, a0y, a0z);
    return hipGetLastError();
}

// ---- the exchange of a row-sharded sweep (one process per GPU): every rank sends its rows of the new iterate and its vote in
// ONE message, [x rows | y rows | z rows | vote], `per` rows each (rows behind n are padding and travel as zeros).
__global__ __launch_bounds__(256) void grfull_pack_kernel(const double* ox, const double* oy, const double* oz, long long row0, long long per, long long n,
                                                          unsigned* flag, double* send) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < per) {
        const long long r = row0 + i;
        const bool in = r < n;
        send[i] = in ? ox[r] : 0.; send[per + i] = in ? oy[r] : 0.; send[2 * per + i] = in ? oz[r] : 0.;
    }
    if (i == 0) { send[3 * per] = *flag ? 1. : 0.; *flag = 0u; }        // (the flag is ready for the next sweep)
}
// recv: `world` such messages in rank order.  Puts every rank's rows in their place in the full columns, and turns the votes into
// the stop word of the queued sweeps: set (for good) once no rank's rows fail the stopping rule.  vote_out <- 1 if some rank's do.
// recv == nullptr (all rows on this device): only the vote, taken from `flag`, which is cleared.
__global__ __launch_bounds__(256) void grfull_unpack_kernel(const double* recv, int world, long long per, long long npad, double* ox, double* oy, double* oz,
                                                            unsigned* flag, unsigned* stop, int* vote_out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (recv != nullptr && idx < npad) {
        const long long r = idx / per, i = idx - r * per;
        const double* msg = recv + r * (3 * per + 1);
        ox[idx] = msg[i]; oy[idx] = msg[per + i]; oz[idx] = msg[2 * per + i];
    }
    if (idx == 0) {
        bool any = false;
        if (recv != nullptr) { for (int r = 0; r < world; r++) any = any || recv[(long long)r * (3 * per + 1) + 3 * per] != 0.; }
        else { any = *flag != 0u; *flag = 0u; }
        if (!any) *stop = 1u;
        *vote_out = any ? 1 : 0;
    }
}
hipError_t launch_grfull_pack(const double* ox, const double* oy, const double* oz, long long row0, long long per, long long n, unsigned* flag, double* send, hipStream_t st) {
    const long long nb = per > 0 ? (per + 255) / 256 : 1;
    hipLaunchKernelGGL(grfull_pack_kernel, dim3((unsigned)nb), dim3(256), 0, st, ox, oy, oz, row0, per, n, flag, send);
    return hipGetLastError();
}
hipError_t launch_grfull_unpack(const double* recv, int world, long long per, double* ox, double* oy, double* oz, unsigned* flag, unsigned* stop, int* vote_out, hipStream_t st) {
    const long long npad = recv ? per * world : 1;
    hipLaunchKernelGGL(grfull_unpack_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st, recv, world, per, npad, ox, oy, oz, flag, stop, vote_out);
    return hipGetLastError();
}

hipError_t launch_grfull_vote(unsigned* ctl, hipStream_t st) {
    hipLaunchKernelGGL(grfull_vote_kernel, dim3(1), dim3(64), 0, st, ctl);
    return hipGetLastError();
}

hipError_t launch_grfull_finish_select(const View& v, double* const b0[3], double* const b1[3], const unsigned* ctl, unsigned* status, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    long long nb = (v.n + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(grfull_finish_select_kernel, dim3((unsigned)nb), dim3(256), 0, st, v, b0[0], b0[1], b0[2], b1[0], b1[1], b1[2], ctl, status);
    return hipGetLastError();
}

hipError_t launch_grfull_finish(const View& v, const double* ax, const double* ay, const double* az, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    long long nb = (v.n + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(grfull_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, v, ax, ay, az);
    return hipGetLastError();
}

}  // namespace rebxamd

// operator_kernels.hip -- the operators modify_mass (reference src/modify_mass.c:63-73) and
// modify_orbits_direct (src/modify_orbits_direct.c:81-132 through rebxtools_com_ptm,
// src/rebxtools.c:137-215) on device columns.
//
// Both lean on REBOUND routines in the reference (reb_move_to_com, reb_tools_particle_to_orbit_err,
// reb_tools_orbit_to_particle); their arithmetic is restated here from the published formulas
// (Murray & Dermott eqs. 2.122 and 2.36), as in the CPU checker that replays the reference's
// archives with them.
//
// modify_orbits_direct walks the particles backwards and lets every massive body push on the
// bodies before it, a sequential chain.  With the massive bodies in a prefix of at most 64
// (star, planets) followed by massless ones the chain is short: the massless tail is independent
// (its pushes are zero) and runs in parallel against the unmodified head, then one thread walks
// the head.
//
// With massive bodies anywhere (or a massive prefix longer than 64) the chain is long: body i is
// stepped from the state the bodies behind it have already shifted it to, about a centre of mass
// those shifts have moved.  Unrolled, both are suffix sums of the pushes s_k = massratio_k * diff_k
// of the bodies k > i, so the chain is a fixed-point problem  s = F(suffix sums of s)  whose map F
// is a parallel pass and whose coupling is weak (mass ratio times dt/tau per body): it is iterated
// from s = 0 until the pushes stop moving (two to four rounds in practice), with the sequential
// walk (one thread, running sums) as the fallback should the rounds not contract.
#include "internal.h"
#include "scan_ops.h"
#include <float.h>

namespace rebxamd {

static constexpr int OB = 256;
static constexpr double PI_D = 3.14159265358979323846;

struct PV { double x, y, z, vx, vy, vz, m; };
struct Elements { double d, a, e, inc, Omega, omega, f; };

__device__ __forceinline__ double acos2_dev(double num, double denom, double disambiguator) {
#pragma clang fp contract(off)
    const double cosine = num / denom;
    if (cosine > -1. && cosine < 1.) {
        const double val = acos(cosine);
        return disambiguator < 0. ? -val : val;
    }
    return (cosine <= -1.) ? PI_D : 0.;
}

// orbital elements of p about primary; false when undefined (massless primary, coincident bodies)
__device__ __forceinline__ bool particle_to_orbit_dev(double G, const PV& p, const PV& primary, Elements& o) {
#pragma clang fp contract(off)
    const double TINY = 1.E-308, MIN_INC = 1.e-8;
    if (primary.m <= TINY) return false;
    const double mu = G * (p.m + primary.m);
    const double dx = p.x - primary.x, dy = p.y - primary.y, dz = p.z - primary.z;
    const double dvx = p.vx - primary.vx, dvy = p.vy - primary.vy, dvz = p.vz - primary.vz;
    o.d = sqrt(dx * dx + dy * dy + dz * dz);
    const double vsquared = dvx * dvx + dvy * dvy + dvz * dvz;
    const double vcircsquared = mu / o.d;
    o.a = -mu / (vsquared - 2. * vcircsquared);
    const double hx = (dy * dvz - dz * dvy), hy = (dz * dvx - dx * dvz), hz = (dx * dvy - dy * dvx);
    const double h = sqrt(hx * hx + hy * hy + hz * hz);
    const double vdiffsquared = vsquared - vcircsquared;
    if (o.d <= TINY) return false;
    const double vr = (dx * dvx + dy * dvy + dz * dvz) / o.d;
    const double rvr = o.d * vr;
    const double muinv = 1. / mu;
    const double ex = muinv * (vdiffsquared * dx - rvr * dvx);
    const double ey = muinv * (vdiffsquared * dy - rvr * dvy);
    const double ez = muinv * (vdiffsquared * dz - rvr * dvz);
    o.e = sqrt(ex * ex + ey * ey + ez * ez);
    o.inc = acos2_dev(hz, h, 1.);
    const double nx = -hy, ny = hx;
    const double n = sqrt(nx * nx + ny * ny);
    o.Omega = acos2_dev(nx, n, ny);
    if (o.inc < MIN_INC || o.inc > PI_D - MIN_INC) {
        const double theta = acos2_dev(dx, o.d, dy);
        const double pomega = acos2_dev(ex, o.e, ey);
        if (o.inc < PI_D / 2.) { o.omega = pomega - o.Omega; o.f = theta - pomega; }
        else { o.omega = o.Omega - pomega; o.f = pomega - theta; }
    } else {
        const double wpf = acos2_dev(nx * dx + ny * dy, n * o.d, dz);
        o.omega = acos2_dev(nx * ex + ny * ey, n * o.e, ez);
        o.f = wpf - o.omega;
    }
    return true;
}

__device__ __forceinline__ void orbit_to_particle_dev(double G, const PV& primary, const Elements& o, PV& p) {
#pragma clang fp contract(off)
    const double a = o.a, e = o.e, f = o.f;
    const double r = a * (1 - e * e) / (1 + e * cos(f));
    const double v0 = sqrt(G * (p.m + primary.m) / a / (1. - e * e));
    const double cO = cos(o.Omega), sO = sin(o.Omega), co = cos(o.omega), so = sin(o.omega), cf = cos(f), sf = sin(f), ci = cos(o.inc), si = sin(o.inc);
    p.x = primary.x + r * (cO * (co * cf - so * sf) - sO * (so * cf + co * sf) * ci);
    p.y = primary.y + r * (sO * (co * cf - so * sf) + cO * (so * cf + co * sf) * ci);
    p.z = primary.z + r * (so * cf + co * sf) * si;
    p.vx = primary.vx + v0 * ((e + cf) * (-ci * co * sO - cO * so) - sf * (co * cO - ci * so * sO));
    p.vy = primary.vy + v0 * ((e + cf) * (ci * co * cO - sO * so) - sf * (co * sO + ci * so * cO));
    p.vz = primary.vz + v0 * ((e + cf) * co * si - sf * si * so);
}

struct DirectArgs {
    int coordinates, refindex, has_p;
    double p_param, dt, G;
    const double *tau_omega, *tau_Omega;
};

__device__ __forceinline__ bool set(const double* col, long long i, double& out) {
    if (col == nullptr) return false;
    const double v = col[i];
    if (!is_set(v)) return false;
    out = v;
    return true;
}

// modify_orbits_direct.c:81-119 for body i; returns the new state (p unchanged when the orbit is undefined)
__device__ __forceinline__ PV direct_step(const View& v, const DirectArgs& a, long long i, const PV& p, const PV& primary) {
#pragma clang fp contract(off)
    Elements o;
    if (!particle_to_orbit_dev(a.G, p, primary, o)) return p;
    const double a0 = o.a, e0 = o.e, inc0 = o.inc;
    double t;
    const bool has_te = set(v.te, i, t);
    const double te = t;
    if (set(v.ta, i, t)) o.a += a0 * a.dt / t;
    if (has_te) o.e += e0 * a.dt / te;
    if (set(v.ti, i, t)) o.inc += inc0 * a.dt / t;
    if (set(a.tau_omega, i, t)) o.omega += 2. * PI_D * a.dt / t;
    if (set(a.tau_Omega, i, t)) o.Omega += 2. * PI_D * a.dt / t;
    if (has_te && a.has_p) o.a += 2. * a0 * e0 * e0 * a.p_param * a.dt / te;
    PV out = p;
    orbit_to_particle_dev(a.G, primary, o, out);
    return out;
}

__device__ __forceinline__ PV load_pv(const View& v, long long i) {
    PV p; p.x = v.x[i]; p.y = v.y[i]; p.z = v.z[i]; p.vx = v.vx[i]; p.vy = v.vy[i]; p.vz = v.vz[i]; p.m = v.m[i];
    return p;
}
__device__ __forceinline__ void store_pv(const View& v, long long i, const PV& p) {
    double* x = const_cast<double*>(v.x); double* y = const_cast<double*>(v.y); double* z = const_cast<double*>(v.z);
    double* vx = const_cast<double*>(v.vx); double* vy = const_cast<double*>(v.vy); double* vz = const_cast<double*>(v.vz);
    x[i] = p.x; y[i] = p.y; z[i] = p.z; vx[i] = p.vx; vy[i] = p.vy; vz[i] = p.vz;
}

// centre of mass of the first K bodies, folded pair by pair as reb_get_com does; hc[0..6] = x y z vx vy vz m
__global__ void head_com_kernel(View v, long long K, double* hc) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PV c = {0., 0., 0., 0., 0., 0., 0.};
    for (long long i = 0; i < K; i++) {
        const PV b = load_pv(v, i);
        c.x = c.x * c.m + b.x * b.m; c.y = c.y * c.m + b.y * b.m; c.z = c.z * c.m + b.z * b.m;
        c.vx = c.vx * c.m + b.vx * b.m; c.vy = c.vy * c.m + b.vy * b.m; c.vz = c.vz * c.m + b.vz * b.m;
        c.m += b.m;
        if (c.m > 0.) { c.x /= c.m; c.y /= c.m; c.z /= c.m; c.vx /= c.m; c.vy /= c.m; c.vz /= c.m; }
    }
    hc[0] = c.x; hc[1] = c.y; hc[2] = c.z; hc[3] = c.vx; hc[4] = c.vy; hc[5] = c.vz; hc[6] = c.m;
    for (int k = 7; k < 16; k++) hc[k] = 0.;
}

// the massless tail: independent bodies against the unmodified head
__global__ __launch_bounds__(OB) void direct_tail_kernel(View v, long long K, DirectArgs a, const double* __restrict__ hc) {
    const long long i = K + (long long)blockIdx.x * OB + threadIdx.x;
    if (i >= v.n) return;
    PV primary;
    if (a.coordinates == REBX_COORDINATES_PARTICLE) primary = load_pv(v, a.refindex);
    else { primary.x = hc[0]; primary.y = hc[1]; primary.z = hc[2]; primary.vx = hc[3]; primary.vy = hc[4]; primary.vz = hc[5]; primary.m = hc[6]; }
    const PV p = load_pv(v, i);
    store_pv(v, i, direct_step(v, a, i, p, primary));
}

// the head: one thread walks it backwards with the reference's back-reactions
// (rebxtools.c:163-214); barycentric pushes that also reach the tail are summed into hc[8..13]
__global__ void direct_head_kernel(View v, long long K, DirectArgs a, double* hc) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PV com;
    if (a.coordinates == REBX_COORDINATES_PARTICLE) com = load_pv(v, a.refindex);
    else { com.x = hc[0]; com.y = hc[1]; com.z = hc[2]; com.vx = hc[3]; com.vy = hc[4]; com.vz = hc[5]; com.m = hc[6]; }
    double sh[6] = {0., 0., 0., 0., 0., 0.};
    const int skip = a.coordinates == REBX_COORDINATES_JACOBI ? 0 : (a.coordinates == REBX_COORDINATES_PARTICLE ? a.refindex : -1);
    for (long long i = K - 1; i >= 0; i--) {
        if (i == skip) continue;
        PV p = load_pv(v, i);
        if (a.coordinates == REBX_COORDINATES_JACOBI) {          // reb_get_com_without_particle
            com.x = com.x * com.m - p.x * p.m; com.y = com.y * com.m - p.y * p.m; com.z = com.z * com.m - p.z * p.m;
            com.vx = com.vx * com.m - p.vx * p.m; com.vy = com.vy * com.m - p.vy * p.m; com.vz = com.vz * com.m - p.vz * p.m;
            com.m -= p.m;
            if (com.m > 0.) { com.x /= com.m; com.y /= com.m; com.z /= com.m; com.vx /= com.m; com.vy /= com.m; com.vz /= com.m; }
        }
        const PV mod = direct_step(v, a, i, p, com);
        const double d[6] = {mod.x - p.x, mod.y - p.y, mod.z - p.z, mod.vx - p.vx, mod.vy - p.vy, mod.vz - p.vz};
        store_pv(v, i, mod);
        double* cols[6] = {const_cast<double*>(v.x), const_cast<double*>(v.y), const_cast<double*>(v.z),
                           const_cast<double*>(v.vx), const_cast<double*>(v.vy), const_cast<double*>(v.vz)};
        if (a.coordinates == REBX_COORDINATES_BARYCENTRIC) {
            const double mr = p.m / com.m;
            for (long long j = 0; j < K; j++) for (int c = 0; c < 6; c++) cols[c][j] -= mr * d[c];
            for (int c = 0; c < 6; c++) sh[c] += mr * d[c];
        } else if (a.coordinates == REBX_COORDINATES_JACOBI) {
            const double mr = p.m / (com.m + p.m);
            for (long long j = 0; j < i + 1; j++) for (int c = 0; c < 6; c++) cols[c][j] -= mr * d[c];
        } else {
            const double mr = p.m / (com.m + p.m);
            for (int c = 0; c < 6; c++) { cols[c][i] -= mr * d[c]; cols[c][a.refindex] -= mr * d[c]; }
        }
    }
    for (int c = 0; c < 6; c++) hc[8 + c] = sh[c];
}

// x_j -= s (six columns) for j in [lo, n)
__global__ __launch_bounds__(OB) void shift_kernel(View v, long long lo, const double* __restrict__ s) {
#pragma clang fp contract(off)
    const long long i = lo + (long long)blockIdx.x * OB + threadIdx.x;
    if (i >= v.n) return;
    PV p = load_pv(v, i);
    p.x -= s[0]; p.y -= s[1]; p.z -= s[2]; p.vx -= s[3]; p.vy -= s[4]; p.vz -= s[5];
    store_pv(v, i, p);
}

// ---- modify_mass
__global__ __launch_bounds__(OB) void mass_growth_kernel(double* m, const double* __restrict__ tau_mass, double dt, long long n) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * OB + threadIdx.x;
    if (i >= n) return;
    const double t = tau_mass[i];
    if (is_set(t)) { const double mi = m[i]; m[i] = mi + mi * dt / t; }
}

// per-workgroup sums of (m, m x, m y, m z, m vx, m vy, m vz)
__global__ __launch_bounds__(OB) void com_partial_kernel(View v, double* partials) {
    __shared__ double red[OB / 64][7];
    const long long i = (long long)blockIdx.x * OB + threadIdx.x;
    double q[7] = {0., 0., 0., 0., 0., 0., 0.};
    if (i < v.n) {
        const PV p = load_pv(v, i);
        q[0] = p.m; q[1] = p.m * p.x; q[2] = p.m * p.y; q[3] = p.m * p.z; q[4] = p.m * p.vx; q[5] = p.m * p.vy; q[6] = p.m * p.vz;
    }
#pragma unroll
    for (int c = 0; c < 7; c++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q[c] += __shfl_down(q[c], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) for (int c = 0; c < 7; c++) red[wave][c] = q[c];
    __syncthreads();
    if (threadIdx.x < 7) {
        double s = 0.;
        for (int w = 0; w < OB / 64; w++) s += red[w][threadIdx.x];
        partials[(long long)blockIdx.x * 7 + threadIdx.x] = s;
    }
}

// fixed-order total of the partial sums -> centre of mass in s[0..5] (position, velocity)
__global__ __launch_bounds__(1024) void com_final_kernel(const double* partials, long long nb, double* s) {
    __shared__ double red[16][7];
    double q[7] = {0., 0., 0., 0., 0., 0., 0.};
    for (long long k = threadIdx.x; k < nb; k += 1024)
        for (int c = 0; c < 7; c++) q[c] += partials[k * 7 + c];
#pragma unroll
    for (int c = 0; c < 7; c++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q[c] += __shfl_down(q[c], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) for (int c = 0; c < 7; c++) red[wave][c] = q[c];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[7];
        for (int c = 0; c < 7; c++) { t[c] = 0.; for (int w = 0; w < 16; w++) t[c] += red[w][c]; }
        const double M = t[0];
        for (int c = 0; c < 6; c++) s[c] = M > 0. ? t[c + 1] / M : t[c + 1];
    }
}

long long operator_partial_doubles(long long n) { return 7 * ((n + OB - 1) / OB) + 16; }

hipError_t launch_modify_mass(const View& v, const double* tau_mass, double dt, double* ws, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    const long long nb = (v.n + OB - 1) / OB;
    double* shift = ws + 7 * nb;
    hipLaunchKernelGGL(mass_growth_kernel, dim3((unsigned)nb), dim3(OB), 0, st, const_cast<double*>(v.m), tau_mass, dt, v.n);
    hipLaunchKernelGGL(com_partial_kernel, dim3((unsigned)nb), dim3(OB), 0, st, v, ws);
    hipLaunchKernelGGL(com_final_kernel, dim3(1), dim3(1024), 0, st, ws, nb, shift);
    hipLaunchKernelGGL(shift_kernel, dim3((unsigned)nb), dim3(OB), 0, st, v, 0LL, shift);
    return hipGetLastError();
}

hipError_t launch_modify_orbits_direct(const View& v, long long K, int coordinates, int refindex, double p_param, int has_p,
                                       const double* tau_omega, const double* tau_Omega, double dt, double G, double* hc16, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    DirectArgs a;
    a.coordinates = coordinates; a.refindex = refindex; a.has_p = has_p; a.p_param = p_param; a.dt = dt; a.G = G;
    a.tau_omega = tau_omega; a.tau_Omega = tau_Omega;
    hipLaunchKernelGGL(head_com_kernel, dim3(1), dim3(64), 0, st, v, K, hc16);
    if (v.n > K) {
        const long long nb = (v.n - K + OB - 1) / OB;
        hipLaunchKernelGGL(direct_tail_kernel, dim3((unsigned)nb), dim3(OB), 0, st, v, K, a, hc16);
    }
    hipLaunchKernelGGL(direct_head_kernel, dim3(1), dim3(64), 0, st, v, K, a, hc16);
    if (coordinates == REBX_COORDINATES_BARYCENTRIC && v.n > K) {
        const long long nb = (v.n - K + OB - 1) / OB;
        hipLaunchKernelGGL(shift_kernel, dim3((unsigned)nb), dim3(OB), 0, st, v, K, hc16 + 8);
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------ modify_orbits_direct, massive bodies anywhere
//
// Reference loop (src/rebxtools.c:164-214), i = N-1 .. 0, with the pushes written out:
//   state of body i when its turn comes      p_i - S_i,          S_i = sum_{k>i} s_k
//   Jacobi: reference point of body i        (P_i + T_i) / M_i,  M_i = sum_{k<i} m_k,  P_i = sum_{k<i} m_k p_k,
//           (the centre of mass of all bodies, taken before the loop, with the shifted bodies i..N-1
//           removed one by one: reb_get_com_without_particle)     T_i = sum_{k>=i} m_k S_k
//   barycentric: the centre of mass of all bodies before the loop; PARTICLE: the primary before the loop
//   diff_i = step(p_i - S_i) - (p_i - S_i),  s_i = massratio_i diff_i
//   body i ends at   Jacobi, PARTICLE: (p_i - S_i) + diff_i - s_i      barycentric: p_i + diff_i - sum_k s_k
// Columns of the workspace are stored one after the other (component c of body i at [c * n + i]).
struct ChainArgs {
    DirectArgs a;
    long long n;
    double* s;          // 6 columns: pushes, current iterate
    double* S;          // 6 columns: S_i
    double* T;          // 6 columns: T_i (Jacobi)
    double* P;          // 7 columns: M_i, P_i (Jacobi)
    double* table;      // workgroup totals of the scan in flight (7 per workgroup)
    double* tot;        // [0..6] sum of m, m p over all bodies; [8..13] sum of the pushes; [16..21] scratch totals
    double* delta;      // per workgroup: sum |s_new - s_old| (6), sum massratio * |state| for positions and velocities (2)
    double* conv;       // [0] relative change of the round before
    unsigned* ctl;      // [0] the pushes have stopped moving, [1] rounds run
};

enum { CHAIN_P = 0, CHAIN_S = 1, CHAIN_T = 2 };

template <int WHAT>
__device__ __forceinline__ void chain_elem(const View& v, const ChainArgs& c, long long i, double (&q)[7]) {
#pragma clang fp contract(off)
    if (i >= c.n) {
#pragma unroll
        for (int k = 0; k < 7; k++) q[k] = 0.;
        return;
    }
    if (WHAT == CHAIN_P) {
        const double m = v.m[i];
        q[0] = m; q[1] = m * v.x[i]; q[2] = m * v.y[i]; q[3] = m * v.z[i]; q[4] = m * v.vx[i]; q[5] = m * v.vy[i]; q[6] = m * v.vz[i];
    } else if (WHAT == CHAIN_S) {
#pragma unroll
        for (int k = 0; k < 6; k++) q[k] = c.s[k * c.n + i];
        q[6] = 0.;
    } else {
        const double m = v.m[i];
#pragma unroll
        for (int k = 0; k < 6; k++) q[k] = m * c.S[k * c.n + i];
        q[6] = 0.;
    }
}

template <int WHAT>
__global__ __launch_bounds__(SB) void chain_totals(View v, ChainArgs c, int skip_when_done) {
    __shared__ double lds[SW][7];
    if (skip_when_done && c.ctl[0]) return;
    double q[7];
    chain_elem<WHAT>(v, c, (long long)blockIdx.x * SB + threadIdx.x, q);
    block_total<7>(q, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; k++) c.table[(long long)blockIdx.x * 7 + k] = q[k];
    }
}

// the table of workgroup totals scanned by scan_of_totals; then every workgroup scans its own elements on top of its entry
template <int WHAT>
__global__ __launch_bounds__(SB) void chain_write(View v, ChainArgs c, int skip_when_done) {
    __shared__ double lds[SW][7];
    if (skip_when_done && c.ctl[0]) return;
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    double q[7], own[7], tot[7];
    chain_elem<WHAT>(v, c, i, q);
#pragma unroll
    for (int k = 0; k < 7; k++) own[k] = q[k];
    block_scan<7, WHAT != CHAIN_P>(q, tot, lds);
    if (i >= c.n) return;
    const double* base = c.table + (long long)blockIdx.x * 7;
    if (WHAT == CHAIN_P) {
#pragma unroll
        for (int k = 0; k < 7; k++) c.P[k * c.n + i] = base[k] + (q[k] - own[k]);          // bodies before i
    } else if (WHAT == CHAIN_S) {
#pragma unroll
        for (int k = 0; k < 6; k++) c.S[k * c.n + i] = base[k] + (q[k] - own[k]);          // bodies behind i
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) c.T[k * c.n + i] = base[k] + q[k];                     // body i and the bodies behind it
    }
}

__global__ void chain_begin(ChainArgs c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { c.ctl[0] = 0; c.ctl[1] = 0; c.conv[0] = 0.; }
}

// One round of the map (FINAL = false: new pushes and how far they moved) or the last pass (FINAL = true: the bodies' new states).
template <int COORD, bool FINAL>
__global__ __launch_bounds__(SB) void chain_map(View v, ChainArgs c) {
#pragma clang fp contract(off)
    __shared__ double lds[SW][8];
    if (!FINAL && c.ctl[0]) return;
    const long long i = (long long)blockIdx.x * SB + threadIdx.x, n = c.n;
    double q[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    if (i < n) {
        const PV orig = load_pv(v, i);
        PV cur = orig;
        double Si[6] = {0., 0., 0., 0., 0., 0.};
        if (COORD != REBX_COORDINATES_PARTICLE) {
#pragma unroll
            for (int k = 0; k < 6; k++) Si[k] = c.S[k * n + i];
            cur.x -= Si[0]; cur.y -= Si[1]; cur.z -= Si[2]; cur.vx -= Si[3]; cur.vy -= Si[4]; cur.vz -= Si[5];
        }
        PV com;
        double mr = 0.;
        bool skip = false;
        if (COORD == REBX_COORDINATES_PARTICLE) {
            com = load_pv(v, c.a.refindex);
            skip = i == c.a.refindex;
            mr = orig.m / (com.m + orig.m);
        } else if (COORD == REBX_COORDINATES_BARYCENTRIC) {
            const double M = c.tot[0];
            com.m = M;
            com.x = c.tot[1]; com.y = c.tot[2]; com.z = c.tot[3]; com.vx = c.tot[4]; com.vy = c.tot[5]; com.vz = c.tot[6];
            if (M > 0.) { com.x /= M; com.y /= M; com.z /= M; com.vx /= M; com.vy /= M; com.vz /= M; }
            mr = orig.m / com.m;
        } else {
            const double M = c.P[i];
            com.m = M;
            com.x = c.P[1 * n + i] + c.T[0 * n + i]; com.y = c.P[2 * n + i] + c.T[1 * n + i]; com.z = c.P[3 * n + i] + c.T[2 * n + i];
            com.vx = c.P[4 * n + i] + c.T[3 * n + i]; com.vy = c.P[5 * n + i] + c.T[4 * n + i]; com.vz = c.P[6 * n + i] + c.T[5 * n + i];
            if (M > 0.) { com.x /= M; com.y /= M; com.z /= M; com.vx /= M; com.vy /= M; com.vz /= M; }
            skip = i == 0;
            mr = orig.m / (com.m + orig.m);
        }
        const PV mod = skip ? cur : direct_step(v, c.a, i, cur, com);
        const double d[6] = {mod.x - cur.x, mod.y - cur.y, mod.z - cur.z, mod.vx - cur.vx, mod.vy - cur.vy, mod.vz - cur.vz};
        double sn[6];
#pragma unroll
        for (int k = 0; k < 6; k++) sn[k] = (skip || orig.m == 0.) ? 0. : mr * d[k];
        if (!FINAL) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const double old = c.s[k * n + i];
                q[k] = fabs(sn[k] - old);
                c.s[k * n + i] = sn[k];
            }
            if (!skip && orig.m != 0.) {
                q[6] = fabs(mr) * fmax(fabs(cur.x - com.x), fmax(fabs(cur.y - com.y), fabs(cur.z - com.z)));
                q[7] = fabs(mr) * fmax(fabs(cur.vx - com.vx), fmax(fabs(cur.vy - com.vy), fabs(cur.vz - com.vz)));
            }
        } else {
            PV out = mod;
            if (COORD == REBX_COORDINATES_BARYCENTRIC) {
                // every body takes every push: the ones behind it already sit in `cur`, the rest is the total minus those
                out.x -= c.tot[8] - Si[0]; out.y -= c.tot[9] - Si[1]; out.z -= c.tot[10] - Si[2];
                out.vx -= c.tot[11] - Si[3]; out.vy -= c.tot[12] - Si[4]; out.vz -= c.tot[13] - Si[5];
            } else if (!skip) {
                out.x -= sn[0]; out.y -= sn[1]; out.z -= sn[2]; out.vx -= sn[3]; out.vy -= sn[4]; out.vz -= sn[5];
            }
            if (!(COORD == REBX_COORDINATES_PARTICLE && skip)) store_pv(v, i, out);
        }
    }
    if (!FINAL) {
        block_total<8>(q, lds);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) c.delta[(long long)blockIdx.x * 8 + k] = q[k];
        }
    }
}

// Adds up what the round moved and decides whether the pushes have settled: the change of every component
// within rounding of the pushes themselves (64 ulp of sum massratio |state|), or no longer shrinking at a
// level (1e-10 of that sum) that only the rounding of the orbit conversions explains.
__global__ __launch_bounds__(SOT) void chain_check(ChainArgs c, long long nb) {
    __shared__ double red[SOT / 64][8];
    if (c.ctl[0]) return;
    double q[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    for (long long b = threadIdx.x; b < nb; b += SOT)
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] += c.delta[b * 8 + k];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = wave_total(q[k], lane);
    if (lane == 0) for (int k = 0; k < 8; k++) red[wave][k] = q[k];
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t[8];
    for (int k = 0; k < 8; k++) { t[k] = 0.; for (int w = 0; w < SOT / 64; w++) t[k] += red[w][k]; }
    double rel = 0.;
    bool settled = true;
    for (int k = 0; k < 6; k++) {
        const double size = t[k < 3 ? 6 : 7];
        if (!(t[k] <= 64. * DBL_EPSILON * size)) settled = false;
        const double r = size > 0. ? t[k] / size : (t[k] > 0. ? HUGE_VAL : 0.);
        if (!(r <= rel)) rel = r;           // (a NaN sticks)
    }
    const unsigned rounds = c.ctl[1] + 1u;
    const double before = c.conv[0];
    if (!settled && rounds >= 2u && rel <= 1e-10 && rel > 0.25 * before) settled = true;
    c.conv[0] = rel;
    c.ctl[1] = rounds;
    if (settled) c.ctl[0] = 1u;
}

// The reference's walk, one thread, with the shifts carried as running sums: leaves the pushes in c.s.
template <int COORD>
__global__ void chain_sequential(View v, ChainArgs c) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long n = c.n;
    PV com;
    com.m = c.tot[0];
    com.x = c.tot[1]; com.y = c.tot[2]; com.z = c.tot[3]; com.vx = c.tot[4]; com.vy = c.tot[5]; com.vz = c.tot[6];
    if (com.m > 0.) { com.x /= com.m; com.y /= com.m; com.z /= com.m; com.vx /= com.m; com.vy /= com.m; com.vz /= com.m; }
    double S[6] = {0., 0., 0., 0., 0., 0.};
    for (long long i = n - 1; i >= 0; i--) {
        PV p = load_pv(v, i);
        const double m = p.m;
        p.x -= S[0]; p.y -= S[1]; p.z -= S[2]; p.vx -= S[3]; p.vy -= S[4]; p.vz -= S[5];
        bool skip = false;
        if (COORD == REBX_COORDINATES_JACOBI) {
            skip = i == 0;
            if (!skip) {
                com.x = com.x * com.m - p.x * p.m; com.y = com.y * com.m - p.y * p.m; com.z = com.z * com.m - p.z * p.m;
                com.vx = com.vx * com.m - p.vx * p.m; com.vy = com.vy * com.m - p.vy * p.m; com.vz = com.vz * com.m - p.vz * p.m;
                com.m -= p.m;
                if (com.m > 0.) { com.x /= com.m; com.y /= com.m; com.z /= com.m; com.vx /= com.m; com.vy /= com.m; com.vz /= com.m; }
            }
        }
        double sn[6] = {0., 0., 0., 0., 0., 0.};
        if (!skip && m != 0.) {
            const PV mod = direct_step(v, c.a, i, p, com);
            const double mr = COORD == REBX_COORDINATES_JACOBI ? m / (com.m + m) : m / com.m;
            sn[0] = mr * (mod.x - p.x); sn[1] = mr * (mod.y - p.y); sn[2] = mr * (mod.z - p.z);
            sn[3] = mr * (mod.vx - p.vx); sn[4] = mr * (mod.vy - p.vy); sn[5] = mr * (mod.vz - p.vz);
        }
        for (int k = 0; k < 6; k++) { c.s[k * n + i] = sn[k]; S[k] += sn[k]; }
    }
}

// PARTICLE coordinates: the primary takes every push
__global__ void chain_primary(View v, ChainArgs c) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PV p = load_pv(v, c.a.refindex);
    p.x -= c.tot[8]; p.y -= c.tot[9]; p.z -= c.tot[10]; p.vx -= c.tot[11]; p.vy -= c.tot[12]; p.vz -= c.tot[13];
    store_pv(v, c.a.refindex, p);
}

__global__ __launch_bounds__(SB) void chain_zero(double* s, long long count) {
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    if (i < count) s[i] = 0.;
}

long long direct_chain_doubles(long long n) {
    const long long nb = (n + SB - 1) / SB;
    return 25 * n + 15 * nb + 64;
}

static ChainArgs chain_args(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega, const double* tau_Omega,
                            double dt, double G, double* ws) {
    const long long n = v.n, nb = (n + SB - 1) / SB;
    ChainArgs c;
    c.a.coordinates = coordinates; c.a.refindex = refindex; c.a.has_p = has_p; c.a.p_param = p_param; c.a.dt = dt; c.a.G = G;
    c.a.tau_omega = tau_omega; c.a.tau_Omega = tau_Omega;
    c.n = n;
    c.s = ws; c.S = c.s + 6 * n; c.T = c.S + 6 * n; c.P = c.T + 6 * n;
    c.table = c.P + 7 * n; c.delta = c.table + 7 * nb;
    c.tot = c.delta + 8 * nb; c.conv = c.tot + 32; c.ctl = (unsigned*)(c.conv + 8);
    return c;
}

template <int WHAT>
static void chain_scan(const View& v, const ChainArgs& c, double* totals, bool totals_only, int skip_when_done, hipStream_t st) {
    const long long nb = (c.n + SB - 1) / SB;
    hipLaunchKernelGGL(chain_totals<WHAT>, dim3((unsigned)nb), dim3(SB), 0, st, v, c, skip_when_done);
    // (once the pushes have settled the passes around this one return at once; scanning the stale table again does no harm,
    // the last pass builds table and totals afresh)
    if (WHAT == CHAIN_P) hipLaunchKernelGGL((scan_of_totals<7, false>), dim3(7), dim3(SOT), 0, st, c.table, nb, totals);
    else hipLaunchKernelGGL((scan_of_totals<7, true>), dim3(7), dim3(SOT), 0, st, c.table, nb, totals);
    if (!totals_only) hipLaunchKernelGGL(chain_write<WHAT>, dim3((unsigned)nb), dim3(SB), 0, st, v, c, skip_when_done);
}

template <int COORD>
static void chain_rounds(const View& v, const ChainArgs& c, int rounds, hipStream_t st) {
    const long long nb = (c.n + SB - 1) / SB;
    for (int r = 0; r < rounds; r++) {
        if (COORD != REBX_COORDINATES_PARTICLE && r > 0) {
            chain_scan<CHAIN_S>(v, c, c.tot + 8, false, 1, st);
            if (COORD == REBX_COORDINATES_JACOBI) chain_scan<CHAIN_T>(v, c, c.tot + 16, false, 1, st);
        }
        hipLaunchKernelGGL((chain_map<COORD, false>), dim3((unsigned)nb), dim3(SB), 0, st, v, c);
        hipLaunchKernelGGL(chain_check, dim3(1), dim3(SOT), 0, st, c, nb);
    }
}

// Queues the rounds of the fixed-point iteration (PARTICLE coordinates: the one pass that forms the pushes).
// `ctl_out`: where the host reads {settled, rounds run} from afterwards.
hipError_t launch_direct_chain_rounds(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega,
                                      const double* tau_Omega, double dt, double G, double* ws, int max_rounds, const unsigned** ctl_out, hipStream_t st) {
    const ChainArgs c = chain_args(v, coordinates, refindex, p_param, has_p, tau_omega, tau_Omega, dt, G, ws);
    const long long n = v.n;
    hipLaunchKernelGGL(chain_begin, dim3(1), dim3(64), 0, st, c);
    // s, S and T start at zero (first round: nobody has been pushed yet)
    hipLaunchKernelGGL(chain_zero, dim3((unsigned)((18 * n + SB - 1) / SB)), dim3(SB), 0, st, c.s, 18 * n);
    if (coordinates != REBX_COORDINATES_PARTICLE) chain_scan<CHAIN_P>(v, c, c.tot, coordinates == REBX_COORDINATES_BARYCENTRIC, 0, st);
    if (coordinates == REBX_COORDINATES_JACOBI) chain_rounds<REBX_COORDINATES_JACOBI>(v, c, max_rounds, st);
    else if (coordinates == REBX_COORDINATES_BARYCENTRIC) chain_rounds<REBX_COORDINATES_BARYCENTRIC>(v, c, max_rounds, st);
    else chain_rounds<REBX_COORDINATES_PARTICLE>(v, c, 1, st);
    *ctl_out = c.ctl;
    return hipGetLastError();
}

// The fallback: the pushes from the sequential walk (Jacobi and barycentric coordinates).
hipError_t launch_direct_chain_sequential(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega,
                                          const double* tau_Omega, double dt, double G, double* ws, hipStream_t st) {
    const ChainArgs c = chain_args(v, coordinates, refindex, p_param, has_p, tau_omega, tau_Omega, dt, G, ws);
    if (coordinates == REBX_COORDINATES_JACOBI) hipLaunchKernelGGL(chain_sequential<REBX_COORDINATES_JACOBI>, dim3(1), dim3(64), 0, st, v, c);
    else hipLaunchKernelGGL(chain_sequential<REBX_COORDINATES_BARYCENTRIC>, dim3(1), dim3(64), 0, st, v, c);
    return hipGetLastError();
}

// The last pass: sums of the settled pushes, then every body's new state.
hipError_t launch_direct_chain_finish(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega,
                                      const double* tau_Omega, double dt, double G, double* ws, hipStream_t st) {
    const ChainArgs c = chain_args(v, coordinates, refindex, p_param, has_p, tau_omega, tau_Omega, dt, G, ws);
    const long long nb = (v.n + SB - 1) / SB;
    chain_scan<CHAIN_S>(v, c, c.tot + 8, coordinates == REBX_COORDINATES_PARTICLE, 0, st);
    if (coordinates == REBX_COORDINATES_JACOBI) {
        chain_scan<CHAIN_T>(v, c, c.tot + 16, false, 0, st);
        hipLaunchKernelGGL((chain_map<REBX_COORDINATES_JACOBI, true>), dim3((unsigned)nb), dim3(SB), 0, st, v, c);
    } else if (coordinates == REBX_COORDINATES_BARYCENTRIC) {
        hipLaunchKernelGGL((chain_map<REBX_COORDINATES_BARYCENTRIC, true>), dim3((unsigned)nb), dim3(SB), 0, st, v, c);
    } else {
        hipLaunchKernelGGL((chain_map<REBX_COORDINATES_PARTICLE, true>), dim3((unsigned)nb), dim3(SB), 0, st, v, c);
        hipLaunchKernelGGL(chain_primary, dim3(1), dim3(64), 0, st, v, c);
    }
    return hipGetLastError();
}

}  // namespace rebxamd

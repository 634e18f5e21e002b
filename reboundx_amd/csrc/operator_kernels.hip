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
#include "internal.h"
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

}  // namespace rebxamd

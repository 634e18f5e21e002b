// step_kernels.hip -- elementwise velocity/position updates of the "integrate a force across a
// timestep" schemes (reference src/integrator_euler.c, integrator_rk2.c, integrator_rk4.c,
// integrator_implicit_midpoint.c) and of the kick/drift steppers (src/steppers.c:112-133).
// All are HBM-bound one-pass kernels over column triples.  EXACT keeps the reference's
// expression order without FMA contraction (bitwise equal to a non-FMA CPU build).
#include "internal.h"
#include "effects.h"

namespace rebxamd {

static constexpr int TB = 256;

struct Cols3 { double *x, *y, *z; };
struct CCols3 { const double *x, *y, *z; };

template <bool EXACT>
__device__ __forceinline__ double axpy1(double v, double c, double a) {
    if (EXACT) {
#pragma clang fp contract(off)
        return v + c * a;
    } else {
        return fma(c, a, v);
    }
}

// out = v + coef * a   (out may alias v: kick, drift, euler)
template <bool EXACT>
__global__ __launch_bounds__(TB) void lincomb_kernel(Cols3 out, CCols3 v, CCols3 a, double coef, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    out.x[i] = axpy1<EXACT>(v.x[i], coef, a.x[i]);
    out.y[i] = axpy1<EXACT>(v.y[i], coef, a.y[i]);
    out.z[i] = axpy1<EXACT>(v.z[i], coef, a.z[i]);
}

// integrator_rk2.c:61-66:  v += b1*a1 + b2*a2
template <bool EXACT>
__global__ __launch_bounds__(TB) void rk2_final_kernel(Cols3 v, CCols3 a1, CCols3 a2, double b1, double b2, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    if (EXACT) {
#pragma clang fp contract(off)
        v.x[i] += b1 * a1.x[i] + b2 * a2.x[i];
        v.y[i] += b1 * a1.y[i] + b2 * a2.y[i];
        v.z[i] += b1 * a1.z[i] + b2 * a2.z[i];
    } else {
        v.x[i] += b1 * a1.x[i] + b2 * a2.x[i];
        v.y[i] += b1 * a1.y[i] + b2 * a2.y[i];
        v.z[i] += b1 * a1.z[i] + b2 * a2.z[i];
    }
}

// integrator_rk4.c:74-82:  stage velocity for k4 = v + dt*k3;  k3 += k2;  k2's accelerations reset
template <bool EXACT>
__global__ __launch_bounds__(TB) void rk4_mid_kernel(Cols3 sv, CCols3 v, Cols3 k2a, Cols3 k3a, double dt, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const double k3x = k3a.x[i], k3y = k3a.y[i], k3z = k3a.z[i];
    sv.x[i] = axpy1<EXACT>(v.x[i], dt, k3x);
    sv.y[i] = axpy1<EXACT>(v.y[i], dt, k3y);
    sv.z[i] = axpy1<EXACT>(v.z[i], dt, k3z);
    k3a.x[i] = k3x + k2a.x[i]; k3a.y[i] = k3y + k2a.y[i]; k3a.z[i] = k3z + k2a.z[i];
    k2a.x[i] = 0.; k2a.y[i] = 0.; k2a.z[i] = 0.;
}

// integrator_rk4.c:85-90:  v += dt6*(a1 + a4 + 2.*(k2+k3))
template <bool EXACT>
__global__ __launch_bounds__(TB) void rk4_final_kernel(Cols3 v, CCols3 a1, CCols3 a4, CCols3 k23, double dt6, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    if (EXACT) {
#pragma clang fp contract(off)
        v.x[i] += dt6 * (a1.x[i] + a4.x[i] + 2. * k23.x[i]);
        v.y[i] += dt6 * (a1.y[i] + a4.y[i] + 2. * k23.y[i]);
        v.z[i] += dt6 * (a1.z[i] + a4.z[i] + 2. * k23.z[i]);
    } else {
        v.x[i] += dt6 * (a1.x[i] + a4.x[i] + 2. * k23.x[i]);
        v.y[i] += dt6 * (a1.y[i] + a4.y[i] + 2. * k23.y[i]);
        v.z[i] += dt6 * (a1.z[i] + a4.z[i] + 2. * k23.z[i]);
    }
}

// integrator_implicit_midpoint.c:101-114: new estimate of the final velocities from the
// accelerations at the averaged state; squared change and squared size (compare(), :47-62) as
// per-workgroup partial sums; then the averaged state for the next round (avg_particles, :32-45).
template <bool EXACT>
__global__ __launch_bounds__(TB) void im_update_kernel(Cols3 fin, CCols3 orig, Cols3 avgv, Cols3 avga, double dt, long long n, double* partials) {
    __shared__ double red[2][TB / 64];
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    double d2 = 0., t2 = 0.;
    if (i < n) {
        const double ox = orig.x[i], oy = orig.y[i], oz = orig.z[i];
        const double fx = axpy1<EXACT>(ox, dt, avga.x[i]);
        const double fy = axpy1<EXACT>(oy, dt, avga.y[i]);
        const double fz = axpy1<EXACT>(oz, dt, avga.z[i]);
        const double dx = fx - fin.x[i], dy = fy - fin.y[i], dz = fz - fin.z[i];
        d2 = dx * dx + dy * dy + dz * dz;
        t2 = fx * fx + fy * fy + fz * fz;
        fin.x[i] = fx; fin.y[i] = fy; fin.z[i] = fz;
        avgv.x[i] = 0.5 * (ox + fx); avgv.y[i] = 0.5 * (oy + fy); avgv.z[i] = 0.5 * (oz + fz);
        avga.x[i] = 0.; avga.y[i] = 0.; avga.z[i] = 0.;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_down(d2, off, 64); t2 += __shfl_down(t2, off, 64); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = d2; red[1][wave] = t2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0., b = 0.;
        for (int w = 0; w < TB / 64; w++) { a += red[0][w]; b += red[1][w]; }
        partials[2 * (long long)blockIdx.x] = a;
        partials[2 * (long long)blockIdx.x + 1] = b;
    }
}

// fixed-order sum of the per-workgroup pairs -> out[0], out[1]
__global__ __launch_bounds__(1024) void im_sum_kernel(const double* partials, long long nb, double* out) {
    __shared__ double red[2][16];
    double a = 0., b = 0.;
    for (long long k = threadIdx.x; k < nb; k += 1024) { a += partials[2 * k]; b += partials[2 * k + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0., t = 0.;
        for (int w = 0; w < 16; w++) { s += red[0][w]; t += red[1][w]; }
        out[0] = s; out[1] = t;
    }
}

// ---------------------------------------------------------------- Wisdom-Holman drift and kick
// for one dominant mass at index 0 and massless bodies behind it (the layout of a debris-disk
// run).  REBOUND owns this step in a host run; on the device it is what keeps a multi-step run
// resident.  In Jacobi coordinates massless bodies do not move the interior centre of mass, so
// body i >= 1 is propagated on a Kepler orbit about body 0 with mass parameter G*m0 (universal
// variables, Stumpff series), and body 0 drifts linearly.

// c_n(z) = sum_k (-z)^k / (2k+n)!  for n = 0..3: series after quartering the argument until
// |z| <= 0.1 (nine terms leave < 1e-23), then the duplication formulas.
#define REBXAMD_STUMPFF_BODY                                                                         \
    c3 = 1. / 121645100408832000.; c2 = 1. / 6402373705728000.;                                      \
    c3 = 1. / 355687428096000. - z * c3; c2 = 1. / 20922789888000. - z * c2;                         \
    c3 = 1. / 1307674368000. - z * c3;   c2 = 1. / 87178291200. - z * c2;                            \
    c3 = 1. / 6227020800. - z * c3;      c2 = 1. / 479001600. - z * c2;                              \
    c3 = 1. / 39916800. - z * c3;        c2 = 1. / 3628800. - z * c2;                                \
    c3 = 1. / 362880. - z * c3;          c2 = 1. / 40320. - z * c2;                                  \
    c3 = 1. / 5040. - z * c3;            c2 = 1. / 720. - z * c2;                                    \
    c3 = 1. / 120. - z * c3;             c2 = 1. / 24. - z * c2;                                     \
    c3 = 1. / 6. - z * c3;               c2 = 1. / 2. - z * c2;                                      \
    c1 = 1. - z * c3; c0 = 1. - z * c2;                                                              \
    for (; n > 0; n--) {                                                                             \
        z *= 4.;                                                                                     \
        c3 = 0.25 * (c2 + c0 * c3);                                                                  \
        c2 = 0.5 * c1 * c1;                                                                          \
        c1 = c0 * c1;                                                                                \
        c0 = 2. * c0 * c0 - 1.;                                                                      \
    }

template <bool EXACT>
__device__ __forceinline__ void stumpff_dev(double z, double (&c)[4]) {
    int n = 0;
    while (fabs(z) > 0.1) { z *= 0.25; n++; }
    double c3, c2, c1, c0;
    if (EXACT) {
#pragma clang fp contract(off)
        REBXAMD_STUMPFF_BODY
    } else {
        REBXAMD_STUMPFF_BODY
    }
    c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
}

template <bool EXACT>
__device__ __forceinline__ void gfuncs_dev(double beta, double X, double (&G)[4]) {
    double c[4];
    const double X2 = X * X;
    stumpff_dev<EXACT>(beta * X2, c);
    if (EXACT) {
#pragma clang fp contract(off)
        G[0] = c[0]; G[1] = X * c[1]; G[2] = X2 * c[2]; G[3] = X2 * X * c[3];
    } else {
        G[0] = c[0]; G[1] = X * c[1]; G[2] = X2 * c[2]; G[3] = X2 * X * c[3];
    }
}

// two-body propagation of the relative state (x, v) about mass parameter M over dt.
// Reference-order arithmetic: Newton on the universal Kepler equation from the second-order
// guess, IEEE division and square root, iterated to a fixed point (the checker's algorithm).
__device__ __forceinline__ void kepler_exact(double (&q)[6], double M, double dt) {
#pragma clang fp contract(off)
    const double x0 = q[0], y0 = q[1], z0 = q[2], vx0 = q[3], vy0 = q[4], vz0 = q[5];
    const double r0 = sqrt(x0 * x0 + y0 * y0 + z0 * z0);
    const double r0i = 1. / r0;
    const double v2 = vx0 * vx0 + vy0 * vy0 + vz0 * vz0;
    const double beta = 2. * M * r0i - v2;
    const double eta0 = x0 * vx0 + y0 * vy0 + z0 * vz0;
    const double zeta0 = M - beta * r0;
    double X = dt * r0i * (1. - 0.5 * dt * eta0 * r0i * r0i);
    double G[4];
    double lo = -HUGE_VAL, hi = HUGE_VAL, prev1 = __builtin_nan(""), prev2 = __builtin_nan("");
    for (int it = 0; it < 60; it++) {
        gfuncs_dev<true>(beta, X, G);
        const double s = r0 * X + eta0 * G[2] + zeta0 * G[3] - dt;
        const double sp = r0 + eta0 * G[1] + zeta0 * G[2];
        double Xn = X - s / sp;
        if (s > 0.) hi = X; else lo = X;
        if (!(Xn > lo && Xn < hi)) Xn = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : Xn;
        if (Xn == X || Xn == prev1 || Xn == prev2) { X = Xn; break; }
        prev2 = prev1; prev1 = X; X = Xn;
    }
    gfuncs_dev<true>(beta, X, G);
    const double r = r0 + eta0 * G[1] + zeta0 * G[2];
    const double ri = 1. / r;
    const double fm1 = -M * G[2] * r0i, g = dt - M * G[3], fd = -M * G[1] * r0i * ri, gdm1 = -M * G[2] * ri;
    q[0] += fm1 * x0 + g * vx0; q[1] += fm1 * y0 + g * vy0; q[2] += fm1 * z0 + g * vz0;
    q[3] += fd * x0 + gdm1 * vx0; q[4] += fd * y0 + gdm1 * vy0; q[5] += fd * z0 + gdm1 * vz0;
}

// Default arithmetic: refined hardware reciprocals, third-order starting guess and Halley's
// method (cubic convergence): a step shorter than 1e-6 of X leaves an error below rounding, so
// the usual cost is two evaluations of the G functions plus the final one.  The bracket of the
// root (the residual is monotonic in X: its derivative is r > 0) guards every step.
__device__ __forceinline__ void kepler_fast(double (&q)[6], double M, double dt) {
    const double x0 = q[0], y0 = q[1], z0 = q[2], vx0 = q[3], vy0 = q[4], vz0 = q[5];
    const double r2 = x0 * x0 + y0 * y0 + z0 * z0;
    const double r0i = fast_rsqrt(r2);
    const double r0 = r2 * r0i;
    const double v2 = vx0 * vx0 + vy0 * vy0 + vz0 * vz0;
    const double beta = 2. * M * r0i - v2;
    const double eta0 = x0 * vx0 + y0 * vy0 + z0 * vz0;
    const double zeta0 = M - beta * r0;
    // series inversion of r0 X + eta0 X^2/2 + zeta0 X^3/6 = dt
    const double u = dt * r0i, a = 0.5 * eta0 * r0i, b = zeta0 * r0i * (1. / 6.);
    double X = u * (1. - u * (a - u * (2. * a * a - b)));
    double G[4];
    double lo = -HUGE_VAL, hi = HUGE_VAL, prev = __builtin_nan("");
    for (int it = 0; it < 60; it++) {
        gfuncs_dev<false>(beta, X, G);
        const double s = r0 * X + eta0 * G[2] + zeta0 * G[3] - dt;
        const double sp = r0 + eta0 * G[1] + zeta0 * G[2];
        const double spp = eta0 * G[0] + zeta0 * G[1];
        if (s > 0.) hi = X; else lo = X;
        // Halley: dX = -s / (s' - s s'' / (2 s'))
        const double spi = fast_rcp(sp);
        const double dX = -s * fast_rcp(sp - 0.5 * s * spp * spi);
        double Xn = X + dX;
        const bool inside = Xn > lo && Xn < hi;
        if (!inside) Xn = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : X - s * spi;
        const bool done = (inside && fabs(dX) <= 1e-6 * fabs(Xn)) || Xn == X || Xn == prev;
        prev = X; X = Xn;
        if (done) break;
    }
    gfuncs_dev<false>(beta, X, G);
    const double r = r0 + eta0 * G[1] + zeta0 * G[2];
    const double ri = fast_rcp(r);
    const double fm1 = -M * G[2] * r0i, g = dt - M * G[3], fd = -M * G[1] * r0i * ri, gdm1 = -M * G[2] * ri;
    q[0] += fm1 * x0 + g * vx0; q[1] += fm1 * y0 + g * vy0; q[2] += fm1 * z0 + g * vz0;
    q[3] += fd * x0 + gdm1 * vx0; q[4] += fd * y0 + gdm1 * vy0; q[5] += fd * z0 + gdm1 * vz0;
}

template <bool EXACT>
__device__ __forceinline__ void kepler_dev(double (&q)[6], double M, double dt) {
    if (EXACT) kepler_exact(q, M, dt); else kepler_fast(q, M, dt);
}

// star[0..5] <- state of body 0 before the drift; body 0 drifts linearly; star[6..8] <- its new
// position.  With accelerations given: star[9..11] <- a_0, and a_0 is cleared for the next force pass.
__global__ void wh_star_drift_kernel(Cols3 x, CCols3 v, Cols3 a, double h, double* star) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double x0 = x.x[0], y0 = x.y[0], z0 = x.z[0], vx = v.x[0], vy = v.y[0], vz = v.z[0];
    star[0] = x0; star[1] = y0; star[2] = z0; star[3] = vx; star[4] = vy; star[5] = vz;
    const double nx = x0 + h * vx, ny = y0 + h * vy, nz = z0 + h * vz;
    x.x[0] = nx; x.y[0] = ny; x.z[0] = nz;
    star[6] = nx; star[7] = ny; star[8] = nz;
    if (a.x != nullptr) {
        star[9] = a.x[0]; star[10] = a.y[0]; star[11] = a.z[0];
        a.x[0] = 0.; a.y[0] = 0.; a.z[0] = 0.;
    }
}

// KICK: v_i += dtk*(a_i - a_0) first (Jacobi kick with massless bodies; a_i is cleared afterwards so
// that the next force pass accumulates onto zero), then the Kepler drift over h.
template <bool EXACT, bool KICK>
__global__ __launch_bounds__(TB) void wh_drift_kernel(Cols3 x, Cols3 v, Cols3 a, double mu, double h, double dtk, long long n, const double* __restrict__ star) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x + 1;
    if (i >= n) return;
    double q[6];
    {
#pragma clang fp contract(off)
        double vx = v.x[i], vy = v.y[i], vz = v.z[i];
        if (KICK) {
            const double ax = a.x[i], ay = a.y[i], az = a.z[i];
            a.x[i] = 0.; a.y[i] = 0.; a.z[i] = 0.;
            if (EXACT) {
                vx += dtk * (ax - star[9]); vy += dtk * (ay - star[10]); vz += dtk * (az - star[11]);
            } else {
                vx = fma(dtk, ax - star[9], vx); vy = fma(dtk, ay - star[10], vy); vz = fma(dtk, az - star[11], vz);
            }
        }
        q[0] = x.x[i] - star[0]; q[1] = x.y[i] - star[1]; q[2] = x.z[i] - star[2];
        q[3] = vx - star[3]; q[4] = vy - star[4]; q[5] = vz - star[5];
    }
    kepler_dev<EXACT>(q, mu, h);
    {
#pragma clang fp contract(off)
        x.x[i] = q[0] + star[6]; x.y[i] = q[1] + star[7]; x.z[i] = q[2] + star[8];
        v.x[i] = q[3] + star[3]; v.y[i] = q[4] + star[4]; v.z[i] = q[5] + star[5];
    }
}

static inline dim3 grid_for(long long n) { return dim3((unsigned)((n + TB - 1) / TB)); }

hipError_t launch_lincomb(double* const out[3], const double* const v[3], const double* const a[3], double coef, long long n, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 o = {out[0], out[1], out[2]};
    const CCols3 cv = {v[0], v[1], v[2]}, ca = {a[0], a[1], a[2]};
    if (exact) hipLaunchKernelGGL(lincomb_kernel<true>, grid_for(n), dim3(TB), 0, st, o, cv, ca, coef, n);
    else hipLaunchKernelGGL(lincomb_kernel<false>, grid_for(n), dim3(TB), 0, st, o, cv, ca, coef, n);
    return hipGetLastError();
}

hipError_t launch_rk2_final(double* const v[3], const double* const a1[3], const double* const a2[3], double b1, double b2, long long n, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 o = {v[0], v[1], v[2]};
    const CCols3 c1 = {a1[0], a1[1], a1[2]}, c2 = {a2[0], a2[1], a2[2]};
    if (exact) hipLaunchKernelGGL(rk2_final_kernel<true>, grid_for(n), dim3(TB), 0, st, o, c1, c2, b1, b2, n);
    else hipLaunchKernelGGL(rk2_final_kernel<false>, grid_for(n), dim3(TB), 0, st, o, c1, c2, b1, b2, n);
    return hipGetLastError();
}

hipError_t launch_rk4_mid(double* const sv[3], const double* const v[3], double* const k2a[3], double* const k3a[3], double dt, long long n, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 s = {sv[0], sv[1], sv[2]}, k2 = {k2a[0], k2a[1], k2a[2]}, k3 = {k3a[0], k3a[1], k3a[2]};
    const CCols3 cv = {v[0], v[1], v[2]};
    if (exact) hipLaunchKernelGGL(rk4_mid_kernel<true>, grid_for(n), dim3(TB), 0, st, s, cv, k2, k3, dt, n);
    else hipLaunchKernelGGL(rk4_mid_kernel<false>, grid_for(n), dim3(TB), 0, st, s, cv, k2, k3, dt, n);
    return hipGetLastError();
}

hipError_t launch_rk4_final(double* const v[3], const double* const a1[3], const double* const a4[3], const double* const k23[3], double dt6, long long n, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 o = {v[0], v[1], v[2]};
    const CCols3 c1 = {a1[0], a1[1], a1[2]}, c4 = {a4[0], a4[1], a4[2]}, c23 = {k23[0], k23[1], k23[2]};
    if (exact) hipLaunchKernelGGL(rk4_final_kernel<true>, grid_for(n), dim3(TB), 0, st, o, c1, c4, c23, dt6, n);
    else hipLaunchKernelGGL(rk4_final_kernel<false>, grid_for(n), dim3(TB), 0, st, o, c1, c4, c23, dt6, n);
    return hipGetLastError();
}

long long im_partial_count(long long n) { return (n + TB - 1) / TB; }

hipError_t launch_im_update(double* const fin[3], const double* const orig[3], double* const avgv[3], double* const avga[3], double dt,
                            long long n, double* partials, double* sums, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 f = {fin[0], fin[1], fin[2]}, av = {avgv[0], avgv[1], avgv[2]}, aa = {avga[0], avga[1], avga[2]};
    const CCols3 o = {orig[0], orig[1], orig[2]};
    if (exact) hipLaunchKernelGGL(im_update_kernel<true>, grid_for(n), dim3(TB), 0, st, f, o, av, aa, dt, n, partials);
    else hipLaunchKernelGGL(im_update_kernel<false>, grid_for(n), dim3(TB), 0, st, f, o, av, aa, dt, n, partials);
    hipLaunchKernelGGL(im_sum_kernel, dim3(1), dim3(1024), 0, st, partials, im_partial_count(n), sums);
    return hipGetLastError();
}

// a == nullptr: drift only.  Otherwise kick by dtk with the accelerations in `a` (which are cleared), then drift.
hipError_t launch_wh_drift(double* const x[3], double* const v[3], double* const a[3], double mu, double h, double dtk, long long n,
                           double* star, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 cx = {x[0], x[1], x[2]}, cv = {v[0], v[1], v[2]};
    const Cols3 ca = a ? Cols3{a[0], a[1], a[2]} : Cols3{nullptr, nullptr, nullptr};
    const CCols3 ccv = {v[0], v[1], v[2]};
    hipLaunchKernelGGL(wh_star_drift_kernel, dim3(1), dim3(64), 0, st, cx, ccv, ca, h, star);
    if (n > 1) {
        if (a) {
            if (exact) hipLaunchKernelGGL((wh_drift_kernel<true, true>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, mu, h, dtk, n, star);
            else hipLaunchKernelGGL((wh_drift_kernel<false, true>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, mu, h, dtk, n, star);
        } else {
            if (exact) hipLaunchKernelGGL((wh_drift_kernel<true, false>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, mu, h, dtk, n, star);
            else hipLaunchKernelGGL((wh_drift_kernel<false, false>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, mu, h, dtk, n, star);
        }
    }
    return hipGetLastError();
}

}  // namespace rebxamd

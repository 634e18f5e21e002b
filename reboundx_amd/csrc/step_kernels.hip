// step_kernels.hip -- elementwise velocity/position updates of the "integrate a force across a
// timestep" schemes (reference src/integrator_euler.c, integrator_rk2.c, integrator_rk4.c,
// integrator_implicit_midpoint.c) and of the kick/drift steppers (src/steppers.c:112-133).
// All are HBM-bound one-pass kernels over column triples.  EXACT keeps the reference's
// expression order without FMA contraction (bitwise equal to a non-FMA CPU build).
#include "internal.h"
#include "effects.h"
#include "kepler.h"
#include "wh_head.h"
#include "updates.h"

namespace rebxamd {

static constexpr int TB = 256;

struct Cols3 { double *x, *y, *z; };
struct CCols3 { const double *x, *y, *z; };
// columns a pass reads once and nobody reads again soon: nontemporal, so that they do not push the
// columns being updated out of the caches
__device__ __forceinline__ double once(const double* col, long long i) { return __builtin_nontemporal_load(col + i); }

// out = v + coef * a   (out may alias v: kick, drift, euler)
template <bool EXACT>
__global__ __launch_bounds__(TB) void lincomb_kernel(Cols3 out, CCols3 v, CCols3 a, double coef, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    out.x[i] = axpy1<EXACT>(v.x[i], coef, once(a.x, i));
    out.y[i] = axpy1<EXACT>(v.y[i], coef, once(a.y, i));
    out.z[i] = axpy1<EXACT>(v.z[i], coef, once(a.z, i));
}

// integrator_rk2.c:61-66:  v += b1*a1 + b2*a2
template <bool EXACT>
__global__ __launch_bounds__(TB) void rk2_final_kernel(Cols3 v, CCols3 a1, CCols3 a2, double b1, double b2, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    v.x[i] = rk2_combine<EXACT>(v.x[i], b1, once(a1.x, i), b2, once(a2.x, i));
    v.y[i] = rk2_combine<EXACT>(v.y[i], b1, once(a1.y, i), b2, once(a2.y, i));
    v.z[i] = rk2_combine<EXACT>(v.z[i], b1, once(a1.z, i), b2, once(a2.z, i));
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
    k3a.x[i] = k3x + once(k2a.x, i); k3a.y[i] = k3y + once(k2a.y, i); k3a.z[i] = k3z + once(k2a.z, i);
    k2a.x[i] = 0.; k2a.y[i] = 0.; k2a.z[i] = 0.;
}

// integrator_rk4.c:85-90:  v += dt6*(a1 + a4 + 2.*(k2+k3))
template <bool EXACT>
__global__ __launch_bounds__(TB) void rk4_final_kernel(Cols3 v, CCols3 a1, CCols3 a4, CCols3 k23, double dt6, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    v.x[i] = rk4_combine<EXACT>(v.x[i], dt6, once(a1.x, i), once(a4.x, i), once(k23.x, i));
    v.y[i] = rk4_combine<EXACT>(v.y[i], dt6, once(a1.y, i), once(a4.y, i), once(k23.y, i));
    v.z[i] = rk4_combine<EXACT>(v.z[i], dt6, once(a1.z, i), once(a4.z, i), once(k23.z, i));
}

// integrator_implicit_midpoint.c:101-114: new estimate of the final velocities from the
// accelerations at the averaged state; squared change and squared size (compare(), :47-62) as
// per-workgroup partial sums; then the averaged state for the next round (avg_particles, :32-45).
template <bool EXACT>
__global__ __launch_bounds__(TB) void im_update_kernel(Cols3 fin, CCols3 orig, Cols3 avgv, Cols3 avga, double dt, long long n, double* partials,
                                                       const unsigned* stop) {
    __shared__ double red[2][TB / 64];
    if (stop != nullptr && *stop != 0u) return;        // rounds queued past convergence do nothing
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    double d2 = 0., t2 = 0.;
    if (i < n) {
        const double ox = once(orig.x, i), oy = once(orig.y, i), oz = once(orig.z, i);
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

// Start of a carried implicit-midpoint update in one pass: the velocities at the start of the step
// are kept aside and the accelerations cleared (integrate_force.c:46).
__global__ __launch_bounds__(TB) void im_begin_kernel(Cols3 keep, CCols3 v, Cols3 a, long long n) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    __builtin_nontemporal_store(v.x[i], keep.x + i); __builtin_nontemporal_store(v.y[i], keep.y + i); __builtin_nontemporal_store(v.z[i], keep.z + i);
    __builtin_nontemporal_store(0., a.x + i); __builtin_nontemporal_store(0., a.y + i); __builtin_nontemporal_store(0., a.z + i);
}

// fixed-order sum of the per-workgroup pairs -> out[0], out[1].  With a stop word (rounds queued
// ahead of the host): the round is counted in stop[1] and the reference's test (squared change over
// squared size below machine epsilon squared, integrator_implicit_midpoint.c:47-62) sets stop[0],
// after which the kernels of the rounds still in the queue return at once.
__global__ __launch_bounds__(1024) void im_sum_kernel(const double* partials, long long nb, double* out, unsigned* stop) {
    __shared__ double red[2][16];
    if (stop != nullptr && stop[0] != 0u) return;
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
        if (stop != nullptr) {
            stop[1] += 1u;
            if (s / t < 2.220446049250313e-16 * 2.220446049250313e-16) stop[0] = 1u;
        }
    }
}

// ---------------------------------------------------------------- Wisdom-Holman drift and kick
// for one dominant mass at index 0 and massless bodies behind it (the layout of a debris-disk
// run); the propagation itself is in kepler.h.

// star[0..5] <- state of body 0 before the drift; body 0 drifts linearly; star[6..8] <- its new
// position; star[12] <- G*m_0.  With accelerations given: star[9..11] <- a_0, and a_0 is cleared
// for the next force pass.
__global__ void wh_star_drift_kernel(Cols3 x, CCols3 v, Cols3 a, const double* m, double G, double h, double* star) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    star[12] = G * m[0];
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
__global__ __launch_bounds__(TB) void wh_drift_kernel(Cols3 x, Cols3 v, Cols3 a, double h, double dtk, long long n, const double* __restrict__ star,
                                                      long long first) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x + first;      // (the bodies before `first` are the head)
    if (i >= n) return;
    double s[6] = {x.x[i], x.y[i], x.z[i], v.x[i], v.y[i], v.z[i]};
    double ax = 0., ay = 0., az = 0.;
    if (KICK) {
        ax = a.x[i]; ay = a.y[i]; az = a.z[i];
        a.x[i] = 0.; a.y[i] = 0.; a.z[i] = 0.;
    }
    wh_kick_drift<EXACT, KICK>(s, ax, ay, az, h, dtk, star);
    x.x[i] = s[0]; x.y[i] = s[1]; x.z[i] = s[2];
    v.x[i] = s[3]; v.y[i] = s[4]; v.z[i] = s[5];
}

// ---------------------------------------------------------------- Wisdom-Holman steps of a short massive head (wh_head.h)
template <bool EXACT>
__global__ __launch_bounds__(64) void wh_head_step_kernel(View v, int K, double G, double dtk, double h, double* jac, double* rec) {
    __shared__ WhHeadShared sh;
    wh_head_step_body<EXACT>(v, K, G, dtk, h, jac, rec, sh, threadIdx.x);
}

hipError_t launch_wh_head_step(const View& v, long long K, double G, double dtk, double h, double* jac, double* rec, bool exact, hipStream_t st) {
    if (K < 1 || K > REBXAMD_MAX_HEAD || K > v.n) return hipErrorInvalidValue;
    if (exact) hipLaunchKernelGGL(wh_head_step_kernel<true>, dim3(1), dim3(64), 0, st, v, (int)K, G, dtk, h, jac, rec);
    else hipLaunchKernelGGL(wh_head_step_kernel<false>, dim3(1), dim3(64), 0, st, v, (int)K, G, dtk, h, jac, rec);
    return hipGetLastError();
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
                            long long n, double* partials, double* sums, bool exact, hipStream_t st, const unsigned* stop) {
    if (n <= 0) return hipSuccess;
    const Cols3 f = {fin[0], fin[1], fin[2]}, av = {avgv[0], avgv[1], avgv[2]}, aa = {avga[0], avga[1], avga[2]};
    const CCols3 o = {orig[0], orig[1], orig[2]};
    if (exact) hipLaunchKernelGGL(im_update_kernel<true>, grid_for(n), dim3(TB), 0, st, f, o, av, aa, dt, n, partials, stop);
    else hipLaunchKernelGGL(im_update_kernel<false>, grid_for(n), dim3(TB), 0, st, f, o, av, aa, dt, n, partials, stop);
    if (sums != nullptr) hipLaunchKernelGGL(im_sum_kernel, dim3(1), dim3(1024), 0, st, partials, im_partial_count(n), sums, (unsigned*)nullptr);
    return hipGetLastError();
}

// sums[0..1] <- fixed-order sums of nb (squared change, squared size) pairs
hipError_t launch_im_begin(double* const keep[3], const double* const v[3], double* const a[3], long long n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 k = {keep[0], keep[1], keep[2]}, acc = {a[0], a[1], a[2]};
    const CCols3 cv = {v[0], v[1], v[2]};
    hipLaunchKernelGGL(im_begin_kernel, grid_for(n), dim3(TB), 0, st, k, cv, acc, n);
    return hipGetLastError();
}
hipError_t launch_im_sum(const double* partials, long long nb, double* sums, hipStream_t st, unsigned* stop) {
    hipLaunchKernelGGL(im_sum_kernel, dim3(1), dim3(1024), 0, st, partials, nb, sums, stop);
    return hipGetLastError();
}

hipError_t launch_wh_star(double* const x[3], double* const v[3], const double* m, double G, double h, double* star, hipStream_t st) {
    const Cols3 cx = {x[0], x[1], x[2]};
    const Cols3 ca = {nullptr, nullptr, nullptr};
    const CCols3 ccv = {v[0], v[1], v[2]};
    hipLaunchKernelGGL(wh_star_drift_kernel, dim3(1), dim3(64), 0, st, cx, ccv, ca, m, G, h, star);
    return hipGetLastError();
}

// body 0 back to the position its record holds (star[0..2]: where it was before the drift the record describes)
__global__ void wh_star_restore_kernel(Cols3 x, const double* __restrict__ star) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    x.x[0] = star[0]; x.y[0] = star[1]; x.z[0] = star[2];
}
hipError_t launch_wh_star_restore(double* const x[3], const double* star, hipStream_t st) {
    const Cols3 cx = {x[0], x[1], x[2]};
    hipLaunchKernelGGL(wh_star_restore_kernel, dim3(1), dim3(64), 0, st, cx, star);
    return hipGetLastError();
}

// a == nullptr: drift only.  Otherwise kick by dtk with the accelerations in `a` (which are cleared), then drift.
hipError_t launch_wh_drift(double* const x[3], double* const v[3], double* const a[3], const double* m, double G, double h, double dtk, long long n,
                           double* star, bool exact, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const Cols3 cx = {x[0], x[1], x[2]}, cv = {v[0], v[1], v[2]};
    const Cols3 ca = a ? Cols3{a[0], a[1], a[2]} : Cols3{nullptr, nullptr, nullptr};
    const CCols3 ccv = {v[0], v[1], v[2]};
    hipLaunchKernelGGL(wh_star_drift_kernel, dim3(1), dim3(64), 0, st, cx, ccv, ca, m, G, h, star);
    if (n > 1) {
        if (a) {
            if (exact) hipLaunchKernelGGL((wh_drift_kernel<true, true>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, h, dtk, n, star, 1LL);
            else hipLaunchKernelGGL((wh_drift_kernel<false, true>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, h, dtk, n, star, 1LL);
        } else {
            if (exact) hipLaunchKernelGGL((wh_drift_kernel<true, false>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, h, dtk, n, star, 1LL);
            else hipLaunchKernelGGL((wh_drift_kernel<false, false>), grid_for(n - 1), dim3(TB), 0, st, cx, cv, ca, h, dtk, n, star, 1LL);
        }
    }
    return hipGetLastError();
}

// Kepler drift over h of the bodies [first, n) about the frame that `rec` records (no kick): the closing half drift of the
// massless bodies behind a head of several massive ones.
hipError_t launch_wh_drift_about(double* const x[3], double* const v[3], long long first, long long n, double h, const double* rec, bool exact, hipStream_t st) {
    if (n <= first) return hipSuccess;
    const Cols3 cx = {x[0], x[1], x[2]}, cv = {v[0], v[1], v[2]}, ca = {nullptr, nullptr, nullptr};
    if (exact) hipLaunchKernelGGL((wh_drift_kernel<true, false>), grid_for(n - first), dim3(TB), 0, st, cx, cv, ca, h, 0., n, rec, first);
    else hipLaunchKernelGGL((wh_drift_kernel<false, false>), grid_for(n - first), dim3(TB), 0, st, cx, cv, ca, h, 0., n, rec, first);
    return hipGetLastError();
}

template <bool EXACT>
__global__ __launch_bounds__(64) void wh_head_close_kernel(View v, int K, double G, double h, const double* pre, double* jac, const double* rec_last,
                                                           double* rec_close) {
    __shared__ WhHeadShared sh;
    wh_head_close_body<EXACT>(v, K, G, h, pre, jac, rec_last, rec_close, sh, threadIdx.x);
}
hipError_t launch_wh_head_close(const View& v, long long K, double G, double h, const double* pre, double* jac, const double* rec_last, double* rec_close,
                                bool exact, hipStream_t st) {
    if (K < 2 || K > REBXAMD_MAX_HEAD || K > v.n) return hipErrorInvalidValue;
    if (exact) hipLaunchKernelGGL(wh_head_close_kernel<true>, dim3(1), dim3(64), 0, st, v, (int)K, G, h, pre, jac, rec_last, rec_close);
    else hipLaunchKernelGGL(wh_head_close_kernel<false>, dim3(1), dim3(64), 0, st, v, (int)K, G, h, pre, jac, rec_last, rec_close);
    return hipGetLastError();
}

}  // namespace rebxamd

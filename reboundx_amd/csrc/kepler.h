// kepler.h -- two-body propagation in universal variables (device code), shared by the
// Wisdom-Holman drift kernel and by the streaming pass that carries a whole step.
#pragma once
#include "effects.h"

namespace rebxamd {

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
        // seven terms: the first one left out is below |z|^7/16! = 5e-21 for |z| <= 0.1
        c3 = 1. / 1307674368000.;            c2 = 1. / 87178291200.;
        c3 = 1. / 6227020800. - z * c3;      c2 = 1. / 479001600. - z * c2;
        c3 = 1. / 39916800. - z * c3;        c2 = 1. / 3628800. - z * c2;
        c3 = 1. / 362880. - z * c3;          c2 = 1. / 40320. - z * c2;
        c3 = 1. / 5040. - z * c3;            c2 = 1. / 720. - z * c2;
        c3 = 1. / 120. - z * c3;             c2 = 1. / 24. - z * c2;
        c3 = 1. / 6. - z * c3;               c2 = 1. / 2. - z * c2;
        c1 = 1. - z * c3; c0 = 1. - z * c2;
        for (; n > 0; n--) {
            z *= 4.;
            c3 = 0.25 * (c2 + c0 * c3);
            c2 = 0.5 * c1 * c1;
            c1 = c0 * c1;
            c0 = 2. * c0 * c0 - 1.;
        }
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
// method (cubic convergence): a step shorter than 4e-6 of X leaves an error below rounding, so
// the usual cost is two evaluations of the G functions.  The bracket of the root (the residual is
// monotonic in X: its derivative is r > 0) guards every step.  The G functions at the accepted X come
// from those at the last iterate by their Taylor series to second order in the step (G0' = -beta G1,
// G1' = G0, G2' = G1, G3' = G2; the cubic term is below (4e-6)^3 of the value), not from a third evaluation.
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
    bool fresh = false;                     // G holds the functions at X
    for (int it = 0; it < 60; it++) {
        gfuncs_dev<false>(beta, X, G);
        const double s = r0 * X + eta0 * G[2] + zeta0 * G[3] - dt;
        const double sp = r0 + eta0 * G[1] + zeta0 * G[2];
        const double spp = eta0 * G[0] + zeta0 * G[1];
        if (s > 0.) hi = X; else lo = X;
        // Halley: dX = -s / (s' - s s'' / (2 s')) = -s s' / (s'^2 - s s''/2); the step is a correction, so one
        // Newton refinement of the hardware reciprocal (relative error ~2e-16 of the step) is enough
        const double den = fma(sp, sp, -0.5 * s * spp);
        double y = __builtin_amdgcn_rcp(den);
        y = fma(y, fma(-den, y, 1.0), y);
        const double dX = -(s * sp) * y;
        double Xn = X + dX;
        const bool inside = Xn > lo && Xn < hi;
        if (!inside) Xn = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : X - s * fast_rcp(sp);
        const bool small = inside && fabs(dX) <= 4e-6 * fabs(Xn);
        if (small) {
            const double d = Xn - X, hd = 0.5 * d;
            const double g1 = fma(d, fma(-beta * hd, G[1], G[0]), G[1]);
            const double g2 = fma(d, fma(hd, G[0], G[1]), G[2]);
            const double g3 = fma(d, fma(hd, G[1], G[2]), G[3]);
            G[1] = g1; G[2] = g2; G[3] = g3;
            X = Xn; fresh = true;
            break;
        }
        if (Xn == X) { fresh = true; break; }
        if (Xn == prev) { X = Xn; break; }
        prev = X; X = Xn;
    }
    if (!fresh) gfuncs_dev<false>(beta, X, G);
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

// Kick and drift of one massless body of a Wisdom-Holman step about body 0.  `star` holds body
// 0's state before its drift [0..5], its position after it [6..8], its acceleration [9..11] and
// its mass parameter G*m_0 [12].
// KICK: v += dtk*(a - a_0) first (Jacobi kick of a massless body), then the Kepler drift over h
// relative to body 0.
// v += dtk*(a - a_0): the Jacobi kick of a massless body (a_0 = star[9..11])
template <bool EXACT>
__device__ __forceinline__ void wh_kick(double& vx, double& vy, double& vz, double ax, double ay, double az, double dtk, const double* __restrict__ star) {
#pragma clang fp contract(off)
    if (EXACT) {
        vx += dtk * (ax - star[9]); vy += dtk * (ay - star[10]); vz += dtk * (az - star[11]);
    } else {
        vx = fma(dtk, ax - star[9], vx); vy = fma(dtk, ay - star[10], vy); vz = fma(dtk, az - star[11], vz);
    }
}

template <bool EXACT, bool KICK>
__device__ __forceinline__ void wh_kick_drift(double (&s)[6], double ax, double ay, double az, double h, double dtk,
                                              const double* __restrict__ star) {
    double q[6];
    {
#pragma clang fp contract(off)
        double vx = s[3], vy = s[4], vz = s[5];
        if (KICK) wh_kick<EXACT>(vx, vy, vz, ax, ay, az, dtk, star);
        q[0] = s[0] - star[0]; q[1] = s[1] - star[1]; q[2] = s[2] - star[2];
        q[3] = vx - star[3]; q[4] = vy - star[4]; q[5] = vz - star[5];
    }
    kepler_dev<EXACT>(q, star[12], h);
    {
#pragma clang fp contract(off)
        s[0] = q[0] + star[6]; s[1] = q[1] + star[7]; s[2] = q[2] + star[8];
        s[3] = q[3] + star[3]; s[4] = q[4] + star[4]; s[5] = q[5] + star[5];
    }
}

}  // namespace rebxamd

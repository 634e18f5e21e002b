// effects.h -- per-particle arithmetic of every streaming force, as device functions.
//
// Each force exists twice:
//   *_exact : the reference's expressions in the reference's order with FMA contraction
//             switched off, so that (with IEEE divide/sqrt) a particle's update carries the
//             same bits the reference C code produces;
//   *_fast  : the same mathematics with one reciprocal square root per (particle, source),
//             reciprocals of loop invariants hoisted and contraction allowed.  Deviation from
//             *_exact is a few ulp of the largest term of each component.
//
// All functions return the update for the particle itself in `d` (caller does acc += d) and,
// where the reference also pushes on the source particle, the update for the source in `br`
// (caller does acc[src] += br, i.e. br already carries the sign).
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

namespace rebxamd {

struct V3 { double x, y, z; };

// dynamic state of one particle plus its parameter columns (the unset marker = not set)
struct Part {
    double x, y, z, vx, vy, vz, m;
    double beta, Rt, k1, ta, te, ti;
};

// snapshot of a source / reference body
// (Rt, k1: the source's own R_tides and k1 for tides, 0 when not set; inv_m = 1/m)
typedef SrcRec Src;

// Separation of a particle from a source, shared by every fast-mode force that has the
// same source: d = p - S, ri = 1/|d|.  `src` remembers which source it was computed for.
struct Geo { double dx, dy, dz, ri; int src; };

// Reciprocal square root and reciprocal for the fast mode: the hardware estimate
// (v_rsq_f64 / v_rcp_f64, good to about 2^-23) refined by one third-order step, so the
// error term is the cube of the estimate's (~2^-69) and the result is rounded once or
// twice.  No range scaling: arguments here are squared distances and 1 - O(v^2/c^2), far
// from the subnormal and overflow ranges; a zero distance gives NaN, where the reference
// gives inf/NaN as well.
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);             // 1 - x y^2
    const double p = fma(0.375, e, 0.5);                // 1/2 + 3/8 e
    return fma(y * e, p, y);                            // y (1 + e/2 + 3/8 e^2)
}
__device__ __forceinline__ double fast_rcp(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);                   // 1 - x y
    return fma(y, fma(e, e, e), y);                     // y (1 + e + e^2)
}

// ---------------------------------------------------------------- radiation_forces
// reference src/radiation_forces.c:69-98
__device__ __forceinline__ void rad_exact(const Part& p, const Src& S, double G, double c, V3& d) {
#pragma clang fp contract(off)
    const double mu = G * S.m;
    const double dx = p.x - S.x, dy = p.y - S.y, dz = p.z - S.z;
    const double dr = sqrt(dx * dx + dy * dy + dz * dz);
    const double dvx = p.vx - S.vx, dvy = p.vy - S.vy, dvz = p.vz - S.vz;
    const double rdot = (dx * dvx + dy * dvy + dz * dvz) / dr;
    const double a_rad = p.beta * mu / (dr * dr);
    d.x = a_rad * ((1. - rdot / c) * dx / dr - dvx / c);
    d.y = a_rad * ((1. - rdot / c) * dy / dr - dvy / c);
    d.z = a_rad * ((1. - rdot / c) * dz / dr - dvz / c);
}
__device__ __forceinline__ void rad_fast(const Part& p, const Src& S, const Geo& g, double G, double invc, V3& d) {
    const double dvx = p.vx - S.vx, dvy = p.vy - S.vy, dvz = p.vz - S.vz;
    const double ri = g.ri;
    const double rdot = (g.dx * dvx + g.dy * dvy + g.dz * dvz) * ri;
    const double a_rad = p.beta * (G * S.m) * (ri * ri);
    const double k = (1. - rdot * invc) * ri;
    d.x = a_rad * (k * g.dx - dvx * invc);
    d.y = a_rad * (k * g.dy - dvy * invc);
    d.z = a_rad * (k * g.dz - dvz * invc);
}

// ---------------------------------------------------------------- gravitational_harmonics
// reference src/gravitational_harmonics.c:61-85
__device__ __forceinline__ void j2_exact(const Part& p, const Src& S, double G, double J2, double R_eq, V3& d, V3& br) {
#pragma clang fp contract(off)
    const double dx = p.x - S.x, dy = p.y - S.y, dz = p.z - S.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double r = sqrt(r2);
    const double costheta2 = dz * dz / r2;
    const double prefac = 3. * J2 * R_eq * R_eq / r2 / r2 / r / 2.;
    const double fac = 5. * costheta2 - 1.;
    d.x = G * S.m * prefac * fac * dx;
    d.y = G * S.m * prefac * fac * dy;
    d.z = G * S.m * prefac * (fac - 2.) * dz;
    br.x = -(G * p.m * prefac * fac * dx);
    br.y = -(G * p.m * prefac * fac * dy);
    br.z = -(G * p.m * prefac * (fac - 2.) * dz);
}
__device__ __forceinline__ void j2_fast(const Part& p, const Src& S, const Geo& g, double G, double J2, double R_eq, V3& d, V3& br) {
    const double ri = g.ri, ri2 = ri * ri;
    const double costheta2 = g.dz * g.dz * ri2;
    const double prefac = (1.5 * J2 * R_eq * R_eq) * (ri2 * ri2 * ri);
    const double fac = 5. * costheta2 - 1.;
    const double fx = prefac * fac * g.dx, fy = prefac * fac * g.dy, fz = prefac * (fac - 2.) * g.dz;
    const double gs = G * S.m, gp = -(G * p.m);
    d.x = gs * fx; d.y = gs * fy; d.z = gs * fz;
    br.x = gp * fx; br.y = gp * fy; br.z = gp * fz;
}
// reference src/gravitational_harmonics.c:99-123
__device__ __forceinline__ void j4_exact(const Part& p, const Src& S, double G, double J4, double R_eq, V3& d, V3& br) {
#pragma clang fp contract(off)
    const double dx = p.x - S.x, dy = p.y - S.y, dz = p.z - S.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double r = sqrt(r2);
    const double costheta2 = dz * dz / r2;
    const double prefac = 5. * J4 * R_eq * R_eq * R_eq * R_eq / r2 / r2 / r2 / r / 8.;
    const double fac = 63. * costheta2 * costheta2 - 42. * costheta2 + 3.;
    d.x = G * S.m * prefac * fac * dx;
    d.y = G * S.m * prefac * fac * dy;
    d.z = G * S.m * prefac * (fac + 12. - 28. * costheta2) * dz;
    br.x = -(G * p.m * prefac * fac * dx);
    br.y = -(G * p.m * prefac * fac * dy);
    br.z = -(G * p.m * prefac * (fac + 12. - 28. * costheta2) * dz);
}
__device__ __forceinline__ void j4_fast(const Part& p, const Src& S, const Geo& g, double G, double J4, double R_eq, V3& d, V3& br) {
    const double ri = g.ri, ri2 = ri * ri;
    const double costheta2 = g.dz * g.dz * ri2;
    const double Re2 = R_eq * R_eq;
    const double prefac = (0.625 * J4 * Re2 * Re2) * (ri2 * ri2 * ri2 * ri);
    const double fac = 63. * costheta2 * costheta2 - 42. * costheta2 + 3.;
    const double fx = prefac * fac * g.dx, fy = prefac * fac * g.dy, fz = prefac * (fac + 12. - 28. * costheta2) * g.dz;
    const double gs = G * S.m, gp = -(G * p.m);
    d.x = gs * fx; d.y = gs * fy; d.z = gs * fz;
    br.x = gp * fx; br.y = gp * fy; br.z = gp * fz;
}

// ---------------------------------------------------------------- gr_potential
// reference src/gr_potential.c:69-87 ; prefac1 = 6 (G m0)^2 / c^2
__device__ __forceinline__ void grpot_exact(const Part& p, const Src& S, double prefac1, V3& d, V3& br) {
#pragma clang fp contract(off)
    const double dx = p.x - S.x, dy = p.y - S.y, dz = p.z - S.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double prefac = prefac1 / (r2 * r2);
    d.x = -(prefac * dx); d.y = -(prefac * dy); d.z = -(prefac * dz);
    br.x = p.m / S.m * prefac * dx;
    br.y = p.m / S.m * prefac * dy;
    br.z = p.m / S.m * prefac * dz;
}
__device__ __forceinline__ void grpot_fast(const Part& p, const Src& S, const Geo& g, double prefac1, V3& d, V3& br) {
    const double ri2 = g.ri * g.ri;
    const double prefac = prefac1 * (ri2 * ri2);
    const double fx = prefac * g.dx, fy = prefac * g.dy, fz = prefac * g.dz;
    const double w = p.m * S.inv_m;
    d.x = -fx; d.y = -fy; d.z = -fz;
    br.x = w * fx; br.y = w * fy; br.z = w * fz;
}

// ---------------------------------------------------------------- central_force
// reference src/central_force.c:63-83
__device__ __forceinline__ void central_exact(const Part& p, const Src& S, double A, double gamma, V3& d, V3& br) {
#pragma clang fp contract(off)
    const double dx = p.x - S.x, dy = p.y - S.y, dz = p.z - S.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double prefac = A * pow(r2, (gamma - 1.) / 2.);
    d.x = prefac * dx; d.y = prefac * dy; d.z = prefac * dz;
    br.x = -(p.m / S.m * prefac * dx);
    br.y = -(p.m / S.m * prefac * dy);
    br.z = -(p.m / S.m * prefac * dz);
}
__device__ __forceinline__ void central_fast(const Part& p, const Src& S, const Geo& g, double A, double gamma, V3& d, V3& br) {
    const double r2 = g.dx * g.dx + g.dy * g.dy + g.dz * g.dz;
    const double prefac = A * pow(r2, 0.5 * (gamma - 1.));
    d.x = prefac * g.dx; d.y = prefac * g.dy; d.z = prefac * g.dz;
    const double w = -(p.m * S.inv_m);
    br.x = w * d.x; br.y = w * d.y; br.z = w * d.z;
}

// ---------------------------------------------------------------- tides_precession
// reference src/tides_precession.c:65-116.  Returns false when the body is skipped
// (mass ratio below DBL_MIN: massless bodies feel and exert nothing).
__device__ __forceinline__ bool tides_exact(const Part& p, const Src& S, double G, V3& d, V3& br) {
#pragma clang fp contract(off)
    const double m0 = S.m;
    const double fac0 = S.k1 * S.Rt * S.Rt * S.Rt * S.Rt * S.Rt;
    const double mratio = p.m / m0;
    if (mratio < DBL_MIN) return false;
    double fac = 0.;
    fac += fac0 * mratio;
    const double Rp = is_set(p.Rt) ? p.Rt : 0.;
    const double k1p = is_set(p.k1) ? p.k1 : 0.;
    fac += k1p * Rp * Rp * Rp * Rp * Rp / mratio;
    const double dx = p.x - S.x, dy = p.y - S.y, dz = p.z - S.z;
    const double dr2 = dx * dx + dy * dy + dz * dz;
    const double prefac = -3 * G * (m0 + p.m) / (dr2 * dr2 * dr2 * dr2) * fac;
    d.x = prefac * dx; d.y = prefac * dy; d.z = prefac * dz;
    br.x = -(mratio * prefac * dx);
    br.y = -(mratio * prefac * dy);
    br.z = -(mratio * prefac * dz);
    return true;
}
__device__ __forceinline__ bool tides_fast(const Part& p, const Src& S, const Geo& g, double G, V3& d, V3& br) {
    const double m0 = S.m;
    const double R02 = S.Rt * S.Rt;
    const double fac0 = S.k1 * R02 * R02 * S.Rt;
    const double mratio = p.m * S.inv_m;
    if (mratio < DBL_MIN) return false;
    const double Rp = is_set(p.Rt) ? p.Rt : 0.;
    const double k1p = is_set(p.k1) ? p.k1 : 0.;
    const double Rp2 = Rp * Rp;
    const double fac = fac0 * mratio + (k1p * Rp2 * Rp2 * Rp) * (m0 * fast_rcp(p.m));
    const double ri2 = g.ri * g.ri, ri4 = ri2 * ri2;
    const double prefac = -3. * G * (m0 + p.m) * fac * (ri4 * ri4);
    d.x = prefac * g.dx; d.y = prefac * g.dy; d.z = prefac * g.dz;
    br.x = -(mratio * d.x); br.y = -(mratio * d.y); br.z = -(mratio * d.z);
    return true;
}

// ---------------------------------------------------------------- modify_orbits_forces
// reference src/modify_orbits_forces.c:77-118 ; force on p relative to `ref` (a particle or a
// centre of mass).  Unset timescales act as +infinity.
__device__ __forceinline__ void mof_exact(const Part& p, const Src& ref, V3& a) {
#pragma clang fp contract(off)
    const double tau_a = is_set(p.ta) ? p.ta : INFINITY;
    const double tau_e = is_set(p.te) ? p.te : INFINITY;
    const double tau_inc = is_set(p.ti) ? p.ti : INFINITY;
    const double dvx = p.vx - ref.vx, dvy = p.vy - ref.vy, dvz = p.vz - ref.vz;
    const double dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    a.x = dvx / (2. * tau_a);
    a.y = dvy / (2. * tau_a);
    a.z = dvz / (2. * tau_a);
    if (tau_e < INFINITY || tau_inc < INFINITY) {
        const double vdotr = dx * dvx + dy * dvy + dz * dvz;
        const double prefac = 2 * vdotr / r2 / tau_e;
        a.x += prefac * dx;
        a.y += prefac * dy;
        a.z += prefac * dz + 2. * dvz / tau_inc;
    }
}
// Reaction of a body's force (one component) on the bodies before it in a prefix-sum force, given
// the share f of the force that falls back on them; rounded the same way wherever it is formed.
__device__ __forceinline__ double reaction(double f, double a) {
#pragma clang fp contract(off)
    return f * a;
}
// a / b from y = fast_rcp(b): the quotient estimate is corrected with its exact residual, which
// leaves it within an ulp of the IEEE quotient (nearly always equal to it) at a third of the cost
// of the hardware division sequence; y is shared by the quotients over the same b.
__device__ __forceinline__ double div_refined(double a, double b, double y) {
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

// Default arithmetic: the reference's seven quotients, each formed by div_refined (four
// reciprocals shared among them) instead of IEEE division sequences.  The terms along dv and
// along d can cancel, so each quotient has to be good to the last bit or so: plain
// reciprocal-multiply forms cost 1e-12 of the update after cancellation.  An unset timescale
// acts as +infinity (its term is exactly zero).
__device__ __forceinline__ void mof_fast(const Part& p, const Src& ref, V3& a) {
#pragma clang fp contract(off)
    // (no contraction here either: after cancellation between the terms, a differently rounded
    // product shows at 1e-12 of the update even when it is the more accurate one)
    const double tau_a = is_set(p.ta) ? p.ta : INFINITY;
    const double tau_e = is_set(p.te) ? p.te : INFINITY;
    const double tau_inc = is_set(p.ti) ? p.ti : INFINITY;
    const double dvx = p.vx - ref.vx, dvy = p.vy - ref.vy, dvz = p.vz - ref.vz;
    const double dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    a.x = a.y = a.z = 0.;
    if (fabs(tau_a) != INFINITY) {             // (true for a NaN value too: it has to come out as NaN)
        const double ta2 = 2. * tau_a, ya = fast_rcp(ta2);
        a.x = div_refined(dvx, ta2, ya); a.y = div_refined(dvy, ta2, ya); a.z = div_refined(dvz, ta2, ya);
    }
    if (tau_e < INFINITY || tau_inc < INFINITY) {
        const double vdotr = dx * dvx + dy * dvy + dz * dvz;
        double prefac = 0., zi = 0.;
        if (fabs(tau_e) != INFINITY) prefac = div_refined(div_refined(2 * vdotr, r2, fast_rcp(r2)), tau_e, fast_rcp(tau_e));
        if (fabs(tau_inc) != INFINITY) zi = div_refined(2. * dvz, tau_inc, fast_rcp(tau_inc));
        a.x += prefac * dx;
        a.y += prefac * dy;
        a.z += prefac * dz + zi;
    }
}

// ---------------------------------------------------------------- gr (1PN, single dominant mass)
// Jacobi-coordinate state of one body: position, velocity, Newtonian acceleration.
struct Jac { double x, y, z, vx, vy, vz, ax, ay, az; };

// reference src/gr.c:110-156.  out = post-Newtonian acceleration in Jacobi coordinates;
// returns the number of inner iterations used.
__device__ __forceinline__ int gr_term_exact(const Jac& p, double mu, double C2, int max_iterations, V3& out) {
#pragma clang fp contract(off)
    double vix = p.vx, viy = p.vy, viz = p.vz;
    double vi2 = vix * vix + viy * viy + viz * viz;
    const double ri = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
    int q = 0;
    double A = (0.5 * vi2 + 3. * mu / ri) / C2;
    for (q = 0; q < max_iterations; q++) {
        const double ox = vix, oy = viy, oz = viz;
        vix = p.vx / (1. - A);
        viy = p.vy / (1. - A);
        viz = p.vz / (1. - A);
        vi2 = vix * vix + viy * viy + viz * viz;
        A = (0.5 * vi2 + 3. * mu / ri) / C2;
        const double dvx = vix - ox, dvy = viy - oy, dvz = viz - oz;
        if ((dvx * dvx + dvy * dvy + dvz * dvz) / vi2 < DBL_EPSILON * DBL_EPSILON) break;
    }
    const double B = (mu / ri - 1.5 * vi2) * mu / (ri * ri * ri) / C2;
    const double rdotrdot = p.x * p.vx + p.y * p.vy + p.z * p.vz;
    const double vdx = p.ax + B * p.x, vdy = p.ay + B * p.y, vdz = p.az + B * p.z;
    const double vdotvdot = vix * vdx + viy * vdy + viz * vdz;
    const double D = (vdotvdot - 3. * mu / (ri * ri * ri) * rdotrdot) / C2;
    out.x = B * (1. - A) * p.x - A * p.ax - D * vix;
    out.y = B * (1. - A) * p.y - A * p.ay - D * viy;
    out.z = B * (1. - A) * p.z - A * p.az - D * viz;
    return q;
}
// Fast mode.  The reference iterates the vector v~ = v/(1-A(v~)); since v~ stays parallel to v
// that is the scalar recurrence s <- 1/(1 - (s^2 v^2/2 + 3 mu/r)/c^2) with v~ = s v, and its
// stopping rule |dv~|^2/v~^2 < eps^2 reads |ds| < eps s.  Same iterates, a third of the work.
__device__ __forceinline__ int gr_term_fast(const Jac& p, double rinv, double mu, double invC2, int max_iterations, V3& out) {
    const double v2 = p.vx * p.vx + p.vy * p.vy + p.vz * p.vz;
    const double mur = mu * rinv;
    const double k3 = 3. * mur * invC2, kh = 0.5 * v2 * invC2;
    int q = 0;
    double s = 1.;
    double A = kh + k3;
    for (q = 0; q < max_iterations; q++) {
        const double sn = fast_rcp(1. - A);
        A = fma(kh, sn * sn, k3);
        const double ds = sn - s;
        s = sn;
        if (fabs(ds) < DBL_EPSILON * sn) break;
    }
    const double vix = p.vx * s, viy = p.vy * s, viz = p.vz * s;
    const double vi2 = v2 * (s * s);
    const double mur3 = mur * rinv * rinv;         // mu / r^3
    const double B = (mur - 1.5 * vi2) * mur3 * invC2;
    const double rdotrdot = p.x * p.vx + p.y * p.vy + p.z * p.vz;
    const double vdx = p.ax + B * p.x, vdy = p.ay + B * p.y, vdz = p.az + B * p.z;
    const double vdotvdot = vix * vdx + viy * vdy + viz * vdz;
    const double D = (vdotvdot - 3. * mur3 * rdotrdot) * invC2;
    const double BA = B * (1. - A);
    out.x = BA * p.x - A * p.ax - D * vix;
    out.y = BA * p.y - A * p.ay - D * viy;
    out.z = BA * p.z - A * p.az - D * viz;
    return q;
}

// Newtonian pull of massive body u on body t, in the orientation the reference's pair loop
// (src/gr.c:86-103) gives it: u<t -> d = p_u - p_t and += ; u>t -> d = p_t - p_u and -= .
__device__ __forceinline__ void newton_pair_exact(double tx, double ty, double tz, double ux, double uy, double uz, double um,
                                                  bool u_before_t, double G, double& ax, double& ay, double& az) {
#pragma clang fp contract(off)
    if (u_before_t) {
        const double dx = ux - tx, dy = uy - ty, dz = uz - tz;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double r = sqrt(r2);
        const double prefac = G / (r2 * r);
        ax += prefac * um * dx; ay += prefac * um * dy; az += prefac * um * dz;
    } else {
        const double dx = tx - ux, dy = ty - uy, dz = tz - uz;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double r = sqrt(r2);
        const double prefac = G / (r2 * r);
        ax -= prefac * um * dx; ay -= prefac * um * dy; az -= prefac * um * dz;
    }
}
__device__ __forceinline__ void geo_compute(const Part& p, const Src& S, int src, Geo& g) {
    g.dx = p.x - S.x; g.dy = p.y - S.y; g.dz = p.z - S.z;
    g.ri = fast_rsqrt(g.dx * g.dx + g.dy * g.dy + g.dz * g.dz);
    g.src = src;
}

__device__ __forceinline__ void newton_pair_fast(double tx, double ty, double tz, double ux, double uy, double uz, double um,
                                                 double G, double& ax, double& ay, double& az) {
    const double dx = ux - tx, dy = uy - ty, dz = uz - tz;
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double ri = fast_rsqrt(r2);
    const double w = (G * um) * (ri * ri * ri);
    ax += w * dx; ay += w * dy; az += w * dz;
}

}  // namespace rebxamd

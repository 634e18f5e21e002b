/* rebx_whfast.c -- a Wisdom-Holman integrator in Jacobi coordinates for the checker.
 * TEST INFRASTRUCTURE ONLY; the product never links or calls it.
 *
 * Why it exists: the reference's test suite holds trajectory archives written by REBOUND's
 * WHFast (reboundx/test/binaries/<force>.sa) and replays them (test_machine_independent.py).
 * REBOUND itself is a third-party dependency that is not part of the reference tree, so the
 * replay is done with this restatement of the published algorithm (Rein & Tamayo 2015,
 * MNRAS 452, 376: democratic-free Jacobi splitting, drift-kick-drift, universal-variable
 * Kepler solver) with REBOUND's defaults: no symplectic correctors, coordinates recomputed and
 * particles synchronised every step, the star-planet1 pair left out of the direct gravity sum
 * and handled by the Kepler step.  It is validated by the archives themselves: the force-free
 * archive is reproduced to roundoff (tests/test_oracle.py).
 *
 * One step of length dt:
 *   Jacobi <- inertial (positions, velocities)
 *   Kepler drift dt/2 of every Jacobi body i >= 1 about mass G*eta_i, eta_i = m_0 + ... + m_i;
 *   centre of mass drifts linearly
 *   inertial <- Jacobi
 *   accelerations = direct gravity without the (0,1) pair + additional forces
 *   Jacobi accelerations; kick dt: v_i += dt*(a_i + [i >= 2] G*eta_i x_i/|x_i|^3)
 *   Kepler drift dt/2, centre of mass drift, inertial <- Jacobi
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "rebx_oracle.h"

typedef struct { double x, y, z, vx, vy, vz, ax, ay, az, m; } jbody;

/* ---- Stumpff functions c0..c3 of z (c_n = sum_k (-z)^k / (2k+n)!): series after quartering
 * the argument, then the duplication formulas. */
static void stumpff(double z, double c[4]) {
    int n = 0;
    while (fabs(z) > 0.1) { z *= 0.25; n++; }
    double c3 = 0., c2 = 0.;
    /* Horner, terms up to z^8: |z| <= 0.1 makes the next term < 1e-23 relative */
    {
        const double a3[] = {1. / 6., 1. / 120., 1. / 5040., 1. / 362880., 1. / 39916800., 1. / 6227020800., 1. / 1307674368000.,
                             1. / 355687428096000., 1. / 121645100408832000.};
        const double a2[] = {1. / 2., 1. / 24., 1. / 720., 1. / 40320., 1. / 3628800., 1. / 479001600., 1. / 87178291200.,
                             1. / 20922789888000., 1. / 6402373705728000.};
        for (int k = 8; k >= 0; k--) { c3 = a3[k] - z * c3; c2 = a2[k] - z * c2; }
    }
    double c1 = 1. - z * c3, c0 = 1. - z * c2;
    for (; n > 0; n--) {
        z *= 4.;
        c3 = 0.25 * (c2 + c0 * c3);
        c2 = 0.5 * c1 * c1;
        c1 = c0 * c1;
        c0 = 2. * c0 * c0 - 1.;
    }
    c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
}

/* G_n(beta, X) = X^n c_n(beta X^2), n = 0..3 */
static void gfuncs(double beta, double X, double G[4]) {
    double c[4];
    const double X2 = X * X;
    stumpff(beta * X2, c);
    G[0] = c[0]; G[1] = X * c[1]; G[2] = X2 * c[2]; G[3] = X2 * X * c[3];
}

/* Two-body propagation of (x, v) about a fixed mass parameter M over dt. */
static void kepler(jbody* p, double M, double dt) {
    const double x0 = p->x, y0 = p->y, z0 = p->z, vx0 = p->vx, vy0 = p->vy, vz0 = p->vz;
    const double r0 = sqrt(x0 * x0 + y0 * y0 + z0 * z0);
    const double r0i = 1. / r0;
    const double v2 = vx0 * vx0 + vy0 * vy0 + vz0 * vz0;
    const double beta = 2. * M * r0i - v2;
    const double eta0 = x0 * vx0 + y0 * vy0 + z0 * vz0;
    const double zeta0 = M - beta * r0;
    double G[4];
    /* universal Kepler equation: r0 X + eta0 G2 + zeta0 G3 = dt.  Newton from the
     * second-order guess; bracket-and-bisect if Newton wanders (never for |dt| << period) */
    double X = dt * r0i * (1. - 0.5 * dt * eta0 * r0i * r0i);
    double lo = -HUGE_VAL, hi = HUGE_VAL;
    double prev1 = NAN, prev2 = NAN;
    int converged = 0;
    for (int it = 0; it < 60; it++) {
        gfuncs(beta, X, G);
        const double s = r0 * X + eta0 * G[2] + zeta0 * G[3] - dt;
        const double sp = r0 + eta0 * G[1] + zeta0 * G[2];          /* = r > 0 */
        if (s > 0.) hi = X; else lo = X;
        double Xn = X - s / sp;
        if (!(Xn > lo && Xn < hi)) Xn = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : Xn;
        if (Xn == X || Xn == prev1 || Xn == prev2) { X = Xn; converged = 1; break; }   /* fixed point or 2-cycle at roundoff */
        prev2 = prev1; prev1 = X; X = Xn;
    }
    (void)converged;
    gfuncs(beta, X, G);
    const double r = r0 + eta0 * G[1] + zeta0 * G[2];
    const double ri = 1. / r;
    const double fm1 = -M * G[2] * r0i;          /* f - 1 */
    const double g = dt - M * G[3];
    const double fd = -M * G[1] * r0i * ri;
    const double gdm1 = -M * G[2] * ri;          /* gdot - 1 */
    p->x += fm1 * x0 + g * vx0;  p->y += fm1 * y0 + g * vy0;  p->z += fm1 * z0 + g * vz0;
    p->vx += fd * x0 + gdm1 * vx0; p->vy += fd * y0 + gdm1 * vy0; p->vz += fd * z0 + gdm1 * vz0;
}

/* ---- Jacobi <-> inertial (same recurrences as the force path's helpers) */
static void to_jacobi_posvel(const jbody* ps, jbody* pj, int N) {
    double eta = ps[0].m;
    double sx = eta * ps[0].x, sy = eta * ps[0].y, sz = eta * ps[0].z;
    double svx = eta * ps[0].vx, svy = eta * ps[0].vy, svz = eta * ps[0].vz;
    for (int i = 1; i < N; i++) {
        const double ei = 1. / eta, mi = ps[i].m;
        pj[i].m = mi;
        pj[i].x = ps[i].x - sx * ei; pj[i].y = ps[i].y - sy * ei; pj[i].z = ps[i].z - sz * ei;
        pj[i].vx = ps[i].vx - svx * ei; pj[i].vy = ps[i].vy - svy * ei; pj[i].vz = ps[i].vz - svz * ei;
        eta += mi;
        sx += mi * ps[i].x; sy += mi * ps[i].y; sz += mi * ps[i].z;
        svx += mi * ps[i].vx; svy += mi * ps[i].vy; svz += mi * ps[i].vz;
    }
    const double Mi = 1. / eta;
    pj[0].m = eta;
    pj[0].x = sx * Mi; pj[0].y = sy * Mi; pj[0].z = sz * Mi;
    pj[0].vx = svx * Mi; pj[0].vy = svy * Mi; pj[0].vz = svz * Mi;
}

static void to_jacobi_acc(const jbody* ps, jbody* pj, int N) {
    double eta = ps[0].m;
    double sx = eta * ps[0].ax, sy = eta * ps[0].ay, sz = eta * ps[0].az;
    for (int i = 1; i < N; i++) {
        const double ei = 1. / eta, mi = ps[i].m;
        pj[i].ax = ps[i].ax - sx * ei; pj[i].ay = ps[i].ay - sy * ei; pj[i].az = ps[i].az - sz * ei;
        eta += mi;
        sx += mi * ps[i].ax; sy += mi * ps[i].ay; sz += mi * ps[i].az;
    }
    const double Mi = 1. / eta;
    pj[0].ax = sx * Mi; pj[0].ay = sy * Mi; pj[0].az = sz * Mi;
}

static void to_inertial(jbody* ps, const jbody* pj, int N, int with_vel) {
    double eta = pj[0].m;
    double sx = pj[0].x * eta, sy = pj[0].y * eta, sz = pj[0].z * eta;
    double svx = pj[0].vx * eta, svy = pj[0].vy * eta, svz = pj[0].vz * eta;
    for (int i = N - 1; i > 0; i--) {
        const double ei = 1. / eta, mi = ps[i].m;
        sx = (sx - mi * pj[i].x) * ei; sy = (sy - mi * pj[i].y) * ei; sz = (sz - mi * pj[i].z) * ei;
        ps[i].x = pj[i].x + sx; ps[i].y = pj[i].y + sy; ps[i].z = pj[i].z + sz;
        if (with_vel) {
            svx = (svx - mi * pj[i].vx) * ei; svy = (svy - mi * pj[i].vy) * ei; svz = (svz - mi * pj[i].vz) * ei;
            ps[i].vx = pj[i].vx + svx; ps[i].vy = pj[i].vy + svy; ps[i].vz = pj[i].vz + svz;
        }
        eta -= mi;
        sx *= eta; sy *= eta; sz *= eta;
        if (with_vel) { svx *= eta; svy *= eta; svz *= eta; }
    }
    const double Mi = 1. / eta;
    ps[0].x = sx * Mi; ps[0].y = sy * Mi; ps[0].z = sz * Mi;
    if (with_vel) { ps[0].vx = svx * Mi; ps[0].vy = svy * Mi; ps[0].vz = svz * Mi; }
}

static void drift(jbody* pj, double m0, int N, double G, double h) {
    double eta = m0;
    for (int i = 1; i < N; i++) {
        eta += pj[i].m;
        kepler(&pj[i], G * eta, h);
    }
    pj[0].x += h * pj[0].vx; pj[0].y += h * pj[0].vy; pj[0].z += h * pj[0].vz;
}

/* direct gravity without the pair (0,1): that pair is the Kepler problem of Jacobi body 1 */
static void gravity_without_01(jbody* ps, int N, double G) {
    for (int i = 0; i < N; i++) { ps[i].ax = 0.; ps[i].ay = 0.; ps[i].az = 0.; }
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < N; j++) {
            if (i == j) continue;
            if ((i == 0 && j == 1) || (i == 1 && j == 0)) continue;
            if (ps[j].m == 0.) continue;
            const double dx = ps[i].x - ps[j].x, dy = ps[i].y - ps[j].y, dz = ps[i].z - ps[j].z;
            const double r = sqrt(dx * dx + dy * dy + dz * dz);
            const double pre = -G * ps[j].m / (r * r * r);
            ps[i].ax += pre * dx; ps[i].ay += pre * dy; ps[i].az += pre * dz;
        }
    }
}

static void pack(const orc_sim* s, jbody* ps) {
    for (int i = 0; i < s->N; i++) {
        ps[i].x = s->x[i]; ps[i].y = s->y[i]; ps[i].z = s->z[i]; ps[i].vx = s->vx[i]; ps[i].vy = s->vy[i]; ps[i].vz = s->vz[i];
        ps[i].ax = s->ax[i]; ps[i].ay = s->ay[i]; ps[i].az = s->az[i]; ps[i].m = s->m[i];
    }
}
static void unpack(const orc_sim* s, const jbody* ps) {
    for (int i = 0; i < s->N; i++) {
        s->x[i] = ps[i].x; s->y[i] = ps[i].y; s->z[i] = ps[i].z; s->vx[i] = ps[i].vx; s->vy[i] = ps[i].vy; s->vz[i] = ps[i].vz;
        s->ax[i] = ps[i].ax; s->ay[i] = ps[i].ay; s->az[i] = ps[i].az;
    }
}

/* nsteps steps of length dt.  `f` (may be NULL) adds the additional forces to the acceleration
 * columns of `s`, which hold the mid-step positions and (also for velocity-dependent forces)
 * the mid-step velocities when it is called.  `pre`/`post` (may be NULL) are operators applied to
 * the synchronised state before and after each step with the step length as argument
 * (REBOUNDx's pre/post timestep modifications). */
int orc_whfast_steps(orc_sim* s, double dt, long nsteps, orc_force_fn f, orc_step_fn pre, orc_step_fn post, void* user) {
    const int N = s->N;
    if (N < 2) return -1;
    jbody* ps = (jbody*)malloc((size_t)N * sizeof(jbody));
    jbody* pj = (jbody*)malloc((size_t)N * sizeof(jbody));
    if (!ps || !pj) { free(ps); free(pj); return -1; }
    const double G = s->G;
    for (long k = 0; k < nsteps; k++) {
        if (pre) pre(s, dt, user);
        pack(s, ps);
        to_jacobi_posvel(ps, pj, N);
        drift(pj, ps[0].m, N, G, 0.5 * dt);
        to_inertial(ps, pj, N, 1);
        gravity_without_01(ps, N, G);
        if (f) { unpack(s, ps); f(s, user); pack(s, ps); }
        to_jacobi_acc(ps, pj, N);
        {
            double eta = ps[0].m;
            for (int i = 1; i < N; i++) {
                eta += ps[i].m;
                pj[i].vx += dt * pj[i].ax; pj[i].vy += dt * pj[i].ay; pj[i].vz += dt * pj[i].az;
                if (i > 1) {
                    const double r2i = 1. / (pj[i].x * pj[i].x + pj[i].y * pj[i].y + pj[i].z * pj[i].z);
                    const double pre3 = dt * sqrt(r2i) * r2i * G * eta;
                    pj[i].vx += pre3 * pj[i].x; pj[i].vy += pre3 * pj[i].y; pj[i].vz += pre3 * pj[i].z;
                }
            }
        }
        drift(pj, ps[0].m, N, G, 0.5 * dt);
        to_inertial(ps, pj, N, 1);
        unpack(s, ps);
        if (post) post(s, dt, user);
    }
    free(ps); free(pj);
    return 0;
}

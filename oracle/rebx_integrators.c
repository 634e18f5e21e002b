/* rebx_integrators.c -- CPU restatement of the reference's "integrate a force across a
 * timestep" operator and the kick/drift steppers.  TEST INFRASTRUCTURE ONLY (the checker the
 * HIP path is compared with); the product never links or calls it.
 *
 * Follows src/integrate_force.c:34-78, src/integrator_euler.c:33-41, src/integrator_rk2.c:38-67,
 * src/integrator_rk4.c:40-91, src/integrator_implicit_midpoint.c:32-124, src/steppers.c:112-133.
 * The reference works on array-of-structure copies of the particle array whose positions and
 * masses equal the originals; here a stage state shares the position and mass columns and owns
 * velocity and acceleration columns.  Every floating-point expression is written as in the
 * reference (no contraction: see the Makefile).
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include "rebx_oracle.h"

typedef struct { double *vx, *vy, *vz, *ax, *ay, *az; } stage;

static int stage_alloc(stage* k, int N) {
    double* b = (double*)malloc((size_t)(N > 0 ? N : 1) * 6 * sizeof(double));
    if (!b) return -1;
    k->vx = b; k->vy = b + N; k->vz = b + 2 * N; k->ax = b + 3 * N; k->ay = b + 4 * N; k->az = b + 5 * N;
    return 0;
}
static void stage_free(stage* k) { free(k->vx); }

/* memcpy(k, sim->particles, ...) of the reference: velocities AND accelerations are copied */
static void stage_copy(stage* k, const orc_sim* s) {
    const size_t b = (size_t)s->N * sizeof(double);
    memcpy(k->vx, s->vx, b); memcpy(k->vy, s->vy, b); memcpy(k->vz, s->vz, b);
    memcpy(k->ax, s->ax, b); memcpy(k->ay, s->ay, b); memcpy(k->az, s->az, b);
}
static orc_sim stage_sim(const orc_sim* s, const stage* k) {
    orc_sim t = *s;
    t.vx = k->vx; t.vy = k->vy; t.vz = k->vz; t.ax = k->ax; t.ay = k->ay; t.az = k->az;
    return t;
}

/* integrator_euler.c:33-41 */
static void euler(orc_sim* s, double dt, orc_force_fn f, void* user) {
    f(s, user);
    for (int i = 0; i < s->N; i++) {
        s->vx[i] += dt * s->ax[i];
        s->vy[i] += dt * s->ay[i];
        s->vz[i] += dt * s->az[i];
    }
}

/* integrator_rk2.c:38-67 */
static int rk2(orc_sim* s, double dt, orc_force_fn f, void* user) {
    const int N = s->N;
    stage k2;
    if (stage_alloc(&k2, N)) return -1;
    stage_copy(&k2, s);
    orc_sim s2 = stage_sim(s, &k2);
    f(s, user);
    const double a21 = 2. * dt / 3.;
    for (int i = 0; i < N; i++) {
        k2.vx[i] = s->vx[i] + a21 * s->ax[i];
        k2.vy[i] = s->vy[i] + a21 * s->ay[i];
        k2.vz[i] = s->vz[i] + a21 * s->az[i];
    }
    f(&s2, user);
    const double b1 = dt / 4.;
    const double b2 = 3. * dt / 4.;
    for (int i = 0; i < N; i++) {
        s->vx[i] += b1 * s->ax[i] + b2 * k2.ax[i];
        s->vy[i] += b1 * s->ay[i] + b2 * k2.ay[i];
        s->vz[i] += b1 * s->az[i] + b2 * k2.az[i];
    }
    stage_free(&k2);
    return 0;
}

/* integrator_rk4.c:40-91 */
static int rk4(orc_sim* s, double dt, orc_force_fn f, void* user) {
    const int N = s->N;
    orc_reset_accelerations(s);
    stage k2, k3;
    if (stage_alloc(&k2, N)) return -1;
    if (stage_alloc(&k3, N)) { stage_free(&k2); return -1; }
    stage_copy(&k2, s);
    stage_copy(&k3, s);
    orc_sim s2 = stage_sim(s, &k2), s3 = stage_sim(s, &k3);
    const double dt2 = dt / 2.;
    f(s, user);                                   /* k1 = sim.particles.a */
    for (int i = 0; i < N; i++) {
        k2.vx[i] = s->vx[i] + dt2 * s->ax[i];
        k2.vy[i] = s->vy[i] + dt2 * s->ay[i];
        k2.vz[i] = s->vz[i] + dt2 * s->az[i];
    }
    f(&s2, user);
    for (int i = 0; i < N; i++) {
        k3.vx[i] = s->vx[i] + dt2 * k2.ax[i];
        k3.vy[i] = s->vy[i] + dt2 * k2.ay[i];
        k3.vz[i] = s->vz[i] + dt2 * k2.az[i];
    }
    f(&s3, user);
    for (int i = 0; i < N; i++) {                 /* k2+k3 kept in k3; k2 reused for k4 */
        k2.vx[i] = s->vx[i] + dt * k3.ax[i];
        k2.vy[i] = s->vy[i] + dt * k3.ay[i];
        k2.vz[i] = s->vz[i] + dt * k3.az[i];
        k3.ax[i] += k2.ax[i];
        k3.ay[i] += k2.ay[i];
        k3.az[i] += k2.az[i];
    }
    orc_reset_accelerations(&s2);
    f(&s2, user);
    const double dt6 = dt / 6.;
    for (int i = 0; i < N; i++) {
        s->vx[i] += dt6 * (s->ax[i] + k2.ax[i] + 2. * k3.ax[i]);
        s->vy[i] += dt6 * (s->ay[i] + k2.ay[i] + 2. * k3.ay[i]);
        s->vz[i] += dt6 * (s->az[i] + k2.az[i] + 2. * k3.az[i]);
    }
    stage_free(&k2);
    stage_free(&k3);
    return 0;
}

/* integrator_implicit_midpoint.c:32-124; returns the loop counter n (10 = not converged) */
static int implicit_midpoint(orc_sim* s, double dt, orc_force_fn f, void* user) {
    const int N = s->N;
    stage fin, prev, avg;
    if (stage_alloc(&fin, N)) return -1;
    if (stage_alloc(&prev, N)) { stage_free(&fin); return -1; }
    if (stage_alloc(&avg, N)) { stage_free(&fin); stage_free(&prev); return -1; }
    stage_copy(&fin, s);
    stage_copy(&avg, s);
    /* positions and masses of ps_avg: 0.5*(x + x) == x exactly, so sharing the columns is the
     * same as the reference's avg_particles (:32-45) for them */
    orc_sim savg = stage_sim(s, &avg);
    int n;
    for (n = 0; n < 10; n++) {
        const size_t b = (size_t)N * sizeof(double);
        memcpy(prev.vx, fin.vx, b); memcpy(prev.vy, fin.vy, b); memcpy(prev.vz, fin.vz, b);
        f(&savg, user);
        for (int i = 0; i < N; i++) {
            fin.vx[i] = s->vx[i] + dt * avg.ax[i];
            fin.vy[i] = s->vy[i] + dt * avg.ay[i];
            fin.vz[i] = s->vz[i] + dt * avg.az[i];
        }
        double tot2 = 0., deltatot2 = 0.;        /* compare(), :47-62 */
        for (int i = 0; i < N; i++) {
            const double dvx = fin.vx[i] - prev.vx[i];
            const double dvy = fin.vy[i] - prev.vy[i];
            const double dvz = fin.vz[i] - prev.vz[i];
            deltatot2 += dvx * dvx + dvy * dvy + dvz * dvz;
            tot2 += fin.vx[i] * fin.vx[i] + fin.vy[i] * fin.vy[i] + fin.vz[i] * fin.vz[i];
        }
        if (deltatot2 / tot2 < DBL_EPSILON * DBL_EPSILON) break;
        for (int i = 0; i < N; i++) {            /* avg_particles(ps_avg, ps_orig, ps_final) */
            avg.vx[i] = 0.5 * (s->vx[i] + fin.vx[i]);
            avg.vy[i] = 0.5 * (s->vy[i] + fin.vy[i]);
            avg.vz[i] = 0.5 * (s->vz[i] + fin.vz[i]);
            avg.ax[i] = 0.; avg.ay[i] = 0.; avg.az[i] = 0.;
        }
    }
    for (int i = 0; i < N; i++) { s->vx[i] = fin.vx[i]; s->vy[i] = fin.vy[i]; s->vz[i] = fin.vz[i]; }
    stage_free(&fin); stage_free(&prev); stage_free(&avg);
    return n;
}

/* integrate_force.c:34-78.  Returns the implicit-midpoint loop counter (0 for the explicit
 * schemes), -1 on allocation failure. */
int orc_integrate_force(orc_sim* s, int integrator, double dt, orc_force_fn f, void* user) {
    orc_reset_accelerations(s);
    switch (integrator) {
    case ORC_INTEGRATOR_IMPLICIT_MIDPOINT: return implicit_midpoint(s, dt, f, user);
    case ORC_INTEGRATOR_RK2: return rk2(s, dt, f, user);
    case ORC_INTEGRATOR_RK4: return rk4(s, dt, f, user);
    case ORC_INTEGRATOR_EULER: euler(s, dt, f, user); return 0;
    default: return 0;      /* REBX_INTEGRATOR_NONE and unknown values: nothing */
    }
}

/* steppers.c:112-121 */
void orc_drift_step(orc_sim* s, double dt) {
    for (int i = 0; i < s->N; i++) {
        s->x[i] += dt * s->vx[i];
        s->y[i] += dt * s->vy[i];
        s->z[i] += dt * s->vz[i];
    }
}

/* steppers.c:123-133, without the call into REBOUND's gravity (the accelerations are
 * whatever the caller left in the columns) */
void orc_kick_step(orc_sim* s, double dt) {
    for (int i = 0; i < s->N; i++) {
        s->vx[i] += dt * s->ax[i];
        s->vy[i] += dt * s->ay[i];
        s->vz[i] += dt * s->az[i];
    }
}

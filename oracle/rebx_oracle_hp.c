/*
 * rebx_oracle_hp.c -- the two forces whose reference implementation is a long sequential
 * recurrence (gr in Jacobi coordinates, modify_orbits_forces in Jacobi / barycentric
 * coordinates), evaluated in IEEE binary128 (113-bit significand, gcc's __float128).
 * TEST INFRASTRUCTURE ONLY (see rebx_oracle.h).
 *
 * Purpose: a yardstick for rounding error.  The fp64 checker (rebx_oracle.c) follows the
 * reference's sequential order of operations; the HIP library evaluates the same sums as
 * parallel prefix/suffix sums.  Both are roundings of the same real-number result; this file
 * computes that result to ~1e-30, so that a test can measure how far EACH of the two fp64
 * evaluations is from it.
 *
 * Formulas (same as rebx_oracle.c, same reference lines):
 *   gr                    src/gr.c:73-171  (Newtonian pair sum :86-103, Jacobi transform :108,
 *                         post-Newtonian term :110-156, back transform :162)
 *   modify_orbits_forces  src/modify_orbits_forces.c:77-118 + src/rebxtools.c:45-126
 * The REBOUND helpers are written in closed form: the Jacobi coordinate of body i is the body
 * minus (sum_{k<i} m_k q_k) / (sum_{k<i} m_k); the back transform of Jacobi accelerations is
 * a_i = aJ_i - sum_{k>=i} (m_k / sum_{l<=k} m_l) aJ_k  (aJ_0 = 0), a_0 = - sum_{k>=1} (...);
 * the centre of mass "without the bodies from i on" is the same interior sum.  In exact
 * arithmetic these equal the reference's recurrences term by term.
 *
 * Results are returned as pairs of doubles (hi, lo) with value hi + lo, i.e. to ~1e-32 relative.
 * Only +, -, *, / of __float128 are used (libgcc); the square root is a double seed refined by
 * Newton steps, so no extra library is needed.
 */
#include "rebx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

typedef __float128 q128;

static q128 q_sqrt(q128 a) {
    if (a <= 0) return 0;
    q128 r = (q128)sqrt((double)a);
    r = (r + a / r) / 2;            /* 53 -> ~106 bits */
    r = (r + a / r) / 2;            /* -> full precision */
    r = (r + a / r) / 2;
    return r;
}

static void split(q128 v, double* hi, double* lo, int i) {
    const double h = (double)v;
    hi[i] = h;
    lo[i] = (double)(v - (q128)h);
}

static inline int has(const double* col, int i) { return col != NULL && !isnan(col[i]); }
static inline int flagged(const uint8_t* col, int i) { return col != NULL && col[i] != 0; }

/* src/gr.c:110-156 for one Jacobi body; the inner iteration runs under the reference's rule
 * (relative change below DBL_EPSILON, at most max_iterations rounds) */
static void gr_term_q(const q128 p[9], q128 mu, q128 C2, int max_iterations, q128 out[3]) {
    q128 vix = p[3], viy = p[4], viz = p[5];
    q128 vi2 = vix * vix + viy * viy + viz * viz;
    const q128 ri = q_sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    q128 A = (vi2 / 2 + 3 * mu / ri) / C2;
    const q128 eps2 = (q128)DBL_EPSILON * (q128)DBL_EPSILON;
    for (int q = 0; q < max_iterations; q++) {
        const q128 ox = vix, oy = viy, oz = viz;
        vix = p[3] / (1 - A); viy = p[4] / (1 - A); viz = p[5] / (1 - A);
        vi2 = vix * vix + viy * viy + viz * viz;
        A = (vi2 / 2 + 3 * mu / ri) / C2;
        const q128 dx = vix - ox, dy = viy - oy, dz = viz - oz;
        if ((dx * dx + dy * dy + dz * dz) / vi2 < eps2) break;
    }
    const q128 B = (mu / ri - 3 * vi2 / 2) * mu / (ri * ri * ri) / C2;
    const q128 rdotrdot = p[0] * p[3] + p[1] * p[4] + p[2] * p[5];
    const q128 vdx = p[6] + B * p[0], vdy = p[7] + B * p[1], vdz = p[8] + B * p[2];
    const q128 vdotvdot = vix * vdx + viy * vdy + viz * vdz;
    const q128 D = (vdotvdot - 3 * mu / (ri * ri * ri) * rdotrdot) / C2;
    out[0] = B * (1 - A) * p[0] - A * p[6] - D * vix;
    out[1] = B * (1 - A) * p[1] - A * p[7] - D * viy;
    out[2] = B * (1 - A) * p[2] - A * p[8] - D * viz;
}

/* The acceleration gr ADDS to every body (not the sum with the incoming accelerations),
 * as (hi, lo) pairs: out[0..2] = hi of x, y, z; out[3..5] = lo.  Each of length N. */
int orc_hp_gr(const orc_sim* s, double c, int max_iterations, double* const out[6]) {
    const int N = s->N;
    if (N <= 0) return 0;
    const q128 G = s->G, C2 = (q128)c * (q128)c;
    q128* an = (q128*)calloc((size_t)N * 3, sizeof(q128));       /* Newtonian accelerations */
    q128* aj = (q128*)calloc((size_t)N * 3, sizeof(q128));       /* post-Newtonian term, Jacobi frame */
    int nm = 0;
    int* massive = (int*)malloc((size_t)N * sizeof(int));
    for (int i = 0; i < N; i++) if (s->m[i] != 0.) massive[nm++] = i;
    /* rows are independent; the threads only shorten the wait (software binary128) */
#pragma omp parallel for schedule(static)
    for (int t = 0; t < N; t++) {                                   /* :86-103 */
        q128 ax = 0, ay = 0, az = 0;
        for (int k = 0; k < nm; k++) {
            const int u = massive[k];
            if (u == t) continue;
            const q128 dx = (q128)s->x[u] - s->x[t], dy = (q128)s->y[u] - s->y[t], dz = (q128)s->z[u] - s->z[t];
            const q128 r2 = dx * dx + dy * dy + dz * dz;
            const q128 w = G * s->m[u] / (r2 * q_sqrt(r2));
            ax += w * dx; ay += w * dy; az += w * dz;
        }
        an[3 * t] = ax; an[3 * t + 1] = ay; an[3 * t + 2] = az;
    }
    free(massive);
    const q128 mu = G * s->m[0];                                    /* :106-107 */
    q128 eta = s->m[0];
    q128 S[9];
    S[0] = eta * s->x[0];  S[1] = eta * s->y[0];  S[2] = eta * s->z[0];
    S[3] = eta * s->vx[0]; S[4] = eta * s->vy[0]; S[5] = eta * s->vz[0];
    S[6] = eta * an[0];    S[7] = eta * an[1];    S[8] = eta * an[2];
    q128* share = (q128*)calloc((size_t)N, sizeof(q128));           /* m_i / (eta_i + m_i) */
    for (int i = 1; i < N; i++) {                                   /* :108 and :110-156 */
        const q128 mi = s->m[i];
        const q128 st[9] = {s->x[i], s->y[i], s->z[i], s->vx[i], s->vy[i], s->vz[i], an[3 * i], an[3 * i + 1], an[3 * i + 2]};
        q128 pj[9];
        for (int k = 0; k < 9; k++) pj[k] = st[k] - S[k] / eta;
        gr_term_q(pj, mu, C2, max_iterations, &aj[3 * i]);
        share[i] = mi / (eta + mi);
        eta += mi;
        for (int k = 0; k < 9; k++) S[k] += mi * st[k];
    }
    q128 suffix[3] = {0, 0, 0};                                     /* :162 */
    for (int i = N - 1; i >= 1; i--) {
        for (int k = 0; k < 3; k++) {
            suffix[k] += share[i] * aj[3 * i + k];
            const q128 a = aj[3 * i + k] - suffix[k];
            split(a, out[k], out[3 + k], i);
        }
    }
    for (int k = 0; k < 3; k++) split(-suffix[k], out[k], out[3 + k], 0);
    free(an); free(aj); free(share);
    return 0;
}

/* src/modify_orbits_forces.c:77-118 */
static void mof_term_q(const q128 p[6], const q128 ref[6], q128 tau_a, int has_ta, q128 tau_e, int has_te, q128 tau_inc, int has_ti, q128 a[3]) {
    q128 d[3], dv[3];
    for (int k = 0; k < 3; k++) { d[k] = p[k] - ref[k]; dv[k] = p[3 + k] - ref[3 + k]; }
    for (int k = 0; k < 3; k++) a[k] = has_ta ? dv[k] / (2 * tau_a) : 0;
    if (has_te || has_ti) {
        const q128 r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const q128 vdotr = d[0] * dv[0] + d[1] * dv[1] + d[2] * dv[2];
        const q128 prefac = has_te ? 2 * vdotr / r2 / tau_e : 0;
        for (int k = 0; k < 3; k++) a[k] += prefac * d[k];
        if (has_ti) a[2] += 2 * dv[2] / tau_inc;
    }
}

/* The acceleration modify_orbits_forces ADDS to every body in Jacobi (0) or barycentric (1)
 * coordinates, as (hi, lo) pairs laid out as in orc_hp_gr. */
int orc_hp_modify_orbits_forces(const orc_sim* s, int coordinates, const double* tau_a, const double* tau_e, const double* tau_inc,
                                double* const out[6]) {
    const int N = s->N;
    if (N <= 0) return 0;
    if (coordinates != ORC_COORD_JACOBI && coordinates != ORC_COORD_BARYCENTRIC) return -2;
    const int jacobi = coordinates == ORC_COORD_JACOBI;
    q128* own = (q128*)calloc((size_t)N * 3, sizeof(q128));
    q128* w = (q128*)calloc((size_t)N, sizeof(q128));
    q128 tot[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < N; i++) {
        const q128 mi = s->m[i];
        tot[0] += mi;
        tot[1] += mi * s->x[i];  tot[2] += mi * s->y[i];  tot[3] += mi * s->z[i];
        tot[4] += mi * s->vx[i]; tot[5] += mi * s->vy[i]; tot[6] += mi * s->vz[i];
    }
    q128 S[7] = {0, 0, 0, 0, 0, 0, 0};                             /* bodies before i */
    for (int i = 0; i < N; i++) {
        const q128 mi = s->m[i];
        const q128 p[6] = {s->x[i], s->y[i], s->z[i], s->vx[i], s->vy[i], s->vz[i]};
        if (!(jacobi && i == 0)) {                                  /* rebxtools.c:72-74: Jacobi skips body 0 */
            const q128* sum = jacobi ? S : tot;
            q128 ref[6];
            for (int k = 0; k < 6; k++) ref[k] = sum[0] > 0 ? sum[1 + k] / sum[0] : sum[1 + k];
            mof_term_q(p, ref, has(tau_a, i) ? (q128)tau_a[i] : 0, has(tau_a, i), has(tau_e, i) ? (q128)tau_e[i] : 0, has(tau_e, i),
                       has(tau_inc, i) ? (q128)tau_inc[i] : 0, has(tau_inc, i), &own[3 * i]);
            w[i] = jacobi ? mi / (S[0] + mi) : mi / tot[0];         /* rebxtools.c:88, :96 */
        }
        S[0] += mi;
        for (int k = 0; k < 6; k++) S[1 + k] += mi * p[k];
    }
    if (jacobi) {
        q128 suffix[3] = {0, 0, 0};                                 /* every j <= i loses w_i a_i (rebxtools.c:96-106) */
        for (int i = N - 1; i >= 0; i--)
            for (int k = 0; k < 3; k++) {
                suffix[k] += w[i] * own[3 * i + k];
                split(own[3 * i + k] - suffix[k], out[k], out[3 + k], i);
            }
    } else {
        q128 all[3] = {0, 0, 0};                                    /* every j loses w_i a_i (rebxtools.c:88-94) */
        for (int i = 0; i < N; i++)
            for (int k = 0; k < 3; k++) all[k] += w[i] * own[3 * i + k];
        for (int i = 0; i < N; i++)
            for (int k = 0; k < 3; k++) split(own[3 * i + k] - all[k], out[k], out[3 + k], i);
    }
    free(own); free(w);
    return 0;
}

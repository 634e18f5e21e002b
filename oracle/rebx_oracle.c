/*
 * rebx_oracle.c -- CPU restatement of the REBOUNDx additional-force path.
 * TEST INFRASTRUCTURE ONLY (see rebx_oracle.h).  Build with -ffp-contract=off.
 *
 * Each force gathers the SoA columns into a small array-of-structs working
 * copy, runs the reference's loop nest in the reference's order, and scatters
 * the accelerations back.  Nothing here is tuned; it is meant to be read next
 * to the cited reference lines.
 */
#include "rebx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* True pi.  The reference header carries a fallback #define with wrong digits
 * (src/reboundx.h:28-30) that is only used when <math.h> hides M_PI; the answer
 * recorded in the reference's own notebook needs the true value. */
#define ORC_PI 3.14159265358979323846

typedef struct body {
    double x, y, z, vx, vy, vz, ax, ay, az, m;
} body;

static body* gather(const orc_sim* s) {
    body* p = (body*)malloc((size_t)(s->N > 0 ? s->N : 1) * sizeof(body));
    for (int i = 0; i < s->N; i++) {
        p[i].x = s->x[i];   p[i].y = s->y[i];   p[i].z = s->z[i];
        p[i].vx = s->vx[i]; p[i].vy = s->vy[i]; p[i].vz = s->vz[i];
        p[i].ax = s->ax[i]; p[i].ay = s->ay[i]; p[i].az = s->az[i];
        p[i].m = s->m[i];
    }
    return p;
}

static void scatter_acc(const orc_sim* s, body* p) {
    for (int i = 0; i < s->N; i++) {
        s->ax[i] = p[i].ax; s->ay[i] = p[i].ay; s->az[i] = p[i].az;
    }
    free(p);
}

/* A column entry stands for "rebx_get_param returned NULL" when it holds a NaN -- any NaN, whatever its sign
 * or payload -- except the one pattern 0x7FF8000000000001, which stands for a parameter that IS in the list
 * and whose value is NaN: the reference computes with it (src/core.c:614-622 returns the pointer). */
static inline int has(const double* col, int i) {
    if (col == NULL) return 0;
    uint64_t u;
    memcpy(&u, &col[i], sizeof u);
    return col[i] == col[i] || u == 0x7FF8000000000001ull;
}
static inline int flagged(const uint8_t* col, int i) { return col != NULL && col[i] != 0; }

/* ------------------------------------------------------------------------
 * REBOUND helpers, restated from REBOUND 3.x (tools.c / transformations.c).
 * REBOUND is not part of the reference tree; call sites are
 * src/rebxtools.c:47,77 and src/gr.c:108,162,217.
 * ---------------------------------------------------------------------- */

/* reb_get_com_of_pair: mass-weighted mean of two bodies; divides only when m>0 */
static body com_of_pair(body a, body b) {
    a.x = a.x * a.m + b.x * b.m;
    a.y = a.y * a.m + b.y * b.m;
    a.z = a.z * a.m + b.z * b.m;
    a.vx = a.vx * a.m + b.vx * b.m;
    a.vy = a.vy * a.m + b.vy * b.m;
    a.vz = a.vz * a.m + b.vz * b.m;
    a.ax = a.ax * a.m + b.ax * b.m;
    a.ay = a.ay * a.m + b.ay * b.m;
    a.az = a.az * a.m + b.az * b.m;
    a.m += b.m;
    if (a.m > 0.) {
        a.x /= a.m;  a.y /= a.m;  a.z /= a.m;
        a.vx /= a.m; a.vy /= a.m; a.vz /= a.m;
        a.ax /= a.m; a.ay /= a.m; a.az /= a.m;
    }
    return a;
}

/* reb_get_com: left fold of com_of_pair over all real particles, from a zero body */
static body com_of_all(const body* p, int N) {
    body com;
    memset(&com, 0, sizeof(com));
    for (int i = 0; i < N; i++) com = com_of_pair(com, p[i]);
    return com;
}

/* reb_get_com_without_particle: down-date a centre of mass by one member */
static body com_without(body com, body p) {
    com.x = com.x * com.m - p.x * p.m;
    com.y = com.y * com.m - p.y * p.m;
    com.z = com.z * com.m - p.z * p.m;
    com.vx = com.vx * com.m - p.vx * p.m;
    com.vy = com.vy * com.m - p.vy * p.m;
    com.vz = com.vz * com.m - p.vz * p.m;
    com.ax = com.ax * com.m - p.ax * p.m;
    com.ay = com.ay * com.m - p.ay * p.m;
    com.az = com.az * com.m - p.az * p.m;
    com.m -= p.m;
    if (com.m > 0.) {
        com.x /= com.m;  com.y /= com.m;  com.z /= com.m;
        com.vx /= com.m; com.vy /= com.m; com.vz /= com.m;
        com.ax /= com.m; com.ay /= com.m; com.az /= com.m;
    }
    return com;
}

/* reb_transformations_inertial_to_jacobi_posvelacc(ps, pj, pmass=ps, N):
 * running mass eta and running m-weighted sums; Jacobi coordinate i is the
 * body minus (sum * 1/eta) of everything interior to it; slot 0 gets the
 * centre of mass and the total mass. */
static void to_jacobi(const body* ps, body* pj, int N, int with_acc) {
    double eta = ps[0].m;
    double sx = eta * ps[0].x, sy = eta * ps[0].y, sz = eta * ps[0].z;
    double svx = eta * ps[0].vx, svy = eta * ps[0].vy, svz = eta * ps[0].vz;
    double sax = eta * ps[0].ax, say = eta * ps[0].ay, saz = eta * ps[0].az;
    for (int i = 1; i < N; i++) {
        const double ei = 1. / eta;
        const body pi = ps[i];
        const double pme = pi.m;
        eta += pme;
        pj[i].m = pme;
        pj[i].x = pi.x - sx * ei;
        pj[i].y = pi.y - sy * ei;
        pj[i].z = pi.z - sz * ei;
        pj[i].vx = pi.vx - svx * ei;
        pj[i].vy = pi.vy - svy * ei;
        pj[i].vz = pi.vz - svz * ei;
        if (with_acc) {
            pj[i].ax = pi.ax - sax * ei;
            pj[i].ay = pi.ay - say * ei;
            pj[i].az = pi.az - saz * ei;
        }
        sx += pme * pi.x;   sy += pme * pi.y;   sz += pme * pi.z;
        svx += pme * pi.vx; svy += pme * pi.vy; svz += pme * pi.vz;
        if (with_acc) { sax += pme * pi.ax; say += pme * pi.ay; saz += pme * pi.az; }
    }
    const double mtot = eta;
    const double mtoti = 1. / mtot;
    pj[0].m = mtot;
    pj[0].x = sx * mtoti;   pj[0].y = sy * mtoti;   pj[0].z = sz * mtoti;
    pj[0].vx = svx * mtoti; pj[0].vy = svy * mtoti; pj[0].vz = svz * mtoti;
    if (with_acc) { pj[0].ax = sax * mtoti; pj[0].ay = say * mtoti; pj[0].az = saz * mtoti; }
}

/* reb_transformations_jacobi_to_inertial_acc(ps, pj, pmass=ps, N): reverse recurrence */
static void jacobi_to_inertial_acc(body* ps, const body* pj, int N) {
    double eta = pj[0].m;
    double sax = pj[0].ax * eta, say = pj[0].ay * eta, saz = pj[0].az * eta;
    for (int i = N - 1; i > 0; i--) {
        const body pji = pj[i];
        const double ei = 1. / eta;
        const double mi = ps[i].m;
        sax = (sax - mi * pji.ax) * ei;
        say = (say - mi * pji.ay) * ei;
        saz = (saz - mi * pji.az) * ei;
        ps[i].ax = pji.ax + sax;
        ps[i].ay = pji.ay + say;
        ps[i].az = pji.az + saz;
        eta -= mi;
        sax *= eta; say *= eta; saz *= eta;
    }
    const double mtoti = 1. / eta;
    ps[0].ax = sax * mtoti;
    ps[0].ay = say * mtoti;
    ps[0].az = saz * mtoti;
}

/* ------------------------------------------------------------------------
 * radiation_forces  (src/radiation_forces.c)
 * ---------------------------------------------------------------------- */

/* :69-98 -- one pass for one source; only particles with beta feel it */
static void radiation_pass(body* p, int N, double G, double c, int src, const double* beta) {
    const body S = p[src];
    const double mu = G * S.m;
    for (int i = 0; i < N; i++) {
        if (i == src) continue;
        if (!has(beta, i)) continue;
        const body q = p[i];
        const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
        const double dr = sqrt(dx * dx + dy * dy + dz * dz);
        const double dvx = q.vx - S.vx, dvy = q.vy - S.vy, dvz = q.vz - S.vz;
        const double rdot = (dx * dvx + dy * dvy + dz * dvz) / dr;
        const double a_rad = beta[i] * mu / (dr * dr);
        p[i].ax += a_rad * ((1. - rdot / c) * dx / dr - dvx / c);
        p[i].ay += a_rad * ((1. - rdot / c) * dy / dr - dvy / c);
        p[i].az += a_rad * ((1. - rdot / c) * dz / dr - dvz / c);
    }
}

/* :100-117 -- every flagged particle is a source; none flagged -> index 0 */
int orc_radiation_forces(const orc_sim* s, double c, const double* beta, const uint8_t* radiation_source) {
    if (s->N <= 0) return 0;
    body* p = gather(s);
    int found = 0;
    for (int i = 0; i < s->N; i++) {
        if (flagged(radiation_source, i)) {
            found = 1;
            radiation_pass(p, s->N, s->G, c, i, beta);
        }
    }
    if (!found) radiation_pass(p, s->N, s->G, c, 0, beta);
    scatter_acc(s, p);
    return 0;
}

double orc_rad_calc_beta(double G, double c, double source_mass, double source_luminosity, double radius, double density, double Q_pr) {
    /* :119-121 */
    return 3. * source_luminosity * Q_pr / (16. * ORC_PI * G * source_mass * c * density * radius);
}
double orc_rad_calc_particle_radius(double G, double c, double source_mass, double source_luminosity, double beta, double density, double Q_pr) {
    /* :122-124 */
    return 3. * source_luminosity * Q_pr / (16. * ORC_PI * G * source_mass * c * density * beta);
}

/* ------------------------------------------------------------------------
 * gravitational_harmonics  (src/gravitational_harmonics.c)
 * ---------------------------------------------------------------------- */

/* :61-85 */
static void j2_pass(body* p, int N, double G, double J2, double R_eq, int src) {
    const body S = p[src];
    for (int i = 0; i < N; i++) {
        if (i == src) continue;
        const body q = p[i];
        const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double r = sqrt(r2);
        const double costheta2 = dz * dz / r2;
        const double prefac = 3. * J2 * R_eq * R_eq / r2 / r2 / r / 2.;
        const double fac = 5. * costheta2 - 1.;
        p[i].ax += G * S.m * prefac * fac * dx;
        p[i].ay += G * S.m * prefac * fac * dy;
        p[i].az += G * S.m * prefac * (fac - 2.) * dz;
        p[src].ax -= G * q.m * prefac * fac * dx;
        p[src].ay -= G * q.m * prefac * fac * dy;
        p[src].az -= G * q.m * prefac * (fac - 2.) * dz;
    }
}

/* :99-123 */
static void j4_pass(body* p, int N, double G, double J4, double R_eq, int src) {
    const body S = p[src];
    for (int i = 0; i < N; i++) {
        if (i == src) continue;
        const body q = p[i];
        const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double r = sqrt(r2);
        const double costheta2 = dz * dz / r2;
        const double prefac = 5. * J4 * R_eq * R_eq * R_eq * R_eq / r2 / r2 / r2 / r / 8.;
        const double fac = 63. * costheta2 * costheta2 - 42. * costheta2 + 3.;
        p[i].ax += G * S.m * prefac * fac * dx;
        p[i].ay += G * S.m * prefac * fac * dy;
        p[i].az += G * S.m * prefac * (fac + 12. - 28. * costheta2) * dz;
        p[src].ax -= G * q.m * prefac * fac * dx;
        p[src].ay -= G * q.m * prefac * fac * dy;
        p[src].az -= G * q.m * prefac * (fac + 12. - 28. * costheta2) * dz;
    }
}

/* :87-97, :125-140 -- all J2 passes first, then all J4 passes */
int orc_gravitational_harmonics(const orc_sim* s, const double* J2, const double* J4, const double* R_eq) {
    if (s->N <= 0) return 0;
    body* p = gather(s);
    for (int i = 0; i < s->N; i++)
        if (has(J2, i) && has(R_eq, i)) j2_pass(p, s->N, s->G, J2[i], R_eq[i], i);
    for (int i = 0; i < s->N; i++)
        if (has(J4, i) && has(R_eq, i)) j4_pass(p, s->N, s->G, J4[i], R_eq[i], i);
    scatter_acc(s, p);
    return 0;
}

/* :142-231 */
double orc_gravitational_harmonics_potential(const orc_sim* s, const double* J2, const double* J4, const double* R_eq) {
    body* p = gather(s);
    const int N = s->N;
    const double G = s->G;
    double H2 = 0., H4 = 0.;
    for (int k = 0; k < N; k++) {
        if (!(has(J2, k) && has(R_eq, k))) continue;
        const body S = p[k];
        double H = 0.;
        for (int i = 0; i < N; i++) {
            if (i == k) continue;
            const body q = p[i];
            const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double r = sqrt(r2);
            const double costheta2 = dz * dz / r2;
            const double prefac = G * q.m * S.m * R_eq[k] * R_eq[k] / r2 / r * J2[k];
            const double P2 = 0.5 * (3. * costheta2 - 1.);
            H += prefac * P2;
        }
        H2 += H;
    }
    for (int k = 0; k < N; k++) {
        if (!(has(J4, k) && has(R_eq, k))) continue;
        const body S = p[k];
        double H = 0.;
        for (int i = 0; i < N; i++) {
            if (i == k) continue;
            const body q = p[i];
            const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double r = sqrt(r2);
            const double costheta2 = dz * dz / r2;
            const double prefac = G * q.m * S.m * R_eq[k] * R_eq[k] * R_eq[k] * R_eq[k] / r2 / r2 / r * J4[k];
            const double P4 = (35. * costheta2 * costheta2 - 30. * costheta2 + 3.) / 8.;
            H += prefac * P4;
        }
        H4 += H;
    }
    free(p);
    return H2 + H4;
}

/* ------------------------------------------------------------------------
 * gr  (src/gr.c)
 * ---------------------------------------------------------------------- */

/* :110-156 -- post-Newtonian term for Jacobi body `p` (pos, vel, Newtonian acc in
 * Jacobi coordinates); writes the Jacobi correction into out; returns iterations used */
static int gr_jacobi_term(const body p, double mu, double C2, int max_iterations, body* out) {
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
    out->ax = B * (1. - A) * p.x - A * p.ax - D * vix;
    out->ay = B * (1. - A) * p.y - A * p.ay - D * viy;
    out->az = B * (1. - A) * p.z - A * p.az - D * viz;
    return q;
}

static int gr_impl(const orc_sim* s, double c, int max_iterations, int literal) {
    const int N = s->N;
    if (N <= 0) return 0;
    const double G = s->G;
    const double C2 = c * c;
    body* particles = gather(s);
    body* ps = (body*)malloc((size_t)N * sizeof(body));
    body* pj = (body*)malloc((size_t)N * sizeof(body));
    memcpy(ps, particles, (size_t)N * sizeof(body));
    for (int i = 0; i < N; i++) { ps[i].ax = 0.; ps[i].ay = 0.; ps[i].az = 0.; }

    if (literal) {
        /* :86-103 as written: every unordered pair once */
        for (int i = 0; i < N; i++) {
            const body pi = ps[i];
            for (int j = i + 1; j < N; j++) {
                const body q = ps[j];
                const double dx = pi.x - q.x, dy = pi.y - q.y, dz = pi.z - q.z;
                const double r2 = dx * dx + dy * dy + dz * dz;
                const double r = sqrt(r2);
                const double prefac = G / (r2 * r);
                ps[i].ax -= prefac * q.m * dx;
                ps[i].ay -= prefac * q.m * dy;
                ps[i].az -= prefac * q.m * dz;
                ps[j].ax += prefac * pi.m * dx;
                ps[j].ay += prefac * pi.m * dy;
                ps[j].az += prefac * pi.m * dz;
            }
        }
    } else {
        /* Same sums, O(N * massive).  In the pair loop above, particle t receives
         * its partners' terms in ascending partner index (first as the `j` of every
         * i<t, then as the `i` of every j>t), and a massless partner adds a signed
         * zero.  Visiting only the massive partners in ascending index therefore
         * yields the same bits (up to the sign of an exact zero). */
        int nm = 0;
        int* massive = (int*)malloc((size_t)N * sizeof(int));
        for (int i = 0; i < N; i++) if (ps[i].m != 0.) massive[nm++] = i;
        for (int t = 0; t < N; t++) {
            const body pt = ps[t];
            double ax = 0., ay = 0., az = 0.;
            for (int k = 0; k < nm; k++) {
                const int u = massive[k];
                if (u == t) continue;
                const body pu = ps[u];
                if (u < t) { /* pair (i=u, j=t): d = p_u - p_t, acc_t += prefac*m_u*d */
                    const double dx = pu.x - pt.x, dy = pu.y - pt.y, dz = pu.z - pt.z;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    const double r = sqrt(r2);
                    const double prefac = G / (r2 * r);
                    ax += prefac * pu.m * dx; ay += prefac * pu.m * dy; az += prefac * pu.m * dz;
                } else {     /* pair (i=t, j=u): d = p_t - p_u, acc_t -= prefac*m_u*d */
                    const double dx = pt.x - pu.x, dy = pt.y - pu.y, dz = pt.z - pu.z;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    const double r = sqrt(r2);
                    const double prefac = G / (r2 * r);
                    ax -= prefac * pu.m * dx; ay -= prefac * pu.m * dy; az -= prefac * pu.m * dz;
                }
            }
            ps[t].ax = ax; ps[t].ay = ay; ps[t].az = az;
        }
        free(massive);
    }

    const double mu = G * ps[0].m;        /* :106-107 source is always index 0 */
    to_jacobi(ps, pj, N, 1);              /* :108 */
    int qmax = 0;
    for (int i = 1; i < N; i++) {
        body corr;
        const int q = gr_jacobi_term(pj[i], mu, C2, max_iterations, &corr);
        if (q > qmax) qmax = q;
        pj[i].ax = corr.ax; pj[i].ay = corr.ay; pj[i].az = corr.az;
    }
    pj[0].ax = 0.; pj[0].ay = 0.; pj[0].az = 0.;   /* :158-160 */
    jacobi_to_inertial_acc(ps, pj, N);              /* :162 */
    for (int i = 0; i < N; i++) {
        particles[i].ax += ps[i].ax;
        particles[i].ay += ps[i].ay;
        particles[i].az += ps[i].az;
    }
    free(ps);
    free(pj);
    scatter_acc(s, particles);
    return qmax;
}

int orc_gr(const orc_sim* s, double c, int max_iterations) { return gr_impl(s, c, max_iterations, 0); }
int orc_gr_literal(const orc_sim* s, double c, int max_iterations) { return gr_impl(s, c, max_iterations, 1); }

double orc_newtonian_energy(const orc_sim* s) {
    double T = 0., V = 0.;
    for (int i = 0; i < s->N; i++) {
        T += 0.5 * s->m[i] * (s->vx[i] * s->vx[i] + s->vy[i] * s->vy[i] + s->vz[i] * s->vz[i]);
        for (int j = i + 1; j < s->N; j++) {
            const double dx = s->x[i] - s->x[j], dy = s->y[i] - s->y[j], dz = s->z[i] - s->z[j];
            V -= s->G * s->m[i] * s->m[j] / sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    return T + V;
}

/* :190-246 (softening = 0) */
double orc_gr_hamiltonian(const orc_sim* s, double c) {
    const int N = s->N;
    const double G = s->G, C2 = c * c;
    body* ps = gather(s);
    body* pj = (body*)malloc((size_t)N * sizeof(body));
    double* mj = (double*)malloc((size_t)N * sizeof(double));
    double V_newt = 0.;
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++) {
            const double dx = ps[i].x - ps[j].x, dy = ps[i].y - ps[j].y, dz = ps[i].z - ps[j].z;
            V_newt -= G * ps[i].m * ps[j].m / sqrt(dx * dx + dy * dy + dz * dz);
        }
    const double mu = G * ps[0].m;
    { /* src/rebxtools.c:27-35 */
        double eta = ps[0].m;
        for (int i = 1; i < N; i++) { mj[i] = ps[i].m * eta; eta += ps[i].m; mj[i] /= eta; }
        mj[0] = eta;
    }
    to_jacobi(ps, pj, N, 0);
    double T = 0.5 * mj[0] * (pj[0].vx * pj[0].vx + pj[0].vy * pj[0].vy + pj[0].vz * pj[0].vz);
    double V_PN = 0.;
    for (int i = 1; i < N; i++) {
        const body p = pj[i];
        const double rdoti2 = p.vx * p.vx + p.vy * p.vy + p.vz * p.vz;
        double vt2 = rdoti2, A, old;
        const double ri = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
        const double vscale2 = mu / ri;
        for (int q = 0; q < 10; q++) {
            old = vt2;
            A = (0.5 * vt2 + 3. * vscale2) / C2;
            vt2 = rdoti2 / ((1. - A) * (1. - A));
            if ((vt2 - old) / vt2 < DBL_EPSILON) break;
        }
        V_PN += mj[i] * (0.5 * mu * mu / (ri * ri) - 0.125 * vt2 * vt2 - 1.5 * mu * vt2 / ri);
        T += 0.5 * mj[i] * vt2;
    }
    V_PN /= C2;
    free(ps); free(pj); free(mj);
    return T + V_newt + V_PN;
}

/* ------------------------------------------------------------------------
 * gr_potential  (src/gr_potential.c:69-98) -- source is always index 0
 * ---------------------------------------------------------------------- */
int orc_gr_potential(const orc_sim* s, double c) {
    const int N = s->N;
    if (N <= 0) return 0;
    const double G = s->G, C2 = c * c;
    body* p = gather(s);
    const body S = p[0];
    const double prefac1 = 6. * (G * S.m) * (G * S.m) / C2;
    for (int i = 1; i < N; i++) {
        const body q = p[i];
        const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double prefac = prefac1 / (r2 * r2);
        p[i].ax -= prefac * dx;
        p[i].ay -= prefac * dy;
        p[i].az -= prefac * dz;
        p[0].ax += q.m / S.m * prefac * dx;
        p[0].ay += q.m / S.m * prefac * dy;
        p[0].az += q.m / S.m * prefac * dz;
    }
    scatter_acc(s, p);
    return 0;
}

/* :100-119 */
double orc_gr_potential_potential(const orc_sim* s, double c) {
    const double C2 = c * c, mu = s->G * s->m[0];
    const double prefac = 3. * mu * mu / C2;
    double H = 0.;
    for (int i = 1; i < s->N; i++) {
        const double dx = s->x[i] - s->x[0], dy = s->y[i] - s->y[0], dz = s->z[i] - s->z[0];
        const double r2 = dx * dx + dy * dy + dz * dz;
        H -= prefac * s->m[i] / r2;
    }
    return H;
}

/* ------------------------------------------------------------------------
 * gr_full  (src/gr_full.c:61-249)
 *
 * The reference keeps rs[N][N] and drs[N][N][3] on the stack and recomputes the
 * two potential sums inside the pair loop (O(N^3)).  The separations are pure
 * functions of the positions (drs[j][i] = -drs[i][j] exactly, rs symmetric) and
 * the potential sums do not depend on the pair, so evaluating them on the fly /
 * once per body gives the same bits in O(N^2) time and O(N) memory.
 * ---------------------------------------------------------------------- */
static inline void sep(const body* p, int i, int j, double* dx, double* dy, double* dz, double* r) {
    /* :78-88 -- drs[i][j] = p_i - p_j is formed for i<j and negated for j>i */
    if (i < j) {
        *dx = p[i].x - p[j].x; *dy = p[i].y - p[j].y; *dz = p[i].z - p[j].z;
    } else {
        *dx = -(p[j].x - p[i].x); *dy = -(p[j].y - p[i].y); *dz = -(p[j].z - p[i].z);
    }
    *r = sqrt((*dx) * (*dx) + (*dy) * (*dy) + (*dz) * (*dz));
}

/* :118-130 hoisted out of the pair loop: a2 sum for rows [row0,row1); a1 = 4*a2 exactly
 * (every term of the reference's a1 sum is the same term scaled by the power of two 4). */
void orc_gr_full_pot(const orc_sim* s, double c, int row0, int row1, double* pot) {
    const int N = s->N;
    const double G = s->G, C2 = c * c;
    body* p = gather(s);
    /* (rows are independent; threads only shorten the wait at large N -- each row's own sum keeps the reference's order) */
#pragma omp parallel for schedule(static) if (row1 - row0 > 256)
    for (int i = row0; i < row1; i++) {
        double s2 = 0.;
        for (int k = 0; k < N; k++) {
            if (k == i) continue;
            double dx, dy, dz, r;
            sep(p, i, k, &dx, &dy, &dz, &r);
            s2 += (1. / (C2)) * G * p[k].m / r;
        }
        pot[i] = s2;
    }
    free(p);
}

/* :104-170 for rows [row0,row1); pot must hold the a2 sums of ALL bodies */
void orc_gr_full_const(const orc_sim* s, double c, int row0, int row1, const double* pot, double* acx, double* acy, double* acz) {
    const int N = s->N;
    const double G = s->G, C2 = c * c;
    body* p = gather(s);
#pragma omp parallel for schedule(static) if (row1 - row0 > 256)
    for (int i = row0; i < row1; i++) {
        double cx = 0., cy = 0., cz = 0.;
        const double vi2 = p[i].vx * p[i].vx + p[i].vy * p[i].vy + p[i].vz * p[i].vz;
        const double a1 = 4. * pot[i];
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            double dx, dy, dz, r;
            sep(p, i, j, &dx, &dy, &dz, &r);
            const double rij2 = r * r;
            const double rij3 = rij2 * r;
            const double a3 = -vi2 / (C2);
            const double vj2 = p[j].vx * p[j].vx + p[j].vy * p[j].vy + p[j].vz * p[j].vz;
            const double a4 = -2. * vj2 / (C2);
            const double a5 = (4. / (C2)) * (p[i].vx * p[j].vx + p[i].vy * p[j].vy + p[i].vz * p[j].vz);
            const double a6_0 = dx * p[j].vx + dy * p[j].vy + dz * p[j].vz;
            const double a6 = (3. / (2. * C2)) * a6_0 * a6_0 / rij2;
            const double factor1 = a1 + pot[j] + a3 + a4 + a5 + a6;
            cx += G * p[j].m * dx * factor1 / rij3;
            cy += G * p[j].m * dy * factor1 / rij3;
            cz += G * p[j].m * dz * factor1 / rij3;
            const double dvx = p[i].vx - p[j].vx, dvy = p[i].vy - p[j].vy, dvz = p[i].vz - p[j].vz;
            const double factor2 = dx * (4. * p[i].vx - 3. * p[j].vx) + dy * (4. * p[i].vy - 3. * p[j].vy) + dz * (4. * p[i].vz - 3. * p[j].vz);
            cx += G * p[j].m * factor2 * dvx / rij3 / (C2);
            cy += G * p[j].m * factor2 * dvy / rij3 / (C2);
            cz += G * p[j].m * factor2 * dvz / rij3 / (C2);
        }
        acx[i] = cx; acy[i] = cy; acz[i] = cz;
    }
    free(p);
}

/* :181-220 for rows [row0,row1): one substitution sweep; returns 1 if any row in range fails
 * the stopping rule (signed ratio, threshold 1e-30), else 0 */
int orc_gr_full_sweep(const orc_sim* s, double c, int row0, int row1,
                      const double* anx, const double* any_, const double* anz,
                      const double* aox, const double* aoy, const double* aoz,
                      const double* acx, const double* acy, const double* acz,
                      double* outx, double* outy, double* outz) {
    const int N = s->N;
    const double G = s->G, C2 = c * c;
    body* p = gather(s);
    int unconverged = 0;
#pragma omp parallel for schedule(static) reduction(|:unconverged) if (row1 - row0 > 256)
    for (int i = row0; i < row1; i++) {
        double nx = 0., ny = 0., nz = 0.;
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            double dx, dy, dz, r;
            sep(p, i, j, &dx, &dy, &dz, &r);
            const double rij3 = r * r * r;
            const double tx = anx[j] + aox[j], ty = any_[j] + aoy[j], tz = anz[j] + aoz[j];
            nx += (G * p[j].m * dx / rij3) * (dx * tx + dy * ty + dz * tz) / (2. * C2) + (7. / (2. * C2)) * G * p[j].m * tx / r;
            ny += (G * p[j].m * dy / rij3) * (dx * tx + dy * ty + dz * tz) / (2. * C2) + (7. / (2. * C2)) * G * p[j].m * ty / r;
            nz += (G * p[j].m * dz / rij3) * (dx * tx + dy * ty + dz * tz) / (2. * C2) + (7. / (2. * C2)) * G * p[j].m * tz / r;
        }
        const double a_new[3] = {acx[i] + nx, acy[i] + ny, acz[i] + nz};
        const double a_old[3] = {aox[i], aoy[i], aoz[i]};
        for (int d = 0; d < 3; d++) {
            const double dev = (fabs(a_new[d]) < 1.e-30) ? 0. : fabs(a_new[d] - a_old[d]) / a_new[d];
            if (dev >= 1.e-30) unconverged = 1;
        }
        outx[i] = a_new[0]; outy[i] = a_new[1]; outz[i] = a_new[2];
    }
    free(p);
    return unconverged;
}

/* :69-102: the Newtonian term is the incoming acceleration, plus the 0-1 pair when REBOUND's
 * gravity routine left it out (WHFast in Jacobi coordinates) */
void orc_gr_full_newton(const orc_sim* s, int gravity_ignore_10, double* anx, double* any_, double* anz) {
    const int N = s->N;
    for (int i = 0; i < N; i++) { anx[i] = s->ax[i]; any_[i] = s->ay[i]; anz[i] = s->az[i]; }
    if (gravity_ignore_10 && N > 1) {
        body* p = gather(s);
        double dx, dy, dz, r;
        sep(p, 0, 1, &dx, &dy, &dz, &r);
        const double prefact = -s->G / (r * r * r);
        const double prefact0 = prefact * p[0].m;
        const double prefact1 = prefact * p[1].m;
        anx[0] += prefact1 * dx; any_[0] += prefact1 * dy; anz[0] += prefact1 * dz;
        anx[1] -= prefact0 * dx; any_[1] -= prefact0 * dy; anz[1] -= prefact0 * dz;
        free(p);
    }
}

int orc_gr_full(const orc_sim* s, double c, int gravity_ignore_10) {
    const int N = s->N;
    if (N <= 0) return 0;
    double* w = (double*)calloc((size_t)13 * (size_t)N, sizeof(double));
    double *anx = w, *any_ = w + N, *anz = w + 2 * N, *pot = w + 3 * N;
    double *acx = w + 4 * N, *acy = w + 5 * N, *acz = w + 6 * N;
    double *aox = w + 7 * N, *aoy = w + 8 * N, *aoz = w + 9 * N;       /* a_new starts at zero (:74-76) */
    double *nwx = w + 10 * N, *nwy = w + 11 * N, *nwz = w + 12 * N;
    orc_gr_full_newton(s, gravity_ignore_10, anx, any_, anz);
    orc_gr_full_pot(s, c, 0, N, pot);
    orc_gr_full_const(s, c, 0, N, pot, acx, acy, acz);
    int sweeps = 0;
    for (int k = 0; k < 10; k++) {          /* :173 the loop bound is the literal 10 */
        sweeps = k + 1;
        const int unconverged = orc_gr_full_sweep(s, c, 0, N, anx, any_, anz, aox, aoy, aoz, acx, acy, acz, nwx, nwy, nwz);
        double* t;
        t = aox; aox = nwx; nwx = t; t = aoy; aoy = nwy; nwy = t; t = aoz; aoz = nwz; nwz = t;
        if (!unconverged) break;
    }
    for (int i = 0; i < N; i++) {            /* :226-230 */
        s->ax[i] += aox[i]; s->ay[i] += aoy[i]; s->az[i] += aoz[i];
    }
    free(w);
    return sweeps;
}

/* ------------------------------------------------------------------------
 * central_force  (src/central_force.c:63-95)
 * ---------------------------------------------------------------------- */
int orc_central_force(const orc_sim* s, const double* Acentral, const double* gammacentral) {
    const int N = s->N;
    if (N <= 0) return 0;
    body* p = gather(s);
    for (int k = 0; k < N; k++) {
        if (!(has(Acentral, k) && has(gammacentral, k))) continue;
        const double A = Acentral[k], gamma = gammacentral[k];
        const body S = p[k];
        for (int i = 0; i < N; i++) {
            if (i == k) continue;
            const body q = p[i];
            const double dx = q.x - S.x, dy = q.y - S.y, dz = q.z - S.z;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double prefac = A * pow(r2, (gamma - 1.) / 2.);
            p[i].ax += prefac * dx;
            p[i].ay += prefac * dy;
            p[i].az += prefac * dz;
            p[k].ax -= q.m / S.m * prefac * dx;
            p[k].ay -= q.m / S.m * prefac * dy;
            p[k].az -= q.m / S.m * prefac * dz;
        }
    }
    scatter_acc(s, p);
    return 0;
}

/* :97-141 */
double orc_central_force_potential(const orc_sim* s, const double* Acentral, const double* gammacentral) {
    double Htot = 0.;
    for (int k = 0; k < s->N; k++) {
        if (!(has(Acentral, k) && has(gammacentral, k))) continue;
        const double A = Acentral[k], gamma = gammacentral[k];
        double H = 0.;
        for (int i = 0; i < s->N; i++) {
            if (i == k) continue;
            const double dx = s->x[i] - s->x[k], dy = s->y[i] - s->y[k], dz = s->z[i] - s->z[k];
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (fabs(gamma + 1.) < DBL_EPSILON) H -= s->m[i] * A * log(sqrt(r2));
            else H -= s->m[i] * A * pow(r2, (gamma + 1.) / 2.) / (gamma + 1.);
        }
        Htot += H;
    }
    return Htot;
}

/* ------------------------------------------------------------------------
 * tides_precession  (src/tides_precession.c:65-130)
 * ---------------------------------------------------------------------- */
static void tides_pass(body* p, int N, double G, const double* R_tides, const double* k1, int src) {
    const double m0 = p[src].m;
    const double R0 = has(R_tides, src) ? R_tides[src] : 0.;
    const double k10 = has(k1, src) ? k1[src] : 0.;
    const double fac0 = k10 * R0 * R0 * R0 * R0 * R0;
    for (int i = 0; i < N; i++) {
        if (i == src) continue;
        const double mratio = p[i].m / m0;
        if (mratio < DBL_MIN) continue;          /* :84-87 massless bodies feel nothing */
        double fac = 0.;
        fac += fac0 * mratio;
        const double Rp = has(R_tides, i) ? R_tides[i] : 0.;
        const double k1p = has(k1, i) ? k1[i] : 0.;
        fac += k1p * Rp * Rp * Rp * Rp * Rp / mratio;
        const double dx = p[i].x - p[src].x, dy = p[i].y - p[src].y, dz = p[i].z - p[src].z;
        const double dr2 = dx * dx + dy * dy + dz * dz;
        const double prefac = -3 * G * (m0 + p[i].m) / (dr2 * dr2 * dr2 * dr2) * fac;
        p[i].ax += prefac * dx;
        p[i].ay += prefac * dy;
        p[i].az += prefac * dz;
        p[src].ax -= mratio * prefac * dx;
        p[src].ay -= mratio * prefac * dy;
        p[src].az -= mratio * prefac * dz;
    }
}

int orc_tides_precession(const orc_sim* s, const double* R_tides, const double* k1, const uint8_t* tides_primary) {
    if (s->N <= 0) return 0;
    body* p = gather(s);
    int found = 0;
    for (int i = 0; i < s->N; i++) {
        if (flagged(tides_primary, i)) {
            found = 1;
            tides_pass(p, s->N, s->G, R_tides, k1, i);
        }
    }
    if (!found) tides_pass(p, s->N, s->G, R_tides, k1, 0);
    scatter_acc(s, p);
    return 0;
}

/* :132-202 -- note the potential looks for `primary`, the force for `tides_primary` */
double orc_tides_precession_potential(const orc_sim* s, const double* R_tides, const double* k1, const uint8_t* primary) {
    int src = 0;
    for (int i = 0; i < s->N; i++) if (flagged(primary, i)) src = i;   /* the last flagged one wins (H = ...) */
    const double m0 = s->m[src];
    const double R0 = has(R_tides, src) ? R_tides[src] : 0.;
    const double k10 = has(k1, src) ? k1[src] : 0.;
    const double fac0 = k10 * R0 * R0 * R0 * R0 * R0;
    double H = 0.;
    for (int i = 0; i < s->N; i++) {
        if (i == src) continue;
        const double mratio = s->m[i] / m0;
        if (mratio < DBL_MIN) continue;
        double fac = 0.;
        fac += fac0 * mratio;
        const double Rp = has(R_tides, i) ? R_tides[i] : 0.;
        const double k1p = has(k1, i) ? k1[i] : 0.;
        fac += k1p * Rp * Rp * Rp * Rp * Rp / mratio;
        const double dx = s->x[i] - s->x[src], dy = s->y[i] - s->y[src], dz = s->z[i] - s->z[src];
        const double dr2 = dx * dx + dy * dy + dz * dz;
        H += -3. / 6. * s->G * (m0 + s->m[i]) * s->m[i] / (dr2 * dr2 * dr2) * fac;
    }
    return H;
}

/* ------------------------------------------------------------------------
 * modify_orbits_forces  (src/modify_orbits_forces.c:77-129, src/rebxtools.c:45-126)
 * ---------------------------------------------------------------------- */

/* modify_orbits_forces.c:77-118 -- force on p relative to `ref`; unset taus are +inf */
static void mof_term(const body* p, const body* ref, double tau_a, double tau_e, double tau_inc, double a[3]) {
    const double dvx = p->vx - ref->vx, dvy = p->vy - ref->vy, dvz = p->vz - ref->vz;
    const double dx = p->x - ref->x, dy = p->y - ref->y, dz = p->z - ref->z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    a[0] = dvx / (2. * tau_a);
    a[1] = dvy / (2. * tau_a);
    a[2] = dvz / (2. * tau_a);
    if (tau_e < INFINITY || tau_inc < INFINITY) {
        const double vdotr = dx * dvx + dy * dvy + dz * dvz;
        const double prefac = 2 * vdotr / r2 / tau_e;
        a[0] += prefac * dx;
        a[1] += prefac * dy;
        a[2] += prefac * dz + 2. * dvz / tau_inc;
    }
}

int orc_modify_orbits_forces(const orc_sim* s, int coordinates, const double* tau_a, const double* tau_e,
                             const double* tau_inc, const uint8_t* primary) {
    const int N = s->N;
    if (N <= 0) return 0;
    body* p = gather(s);
    body com = com_of_all(p, N);                 /* rebxtools.c:47 */
    int refindex = -1;
    if (coordinates == ORC_COORD_JACOBI) {
        refindex = 0;
    } else if (coordinates == ORC_COORD_PARTICLE) {
        for (int i = 0; i < N; i++) {
            if (flagged(primary, i)) { com = p[i]; refindex = i; break; }
        }
        if (refindex < 0) { free(p); return -1; }  /* reference reports an error (and then indexes [-1]) */
    } else if (coordinates != ORC_COORD_BARYCENTRIC) {
        free(p);
        return -2;
    }
    for (int i = N - 1; i >= 0; i--) {           /* :71 backwards */
        if (i == refindex) continue;
        if (coordinates == ORC_COORD_JACOBI) com = com_without(com, p[i]);
        const double ta = has(tau_a, i) ? tau_a[i] : INFINITY;
        const double te = has(tau_e, i) ? tau_e[i] : INFINITY;
        const double ti = has(tau_inc, i) ? tau_inc[i] : INFINITY;
        double a[3];
        mof_term(&p[i], &com, ta, te, ti, a);
        p[i].ax += a[0]; p[i].ay += a[1]; p[i].az += a[2];
        double massratio;
        switch (coordinates) {
        case ORC_COORD_BARYCENTRIC:
            massratio = p[i].m / com.m;
            if (massratio != 0.) {               /* a zero ratio only subtracts signed zeros */
                for (int j = 0; j < N; j++) {
                    p[j].ax -= massratio * a[0]; p[j].ay -= massratio * a[1]; p[j].az -= massratio * a[2];
                }
            }
            break;
        case ORC_COORD_JACOBI:
            massratio = p[i].m / (com.m + p[i].m);   /* back_reactions_inclusive = 1 */
            if (massratio != 0.) {
                for (int j = 0; j < i + 1; j++) {
                    p[j].ax -= massratio * a[0]; p[j].ay -= massratio * a[1]; p[j].az -= massratio * a[2];
                }
            }
            break;
        default: /* PARTICLE */
            massratio = p[i].m / (com.m + p[i].m);
            p[i].ax -= massratio * a[0]; p[i].ay -= massratio * a[1]; p[i].az -= massratio * a[2];
            p[refindex].ax -= massratio * a[0]; p[refindex].ay -= massratio * a[1]; p[refindex].az -= massratio * a[2];
            break;
        }
    }
    scatter_acc(s, p);
    return 0;
}

void orc_reset_accelerations(const orc_sim* s) {
    for (int i = 0; i < s->N; i++) { s->ax[i] = 0.; s->ay[i] = 0.; s->az[i] = 0.; }
}

void orc_newtonian_gravity(const orc_sim* s) {
    const int N = s->N;
    int nm = 0;
    int* massive = (int*)malloc((size_t)(N > 0 ? N : 1) * sizeof(int));
    for (int i = 0; i < N; i++) if (s->m[i] != 0.) massive[nm++] = i;
    for (int t = 0; t < N; t++) {
        double ax = 0., ay = 0., az = 0.;
        for (int k = 0; k < nm; k++) {
            const int u = massive[k];
            if (u == t) continue;
            const double dx = s->x[u] - s->x[t], dy = s->y[u] - s->y[t], dz = s->z[u] - s->z[t];
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double pre = s->G * s->m[u] / (r2 * sqrt(r2));
            ax += pre * dx; ay += pre * dy; az += pre * dz;
        }
        s->ax[t] += ax; s->ay[t] += ay; s->az[t] += az;
    }
    free(massive);
}

/* ------------------------------------------------------------------------
 * Operators modify_mass (src/modify_mass.c:63-73) and modify_orbits_direct
 * (src/modify_orbits_direct.c:81-132 through rebxtools_com_ptm, src/rebxtools.c:137-215).
 * They call REBOUND for the centre-of-mass shift and the orbit <-> Cartesian conversions
 * (reb_move_to_com, reb_tools_particle_to_orbit_err, reb_tools_orbit_to_particle), restated
 * here from REBOUND 3.x's published algorithm (Murray & Dermott eqs. 2.122, 2.36) and pinned by
 * replaying the reference's archives modify_mass.sa / modify_orbits_direct.sa.
 * ---------------------------------------------------------------------- */

static void scatter_state(const orc_sim* s, const body* p) {
    for (int i = 0; i < s->N; i++) {
        s->x[i] = p[i].x; s->y[i] = p[i].y; s->z[i] = p[i].z;
        s->vx[i] = p[i].vx; s->vy[i] = p[i].vy; s->vz[i] = p[i].vz;
        s->m[i] = p[i].m;
    }
}

/* modify_mass.c:63-73 followed by reb_move_to_com */
void orc_modify_mass(const orc_sim* s, const double* tau_mass, double dt) {
    const int N = s->N;
    if (N <= 0) return;
    body* p = gather(s);
    for (int i = 0; i < N; i++)
        if (has(tau_mass, i)) p[i].m += p[i].m * dt / tau_mass[i];
    const body com = com_of_all(p, N);
    for (int i = 0; i < N; i++) {
        p[i].x -= com.x; p[i].y -= com.y; p[i].z -= com.z;
        p[i].vx -= com.vx; p[i].vy -= com.vy; p[i].vz -= com.vz;
    }
    scatter_state(s, p);
    free(p);
}

typedef struct { double d, a, e, inc, Omega, omega, f; } orbit;

/* angle in [-pi, pi] from its cosine num/denom and the sign of `disambiguator` */
static double acos2(double num, double denom, double disambiguator) {
    const double cosine = num / denom;
    if (cosine > -1. && cosine < 1.) {
        const double val = acos(cosine);
        return disambiguator < 0. ? -val : val;
    }
    return (cosine <= -1.) ? ORC_PI : 0.;
}

/* reb_tools_particle_to_orbit_err: returns nonzero when the orbit is undefined */
static int particle_to_orbit(double G, body p, body primary, orbit* o) {
    const double TINY = 1.E-308, MIN_INC = 1.e-8;
    if (primary.m <= TINY) return 1;
    const double mu = G * (p.m + primary.m);
    const double dx = p.x - primary.x, dy = p.y - primary.y, dz = p.z - primary.z;
    const double dvx = p.vx - primary.vx, dvy = p.vy - primary.vy, dvz = p.vz - primary.vz;
    o->d = sqrt(dx * dx + dy * dy + dz * dz);
    const double vsquared = dvx * dvx + dvy * dvy + dvz * dvz;
    const double vcircsquared = mu / o->d;
    o->a = -mu / (vsquared - 2. * vcircsquared);
    const double hx = (dy * dvz - dz * dvy), hy = (dz * dvx - dx * dvz), hz = (dx * dvy - dy * dvx);
    const double h = sqrt(hx * hx + hy * hy + hz * hz);
    const double vdiffsquared = vsquared - vcircsquared;
    if (o->d <= TINY) return 2;
    const double vr = (dx * dvx + dy * dvy + dz * dvz) / o->d;
    const double rvr = o->d * vr;
    const double muinv = 1. / mu;
    const double ex = muinv * (vdiffsquared * dx - rvr * dvx);
    const double ey = muinv * (vdiffsquared * dy - rvr * dvy);
    const double ez = muinv * (vdiffsquared * dz - rvr * dvz);
    o->e = sqrt(ex * ex + ey * ey + ez * ez);
    o->inc = acos2(hz, h, 1.);
    const double nx = -hy, ny = hx;
    const double n = sqrt(nx * nx + ny * ny);
    o->Omega = acos2(nx, n, ny);
    if (o->inc < MIN_INC || o->inc > ORC_PI - MIN_INC) {
        /* nearly planar: longitudes are well defined, node-referenced angles are not */
        const double theta = acos2(dx, o->d, dy);
        const double pomega = acos2(ex, o->e, ey);
        if (o->inc < ORC_PI / 2.) { o->omega = pomega - o->Omega; o->f = theta - pomega; }
        else { o->omega = o->Omega - pomega; o->f = pomega - theta; }
    } else {
        const double wpf = acos2(nx * dx + ny * dy, n * o->d, dz);
        o->omega = acos2(nx * ex + ny * ey, n * o->e, ez);
        o->f = wpf - o->omega;
    }
    return 0;
}

/* reb_tools_orbit_to_particle: Murray & Dermott eq. 2.122 (positions), 2.36 rotated (velocities) */
static body orbit_to_particle(double G, body primary, double m, double a, double e, double inc, double Omega, double omega, double f) {
    body p;
    memset(&p, 0, sizeof(p));
    p.m = m;
    const double r = a * (1 - e * e) / (1 + e * cos(f));
    const double v0 = sqrt(G * (m + primary.m) / a / (1. - e * e));
    const double cO = cos(Omega), sO = sin(Omega), co = cos(omega), so = sin(omega), cf = cos(f), sf = sin(f), ci = cos(inc), si = sin(inc);
    p.x = primary.x + r * (cO * (co * cf - so * sf) - sO * (so * cf + co * sf) * ci);
    p.y = primary.y + r * (sO * (co * cf - so * sf) + cO * (so * cf + co * sf) * ci);
    p.z = primary.z + r * (so * cf + co * sf) * si;
    p.vx = primary.vx + v0 * ((e + cf) * (-ci * co * sO - cO * so) - sf * (co * cO - ci * so * sO));
    p.vy = primary.vy + v0 * ((e + cf) * (ci * co * cO - sO * so) - sf * (co * sO + ci * so * cO));
    p.vz = primary.vz + v0 * ((e + cf) * co * si - sf * si * so);
    return p;
}

/* modify_orbits_direct.c:81-119: one body's new state relative to `primary` */
static body mod_direct_step(double G, body p, body primary, double dt, double p_param, int has_p,
                            const double* tau_a, const double* tau_e, const double* tau_inc, const double* tau_omega, const double* tau_Omega, int i) {
    orbit o;
    if (particle_to_orbit(G, p, primary, &o)) return p;
    const double a0 = o.a, e0 = o.e, inc0 = o.inc;
    if (has(tau_a, i)) o.a += a0 * dt / tau_a[i];
    if (has(tau_e, i)) o.e += e0 * dt / tau_e[i];
    if (has(tau_inc, i)) o.inc += inc0 * dt / tau_inc[i];
    if (has(tau_omega, i)) o.omega += 2. * ORC_PI * dt / tau_omega[i];
    if (has(tau_Omega, i)) o.Omega += 2. * ORC_PI * dt / tau_Omega[i];
    if (has(tau_e, i) && has_p) o.a += 2. * a0 * e0 * e0 * p_param * dt / tau_e[i];
    body out = orbit_to_particle(G, primary, p.m, o.a, o.e, o.inc, o.Omega, o.omega, o.f);
    out.ax = p.ax; out.ay = p.ay; out.az = p.az;
    return out;
}

static void subtract_posvel(body* p, const body* diff, double massratio) {
    p->x -= massratio * diff->x; p->y -= massratio * diff->y; p->z -= massratio * diff->z;
    p->vx -= massratio * diff->vx; p->vy -= massratio * diff->vy; p->vz -= massratio * diff->vz;
}

/* rebxtools_com_ptm (rebxtools.c:137-215) with modify_orbits_direct's step; p_param = NaN when unset */
int orc_modify_orbits_direct(const orc_sim* s, int coordinates, double p_param, const double* tau_a, const double* tau_e, const double* tau_inc,
                             const double* tau_omega, const double* tau_Omega, const uint8_t* primary, double dt) {
    const int N = s->N;
    if (N <= 0) return 0;
    body* p = gather(s);
    body com = com_of_all(p, N);
    int refindex = -1;
    if (coordinates == ORC_COORD_JACOBI) refindex = 0;
    else if (coordinates == ORC_COORD_PARTICLE) {
        for (int i = 0; i < N; i++) if (flagged(primary, i)) { com = p[i]; refindex = i; break; }
        if (refindex < 0) { free(p); return -1; }
    } else if (coordinates != ORC_COORD_BARYCENTRIC) { free(p); return -2; }
    const int has_p = !isnan(p_param);
    for (int i = N - 1; i >= 0; i--) {
        if (i == refindex) continue;
        if (coordinates == ORC_COORD_JACOBI) com = com_without(com, p[i]);
        const body mod = mod_direct_step(s->G, p[i], com, dt, p_param, has_p, tau_a, tau_e, tau_inc, tau_omega, tau_Omega, i);
        body diff;
        diff.x = mod.x - p[i].x; diff.y = mod.y - p[i].y; diff.z = mod.z - p[i].z;
        diff.vx = mod.vx - p[i].vx; diff.vy = mod.vy - p[i].vy; diff.vz = mod.vz - p[i].vz;
        p[i].x = mod.x; p[i].y = mod.y; p[i].z = mod.z; p[i].vx = mod.vx; p[i].vy = mod.vy; p[i].vz = mod.vz;
        double massratio;
        switch (coordinates) {
        case ORC_COORD_BARYCENTRIC:
            massratio = p[i].m / com.m;
            for (int j = 0; j < N; j++) subtract_posvel(&p[j], &diff, massratio);
            break;
        case ORC_COORD_JACOBI:
            massratio = p[i].m / (com.m + p[i].m);
            for (int j = 0; j < i + 1; j++) subtract_posvel(&p[j], &diff, massratio);
            break;
        default:
            massratio = p[i].m / (com.m + p[i].m);
            subtract_posvel(&p[i], &diff, massratio);
            subtract_posvel(&p[refindex], &diff, massratio);
            break;
        }
    }
    scatter_state(s, p);
    free(p);
    return 0;
}

/* gr_full.c:251-377 -- the Hamiltonian gr_full derives from.  The canonical velocities vtilde
 * are found by ten Gauss-Seidel sweeps (each body is updated in place and the next body already
 * sees the new value), then kinetic, Newtonian and post-Newtonian parts are summed. */
double orc_gr_full_hamiltonian(const orc_sim* s, double c) {
    const int N = s->N;
    const double G = s->G, C2 = c * c;
    body* p = gather(s);
    double* vt = (double*)malloc((size_t)(N > 0 ? N : 1) * 3 * sizeof(double));
    for (int j = 0; j < N; j++) { vt[3 * j] = p[j].vx; vt[3 * j + 1] = p[j].vy; vt[3 * j + 2] = p[j].vz; }
    for (int q = 0; q < 10; q++) {
        for (int i = 0; i < N; i++) {
            const body pi = p[i];
            const double vt2 = vt[3 * i] * vt[3 * i] + vt[3 * i + 1] * vt[3 * i + 1] + vt[3 * i + 2] * vt[3 * i + 2];
            const double A = (1. - 0.5 * vt2 / C2);
            double dx = 0., dy = 0., dz = 0.;
            for (int j = 0; j < N; j++) {
                if (j == i) continue;
                const body pj = p[j];
                const double xij = pj.x - pi.x, yij = pj.y - pi.y, zij = pj.z - pi.z;
                const double rij2 = xij * xij + yij * yij + zij * zij;
                const double rij = sqrt(rij2);
                const double rijdotvj = vt[3 * j] * xij + vt[3 * j + 1] * yij + vt[3 * j + 2] * zij;
                const double pfac = pj.m / rij;
                dx += pfac * (6. * vt[3 * i] - 7. * vt[3 * j] - rijdotvj * xij / rij2);
                dy += pfac * (6. * vt[3 * i + 1] - 7. * vt[3 * j + 1] - rijdotvj * yij / rij2);
                dz += pfac * (6. * vt[3 * i + 2] - 7. * vt[3 * j + 2] - rijdotvj * zij / rij2);
            }
            dx *= G / (2. * C2); dy *= G / (2. * C2); dz *= G / (2. * C2);
            vt[3 * i] = (pi.vx + dx) / A;
            vt[3 * i + 1] = (pi.vy + dy) / A;
            vt[3 * i + 2] = (pi.vz + dz) / A;
        }
    }
    double e_kin = 0., e_pot = 0., e_pn = 0.;
    for (int i = 0; i < N; i++) {
        const double vt2 = vt[3 * i] * vt[3 * i] + vt[3 * i + 1] * vt[3 * i + 1] + vt[3 * i + 2] * vt[3 * i + 2];
        e_kin += 0.5 * p[i].m * vt2;
    }
    for (int i = 0; i < N; i++) {
        const body pi = p[i];
        double sumk = 0.;
        for (int k = 0; k < N; k++) {
            if (k == i) continue;
            const double xik = p[k].x - pi.x, yik = p[k].y - pi.y, zik = p[k].z - pi.z;
            sumk -= 2. * G * p[k].m / sqrt(xik * xik + yik * yik + zik * zik);
        }
        const double vt2 = vt[3 * i] * vt[3 * i] + vt[3 * i + 1] * vt[3 * i + 1] + vt[3 * i + 2] * vt[3 * i + 2];
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const body pj = p[j];
            const double xij = pj.x - pi.x, yij = pj.y - pi.y, zij = pj.z - pi.z;
            const double rij2 = xij * xij + yij * yij + zij * zij;
            const double rij = sqrt(rij2);
            const double rijdotvj = vt[3 * j] * xij + vt[3 * j + 1] * yij + vt[3 * j + 2] * zij;
            const double rijdotvi = vt[3 * i] * xij + vt[3 * i + 1] * yij + vt[3 * i + 2] * zij;
            const double vidotvj = vt[3 * i] * vt[3 * j] + vt[3 * i + 1] * vt[3 * j + 1] + vt[3 * i + 2] * vt[3 * j + 2];
            e_pn -= G / (4. * C2) * pi.m * pj.m / rij * (6. * vt2 - 7 * vidotvj - rijdotvi * rijdotvj / rij2 + sumk);
        }
        e_pn -= pi.m / (8. * C2) * vt2 * vt2;
        for (int j = i + 1; j < N; j++) {
            const double dx = pi.x - p[j].x, dy = pi.y - p[j].y, dz = pi.z - p[j].z;
            e_pot -= G * pi.m * p[j].m / sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    free(vt);
    free(p);
    return e_kin + e_pot + e_pn;
}

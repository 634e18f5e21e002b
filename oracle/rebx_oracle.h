/*
 * rebx_oracle.h -- CPU restatement of the REBOUNDx additional-force path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against; it is never linked into or called by the product library.  Only
 * tests/, the smoke check and the CPU-baseline leg of the benchmark load it.
 *
 * Every routine is a plain-C, single-threaded, fp64 restatement of one
 * reference function; the reference location is cited beside each one.
 * Summation and evaluation order follow the reference expression by
 * expression so that, compiled without FMA contraction, results are bitwise
 * what the reference computes for the same inputs.
 *
 * Parity pinning: the reference sources cannot be built in this image
 * (they need rebound.h / librebound from the REBOUND package, pinned as
 * rebound>=3.10.0 in the reference setup.py:111, which is not vendored and not
 * installed).  The oracle is therefore pinned by the reference's own recorded
 * answers (see tests/test_oracle.py) and by the potential/force
 * consistency its conservation tests rely on.  The four REBOUND helpers
 * (centre of mass, Jacobi transforms) are restated from REBOUND 3.x's
 * published algorithm; for those, parity is unpinned at the instruction level.
 *
 * Conventions: particles are passed as SoA columns.  Per-particle double
 * parameters are columns in which NaN means "parameter not set on this
 * particle"; a NULL column means "not set on any particle".  Presence-only
 * parameters (radiation_source, tides_primary, primary) are uint8 columns
 * (non-zero = set) or NULL.
 */
#ifndef REBX_ORACLE_H
#define REBX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sim {
    int N;          /* real particles (N - N_var in the reference) */
    double G;
    double *x, *y, *z;
    double *vx, *vy, *vz;
    double *m;
    double *ax, *ay, *az;   /* accumulated into (+=), as the callback contract demands */
} orc_sim;

enum { ORC_COORD_JACOBI = 0, ORC_COORD_BARYCENTRIC = 1, ORC_COORD_PARTICLE = 2 };

/* src/radiation_forces.c:69-117 */
int orc_radiation_forces(const orc_sim* s, double c, const double* beta, const uint8_t* radiation_source);
/* src/gravitational_harmonics.c:61-140 */
int orc_gravitational_harmonics(const orc_sim* s, const double* J2, const double* J4, const double* R_eq);
/* src/gr.c:73-188; returns the largest inner-iteration count any particle used */
int orc_gr(const orc_sim* s, double c, int max_iterations);
/* same arithmetic with the reference's literal all-pairs double loop (small N only) */
int orc_gr_literal(const orc_sim* s, double c, int max_iterations);
/* src/gr_potential.c:69-98 */
int orc_gr_potential(const orc_sim* s, double c);
/* src/gr_full.c:61-249; returns the number of substitution sweeps executed (1..10) */
int orc_gr_full(const orc_sim* s, double c, int gravity_ignore_10);
/* the phases of gr_full on row ranges (what a row-sharded driver composes); see rebx_oracle.c */
void orc_gr_full_newton(const orc_sim* s, int gravity_ignore_10, double* anx, double* any_, double* anz);
void orc_gr_full_pot(const orc_sim* s, double c, int row0, int row1, double* pot);
void orc_gr_full_const(const orc_sim* s, double c, int row0, int row1, const double* pot, double* acx, double* acy, double* acz);
int orc_gr_full_sweep(const orc_sim* s, double c, int row0, int row1,
                      const double* anx, const double* any_, const double* anz,
                      const double* aox, const double* aoy, const double* aoz,
                      const double* acx, const double* acy, const double* acz,
                      double* outx, double* outy, double* outz);
/* src/central_force.c:63-95 */
int orc_central_force(const orc_sim* s, const double* Acentral, const double* gammacentral);
/* src/tides_precession.c:65-130 */
int orc_tides_precession(const orc_sim* s, const double* R_tides, const double* k1, const uint8_t* tides_primary);
/* src/modify_orbits_forces.c:77-129 + src/rebxtools.c:45-126; returns -1 when
 * PARTICLE coordinates are requested and no particle carries `primary` */
int orc_modify_orbits_forces(const orc_sim* s, int coordinates, const double* tau_a, const double* tau_e,
                             const double* tau_inc, const uint8_t* primary);
/* The same two forces evaluated in IEEE binary128 (rebx_oracle_hp.c): what gr (src/gr.c:73-171) and
 * modify_orbits_forces in Jacobi or barycentric coordinates (src/modify_orbits_forces.c:77-118,
 * src/rebxtools.c:45-126) ADD to every body's acceleration, as pairs of doubles: out[0..2] = leading
 * parts of x, y, z, out[3..5] = remainders (value = hi + lo), each an array of N.  A yardstick for the
 * rounding error of the fp64 evaluations (sequential here, parallel sums on the device). */
int orc_hp_gr(const orc_sim* s, double c, int max_iterations, double* const out[6]);
int orc_hp_modify_orbits_forces(const orc_sim* s, int coordinates, const double* tau_a, const double* tau_e, const double* tau_inc,
                                double* const out[6]);
/* src/core.c:860-866 */
void orc_reset_accelerations(const orc_sim* s);

/* "integrate a force across a timestep" operator (src/integrate_force.c:34-78) with the schemes
 * of src/integrator_{euler,rk2,rk4,implicit_midpoint}.c; enum values of src/reboundx.h:157-163.
 * `f` adds the force's accelerations to the acceleration columns of the state it is handed
 * (the reference's update_accelerations callback). */
enum { ORC_INTEGRATOR_NONE = -1, ORC_INTEGRATOR_IMPLICIT_MIDPOINT = 0, ORC_INTEGRATOR_RK4 = 1, ORC_INTEGRATOR_EULER = 2, ORC_INTEGRATOR_RK2 = 3 };
typedef void (*orc_force_fn)(orc_sim* state, void* user);
int orc_integrate_force(orc_sim* s, int integrator, double dt, orc_force_fn f, void* user);
void orc_drift_step(orc_sim* s, double dt);   /* src/steppers.c:112-121 */
void orc_kick_step(orc_sim* s, double dt);    /* src/steppers.c:123-133 (acceleration columns as given) */

/* operators: src/modify_mass.c:63-73 (+ reb_move_to_com) and src/modify_orbits_direct.c:81-132 with
 * src/rebxtools.c:137-215.  They change positions, velocities and masses in place.  p_param = NaN
 * when the operator's "p" parameter is unset. */
void orc_modify_mass(const orc_sim* s, const double* tau_mass, double dt);
int orc_modify_orbits_direct(const orc_sim* s, int coordinates, double p_param, const double* tau_a, const double* tau_e, const double* tau_inc,
                             const double* tau_omega, const double* tau_Omega, const uint8_t* primary, double dt);

/* Wisdom-Holman steps (Jacobi coordinates, REBOUND's WHFast defaults) used to replay the
 * reference's trajectory archives; see rebx_whfast.c. */
typedef void (*orc_step_fn)(orc_sim* state, double dt, void* user);
int orc_whfast_steps(orc_sim* s, double dt, long nsteps, orc_force_fn f, orc_step_fn pre, orc_step_fn post, void* user);

/* potentials used by the reference's conservation tests */
double orc_gravitational_harmonics_potential(const orc_sim* s, const double* J2, const double* J4, const double* R_eq); /* gravitational_harmonics.c:142-231 */
double orc_central_force_potential(const orc_sim* s, const double* Acentral, const double* gammacentral);               /* central_force.c:97-141 */
double orc_tides_precession_potential(const orc_sim* s, const double* R_tides, const double* k1, const uint8_t* primary); /* tides_precession.c:132-202 */
double orc_gr_potential_potential(const orc_sim* s, double c);                                                          /* gr_potential.c:100-132 */
double orc_gr_hamiltonian(const orc_sim* s, double c);                                                                  /* gr.c:190-260 */
double orc_gr_full_hamiltonian(const orc_sim* s, double c);                                                             /* gr_full.c:251-377 */
double orc_newtonian_energy(const orc_sim* s);

/* src/radiation_forces.c:119-124 */
double orc_rad_calc_beta(double G, double c, double source_mass, double source_luminosity, double radius, double density, double Q_pr);
double orc_rad_calc_particle_radius(double G, double c, double source_mass, double source_luminosity, double beta, double density, double Q_pr);

/* plain Newtonian all-pairs gravity (what REBOUND adds before the hook runs); used to
 * build realistic incoming accelerations in tests.  Only massive particles act as sources. */
void orc_newtonian_gravity(const orc_sim* s);

#ifdef __cplusplus
}
#endif
#endif

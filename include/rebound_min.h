/*
 * rebound_min.h -- the slice of REBOUND's public types that the additional-force
 * path touches, for building and testing this library where the REBOUND package
 * (a third-party dependency of the reference, not part of its tree) is absent.
 *
 * When building against a real REBOUND, define REBXAMD_WITH_REBOUND and put
 * REBOUND's src/ on the include path; reboundx_amd.h then includes the real
 * "rebound.h" instead of this file, and reb_error/reb_warning come from
 * librebound.
 *
 * struct reb_particle has REBOUND 3.x's exact 128-byte layout (the force
 * callback receives arrays of it).  struct reb_simulation here is NOT
 * layout-compatible with REBOUND's: it lists only the members this library
 * reads or writes, which is enough for a standalone driver and for the tests.
 */
#ifndef REBXAMD_REBOUND_MIN_H
#define REBXAMD_REBOUND_MIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct reb_simulation;
struct reb_treecell;

struct reb_particle {
    double x, y, z;
    double vx, vy, vz;
    double ax, ay, az;
    double m;
    double r;
    double lastcollision;
    struct reb_treecell* c;
    uint32_t hash;
    void* ap;                       /* head of the REBOUNDx parameter list */
    struct reb_simulation* sim;
};

struct reb_vec3d { double x, y, z; };

enum REB_INTEGRATOR_MIN {
    REB_INTEGRATOR_IAS15 = 0,
    REB_INTEGRATOR_WHFAST = 1,
    REB_INTEGRATOR_MERCURIUS = 9
};

struct reb_simulation {
    double t;
    double G;
    double dt;
    int N;
    int N_var;
    int N_active;
    int integrator;
    unsigned int gravity_ignore_terms;      /* 1: gravity routine skipped the 0-1 pair (WHFast, Jacobi) */
    int force_is_velocity_dependent;
    struct reb_particle* particles;
    void* extras;
    void (*additional_forces)(struct reb_simulation* r);
    void (*pre_timestep_modifications)(struct reb_simulation* r);
    void (*post_timestep_modifications)(struct reb_simulation* r);
    void (*free_particle_ap)(struct reb_particle* p);
    void (*extras_cleanup)(struct reb_simulation* r);
    /* message queue used by the standalone reb_error / reb_warning below */
    int n_messages;
    char** messages;                        /* each entry: 'e' or 'w' followed by the text */
};

/* Standalone stand-ins for REBOUND's message channel: append to sim->messages. */
void reb_error(struct reb_simulation* r, const char* msg);
void reb_warning(struct reb_simulation* r, const char* msg);
/* Pops the oldest queued message into buf (truncated to len-1); returns 0 if none,
 * 1 for a warning, 2 for an error. */
int rebxamd_pop_message(struct reb_simulation* r, char* buf, int len);
void rebxamd_clear_messages(struct reb_simulation* r);

#ifdef __cplusplus
}
#endif
#endif

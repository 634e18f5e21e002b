/*
 * reboundx_amd.h -- C ABI of librebxamd, an MI355X (gfx950) implementation of
 * REBOUNDx's per-timestep additional-force path.
 *
 * Two groups of entry points:
 *
 *  A. The plugin API.  Same symbol names, argument meaning, struct layouts and
 *     error behaviour as the reference library for this path, so REBOUND's
 *     sim->additional_forces hook, C user programs and ctypes bindings written
 *     for the reference bind to this library unchanged.  Each declaration cites
 *     the reference declaration/definition it replaces (paths relative to the
 *     reference tree).  Host particle arrays (array-of-structs) go in and out.
 *
 *  B. The device-resident API (rebxamd_*).  Plain device pointers and sizes;
 *     particles and per-particle parameters live in HBM as fp64 columns and
 *     never cross PCIe.  This is what group A calls underneath, and what a
 *     device-resident integrator or a multi-GPU driver calls directly.
 *
 * No torch types, no C++ types, no HIP types appear in any signature
 * (streams are passed as void*; pass NULL for the default stream).
 */
#ifndef REBOUNDX_AMD_H
#define REBOUNDX_AMD_H

#include <stdint.h>
#include <stdio.h>

#ifdef REBXAMD_WITH_REBOUND
#include "rebound.h"
#else
#include "rebound_min.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================
 * A. Plugin API (drop-in for the reference on this path)
 * ==================================================================== */

/* src/core.c:41-43 */
extern const char* rebx_build_str;
extern const char* rebx_version_str;
extern const char* rebx_githash_str;

/* src/reboundx.h:51-59 */
enum rebx_param_type {
    REBX_TYPE_NONE, REBX_TYPE_DOUBLE, REBX_TYPE_INT, REBX_TYPE_POINTER,
    REBX_TYPE_FORCE, REBX_TYPE_UINT32, REBX_TYPE_ORBIT
};
/* src/reboundx.h:64-68 */
enum REBX_COORDINATES {
    REBX_COORDINATES_JACOBI, REBX_COORDINATES_BARYCENTRIC, REBX_COORDINATES_PARTICLE
};
/* src/reboundx.h:81-85 */
enum rebx_force_type { REBX_FORCE_NONE, REBX_FORCE_POS, REBX_FORCE_VEL };

/* src/reboundx.h:173-176 -- 16 bytes */
struct rebx_node {
    void* object;
    struct rebx_node* next;
};
/* src/reboundx.h:182-186 -- 24 bytes */
struct rebx_param {
    char* name;
    enum rebx_param_type type;
    void* value;
};
/* src/reboundx.h:191-198 -- 40 bytes; mirrored field by field by ctypes users
 * (reference reboundx/extras.py:284-288) */
struct rebx_force {
    char* name;
    struct rebx_node* ap;
    struct reb_simulation* sim;
    enum rebx_force_type force_type;
    void (*update_accelerations)(struct reb_simulation* const sim, struct rebx_force* const force,
                                 struct reb_particle* const particles, const int N);
};
/* src/reboundx.h:73-76, :90-94 */
enum rebx_timing { REBX_TIMING_PRE = -1, REBX_TIMING_POST = 1 };
enum rebx_operator_type { REBX_OPERATOR_NONE, REBX_OPERATOR_UPDATER, REBX_OPERATOR_RECORDER };
/* src/reboundx.h:203-210 -- 40 bytes (mirrored by ctypes, reference reboundx/extras.py:254-258) */
struct rebx_operator {
    char* name;
    struct rebx_node* ap;
    struct reb_simulation* sim;
    enum rebx_operator_type operator_type;
    void (*step_function)(struct reb_simulation* sim, struct rebx_operator* op, const double dt);
};
/* src/reboundx.h:216-219 */
struct rebx_step {
    struct rebx_operator* op;
    double dt_fraction;
};
/* src/reboundx.h:233-243 -- 56 bytes; Python allocates this struct itself
 * (reference reboundx/extras.py:39-45, :291-297), so the size and field order are frozen.
 * Device state hangs off a side table keyed by the struct's address. */
struct rebx_extras {
    struct reb_simulation* sim;
    struct rebx_node* additional_forces;
    struct rebx_node* pre_timestep_modifications;
    struct rebx_node* post_timestep_modifications;
    struct rebx_node* registered_params;
    struct rebx_node* allocated_forces;
    struct rebx_node* allocated_operators;
};

/* src/reboundx.h:263, src/core.c:119-128 */
struct rebx_extras* rebx_attach(struct reb_simulation* sim);
/* src/reboundx.h:270, src/core.c:135-155 */
void rebx_detach(struct reb_simulation* sim, struct rebx_extras* rebx);
/* src/reboundx.h:271, src/core.c:130-133 */
void rebx_extras_cleanup(struct reb_simulation* sim);
/* src/reboundx.h:277, src/core.c:180-183 */
void rebx_free(struct rebx_extras* rebx);
/* src/core.h:45-46, src/core.c:158-178, :51-92 (called directly by the Python wrapper) */
void rebx_initialize(struct reb_simulation* sim, struct rebx_extras* rebx);
void rebx_register_default_params(struct rebx_extras* rebx);
/* src/core.h:99, src/core.c:799-854 */
void rebx_free_pointers(struct rebx_extras* rebx);
/* src/core.h:95, src/core.c:764-766 (installed as sim->free_particle_ap) */
void rebx_free_particle_ap(struct reb_particle* p);

/* src/reboundx.h:332, src/core.c:189-225 */
struct rebx_force* rebx_create_force(struct rebx_extras* const rebx, const char* name);
/* src/reboundx.h:330, src/core.c:227-277 */
struct rebx_force* rebx_load_force(struct rebx_extras* const rebx, const char* name);
/* src/reboundx.h:328, src/core.c:371-409 */
int rebx_add_force(struct rebx_extras* rebx, struct rebx_force* force);
/* src/reboundx.h:279, src/core.c:654-662 */
int rebx_remove_force(struct rebx_extras* rebx, struct rebx_force* force);
/* src/reboundx.h:358, src/core.c:624-635 */
struct rebx_force* rebx_get_force(struct rebx_extras* const rebx, const char* const name);

/* src/reboundx.h:401-407, src/core.c:512-622, :94-117 */
void* rebx_get_param(struct rebx_extras* rebx, struct rebx_node* ap, const char* const param_name);
struct rebx_param* rebx_get_param_struct(struct rebx_extras* rebx, struct rebx_node* ap, const char* const param_name);
void rebx_set_param_pointer(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const param_name, void* val);
void rebx_set_param_double(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const param_name, double val);
void rebx_set_param_int(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const param_name, int val);
void rebx_set_param_uint32(struct rebx_extras* const rebx, struct rebx_node** apptr, const char* const param_name, uint32_t val);
void rebx_register_param(struct rebx_extras* const rebx, const char* name, enum rebx_param_type type);
/* src/core.h:102, src/core.c:958-966 */
enum rebx_param_type rebx_get_type(struct rebx_extras* rebx, const char* name);
/* src/reboundx.h:542, src/linkedlist.c:97-106 */
int rebx_len(struct rebx_node* head);
/* src/reboundx.h:590, src/core.c:1003-1010 */
void rebx_error(struct rebx_extras* rebx, const char* const msg);

/* src/core.h:52, src/core.c:868-880 -- the function installed in sim->additional_forces.
 * Walks the added forces in list order (the reverse of rebx_add_force order), runs every
 * maximal run of built-in forces as one fused device pass, and calls user-supplied
 * callbacks on the host in between. */
void rebx_additional_forces(struct reb_simulation* sim);

/* Binary parameter files (.rebx): src/reboundx.h:134-152 (messages), :287 and src/output.c:267-288
 * (writer), :296-302 and src/input.c:627-702 (readers).  Same record format as the reference, so
 * files written by either side load in the other; padding bytes in record headers are written as
 * zero.  UINT32 parameter values are written with their 4 bytes (the reference's size table has no
 * entry for the type and writes none). */
enum rebx_input_binary_messages {
    REBX_INPUT_BINARY_WARNING_NONE = 0,
    REBX_INPUT_BINARY_ERROR_NOFILE = 1,
    REBX_INPUT_BINARY_ERROR_CORRUPT = 2,
    REBX_INPUT_BINARY_ERROR_NO_MEMORY = 4,
    REBX_INPUT_BINARY_ERROR_REBX_NOT_LOADED = 8,
    REBX_INPUT_BINARY_ERROR_REGISTERED_PARAM_NOT_LOADED = 16,
    REBX_INPUT_BINARY_WARNING_PARAM_NOT_LOADED = 32,
    REBX_INPUT_BINARY_WARNING_PARTICLE_PARAMS_NOT_LOADED = 64,
    REBX_INPUT_BINARY_WARNING_FORCE_NOT_LOADED = 128,
    REBX_INPUT_BINARY_WARNING_OPERATOR_NOT_LOADED = 256,
    REBX_INPUT_BINARY_WARNING_STEP_NOT_LOADED = 512,
    REBX_INPUT_BINARY_WARNING_ADDITIONAL_FORCE_NOT_LOADED = 1024,
    REBX_INPUT_BINARY_WARNING_FIELD_UNKNOWN = 2048,
    REBX_INPUT_BINARY_WARNING_LIST_UNKNOWN = 4096,
    REBX_INPUT_BINARY_WARNING_PARAM_VALUE_NULL = 8192,
    REBX_INPUT_BINARY_WARNING_VERSION = 16384,
    REBX_INPUT_BINARY_WARNING_FORCE_PARAM_NOT_LOADED = 32768
};
void rebx_output_binary(struct rebx_extras* rebx, char* filename);
void rebx_init_extras_from_binary(struct rebx_extras* rebx, const char* const filename, enum rebx_input_binary_messages* warnings);
struct rebx_extras* rebx_create_extras_from_binary(struct reb_simulation* sim, const char* const filename);

/* Potentials of the position-dependent forces, evaluated on the device from sim->particles
 * (src/reboundx.h:483-504; gravitational_harmonics.c:221-231, central_force.c:122-141,
 * tides_precession.c:182-202, gr_potential.c:122-132). */
double rebx_gravitational_harmonics_potential(struct rebx_extras* const rebx);
double rebx_central_force_potential(struct rebx_extras* const rebx);
double rebx_tides_precession_potential(struct rebx_extras* const rebx);
double rebx_gr_potential_potential(struct rebx_extras* const rebx, const struct rebx_force* const gr_potential);
/* src/reboundx.h:467, src/gr.c:248-260 */
double rebx_gr_hamiltonian(struct rebx_extras* const rebx, const struct rebx_force* const gr);
/* src/reboundx.h:475, src/gr_full.c:251-377 */
double rebx_gr_full_hamiltonian(struct rebx_extras* const rebx, const struct rebx_force* const gr_full);

/* Operators on the additional-force path: the "integrate a force across a timestep" operator and
 * the kick/drift steppers.  src/reboundx.h:331, :329, :327, :280; src/core.c:279-368, :411-510,
 * :637-648, :697-728.  rebx_load_operator knows "integrate_force", "kick", "drift", "modify_mass",
 * "modify_orbits_direct" and "track_min_distance"; the reference's other operators (kepler, jump,
 * interaction, ias15, which wrap REBOUND internals) are not part of this library and report
 * "not found". */
struct rebx_operator* rebx_create_operator(struct rebx_extras* const rebx, const char* name);
struct rebx_operator* rebx_load_operator(struct rebx_extras* const rebx, const char* name);
int rebx_add_operator(struct rebx_extras* rebx, struct rebx_operator* op);
int rebx_add_operator_step(struct rebx_extras* rebx, struct rebx_operator* op, const double dt_fraction, enum rebx_timing timing);
int rebx_remove_operator(struct rebx_extras* rebx, struct rebx_operator* op);
struct rebx_operator* rebx_get_operator(struct rebx_extras* const rebx, const char* const name);
/* src/core.c:882-912: installed as sim->pre/post_timestep_modifications; run every step of the
 * respective list with dt = sim->dt * dt_fraction */
void rebx_pre_timestep_modifications(struct reb_simulation* sim);
void rebx_post_timestep_modifications(struct reb_simulation* sim);
/* src/integrate_force.c:34-78 (operator parameters "force" and "integrator").  A built-in force
 * is integrated on the device in one round trip (rebxamd_integrate_force); a user callback is
 * integrated on the host with the same schemes, calling it on the stage arrays. */
void rebx_integrate_force(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt);
/* src/steppers.c:112-133.  rebx_kick_step uses the accelerations held in sim->particles; when
 * built against REBOUND (REBXAMD_WITH_REBOUND) it first calls reb_update_acceleration as the
 * reference does. */
void rebx_drift_step(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt);
/* src/modify_mass.c:63-73 (per-particle "tau_mass") and src/modify_orbits_direct.c:121-132
 * (per-particle "tau_a", "tau_e", "tau_inc", "tau_omega", "tau_Omega"; operator parameters
 * "coordinates" and "p"): one device round trip each. */
void rebx_modify_mass(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt);
void rebx_modify_orbits_direct(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt);
void rebx_kick_step(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt);
/* src/track_min_distance.c:66-96 (a recorder: per-particle "min_distance", "min_distance_from", "min_distance_orbit").
 * Leaves the state alone and touches only the particles that carry "min_distance": runs on the host, on the particle
 * array the simulation holds.  "min_distance_from" names a particle by its hash; one that is not found is reported
 * (the reference dereferences the NULL that reb_get_particle_by_hash returns).  "min_distance_orbit" is REBOUND's
 * struct reb_orbit: it is filled when the library is built against REBOUND (REBXAMD_WITH_REBOUND) and left alone, with a
 * warning, otherwise. */
void rebx_track_min_distance(struct reb_simulation* const sim, struct rebx_operator* const op, const double dt);
/* src/core.h:61, src/core.c:860-866 */
void rebx_reset_accelerations(struct reb_particle* const ps, const int N);

/* src/core.h:66-74 -- the force callbacks.  Each adds its accelerations to the
 * particle array it is given (which may be a scratch copy, src/integrator_rk4.c:58-83). */
void rebx_gr(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_gr_full(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_gr_potential(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_radiation_forces(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_modify_orbits_forces(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_tides_precession(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_central_force(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);
void rebx_gravitational_harmonics(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);

/* src/core.h:74.  Not implemented (src/ephemeris_forces.c needs the JPL ephemeris files): reports through the
 * simulation's message channel and leaves the accelerations alone; rebx_load_force does not know the name. */
void rebx_ephemeris_forces(struct reb_simulation* const sim, struct rebx_force* const force, struct reb_particle* const particles, const int N);

/* src/reboundx.h:436,448, src/radiation_forces.c:119-124 */
/* src/central_force.c:143-152 (declared src/reboundx.h:458): Acentral that makes the pericentre of p's orbit about
 * `primary` turn at the rate pomegadot under a central force a = Acentral r^gamma:
 * G m_primary pomegadot / (1 + gamma/2) / d^(gamma+2) / n, d = separation, n = mean motion of the osculating orbit.
 * gamma = -2: reb_error and 0 (the precession vanishes there).  G is p.sim->G. */
double rebx_central_force_Acentral(const struct reb_particle p, const struct reb_particle primary, const double pomegadot, const double gamma);
double rebx_rad_calc_beta(const double G, const double c, const double source_mass, const double source_luminosity, const double radius, const double density, const double Q_pr);
double rebx_rad_calc_particle_radius(const double G, const double c, const double source_mass, const double source_luminosity, const double beta, const double density, const double Q_pr);

/* Extensions to the plugin API (no counterpart in the reference). */

/* The STRUCTURE of the per-particle parameter lists (which particle holds which parameter, and where
 * its value lives) is cached and rebuilt when a rebx_set_param_*, add/remove call or a change of the
 * particle array is seen; the VALUES are read again through the cached pointers on every call, as the
 * reference reads them (src/core.c:601-622), so a value overwritten in place through a pointer obtained
 * from rebx_get_param is picked up by itself.  Call this only after editing a list by hand (nodes
 * linked or unlinked without the API). */
void rebxamd_invalidate_params(struct rebx_extras* rebx);
/* Bulk per-particle parameters.  Gives particle i (0 <= i < N) the parameter `name` with value
 * values[i] without building N linked lists: one call and one copy of the column instead of N times
 * rebx_set_param_double (src/core.c:552-565), which at 1e7-1e8 particles costs gigabytes of nodes.
 * `name` is one of the per-particle double parameters the built-in effects and operators read:
 * "beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc", "tau_mass", "tau_omega", "tau_Omega"; it must
 * be registered with type double (the defaults are).  A NaN in values[i] means "not set on particle i";
 * particles with index >= N have it not set.  The values are copied: call again to change them;
 * values == NULL or N == 0 removes the column.  The linked lists stay authoritative: where a particle's
 * own list holds `name` (rebx_set_param_double), that value is used and the column's entry ignored.
 * The column is seen by this library's effects and operators only: rebx_get_param on a particle does
 * not find it (there is no list node).  rebx_output_binary writes its entries as ordinary parameters of the
 * particles, so a file that is read back (by this library or by the reference) holds them in the lists.  Indices follow the
 * particle array at the time of each call; after removing or reordering particles set the column again.
 * Returns 1, or 0 after reporting through the simulation's message channel (unknown or unregistered
 * name, negative N). */
int rebxamd_set_param_column(struct rebx_extras* rebx, const char* name, const double* values, int N);
/* The stored copy of a column set by rebxamd_set_param_column and its length (NULL and 0 if none):
 * for custom force callbacks that want to read it. */
const double* rebxamd_get_param_column(struct rebx_extras* rebx, const char* name, int* N);
/* 0 = fast arithmetic (default), 1 = reference-order arithmetic, no FMA contraction. */
void rebxamd_set_mode(struct rebx_extras* rebx, int mode);
/* Small systems (at most 32768 particles, massive bodies within the first REBXAMD_MAX_HEAD, no gr_full in the batch) are
 * evaluated in place: the kernels read and update the pinned host columns the particles were packed into, over the link,
 * instead of waiting for one copy command per direction (33 against 56 microseconds per call at 1000 particles; the
 * results are the same bits).  on = 0 goes back to the copies for this simulation; the environment variable
 * REBXAMD_HOOK_IN_PLACE=0 does so for the process (default: 1).  No counterpart in the reference. */
void rebxamd_set_in_place(struct rebx_extras* rebx, int on);
/* Milliseconds spent by the last rebx_additional_forces / callback in: [0] packing host
 * AoS -> columns, [1] host->device, [2] kernels, [3] device->host + unpack. */
void rebxamd_last_timings(struct rebx_extras* rebx, double out_ms[4]);

/* ======================================================================
 * B. Device-resident API
 * ==================================================================== */

typedef struct rebxamd_ctx rebxamd_ctx;

enum rebxamd_status {
    REBXAMD_OK = 0,
    REBXAMD_ERR_ARG = 1,            /* NULL / negative size / bad index */
    REBXAMD_ERR_HIP = 2,            /* a HIP runtime call failed; see rebxamd_last_error */
    REBXAMD_ERR_UNSUPPORTED = 3,    /* valid request this build cannot run on the device */
    REBXAMD_ERR_PARAM = 4           /* a required effect parameter is missing (reference: reb_error + return) */
};

enum rebxamd_mode {
    REBXAMD_MODE_FAST = 0,          /* shared reciprocals, FMA contraction allowed; <=1e-12 relative */
    REBXAMD_MODE_EXACT = 1          /* reference expression order, no contraction: the bits of a CPU build of the reference without FMA
                                     * contraction on every body while the chains below apply (sources and reference particles collect
                                     * the pushes one by one in the reference's order; as parallel sums beyond REBXAMD_SEQ_MAX_BODIES) --
                                     * central_force apart, whose pow() is the device library's (<= 4 ulp from glibc's) */
};

/* gr and Jacobi/barycentric modify_orbits_forces accumulate body by body in the reference (src/gr.c:80-167 through
 * REBOUND's Jacobi recurrences; src/rebxtools.c:45-126).  With the massive bodies in a prefix of at most
 * REBXAMD_MAX_HEAD particles followed by massless ones, REBXAMD_MODE_EXACT walks the prefix in that order.  On every
 * other mass layout REBXAMD_MODE_EXACT walks the same recurrences in the reference's order as long as they cover at
 * most REBXAMD_SEQ_MAX_BODIES particles: the whole view, or -- with view->tail_massless set -- its first n_head
 * particles, the massless ones behind them (any number) being measured from the sums those leave (chains on one
 * workgroup: milliseconds at the limit; every body then carries the bits of the sequential evaluation -- under the same
 * limit the pushes of massive bodies beyond the REBXAMD_MAX_HEAD prefix on sources and reference particles are added
 * one by one in the reference's order too); longer
 * chains, REBXAMD_MODE_FAST on such layouts and the particle-shard
 * entry points (rebxamd_shard_*) evaluate the sums as parallel prefix sums, whose results stay within 5e-14 of a body's
 * largest acceleration component of the exactly rounded value but are NOT the sequential order's bits (the sequential
 * order itself drifts from the exact value as the number of bodies grows: 1e-12 at 6000 bodies, 5e-11 at 300 000). */
#define REBXAMD_SEQ_MAX_BODIES 65536

/* One elementary pass of a force.  A reference force expands to one op per source
 * (e.g. gravitational_harmonics with one oblate body -> one J2 op then one J4 op). */
enum rebxamd_op_kind {
    REBXAMD_OP_RADIATION = 1,       /* src, a=c                      radiation_forces.c:69-98 */
    REBXAMD_OP_J2 = 2,              /* src, a=J2, b=R_eq             gravitational_harmonics.c:61-85 */
    REBXAMD_OP_J4 = 3,              /* src, a=J4, b=R_eq             gravitational_harmonics.c:99-123 */
    REBXAMD_OP_GR = 4,              /* a=c, iarg=max_iterations      gr.c:73-171 (source is index 0) */
    REBXAMD_OP_GR_POTENTIAL = 5,    /* a=c                           gr_potential.c:69-87 (source is index 0) */
    REBXAMD_OP_CENTRAL = 6,         /* src, a=Acentral, b=gamma      central_force.c:63-83 */
    REBXAMD_OP_TIDES = 7,           /* src                           tides_precession.c:65-116 */
    REBXAMD_OP_MOF = 8,             /* iarg=coordinates, src=reference particle (PARTICLE only)
                                       modify_orbits_forces.c:77-118 + rebxtools.c:45-126 */
    REBXAMD_OP_GR_FULL = 9          /* a=c, iarg=gravity_ignore_10   gr_full.c:61-231 */
};

typedef struct rebxamd_op {
    int32_t kind;
    int32_t src;
    int32_t iarg;
    int32_t reserved;
    double a;
    double b;
} rebxamd_op;

#define REBXAMD_MAX_OPS 32
#define REBXAMD_UNSET_BITS 0x7FF8000000000000ull        /* what C's NAN and numpy.nan hold: the usual way to write "not set" */
#define REBXAMD_NAN_VALUE_BITS 0x7FF8000000000001ull    /* the one NaN that stands for a parameter whose VALUE is NaN */
#define REBXAMD_MAX_HEAD 64

/* A set of particles resident in device memory, as fp64 columns of length n.
 *
 * Per-particle parameter columns mark "not set on this particle" with a NaN: any NaN, whatever its
 * sign and payload (NAN, numpy.nan, -nan, the 0/0 of host arithmetic, a NaN a computed column
 * inherited), except the single pattern REBXAMD_NAN_VALUE_BITS; a NULL column means not set on any
 * particle.  REBXAMD_NAN_VALUE_BITS is a parameter that IS set and whose value is NaN: it runs through
 * the arithmetic, as a NaN-valued parameter does in the reference (which finds the parameter in the
 * list and computes with it, e.g. src/modify_orbits_forces.c:78-108).  The plugin layer uploads
 * such parameters with that pattern.
 *
 * n_head / tail_massless describe the mass layout: if tail_massless != 0 the
 * caller asserts that every particle with index >= n_head has m == 0 exactly
 * (rebxamd_classify computes the smallest such n_head).  With n_head <=
 * REBXAMD_MAX_HEAD and every op source inside the head, the library runs the
 * head with the reference's sequential semantics in one wavefront and streams
 * the tail with no reductions at all; otherwise it uses block reductions for
 * the back-reactions on source particles. */
typedef struct rebxamd_view {
    int64_t n;
    int64_t n_head;
    int32_t tail_massless;
    int32_t reserved;
    double G;
    const double *x, *y, *z;
    const double *vx, *vy, *vz;
    const double *m;
    double *ax, *ay, *az;
    const double *beta;
    const double *R_tides, *k1;
    const double *tau_a, *tau_e, *tau_inc;
} rebxamd_view;

int rebxamd_create(int device, rebxamd_ctx** out);
void rebxamd_destroy(rebxamd_ctx* ctx);
const char* rebxamd_last_error(rebxamd_ctx* ctx);

/* Finds n_head = 1 + index of the last particle with m != 0 (at least 1).  Synchronises. */
int rebxamd_classify(rebxamd_ctx* ctx, const double* m, int64_t n, int64_t* n_head, void* stream);

/* Applies `ops` in order to the particles of `view`, adding to ax/ay/az, exactly as the
 * reference would by calling the corresponding update_accelerations callbacks one after the
 * other (src/core.c:868-880) -- but in as few passes over HBM as the ops allow.  Asynchronous
 * on `stream`. */
int rebxamd_apply(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode, void* stream);

/* rebxamd_apply for callers that cannot allocate device memory (no counterpart in the reference;
 * for code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc).
 * Same kernels and results as rebxamd_apply, but no context: the small tables, the warning word and
 * the partial sums of the pushes on source particles live in a fixed 4 MiB array inside the
 * library's code object (one copy per GPU, held until the process ends), and the call makes no
 * hipMalloc/hipFree.  Works on the current device, which must be the one that holds the columns of
 * `view`.  The address of the array is looked up on the first call on each device
 * (hipGetSymbolAddress), so make that first call outside graph capture.
 * Limits (REBXAMD_ERR_UNSUPPORTED with a message in errbuf beyond them): nops <= REBXAMD_MAX_OPS; no
 * gr_full; gr and Jacobi/barycentric modify_orbits_forces only when the massive bodies are a prefix
 * of at most REBXAMD_MAX_HEAD particles (view->n_head, view->tail_massless); on layouts that need
 * parallel sums of the pushes (massive bodies beyond such a prefix), (number of ops that push on a
 * source) x ceil(n / 512) x 24 bytes must fit in the array less 80 KiB (3 such ops: n <= 2.9e7).
 * warnings_dev: device word in which kernels raise the rebxamd_fetch_warnings bits (the caller
 * clears it), or NULL (bits are dropped).  errbuf (may be NULL) receives a message on failure.
 * Every call on a GPU shares that GPU's array and one host-side record per GPU (warning word, table
 * index, error text).  The host side of a call is serialised per GPU by a lock inside the library, so
 * host threads may call concurrently; the work they queue is not ordered by it: calls whose kernels
 * overlap on one GPU (two streams) race on the array, and the caller has to order them. */
int rebxamd_apply_static(const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode, uint32_t* warnings_dev, void* stream,
                         char* errbuf, int errbuf_len);

/* Particle-sharded runs (no counterpart in the reference).  `view` holds [the replicated head |
 * this device's slice of the particles] and the slice may contain massive particles.
 * rebxamd_apply_targets applies `ops` (no gr, no gr_full, modify_orbits_forces in PARTICLE
 * coordinates only; sources inside the head) to the particles [target_lo, n) only and, instead
 * of pushing on the source particles, stores each op's total push in sums[3*k..3*k+2] (device
 * memory, nops*3 doubles).  The caller sums `sums` over devices (all-reduce) and hands the result
 * to rebxamd_add_pushes, which adds it to the source particles.  The head's own update is the
 * plain rebxamd_apply on a view of the head alone. */
int rebxamd_apply_targets(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int mode,
                          int64_t target_lo, double* sums, void* stream);
int rebxamd_add_pushes(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, const double* sums, void* stream);

/* modify_orbits_forces in Jacobi or barycentric coordinates when the particles are sharded over
 * devices and the slices hold massive bodies (no counterpart in the reference, which is one
 * sequential loop: src/rebxtools.c:45-126).  Global order = [head | slice of rank 0 | slice of
 * rank 1 | ...]; every device holds [head | its slice] and calls, for the range of the head
 * [0, n_head) and for the range of its slice [n_head, n):
 *   1. rebxamd_shard_sums        sums7 <- (m, m x, m y, m z, m vx, m vy, m vz) over the range
 *      -> all-gather the slice sums;
 *   2. rebxamd_shard_mof_forces  before7 = sums of every body that precedes the range in global
 *      order (NULL = none) for Jacobi, sums of ALL bodies for barycentric; push3 <- summed
 *      back-reaction of the range -> all-gather the slice pushes;
 *   3. rebxamd_shard_mof_apply   after3 = pushes of every body that follows the range (NULL =
 *      none) for Jacobi, pushes of ALL bodies for barycentric; adds to ax/ay/az of the range.
 * All vectors live in device memory.  Steps 2 and 3 of one range must not be separated by other
 * calls on the same context with a different view (the per-body terms are kept in between).
 * reboundx_amd/dist.py (mof_com_sharded) is the driver. */
int rebxamd_shard_sums(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, double* sums7, void* stream);
int rebxamd_shard_mof_forces(rebxamd_ctx* ctx, const rebxamd_view* view, int coordinates, int64_t lo, int64_t hi,
                             const double* before7, double* push3, void* stream);
int rebxamd_shard_mof_apply(rebxamd_ctx* ctx, const rebxamd_view* view, int coordinates, int64_t lo, int64_t hi, const double* after3,
                            void* stream);

/* gr (src/gr.c:73-171) when the particles are sharded over devices and the slices hold massive
 * bodies (no counterpart in the reference, which is one sequential loop).  Same global order
 * and ranges as above; body ids are positions in the global order.  Per range:
 *   1. rebxamd_shard_massive     records <- {x, y, z, G m, id0 + (i - lo)} of the massive bodies of
 *      the range in index order (5 doubles each; capacity 5*(hi-lo)), count <- their number (one
 *      double) -> all-gather counts and records; concatenate head, slice of rank 0, slice of rank 1, ...;
 *   2. rebxamd_shard_gr_newton   Newtonian pull of the gathered records on the bodies of the range
 *      (a record whose id equals the body's own id is skipped; kept inside the context);
 *      sums10 <- (m, m x, m v, m a_newton) over the range -> all-gather the slice sums;
 *   3. rebxamd_shard_gr_forces   before10 = sums of every body that precedes the range (NULL for the
 *      range that starts at body 0); push3 <- summed back-reaction of the range -> all-gather;
 *   4. rebxamd_shard_gr_apply    after3 = pushes of every body that follows the range (NULL = none);
 *      adds to ax/ay/az of the range.
 * Steps 2-4 of one range must not be separated by other calls on the same context that use the
 * same range start.  The cost of step 2 is (bodies of the range) x (massive bodies of the system).
 * reboundx_amd/dist.py (gr_sharded) is the driver. */
int rebxamd_shard_massive(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, int64_t id0, double* records, double* count,
                          void* stream);
int rebxamd_shard_gr_newton(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, int64_t id0, const double* records,
                            int64_t nrecords, double* sums10, int mode, void* stream);
int rebxamd_shard_gr_forces(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int max_iterations, int64_t lo, int64_t hi,
                            const double* before10, double* push3, int mode, void* stream);
int rebxamd_shard_gr_apply(rebxamd_ctx* ctx, const rebxamd_view* view, int64_t lo, int64_t hi, const double* after3, void* stream);

/* src/core.c:860-866 on device columns. */
int rebxamd_reset_accelerations(rebxamd_ctx* ctx, const rebxamd_view* view, void* stream);

/* Device-resident Wisdom-Holman stepping for one dominant mass at index 0 followed by massless
 * bodies (view->n_head = 1, view->tail_massless = 1), so that a many-step run with additional
 * forces never leaves the GPU.  No counterpart in the reference: there REBOUND's WHFast owns
 * the step and calls rebx_additional_forces from its kick (src/core.c:868-880).
 * rebxamd_wh_drift: body 0 drifts linearly, every other body moves on its Kepler orbit about it
 * (mass parameter G*m0) for dt.  rebxamd_wh_steps: nsteps times { drift dt/2; accelerations :=
 * ops (the additional forces alone: the dominant mass is in the drift); v_i += dt*(a_i - a_0);
 * drift dt/2 }, with the two half drifts between consecutive steps merged into one Kepler drift of
 * dt (the state is synchronised when the call returns, not in between).  Positions and velocities
 * are updated in place.  The acceleration columns are scratch: on return body 0's is zero and the others'
 * are either untouched (the forces never left registers) or zero.  Asynchronous (G*m0 is formed on the device).
 * rebxamd_wh_steps also takes a head of several massive bodies (star + planets: 2 <= view->n_head <= 64,
 * view->tail_massless = 1): the head then steps in Jacobi coordinates as REBOUND's WHFast does with its defaults
 * (Rein & Tamayo 2015: body i on a Kepler orbit about G*(m_0 + ... + m_i), the centre of mass drifting linearly,
 * direct gravity among the head bodies without the pair (0,1), Jacobi term on bodies i >= 2), and every massless body
 * about the head's centre of mass, pulled by every head body.  The force stack must then be one streaming run (no
 * gr_full, at most one gr and one Jacobi/barycentric modify_orbits_forces) whose sources are head bodies. */
int rebxamd_wh_drift(rebxamd_ctx* ctx, const rebxamd_view* view, double dt, int mode, void* stream);
int rebxamd_wh_steps(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, double dt, int64_t nsteps,
                     int mode, void* stream);

/* The same run left unsynchronised between calls, for a host that looks in after every step (or every few) and still
 * wants one launch and one Kepler drift per step -- what REBOUND's WHFast does with safe_mode = 0 and
 * reb_integrator_synchronize.  No counterpart in the reference.  One massive body followed by massless ones
 * (view->n_head = 1, view->tail_massless = 1), a force stack of at least one op that is one streaming run whose source
 * is body 0 (no gr_full, no barycentric modify_orbits_forces); or a head of 2 .. 64 massive bodies followed by at least
 * one massless one (view->tail_massless = 1, acceleration columns present) under any stack rebxamd_wh_steps takes behind
 * such a head (the empty one included); anything else: REBXAMD_ERR_UNSUPPORTED, use rebxamd_wh_steps.
 *   rebxamd_wh_run_begin   opens the run (launches nothing; keeps copies of *view and of the ops).  Until the run is
 *                          closed the caller must leave the columns where they are and must not write positions,
 *                          velocities, masses or parameters.
 *   rebxamd_wh_run_steps   nsteps more steps, each ONE pass: { drift dt (dt/2 in the first step of the run) ; forces ;
 *                          kick }.  Between calls the massless bodies sit at the middle of the last step, kicked; the rows
 *                          of the massive bodies are one drift further (the next step's head pass, and behind several
 *                          massive bodies their next step, have already run).
 *   rebxamd_wh_run_end     the closing half drift: the columns hold the synchronised state, bit for bit what
 *                          rebxamd_wh_steps(view, ops, dt, <steps taken>) would have left.  Closes the run.
 * All three are asynchronous on `stream`, which must be the same for the whole run (REBXAMD_ERR_ARG otherwise).  While a
 * run is open the context's tables and records belong to it: rebxamd_apply, rebxamd_apply_targets, rebxamd_add_pushes,
 * rebxamd_integrate_force, rebxamd_wh_drift, rebxamd_wh_steps, rebxamd_wh_batch, rebxamd_modify_orbits_direct and
 * rebxamd_potentials on the same context return REBXAMD_ERR_ARG (use a second context, or close the run). */
int rebxamd_wh_run_begin(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, double dt, int mode, void* stream);
int rebxamd_wh_run_steps(rebxamd_ctx* ctx, int64_t nsteps, void* stream);
int rebxamd_wh_run_end(rebxamd_ctx* ctx, void* stream);

/* Operators on device columns.  Positions, velocities (and masses) of `view` are UPDATED IN PLACE.
 * rebxamd_modify_mass: m_i += m_i*dt/tau_mass_i where tau_mass_i is set (device column, NaN = not
 * set), then positions and velocities are shifted to the centre-of-mass frame
 * (src/modify_mass.c:63-73, which ends with reb_move_to_com).
 * rebxamd_modify_orbits_direct: a, e, inc, omega, Omega of every body's orbit change by
 * dt/tau_* (view->tau_a/tau_e/tau_inc and the two extra columns, NaN = not set; *p_param = the
 * operator's "p", NULL = unset) in the given coordinates with the reference's back-reactions
 * (src/modify_orbits_direct.c:81-132, src/rebxtools.c:137-215).  `primary` = reference particle
 * for REBX_COORDINATES_PARTICLE.  With the massive bodies in a prefix of at most 64 particles
 * followed by massless ones, one thread walks the prefix exactly as the reference does and the
 * massless bodies run in parallel.  Any other layout (the reference's chain, in which every massive
 * body shifts the bodies before it ahead of their own step, is then long): the pushes are found as
 * the fixed point of a parallel map over suffix sums -- rounds until they stop moving -- and one
 * thread walks the chain sequentially should the rounds not settle; results agree with the
 * sequential walk to rounding of the sums (1e-12 of the state).  That path synchronises the stream
 * once.  rebxamd_direct_chain: max_rounds >= 0 sets the number of rounds tried before the
 * sequential walk (default 12; 0 = always walk); *last_rounds (may be NULL) = rounds the last
 * long-chain call took, -1 if it walked. */
int rebxamd_modify_mass(rebxamd_ctx* ctx, const rebxamd_view* view, const double* tau_mass, double dt, void* stream);
int rebxamd_modify_orbits_direct(rebxamd_ctx* ctx, const rebxamd_view* view, int coordinates, int primary, const double* p_param,
                                 const double* tau_omega, const double* tau_Omega, double dt, void* stream);
int rebxamd_direct_chain(rebxamd_ctx* ctx, int max_rounds, int* last_rounds);

/* Potential energy of each op in `ops` (J2, J4, central_force, tides_precession, gr_potential),
 * summed over the particles of `view`: out[k] (host memory, nops doubles) is what the
 * reference's per-source potential routine returns for op k
 * (src/gravitational_harmonics.c:142-193, central_force.c:97-120, tides_precession.c:132-180,
 * gr_potential.c:100-120).  For tides_precession the caller picks the source as the reference's
 * potential does (the last particle flagged "primary", else 0).  Synchronises the stream. */
int rebxamd_potentials(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, double* out, void* stream);

/* The Hamiltonian the gr force derives from, Newtonian part included (src/gr.c:190-246):
 * kinetic + Newtonian pair potential + post-Newtonian term in Jacobi coordinates.  *H in host
 * memory; synchronises the stream.  O(N * massive) for the pair potential. */
int rebxamd_gr_hamiltonian(rebxamd_ctx* ctx, const rebxamd_view* view, double c, double* H, void* stream);

/* The Hamiltonian the gr_full force derives from (src/gr_full.c:251-377): canonical velocities by
 * ten fixed-point sweeps over all pairs, then kinetic + Newtonian + post-Newtonian energy.  The
 * sweeps read the previous iterate (the reference updates in place); both converge to the same
 * fixed point.  *H in host memory; synchronises the stream.  O(N^2) per sweep. */
int rebxamd_gr_full_hamiltonian(rebxamd_ctx* ctx, const rebxamd_view* view, double c, double* H, void* stream);

/* Values of enum rebx_integrator, src/reboundx.h:157-163. */
enum rebxamd_integrator {
    REBXAMD_INTEGRATOR_NONE = -1,
    REBXAMD_INTEGRATOR_IMPLICIT_MIDPOINT = 0,
    REBXAMD_INTEGRATOR_RK4 = 1,
    REBXAMD_INTEGRATOR_EULER = 2,
    REBXAMD_INTEGRATOR_RK2 = 3
};

/* The "integrate a force across a timestep" operator on device columns: replaces
 * rebx_integrate_force (src/integrate_force.c:34-78) and the schemes it dispatches to
 * (src/integrator_euler.c:33-41, integrator_rk2.c:38-67, integrator_rk4.c:40-91,
 * integrator_implicit_midpoint.c:83-124).  `ops` plays the role of the `force` parameter.
 * Integrates dv/dt = a(x, v) over dt at fixed positions: view->vx/vy/vz are UPDATED IN PLACE
 * (the const in rebxamd_view is cast away for them); as in the reference, view->ax/ay/az are
 * cleared first and hold the first-stage accelerations on return (zero for the implicit
 * midpoint rule).  *iterations (may be NULL) receives the implicit-midpoint loop counter; 10
 * means not converged and raises warning bit 4.  The implicit midpoint rule synchronises the
 * stream once per iteration (its convergence test is a global sum); the others are asynchronous. */
int rebxamd_integrate_force(rebxamd_ctx* ctx, const rebxamd_view* view, const rebxamd_op* ops, int nops, int integrator,
                            double dt, int mode, int* iterations, void* stream);

/* v += dt*a on device columns: the velocity update of rebx_kick_step (src/steppers.c:123-133;
 * the accelerations are whatever the columns hold -- gravity is REBOUND's, not this library's). */
int rebxamd_kick(rebxamd_ctx* ctx, const rebxamd_view* view, double dt, int mode, void* stream);
/* x += dt*v on device columns: rebx_drift_step (src/steppers.c:112-121). */
int rebxamd_drift(rebxamd_ctx* ctx, const rebxamd_view* view, double dt, int mode, void* stream);

/* Bits raised by kernels since the last call (synchronises, then clears):
 *  1 = a gr particle used 10 inner iterations (reference warns, gr.c:137-140)
 *  2 = gr_full ran all 10 sweeps without meeting its stopping rule (gr_full.c:221-223)
 *  4 = the implicit midpoint rule used 10 iterations (reference warns, integrator_implicit_midpoint.c:116-118) */
int rebxamd_fetch_warnings(rebxamd_ctx* ctx, uint32_t* bits, void* stream);

/* Timing aid (no counterpart in the reference).  While enabled, every streaming pass launched
 * by rebxamd_apply is bracketed by a HIP event pair recorded on the stream it is launched on.
 * rebxamd_timing_read waits for the recorded events, returns the summed kernel time and the
 * number of launches since the last read, and starts over. */
int rebxamd_timing_enable(rebxamd_ctx* ctx, int on);
/* Kernel selection (no counterpart in the reference).  The streaming passes have instances built
 * for the common force stacks (radiation + harmonics + gr; radiation + harmonics + gr_potential +
 * tides + modify_orbits_forces; tides + modify_orbits_forces + gr_potential; the single forces)
 * that run the stack as straight-line code; every other stack walks its op list at run time.
 * Both compute the same expressions.  on = 0 makes this context always walk the list
 * (default: 1). */
int rebxamd_fixed_stacks(rebxamd_ctx* ctx, int on);
/* Kernel selection (no counterpart in the reference).  on = 0 makes REBXAMD_MODE_EXACT evaluate gr and
 * Jacobi/barycentric modify_orbits_forces with parallel prefix sums on every layout that has massive bodies beyond a
 * short prefix, also below REBXAMD_SEQ_MAX_BODIES particles (default: 1, the reference's order there). */
int rebxamd_sequential_exact(rebxamd_ctx* ctx, int on);
/* Kernel selection (no counterpart in the reference).  rebxamd_grfull on at most 256 bodies -- the sizes the reference's
 * own O(N^3) loops are used at -- runs every pass and every sweep of the evaluation in ONE launch of one workgroup (the
 * same per-row arithmetic in the same order as the queued passes: the same bits, an eighth of the latency).  on = 0
 * makes this context queue the passes at every size (default: 1). */
int rebxamd_grfull_one_launch(rebxamd_ctx* ctx, int on);
/* Kernel selection for rebxamd_wh_steps and rebxamd_integrate_force (no counterpart in the
 * reference).  When the massive bodies are a short prefix, the bodies behind it massless and the
 * force stack one streaming run, the pass over the massless bodies carries the whole update:
 * rebxamd_wh_steps -- forces, kick and Kepler drift (their accelerations never go through memory;
 * the acceleration columns stay zero); rebxamd_integrate_force with the Euler, RK2 or RK4 scheme
 * -- every stage evaluated in registers, velocities and first-stage accelerations written; with
 * the implicit-midpoint scheme -- one pass per round (mean velocity, forces, new estimate, partial
 * sums of the convergence test).
 * The switch also covers rebxamd_apply in default arithmetic on layouts with massive bodies
 * anywhere: when a stack holds gr or Jacobi/barycentric modify_orbits_forces (evaluated with
 * prefix sums) beside streaming forces that push on their sources, the pass of the streaming
 * forces also applies the last of those two forces, so the accelerations are read and written
 * once for both.
 * on = 0 makes this context run force passes and elementwise passes separately (default: 1).
 * Both compute the same expressions (the order of the additions into a_i differs at rounding
 * level in the rebxamd_apply case). */
int rebxamd_carried_steps(rebxamd_ctx* ctx, int on);
/* Steps per pass of rebxamd_wh_steps over the massless bodies behind the massive head (no counterpart in the reference).
 * The massless bodies feel the head but not one another, so one pass can take each of them through a batch of steps in
 * registers once the head has gone through those steps (one wavefront, one step after the other; the tables and the frame
 * record of every step are kept): positions and velocities then make one round trip through memory per batch instead of
 * one per step, and the run is bound by arithmetic instead of memory.  The operations on every body, and so the results,
 * are those of one pass per step, bit for bit.  Applies to a head of one body and to star + planets alike.
 * 1 <= steps_per_pass <= 64; default 32; 1 = one pass per step. */
int rebxamd_wh_batch(rebxamd_ctx* ctx, int steps_per_pass);
int rebxamd_timing_read(rebxamd_ctx* ctx, double* total_ms, int* launches);

/* gr_full on one device (gr_full.c:61-231).  The three phases are also exposed on row
 * ranges [row0,row1) so that a multi-GPU driver can shard rows and all-gather between
 * phases.  Arrays are length-n device columns.
 *   pot   : out pot[i] = sum_{k!=i} (1/c^2) G m_k / r_ik for rows in range (the reference's
 *           a2 sum, gr_full.c:125-130; its a1 sum is exactly 4 pot[i])
 *   const : out ac{x,y,z}[i] = the part that does not depend on the accelerations
 *           (gr_full.c:104-170); needs pot of ALL bodies
 *   sweep : out an{x,y,z}[i] = ac + terms linear in (a_newton + a_old) of ALL bodies
 *           (gr_full.c:181-203); also sets *dev_flag (device uint32) to 1 if any row in range
 *           fails the reference's stopping rule (gr_full.c:206-220)
 */
/* newton: an = incoming accelerations of ALL bodies (+ the 0-1 pair if gravity_ignore_10,
 *         gr_full.c:69-102), a0 = 0 (the first iterate, gr_full.c:74-76)
 * finish: view->a{x,y,z} += a{x,y,z} (gr_full.c:226-230) */
int rebxamd_grfull_newton(rebxamd_ctx* ctx, const rebxamd_view* view, int gravity_ignore_10,
                          double* anx, double* any_, double* anz, double* a0x, double* a0y, double* a0z, void* stream);
int rebxamd_grfull_finish(rebxamd_ctx* ctx, const rebxamd_view* view, const double* ax, const double* ay, const double* az, void* stream);
int rebxamd_grfull_pot(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int64_t row0, int64_t row1, double* pot, int mode, void* stream);
int rebxamd_grfull_const(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int64_t row0, int64_t row1, const double* pot,
                         double* acx, double* acy, double* acz, int mode, void* stream);
int rebxamd_grfull_sweep(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int64_t row0, int64_t row1,
                         const double* anx, const double* any_, const double* anz,      /* a_newton, all bodies */
                         const double* aox, const double* aoy, const double* aoz,       /* a_old, all bodies */
                         const double* acx, const double* acy, const double* acz,       /* a_const, rows in range valid */
                         double* outx, double* outy, double* outz,                      /* a_new, rows in range written */
                         uint32_t* dev_flag, int mode, void* stream);
/* rebxamd_grfull_pot on rows [row0,row1) that ALSO forms the pair sums of the first sweep (gr_full.c:173-199 with
 * a_old = 0: they need neither the potential sums nor a_const), sharing the separations and reciprocal distances of
 * the pairs: one all-pairs pass less per evaluation.  The sweep's sums wait inside the context;
 * rebxamd_grfull_sweep0_finish (same rows, same arguments as rebxamd_grfull_sweep) adds a_const to them, writes the
 * new iterate and raises the flag like any sweep.  One pair of calls at a time per context.  In reference-order
 * arithmetic the two calls are rebxamd_grfull_pot and rebxamd_grfull_sweep. */
int rebxamd_grfull_pot_sweep0(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int64_t row0, int64_t row1,
                              const double* anx, const double* any_, const double* anz, double* pot, int mode, void* stream);
int rebxamd_grfull_sweep0_finish(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int64_t row0, int64_t row1,
                                 const double* anx, const double* any_, const double* anz,
                                 const double* aox, const double* aoy, const double* aoz,
                                 const double* acx, const double* acy, const double* acz,
                                 double* outx, double* outy, double* outz, uint32_t* dev_flag, int mode, void* stream);
/* The per-sweep exchange of a row-sharded driver in two small kernels around its all-gather (no counterpart in the reference).
 * rebxamd_grfull_pack: send <- [ outx rows | outy rows | outz rows | vote ] of rows [row0, row0 + per) (rows >= n travel as
 * zeros), 3*per + 1 doubles; the vote is dev_flag (as raised by the sweep), which is cleared for the next sweep.
 * rebxamd_grfull_unpack: recv = `world` such messages in rank order -> every rank's rows into the full columns
 * (world*per doubles each); *stop_dev is set once no rank's rows fail the stopping rule (the word handed to
 * rebxamd_grfull_stop_word: sweeps queued behind it then return at once); *vote_dev <- 1 while some rank's do.  With
 * recv == NULL (all rows on this device) only the vote is handled, taken from dev_flag. */
int rebxamd_grfull_pack(rebxamd_ctx* ctx, const double* outx, const double* outy, const double* outz, int64_t row0, int64_t per, int64_t n,
                        uint32_t* dev_flag, double* send, void* stream);
int rebxamd_grfull_unpack(rebxamd_ctx* ctx, const double* recv, int world, int64_t per, double* outx, double* outy, double* outz,
                          uint32_t* dev_flag, uint32_t* stop_dev, int32_t* vote_dev, void* stream);
/* Lets a driver queue sweeps ahead of the host (no counterpart in the reference): while stop_dev != NULL,
 * the kernels of every rebxamd_grfull_sweep call on this context return at once, writing nothing, when the
 * device word *stop_dev is non-zero at the time they run.  NULL (default) switches this off. */
int rebxamd_grfull_stop_word(rebxamd_ctx* ctx, const uint32_t* stop_dev);
/* Whole evaluation on one device, including the reference's stopping rule; adds the result
 * to view->a{x,y,z}.  The ten sweeps are queued at once and the stopping rule is evaluated on the device
 * (sweeps past the one that meets it return at once).  *sweeps (may be NULL) receives the number of sweeps
 * run: asking for it waits for the stream; with NULL the call is asynchronous. */
int rebxamd_grfull(rebxamd_ctx* ctx, const rebxamd_view* view, double c, int gravity_ignore_10, int mode, int* sweeps, void* stream);

#ifdef __cplusplus
}
#endif
#endif

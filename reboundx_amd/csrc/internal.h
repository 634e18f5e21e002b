// internal.h -- structures shared by the host side of the device API and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/reboundx_amd.h"

namespace rebxamd {

// "Not set on this particle" in a per-particle parameter column: any NaN -- the quiet NaN of C's NAN and
// numpy.nan, its negative, the 0/0 of host arithmetic, whatever a computed column carries -- except the one
// pattern REBXAMD_NAN_VALUE_BITS, which stands for a parameter that is set and whose value is NaN: that one
// runs through the arithmetic and poisons what it touches, as a NaN-valued parameter does in the reference.
__host__ __device__ __forceinline__ bool is_set(double v) {
    return v == v || __builtin_bit_cast(unsigned long long, v) == REBXAMD_NAN_VALUE_BITS;
}
__host__ __device__ __forceinline__ double unset_value() { return __builtin_bit_cast(double, (unsigned long long)REBXAMD_UNSET_BITS); }

// One elementary pass, with everything that does not depend on the particle precomputed
// on the host (in the same fp64 arithmetic the reference uses for these scalars).
struct DevOp {
    int kind;       // rebxamd_op_kind; REBXAMD_OP_MOF is refined by `iarg` (coordinates)
    int src;        // source / reference particle index (-1: none)
    int iarg;       // max_iterations (gr) or coordinates (mof)
    int slot;       // index of this op's back-reaction partial sums, -1 if it has none
    double a, b;    // as in rebxamd_op
    double c, d;    // derived: radiation 1/c; gr c^2, 1/c^2; gr_potential 6(G m0)^2/c^2 is formed on device
};

struct Program {
    int nops;
    int nslots;
    int need_vel, need_beta, need_tides, need_tau;
    int gr_op;          // index of the (single) gr op in this run, or -1
    int mofcom_op;      // index of the (single) Jacobi/barycentric mof op, or -1
    double G;
    DevOp ops[REBXAMD_MAX_OPS];
};

// Plain copy of rebxamd_view for kernels.
struct View {
    long long n;
    long long n_head;
    const double *x, *y, *z, *vx, *vy, *vz, *m;
    double *ax, *ay, *az;
    const double *beta, *Rt, *k1, *ta, *te, *ti;
};

// State of one source / reference body as the kernels consume it.
struct SrcRec { double x, y, z, vx, vy, vz, m, Rt, k1, inv_m; };

// Small tables in device memory, written by a one-wavefront kernel right before the streaming
// pass and read by it through scalar loads (every lane wants the same entry):
//  - the state of each op's source particle;
//  - what the head pass hands to the tail pass.
struct HeadConsts {
    SrcRec src[REBXAMD_MAX_OPS];
    double src_acc[REBXAMD_MAX_OPS][3];     // acceleration of each op's source as the run of forces finds it (any-mass-layout pass)
    // gr: interior centre of mass seen by every body behind the head, i.e. (sum m*q) * (1/sum m)
    // for q in {pos, vel, Newtonian acc}, plus mu = G*m0.
    double gr_c[9];
    double gr_mu;
    // massive head bodies (x y z m), in index order, for the Newtonian term of gr
    double head[REBXAMD_MAX_HEAD][4];
    int nheadm;
    int head0;              // index of the first massive head body
    // modify_orbits_forces in Jacobi/barycentric coordinates: centre of mass of the head
    double com[7];          // x y z vx vy vz m
    int nbary;              // barycentric: number of head back-reaction terms that follow
    int pad;
    double bary[REBXAMD_MAX_HEAD][3];   // massratio_i * a_i for i = nbary-1 .. 0 (application order)
};

// What the streaming pass behind the head (tail bodies massless) does with the forces it has
// computed, besides or instead of adding them to the acceleration columns:
//   EPI_WH    a whole Wisdom-Holman step: kick by dtk, Kepler drift over h about body 0, whose record
//             (state before the drift, position after it, acceleration, G*m_0) is `star`; the first step of
//             a run also carries its leading half drift (over h0 about the record star0) ahead of the forces;
//   EPI_EULER / EPI_RK2 / EPI_RK4   a whole "integrate the force across dt" update of the velocities
//             (reference src/integrator_euler.c, integrator_rk2.c, integrator_rk4.c): every stage is
//             evaluated on the body while it sits in registers, against the head tables of that stage
//             (tab[0], tab[1], ...); the new velocities and the first-stage accelerations are written.
//             c0, c1, c2: Euler dt; RK2 2dt/3, dt/4, 3dt/4; RK4 dt/2, dt, dt/6 (formed on the host
//             exactly as the separate passes form them).
//   EPI_IM    one round of the implicit-midpoint iteration (src/integrator_implicit_midpoint.c:83-124):
//             the stack is evaluated at the mean of the velocity columns and of the current estimate
//             of the final velocities (aux), the new estimate orig + c0*a replaces the old one, and
//             the squared change and squared size (compare(), :47-62) are left as one pair of partial
//             sums per workgroup in `partials`.  Velocity and acceleration columns are not written.
//   EPI_APPLY (any-mass-layout pass, default arithmetic) the last pass of a prefix-sum force rides along:
//             a_i also receives aJ_i minus the reactions of the bodies k >= i (scan_kernels.hip, pass 5),
//             so the accelerations make one round trip for the streaming forces and that force together
//   EPI_WHB   nsub consecutive Wisdom-Holman steps of every body, which stays in registers in between;
//             an instance of its own so that the one-step pass keeps its register allocation
enum { EPI_NONE = 0, EPI_WH = 1, EPI_EULER = 2, EPI_RK2 = 3, EPI_RK4 = 4, EPI_IM = 5, EPI_APPLY = 6, EPI_WHB = 7 };
// What the per-body pass of a prefix-sum force leaves for the pass that applies it, which works one wavefront per part of a
// wavefront segment of that pass (segment_bodies bodies in SSUB parts, SW segments per workgroup of that pass): the force in
// the body's own frame (aj); per part, the mass of every body before the part (part_mass: the share of a body's force that
// falls back on the bodies before it, m / (eta + m), is formed from it and a prefix sum of the masses over the wavefront); and
// the reactions (3 doubles per row) summed over the workgroups behind each workgroup (behind_blocks), per segment (seg_totals)
// and up to the end of each of the first SSUB - 1 parts of every segment (seg_running), from which every wavefront adds up the
// reactions of the bodies behind its part.  all != nullptr (barycentric coordinates): every body receives the same total instead.
struct ScanTail { const double* aj[3]; const double* part_mass; const double* behind_blocks; const double* seg_totals; const double* seg_running;
                  const double* all; int segment_bodies; };
struct StepArgs { double h, dtk; const double* star; int epi; double c0, c1, c2; double* aux[3]; double* partials;
                  const double* star0; double h0;      // EPI_WH: a leading Kepler drift over h0 about the record star0 (nullptr: none)
                  ScanTail tail;                       // EPI_APPLY
                  // EPI_WH, one-body head: one extra workgroup of the pass fills the tables and the record of body 0
                  // for the NEXT step (next_hc, next_star; drift of body 0 over next_h), so that step needs no head
                  // pass of its own (head_ready: this step's tables were filled that way)
                  HeadConsts* next_hc; double* next_star; double next_h; int head_ready;
                  const unsigned* stop;                // EPI_IM with rounds queued ahead of the host: the pass returns at once when *stop is set
                  int lead_in_head;                    // EPI_WH: this step's head pass also records star0 and drifts body 0 over h0 (no pass of its own)
                  // EPI_WH behind a head of several massive bodies (star + planets): `star` and `star0` then describe the head's
                  // centre of mass instead of body 0 -- the origin of every massless body's Jacobi coordinates -- and the pass adds
                  // the pull of the head bodies (the tables' list of massive bodies) and the Jacobi term G M q/|q|^3 of the kick
                  int wh_head;
                  double* jac;                         // wh_head > 1 with the next step's head pass chained in: the head's Jacobi state between steps
                  // EPI_WH, one-body head, run left unsynchronised between steps: the chained head pass of the next step first records
                  // body 0 in next_lead and drifts it over next_h_lead (that step's leading drift); h == 0 then means "no trailing drift"
                  double* next_lead; double next_h_lead;
                  // EPI_WHB: the pass carries nsub (> 1) consecutive steps of every massless body, which stays in
                  // registers in between -- the bodies are independent of one another and body 0 of them, so only the tables
                  // (tabs[j]) and records (star + 16 j) change from step to step; the trailing drift is h_sub after every step
                  // but the last of the batch, h after that one
                  int nsub; double h_sub; const HeadConsts* tabs;
                  // EPI_WHB with next_hc set: the extra workgroup of the pass takes the head through the NEXT batch of next_nsub steps
                  // (tables next_hc[j], records next_star + 16 j; drift over h_sub after every step but the last, next_h after that
                  // one), so that batch needs no head launch of its own (head_ready)
                  int next_nsub;
                  // EPI_WH behind several massive bodies, run left unsynchronised between steps: the grains make no trailing drift
                  // (h == 0) while the head drifts over head_h after its kick; the head's state between kick and drift is kept in
                  // `pre` by a head launch of this step, in `next_pre` by the chained head step of the next one
                  double head_h; double* pre; double* next_pre; };

enum { WARN_GR_ITER = 1u, WARN_GRFULL_SWEEPS = 2u, WARN_IM_ITER = 4u };

// launchers implemented in stream_kernels.hip
// star != nullptr: a one-body head is recorded in `star`, its acceleration cleared, and drifted over h (Wisdom-Holman runs)
// lead != nullptr (one-body head only): body 0 is first recorded in `lead` and drifted over h_lead (the leading half drift of a run)
hipError_t launch_head(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, bool exact, hipStream_t st,
                       double* star = nullptr, double h = 0., double* lead = nullptr, double h_lead = 0.);
// the one-body head passes of nsub consecutive Wisdom-Holman steps (tables hc[j], records star + 16 j); see head_one_batch_kernel
hipError_t launch_head_batch(const View& v, const Program& prog, HeadConsts* hc, double* star, int nsub, double h, double h_last, double* lead,
                             double h_lead, hipStream_t st);
// star + planets: head passes and head steps of nsub consecutive Wisdom-Holman steps (tables hc[j], frame records star + 16 j); see wh_head_batch_kernel
// (lead != nullptr: the start of the run -- record `lead`, leading drift of the head over h_lead -- ahead of the first step)
hipError_t launch_wh_head_batch(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, long long K, double dtk, double* jac, double* star,
                                int nsub, double h, double h_last, bool exact, hipStream_t st, double* lead = nullptr, double h_lead = 0., double* pre = nullptr);
hipError_t launch_stream(const View& v, const Program& prog, long long lo, const HeadConsts* hc, double* partials,
                         int* nblocks_out, bool backreact, bool exact, bool vec2, bool fixed, unsigned* status, hipStream_t st,
                         const StepArgs* step = nullptr);
hipError_t launch_stage(const View& v, const Program& prog, HeadConsts* hc, hipStream_t st);
// implicit-midpoint update of at most REBXAMD_MAX_HEAD bodies in one launch (out[0] <- index of the round that met the test, 10: none)
hipError_t launch_head_implicit_midpoint(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, double dt, double* w, long long stride, int* out,
                                         bool exact, hipStream_t st);
// Euler / RK2 / RK4 update of at most REBXAMD_MAX_HEAD bodies (v.n of them) in one launch; hc: four tables, one per stage
hipError_t launch_head_integrate(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, int scheme, double c0, double c1, double c2,
                                 double* w, long long stride, bool exact, hipStream_t st);
// chunk_sums: scratch of REBXAMD_MAX_OPS * 64 * 3 doubles
// sums_out: nullptr = add the totals to the source particles; else store 3 doubles per op there
hipError_t launch_finalize(const View& v, const Program& prog, const double* partials, int nblocks, double* chunk_sums, double* sums_out, hipStream_t st);
// reference-order arithmetic, any mass layout: the accelerations of the ops' sources rebuilt in the reference's order (instead of launch_finalize)
// (nchain: bodies that can push -- up to the last massive one)
hipError_t launch_source_replay(const View& v, const Program& prog, const HeadConsts* hc, long long nchain, hipStream_t st);
hipError_t launch_add_pushes(const View& v, const Program& prog, const double* sums, hipStream_t st);
hipError_t launch_last_massive(const double* m, long long n, long long* out_dev, hipStream_t st);
hipError_t launch_reset(const View& v, hipStream_t st);
int stream_block_count(long long lo, long long n, bool vec2);

// gr and Jacobi/barycentric modify_orbits_forces for arbitrary mass layouts (scan_kernels.hip)
size_t scan_workspace_doubles(long long n);
size_t scan_range_doubles(long long n);
hipError_t launch_range_massive(const View& v, double G, double id0, double* ws, double* rec, double* count, hipStream_t st);
hipError_t launch_range_gr_newton(const View& v, const double* rec, long long M, double id0, bool exact, double* ws, double* sums10, hipStream_t st);
hipError_t launch_range_gr_forces(const View& v, const DevOp& op, double G, bool starts_at_origin, const double* before10, const double* m0,
                                  bool exact, double* ws, double* push3, unsigned* status, hipStream_t st);
hipError_t launch_range_gr_apply(const View& v, const double* after3, double* ws, hipStream_t st);
hipError_t launch_range_sums(const View& v, double* ws, double* sums7, hipStream_t st);
hipError_t launch_range_mof_forces(const View& v, const DevOp& op, double G, bool barycentric, bool starts_at_origin, const double* before7,
                                   double* ws, double* push3, hipStream_t st);
hipError_t launch_range_mof_apply(const View& v, bool barycentric, const double* after3, double* ws, hipStream_t st);
hipError_t launch_gr_hamiltonian(const View& v, double G, double c, double* ws, double* out, hipStream_t st);
// tail != nullptr: the pass that applies the force is left to the caller (an EPI_APPLY streaming pass) and described there
hipError_t launch_scan_op(const View& v, const DevOp& op, double G, bool exact, double* ws, unsigned* status, hipStream_t st, ScanTail* tail = nullptr);
// the same two forces in the reference's own order of operations (seq_kernels.hip): reference-order arithmetic,
// at most REBXAMD_SEQ_MAX_BODIES bodies; ws as for launch_scan_op
// nseq: bodies the chains walk (the index behind the last massive body when the bodies after it are massless, else v.n)
hipError_t launch_seq_op(const View& v, const DevOp& op, double G, double* ws, unsigned* status, hipStream_t st, long long nseq);
int stream_bodies_per_block();     // bodies one workgroup of the streaming pass covers
int scan_segment_bodies();         // bodies one wavefront of the prefix-sum passes covers

// velocity/position updates of the force integrators and steppers (step_kernels.hip)
hipError_t launch_lincomb(double* const out[3], const double* const v[3], const double* const a[3], double coef, long long n, bool exact, hipStream_t st);
hipError_t launch_rk2_final(double* const v[3], const double* const a1[3], const double* const a2[3], double b1, double b2, long long n, bool exact, hipStream_t st);
hipError_t launch_rk4_mid(double* const sv[3], const double* const v[3], double* const k2a[3], double* const k3a[3], double dt, long long n, bool exact, hipStream_t st);
hipError_t launch_rk4_final(double* const v[3], const double* const a1[3], const double* const a4[3], const double* const k23[3], double dt6, long long n, bool exact, hipStream_t st);
// body 0 alone: records it in `star` and drifts it over h
hipError_t launch_wh_star_restore(double* const x[3], const double* star, hipStream_t st);
hipError_t launch_wh_star(double* const x[3], double* const v[3], const double* m, double G, double h, double* star, hipStream_t st);
hipError_t launch_wh_drift(double* const x[3], double* const v[3], double* const a[3], const double* m, double G, double h, double dtk, long long n,
                           double* star, bool exact, hipStream_t st);
// Wisdom-Holman steps of a head of K <= 64 massive bodies in Jacobi coordinates (one wavefront; state between steps in `jac`,
// 8*64 doubles).  begin: inertial -> Jacobi, record of the centre of mass in `lead`, drift over h0, Jacobi -> inertial columns.
// step (after the head pass has left the additional accelerations of the head bodies in their columns): gravity among the head
// bodies without the pair (0,1), Jacobi kick over dtk, record of the centre of mass and its acceleration in `rec`, drift over h,
// Jacobi -> inertial columns, acceleration columns of the head cleared.
// closing half drift of the massless bodies [first, n) about the frame in `rec`; end of an unsynchronised run behind a head of
// several massive bodies (see wh_head_close_body)
hipError_t launch_wh_drift_about(double* const x[3], double* const v[3], long long first, long long n, double h, const double* rec, bool exact, hipStream_t st);
hipError_t launch_wh_head_close(const View& v, long long K, double G, double h, const double* pre, double* jac, const double* rec_last, double* rec_close,
                                bool exact, hipStream_t st);
hipError_t launch_wh_head_step(const View& v, long long K, double G, double dtk, double h, double* jac, double* rec, bool exact, hipStream_t st);
long long im_partial_count(long long n);
hipError_t launch_im_update(double* const fin[3], const double* const orig[3], double* const avgv[3], double* const avga[3], double dt,
                            long long n, double* partials, double* sums, bool exact, hipStream_t st, const unsigned* stop = nullptr);
// stop (two words, device): [0] set once the round's test is met, [1] rounds counted; kernels given the word return at once when [0] is set
hipError_t launch_im_sum(const double* partials, long long nb, double* sums, hipStream_t st, unsigned* stop = nullptr);
hipError_t launch_im_begin(double* const keep[3], const double* const v[3], double* const a[3], long long n, hipStream_t st);

// operators modify_mass and modify_orbits_direct (operator_kernels.hip)
long long operator_partial_doubles(long long n);
long long direct_chain_doubles(long long n);
hipError_t launch_direct_chain_rounds(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega,
                                      const double* tau_Omega, double dt, double G, double* ws, int max_rounds, const unsigned** ctl_out, hipStream_t st);
hipError_t launch_direct_chain_sequential(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega,
                                          const double* tau_Omega, double dt, double G, double* ws, hipStream_t st);
hipError_t launch_direct_chain_finish(const View& v, int coordinates, int refindex, double p_param, int has_p, const double* tau_omega,
                                      const double* tau_Omega, double dt, double G, double* ws, hipStream_t st);
hipError_t launch_modify_mass(const View& v, const double* tau_mass, double dt, double* ws, hipStream_t st);
hipError_t launch_modify_orbits_direct(const View& v, long long K, int coordinates, int refindex, double p_param, int has_p,
                                       const double* tau_omega, const double* tau_Omega, double dt, double G, double* hc16, hipStream_t st);

// potential energies of the position-dependent forces (potential_kernels.hip)
long long potential_block_count(long long n);
hipError_t launch_potentials(const View& v, const Program& prog, double* partials, double* out, hipStream_t st);

// gr_full launchers implemented in grfull_kernels.hip
// pass: 0 = potential sums, 1 = acceleration-independent part, 2 = one substitution sweep
size_t grfull_partial_bytes(long long n, long long nrows);
hipError_t launch_grfull_pass(int pass, const View& v, double G, double c, long long row0, long long row1, const double* pot,
                              const double* anx, const double* any_, const double* anz,
                              const double* aox, const double* aoy, const double* aoz,
                              const double* acx, const double* acy, const double* acz,
                              double* o0, double* o1, double* o2, unsigned* flag, double* part, bool exact, hipStream_t st,
                              const unsigned* stop = nullptr);
// sweeps queued ahead of the host: ctl = {stop, sweeps run, vote of the sweep just run}
hipError_t launch_grfull_sweep0_pot(const View& v, double G, double c, long long row0, long long row1,
                                    const double* anx, const double* any_, const double* anz, double* part4, hipStream_t st);
hipError_t launch_grfull_sweep0_pot_combine(bool sweep, const View& v, long long row0, long long row1, const double* part4,
                                            const double* acx, const double* acy, const double* acz, const double* aox, const double* aoy, const double* aoz,
                                            double* o0, double* o1, double* o2, unsigned* flag, hipStream_t st);
hipError_t launch_grfull_pack(const double* ox, const double* oy, const double* oz, long long row0, long long per, long long n, unsigned* flag, double* send, hipStream_t st);
hipError_t launch_grfull_unpack(const double* recv, int world, long long per, double* ox, double* oy, double* oz, unsigned* flag, unsigned* stop, int* vote_out, hipStream_t st);
hipError_t launch_grfull_vote(unsigned* ctl, hipStream_t st);
hipError_t launch_grfull_finish_select(const View& v, double* const b0[3], double* const b1[3], const unsigned* ctl, unsigned* status, hipStream_t st);
size_t grfull_hamiltonian_doubles(long long n);
hipError_t launch_grfull_hamiltonian(const View& v, double G, double c, double* ws, double** out, hipStream_t st);
// a whole gr_full evaluation of at most grfull_small_limit() bodies in one launch (w: 13 columns of n doubles; ctl: 3 words)
hipError_t launch_grfull_small(const View& v, double G, double c, int gravity_ignore_10, double* w, unsigned* ctl, unsigned* status, bool exact, hipStream_t st);
int grfull_small_limit();
hipError_t launch_grfull_prepare(const View& v, double G, int gravity_ignore_10, double* anx, double* any_, double* anz,
                                 double* a0x, double* a0y, double* a0z, hipStream_t st);
hipError_t launch_grfull_finish(const View& v, const double* ax, const double* ay, const double* az, hipStream_t st);

}  // namespace rebxamd

// stream_kernels.hip -- the per-particle forces as HBM-streaming kernels for gfx950.
//
// Layout: every particle quantity is an fp64 column; one thread owns two consecutive
// particles and moves each column with one 16-byte access, so a wavefront reads 1 KiB
// per column per instruction.  All forces stacked in one call are evaluated on the
// particle while it sits in registers: positions, velocities, mass and parameters are
// read once, accelerations are read once and written once.
//
// Sources (the star, a reference particle, ...) are few; their state is staged in LDS once
// per workgroup and read by broadcast.
//
// Two regimes (picked by the host, see device_api.hip):
//   head + tail : the massive bodies form a short prefix ("head", <= 64 bodies) followed by
//       massless particles.  head_kernel runs the head in one wavefront with the reference's
//       sequential semantics (ordered back-reaction sums, sequential Jacobi recurrences);
//       stream_kernel<BACKREACT=false> streams the tail, which by construction pushes on
//       nothing.
//   general : any mass layout.  stream_kernel<BACKREACT=true> streams all particles and leaves
//       per-workgroup partial sums of every back-reaction; finalize_kernel adds them to the
//       source particles in a fixed order (deterministic, no atomics).
#include "internal.h"
#include "effects.h"
#include "kepler.h"
#include "updates.h"
#include "wave_ops.h"
#include "scan_ops.h"
#include "wh_head.h"

namespace rebxamd {

static constexpr int BLOCK = 256;
static constexpr int WAVES = BLOCK / 64;
#ifndef STREAM_WAVES_PER_SIMD
#define STREAM_WAVES_PER_SIMD 4
#endif

// ------------------------------------------------------------------ small helpers

__device__ __forceinline__ double ld_or_nan(const double* col, long long i) {
    return col ? col[i] : unset_value();
}

__device__ __forceinline__ void load_src(const View& v, int idx, Src& S) {
    S.x = v.x[idx]; S.y = v.y[idx]; S.z = v.z[idx];
    S.vx = v.vx ? v.vx[idx] : 0.; S.vy = v.vy ? v.vy[idx] : 0.; S.vz = v.vz ? v.vz[idx] : 0.;
    S.m = v.m ? v.m[idx] : 0.;
    const double rt = ld_or_nan(v.Rt, idx), k1 = ld_or_nan(v.k1, idx);
    S.Rt = is_set(rt) ? rt : 0.;
    S.k1 = is_set(k1) ? k1 : 0.;
    S.inv_m = 1. / S.m;
}

// sum over the wavefront (every lane of the wavefront must call it)
__device__ __forceinline__ double wave_sum(double v) { return wave_total(v, (int)(threadIdx.x & 63)); }

// One force on one particle.  Returns true when the particle was acted on; `d` is its own
// update, `br` the push on the source (already signed).  Fast mode reuses the separation
// from the source (`g`) when the previous force on this particle had the same source.
template <bool EXACT>
__device__ __forceinline__ bool sourced_effect(const int kind, const DevOp& op, const Part& p, const Src& S, double G, Geo& g, V3& d, V3& br) {
    br.x = br.y = br.z = 0.;
    if (!EXACT) {
        if (kind == REBXAMD_OP_RADIATION && !is_set(p.beta)) return false;
        if (g.src != op.src) geo_compute(p, S, op.src, g);
    }
    switch (kind) {
    case REBXAMD_OP_RADIATION:
        if (EXACT) { if (!is_set(p.beta)) return false; rad_exact(p, S, G, op.a, d); }
        else rad_fast(p, S, g, G, op.c, d);
        return true;
    case REBXAMD_OP_J2:
        if (EXACT) j2_exact(p, S, G, op.a, op.b, d, br); else j2_fast(p, S, g, G, op.a, op.b, d, br);
        return true;
    case REBXAMD_OP_J4:
        if (EXACT) j4_exact(p, S, G, op.a, op.b, d, br); else j4_fast(p, S, g, G, op.a, op.b, d, br);
        return true;
    case REBXAMD_OP_GR_POTENTIAL: {
        if (EXACT) {
            const double prefac1 = [&] {
#pragma clang fp contract(off)
                return 6. * (G * S.m) * (G * S.m) / op.c;     // op.c = c*c
            }();
            grpot_exact(p, S, prefac1, d, br);
        } else {
            const double gm = G * S.m;
            grpot_fast(p, S, g, 6. * gm * gm * op.d, d, br);      // op.d = 1/c^2
        }
        return true;
    }
    case REBXAMD_OP_CENTRAL:
        if (EXACT) central_exact(p, S, op.a, op.b, d, br); else central_fast(p, S, g, op.a, op.b, d, br);
        return true;
    case REBXAMD_OP_TIDES:
        return EXACT ? tides_exact(p, S, G, d, br) : tides_fast(p, S, g, G, d, br);
    default:
        return false;
    }
}

// modify_orbits_forces relative to a particle (reference src/rebxtools.c:108-121):
// p.a += a; p.a -= w*a; ref.a -= w*a with w = m/(m_ref+m).
template <bool EXACT>
__device__ __forceinline__ void mof_particle(const Part& p, const Src& ref, V3& acc, V3& br) {
    V3 a;
    if (EXACT) mof_exact(p, ref, a); else mof_fast(p, ref, a);
    {
#pragma clang fp contract(off)
        acc.x += a.x; acc.y += a.y; acc.z += a.z;
        const double w = p.m / (ref.m + p.m);
        const double wx = w * a.x, wy = w * a.y, wz = w * a.z;
        acc.x -= wx; acc.y -= wy; acc.z -= wz;
        br.x = -wx; br.y = -wy; br.z = -wz;
    }
}

// ------------------------------------------------------------------ sequential recurrences
// (restated from REBOUND 3.x tools.c / transformations.c; run by one lane over the head)

struct Com { double x, y, z, vx, vy, vz, m; };

__device__ __forceinline__ void com_add(Com& a, const Src& b) {
#pragma clang fp contract(off)
    a.x = a.x * a.m + b.x * b.m;   a.y = a.y * a.m + b.y * b.m;   a.z = a.z * a.m + b.z * b.m;
    a.vx = a.vx * a.m + b.vx * b.m; a.vy = a.vy * a.m + b.vy * b.m; a.vz = a.vz * a.m + b.vz * b.m;
    a.m += b.m;
    if (a.m > 0.) {
        a.x /= a.m; a.y /= a.m; a.z /= a.m; a.vx /= a.m; a.vy /= a.m; a.vz /= a.m;
    }
}
__device__ __forceinline__ void com_sub(Com& a, const Src& b) {
#pragma clang fp contract(off)
    a.x = a.x * a.m - b.x * b.m;   a.y = a.y * a.m - b.y * b.m;   a.z = a.z * a.m - b.z * b.m;
    a.vx = a.vx * a.m - b.vx * b.m; a.vy = a.vy * a.m - b.vy * b.m; a.vz = a.vz * a.m - b.vz * b.m;
    a.m -= b.m;
    if (a.m > 0.) {
        a.x /= a.m; a.y /= a.m; a.z /= a.m; a.vx /= a.m; a.vy /= a.m; a.vz /= a.m;
    }
}

// ------------------------------------------------------------------ head kernel
// One wavefront; lane t owns head particle t.

struct HeadShared {
    Src hp[REBXAMD_MAX_HEAD];
    double br[REBXAMD_MAX_HEAD][3];
    int brf[REBXAMD_MAX_HEAD];
    double na[REBXAMD_MAX_HEAD][3];     // Newtonian acc (inertial), later inertial gr acc
    double pj[REBXAMD_MAX_HEAD][9];     // Jacobi pos, vel, acc
    Com com[REBXAMD_MAX_HEAD + 1];
};

// star != nullptr (Wisdom-Holman runs, one head body): instead of storing body 0's acceleration the
// pass records body 0 for the stepping pass that follows -- star[0..5] its state, star[9..11] its
// acceleration (the column is cleared), star[12] its mass parameter G*m_0 -- drifts it linearly
// over h and records the new position in star[6..8].
// (a device function of one wavefront, t = lane: the kernel below calls it, and so does the extra workgroup of a stepping pass
// behind a head of several massive bodies, which runs the head pass of the next step)
template <bool EXACT>
__device__ __forceinline__ void head_body(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, double* star, double h, HeadShared& sh, const int t) {
    const int K = (int)(v.n_head < v.n ? v.n_head : v.n);
    const bool valid = t < K;
    const double G = prog.G;

    Part p;
    V3 acc = {0., 0., 0.};
    if (valid) {
        p.x = v.x[t]; p.y = v.y[t]; p.z = v.z[t];
        p.vx = v.vx ? v.vx[t] : 0.; p.vy = v.vy ? v.vy[t] : 0.; p.vz = v.vz ? v.vz[t] : 0.;
        p.m = v.m ? v.m[t] : 0.;
        p.beta = ld_or_nan(v.beta, t);
        p.Rt = ld_or_nan(v.Rt, t); p.k1 = ld_or_nan(v.k1, t);
        p.ta = ld_or_nan(v.ta, t); p.te = ld_or_nan(v.te, t); p.ti = ld_or_nan(v.ti, t);
        acc.x = v.ax[t]; acc.y = v.ay[t]; acc.z = v.az[t];
        load_src(v, t, sh.hp[t]);
    }
    wave_sync();
    // tables for the tail pass: every op's source (all inside the head) and the massive head bodies
    if (t < prog.nops) {
        int s = prog.ops[t].src;
        if (prog.ops[t].kind == REBXAMD_OP_GR) s = 0;
        if (s >= 0) hc->src[t] = sh.hp[s];
    }
    {
        const double mt = valid ? p.m : 0.;
        const unsigned long long mask = __ballot(mt != 0.);
        if (mt != 0.) {
            const int c = __popcll(mask & ((1ull << t) - 1ull));
            hc->head[c][0] = p.x; hc->head[c][1] = p.y; hc->head[c][2] = p.z; hc->head[c][3] = mt;
        }
        if (t == 0) { hc->nheadm = __popcll(mask); hc->head0 = mask ? __ffsll((long long)mask) - 1 : -1; }
    }

    Geo geo; geo.src = -1; geo.dx = geo.dy = geo.dz = geo.ri = 0.;
    for (int o = 0; o < prog.nops; o++) {
        const DevOp op = prog.ops[o];
        if (op.kind == REBXAMD_OP_GR) {
            // --- Newtonian acceleration among the head bodies (massive partners, ascending)
            double nx = 0., ny = 0., nz = 0.;
            if (valid) {
                for (int u = 0; u < K; u++) {
                    if (u == t || sh.hp[u].m == 0.) continue;
                    if (EXACT) newton_pair_exact(p.x, p.y, p.z, sh.hp[u].x, sh.hp[u].y, sh.hp[u].z, sh.hp[u].m, u < t, G, nx, ny, nz);
                    else newton_pair_fast(p.x, p.y, p.z, sh.hp[u].x, sh.hp[u].y, sh.hp[u].z, sh.hp[u].m, G, nx, ny, nz);
                }
                sh.na[t][0] = nx; sh.na[t][1] = ny; sh.na[t][2] = nz;
            }
            wave_sync();
            // --- inertial -> Jacobi (sequential prefix recurrence), lane 0
            if (t == 0) {
#pragma clang fp contract(off)
                double eta = sh.hp[0].m;
                double s[9];
                s[0] = eta * sh.hp[0].x;  s[1] = eta * sh.hp[0].y;  s[2] = eta * sh.hp[0].z;
                s[3] = eta * sh.hp[0].vx; s[4] = eta * sh.hp[0].vy; s[5] = eta * sh.hp[0].vz;
                s[6] = eta * sh.na[0][0]; s[7] = eta * sh.na[0][1]; s[8] = eta * sh.na[0][2];
                for (int i = 1; i < K; i++) {
                    const double ei = 1. / eta;
                    const Src q = sh.hp[i];
                    const double qa0 = sh.na[i][0], qa1 = sh.na[i][1], qa2 = sh.na[i][2];
                    eta += q.m;
                    sh.pj[i][0] = q.x - s[0] * ei;  sh.pj[i][1] = q.y - s[1] * ei;  sh.pj[i][2] = q.z - s[2] * ei;
                    sh.pj[i][3] = q.vx - s[3] * ei; sh.pj[i][4] = q.vy - s[4] * ei; sh.pj[i][5] = q.vz - s[5] * ei;
                    sh.pj[i][6] = qa0 - s[6] * ei;  sh.pj[i][7] = qa1 - s[7] * ei;  sh.pj[i][8] = qa2 - s[8] * ei;
                    s[0] += q.m * q.x;  s[1] += q.m * q.y;  s[2] += q.m * q.z;
                    s[3] += q.m * q.vx; s[4] += q.m * q.vy; s[5] += q.m * q.vz;
                    s[6] += q.m * qa0;  s[7] += q.m * qa1;  s[8] += q.m * qa2;
                }
                // what every (massless) body behind the head subtracts: s * (1/eta)
                const double ei = 1. / eta;
                for (int k = 0; k < 9; k++) hc->gr_c[k] = s[k] * ei;
                hc->gr_mu = G * sh.hp[0].m;
                sh.com[0].m = eta;      // total mass, start of the reverse recurrence
            }
            wave_sync();
            // --- post-Newtonian term per Jacobi body
            if (valid && t >= 1) {
                Jac j;
                j.x = sh.pj[t][0]; j.y = sh.pj[t][1]; j.z = sh.pj[t][2];
                j.vx = sh.pj[t][3]; j.vy = sh.pj[t][4]; j.vz = sh.pj[t][5];
                j.ax = sh.pj[t][6]; j.ay = sh.pj[t][7]; j.az = sh.pj[t][8];
                V3 corr;
                const double mu = G * sh.hp[0].m;
                const int q = EXACT ? gr_term_exact(j, mu, op.c, op.iarg, corr)
                                    : gr_term_fast(j, fast_rsqrt(j.x * j.x + j.y * j.y + j.z * j.z), mu, op.d, op.iarg, corr);
                if (q == 10) atomicOr(status, WARN_GR_ITER);
                sh.pj[t][6] = corr.x; sh.pj[t][7] = corr.y; sh.pj[t][8] = corr.z;
            }
            wave_sync();
            // --- Jacobi -> inertial accelerations (sequential reverse recurrence), lane 0
            if (t == 0) {
#pragma clang fp contract(off)
                double eta = sh.com[0].m;
                double s0 = 0. * eta, s1 = 0. * eta, s2 = 0. * eta;   // Jacobi slot 0 carries zero acceleration
                for (int i = K - 1; i > 0; i--) {
                    const double ei = 1. / eta;
                    const double mi = sh.hp[i].m;
                    s0 = (s0 - mi * sh.pj[i][6]) * ei;
                    s1 = (s1 - mi * sh.pj[i][7]) * ei;
                    s2 = (s2 - mi * sh.pj[i][8]) * ei;
                    sh.na[i][0] = sh.pj[i][6] + s0;
                    sh.na[i][1] = sh.pj[i][7] + s1;
                    sh.na[i][2] = sh.pj[i][8] + s2;
                    eta -= mi;
                    s0 *= eta; s1 *= eta; s2 *= eta;
                }
                const double mi0 = 1. / eta;
                sh.na[0][0] = s0 * mi0; sh.na[0][1] = s1 * mi0; sh.na[0][2] = s2 * mi0;
            }
            wave_sync();
            if (valid) { acc.x += sh.na[t][0]; acc.y += sh.na[t][1]; acc.z += sh.na[t][2]; }
            wave_sync();
        } else if (op.kind == REBXAMD_OP_MOF && op.iarg != REBX_COORDINATES_PARTICLE) {
            const bool jacobi = (op.iarg == REBX_COORDINATES_JACOBI);
            // --- centres of mass: com[K] = all head bodies; Jacobi: com[i] = bodies 0..i-1
            if (t == 0) {
                Com c = {0., 0., 0., 0., 0., 0., 0.};
                for (int i = 0; i < K; i++) com_add(c, sh.hp[i]);
                sh.com[K] = c;
                hc->com[0] = c.x; hc->com[1] = c.y; hc->com[2] = c.z;
                hc->com[3] = c.vx; hc->com[4] = c.vy; hc->com[5] = c.vz; hc->com[6] = c.m;
                if (jacobi) {
                    for (int i = K - 1; i >= 1; i--) { com_sub(c, sh.hp[i]); sh.com[i] = c; }
                }
                hc->nbary = jacobi ? 0 : K;
            }
            wave_sync();
            // --- each body's own force and its mass-weighted reaction
            const bool acts = valid && (jacobi ? t >= 1 : true);
            V3 a = {0., 0., 0.};
            if (acts) {
                const Com c = jacobi ? sh.com[t] : sh.com[K];
                Src ref; ref.x = c.x; ref.y = c.y; ref.z = c.z; ref.vx = c.vx; ref.vy = c.vy; ref.vz = c.vz; ref.m = c.m;
                if (EXACT) mof_exact(p, ref, a); else mof_fast(p, ref, a);
                double w;
                {
#pragma clang fp contract(off)
                    w = jacobi ? p.m / (c.m + p.m) : p.m / c.m;
                    sh.br[t][0] = w * a.x; sh.br[t][1] = w * a.y; sh.br[t][2] = w * a.z;
                }
                if (!jacobi) {
                    hc->bary[K - 1 - t][0] = sh.br[t][0];
                    hc->bary[K - 1 - t][1] = sh.br[t][1];
                    hc->bary[K - 1 - t][2] = sh.br[t][2];
                }
            }
            wave_sync();
            // --- reference order: i descending; body j gains its own force at i==j and loses
            //     w_i a_i for every i >= j (Jacobi, i>=1) or for every i (barycentric)
            if (valid) {
#pragma clang fp contract(off)
                const int lowest = jacobi ? (t > 1 ? t : 1) : 0;
                for (int i = K - 1; i >= lowest; i--) {
                    if (i == t) { acc.x += a.x; acc.y += a.y; acc.z += a.z; }
                    acc.x -= sh.br[i][0]; acc.y -= sh.br[i][1]; acc.z -= sh.br[i][2];
                }
            }
            wave_sync();
        } else {
            // --- forces with one source / reference particle
            const int src = op.src;
            const Src S = sh.hp[src];
            V3 d = {0., 0., 0.}, br = {0., 0., 0.};
            bool acted = false;
            if (valid && t != src) {
                if (op.kind == REBXAMD_OP_MOF) {
                    mof_particle<EXACT>(p, S, acc, br);
                    acted = true;
                } else {
                    acted = sourced_effect<EXACT>(op.kind, op, p, S, G, geo, d, br);
                    if (acted) {
#pragma clang fp contract(off)
                        acc.x += d.x; acc.y += d.y; acc.z += d.z;
                    }
                }
            }
            sh.br[t][0] = br.x; sh.br[t][1] = br.y; sh.br[t][2] = br.z;
            sh.brf[t] = acted ? 1 : 0;
            wave_sync();
            if (valid && t == src && op.kind != REBXAMD_OP_RADIATION) {
#pragma clang fp contract(off)
                if (op.kind == REBXAMD_OP_MOF) {        // the reference walks this loop backwards
                    for (int i = K - 1; i >= 0; i--)
                        if (sh.brf[i]) { acc.x += sh.br[i][0]; acc.y += sh.br[i][1]; acc.z += sh.br[i][2]; }
                } else {
                    for (int i = 0; i < K; i++)
                        if (sh.brf[i]) { acc.x += sh.br[i][0]; acc.y += sh.br[i][1]; acc.z += sh.br[i][2]; }
                }
            }
            wave_sync();
        }
    }
    if (star != nullptr) {
        if (t == 0 && valid) {
#pragma clang fp contract(off)
            star[0] = p.x; star[1] = p.y; star[2] = p.z; star[3] = p.vx; star[4] = p.vy; star[5] = p.vz;
            star[9] = acc.x; star[10] = acc.y; star[11] = acc.z; star[12] = G * p.m;
            v.ax[0] = 0.; v.ay[0] = 0.; v.az[0] = 0.;
            const double nx = p.x + h * p.vx, ny = p.y + h * p.vy, nz = p.z + h * p.vz;
            const_cast<double*>(v.x)[0] = nx; const_cast<double*>(v.y)[0] = ny; const_cast<double*>(v.z)[0] = nz;
            star[6] = nx; star[7] = ny; star[8] = nz;
        }
        return;
    }
    if (valid) { v.ax[t] = acc.x; v.ay[t] = acc.y; v.az[t] = acc.z; }
}
template <bool EXACT>
__global__ __launch_bounds__(64) void head_kernel(View v, Program prog, HeadConsts* hc, unsigned* status, double* star, double h) {
    __shared__ HeadShared sh;
    head_body<EXACT>(v, prog, hc, status, star, h, sh, threadIdx.x);
}

// ------------------------------------------------------------------ force integrators on a head alone
// Euler, RK2 and RK4 (integrator_euler.c:33-41, integrator_rk2.c:38-67, integrator_rk4.c:40-91) on at most 64 bodies: every
// stage's head pass and every velocity update in ONE launch of one wavefront, instead of a launch per pass and per update
// (a dozen for RK4).  The stages go through head_body and the update functions of updates.h exactly as the separate kernels
// do, and stage s leaves its tables in hc[s] as they do -- for the pass over the massless bodies behind the head, which
// evaluates all stages per body.  w: nine columns of `stride` doubles (stage velocities, two sets of stage accelerations).
// c0, c1, c2: the scheme's coefficients as the host forms them (RK2: 2dt/3, dt/4, 3dt/4; RK4: dt/2, dt, dt/6; Euler: dt).
template <bool EXACT>
__global__ __launch_bounds__(64) void head_integrate_kernel(View v, Program prog, HeadConsts* hc, unsigned* status, int scheme, double c0, double c1, double c2,
                                                            double* w, long long stride) {
    __shared__ HeadShared sh;
    const int t = threadIdx.x;
    const int K = (int)(v.n_head < v.n ? v.n_head : v.n);
    const bool valid = t < K;
    double* const vx = const_cast<double*>(v.vx); double* const vy = const_cast<double*>(v.vy); double* const vz = const_cast<double*>(v.vz);
    double* const sv[3] = {w, w + stride, w + 2 * stride};
    double* const ka[3] = {w + 3 * stride, w + 4 * stride, w + 5 * stride};
    double* const kb[3] = {w + 6 * stride, w + 7 * stride, w + 8 * stride};
    // integrate_force.c:46: the accelerations are cleared first; the stage copies are made after that, so theirs start at zero
    if (valid) {
        v.ax[t] = 0.; v.ay[t] = 0.; v.az[t] = 0.;
        for (int k = 0; k < 3; k++) { ka[k][t] = 0.; kb[k][t] = 0.; }
    }
    wave_sync();
    head_body<EXACT>(v, prog, hc, status, nullptr, 0., sh, t);                      // k1 -> the accelerations of v
    wave_sync();
    if (scheme == REBXAMD_INTEGRATOR_EULER) {
        if (valid) { vx[t] = axpy1<EXACT>(vx[t], c0, v.ax[t]); vy[t] = axpy1<EXACT>(vy[t], c0, v.ay[t]); vz[t] = axpy1<EXACT>(vz[t], c0, v.az[t]); }
        return;
    }
    View sa = v, sb = v;                    // stage states share positions, masses, parameters
    sa.vx = sv[0]; sa.vy = sv[1]; sa.vz = sv[2]; sa.ax = ka[0]; sa.ay = ka[1]; sa.az = ka[2];
    sb.vx = sv[0]; sb.vy = sv[1]; sb.vz = sv[2]; sb.ax = kb[0]; sb.ay = kb[1]; sb.az = kb[2];
    if (valid) { sv[0][t] = axpy1<EXACT>(vx[t], c0, v.ax[t]); sv[1][t] = axpy1<EXACT>(vy[t], c0, v.ay[t]); sv[2][t] = axpy1<EXACT>(vz[t], c0, v.az[t]); }
    wave_sync();
    head_body<EXACT>(sa, prog, hc + 1, status, nullptr, 0., sh, t);                 // k2 -> ka
    wave_sync();
    if (scheme == REBXAMD_INTEGRATOR_RK2) {
        if (valid) {
            vx[t] = rk2_combine<EXACT>(vx[t], c1, v.ax[t], c2, ka[0][t]);
            vy[t] = rk2_combine<EXACT>(vy[t], c1, v.ay[t], c2, ka[1][t]);
            vz[t] = rk2_combine<EXACT>(vz[t], c1, v.az[t], c2, ka[2][t]);
        }
        return;
    }
    if (valid) { sv[0][t] = axpy1<EXACT>(vx[t], c0, ka[0][t]); sv[1][t] = axpy1<EXACT>(vy[t], c0, ka[1][t]); sv[2][t] = axpy1<EXACT>(vz[t], c0, ka[2][t]); }
    wave_sync();
    head_body<EXACT>(sb, prog, hc + 2, status, nullptr, 0., sh, t);                 // k3 -> kb
    wave_sync();
    if (valid) {                            // stage velocity for k4 = v + dt k3;  kb = k3 + k2;  ka = 0
        for (int k = 0; k < 3; k++) {
            const double k3 = kb[k][t];
            sv[k][t] = axpy1<EXACT>(k == 0 ? vx[t] : (k == 1 ? vy[t] : vz[t]), c1, k3);
            kb[k][t] = k3 + ka[k][t];
            ka[k][t] = 0.;
        }
    }
    wave_sync();
    head_body<EXACT>(sa, prog, hc + 3, status, nullptr, 0., sh, t);                 // k4 -> ka
    wave_sync();
    if (valid) {
        vx[t] = rk4_combine<EXACT>(vx[t], c2, v.ax[t], ka[0][t], kb[0][t]);
        vy[t] = rk4_combine<EXACT>(vy[t], c2, v.ay[t], ka[1][t], kb[1][t]);
        vz[t] = rk4_combine<EXACT>(vz[t], c2, v.az[t], ka[2][t], kb[2][t]);
    }
}

// Implicit midpoint (integrator_implicit_midpoint.c:83-124) on at most 64 bodies, every round in the same launch: forces at the
// averaged velocities (head_body), the new estimate of the final velocities, the reference's test on the summed squared change
// and squared size (compare(), :47-62) -- formed as the separate kernels form them for one workgroup: a wavefront's shuffle tree
// -- and the averaged state for the next round; at most ten rounds.  out[0] <- the index of the round that met the test (10: none
// did, and the warning is raised).  w: nine columns of `stride` doubles.
template <bool EXACT>
__global__ __launch_bounds__(64) void head_implicit_midpoint_kernel(View v, Program prog, HeadConsts* hc, unsigned* status, double dt, double* w, long long stride, int* out) {
    __shared__ HeadShared sh;
    const int t = threadIdx.x;
    const int K = (int)(v.n_head < v.n ? v.n_head : v.n);
    const bool valid = t < K;
    double* const vx = const_cast<double*>(v.vx); double* const vy = const_cast<double*>(v.vy); double* const vz = const_cast<double*>(v.vz);
    double* const sv[3] = {w, w + stride, w + 2 * stride};
    double* const ka[3] = {w + 3 * stride, w + 4 * stride, w + 5 * stride};
    double ox = 0., oy = 0., oz = 0., fx = 0., fy = 0., fz = 0.;
    if (valid) {
        v.ax[t] = 0.; v.ay[t] = 0.; v.az[t] = 0.;                      // integrate_force.c:46
        ox = vx[t]; oy = vy[t]; oz = vz[t];
        fx = ox; fy = oy; fz = oz;                                      // estimate of the final velocities
        sv[0][t] = ox; sv[1][t] = oy; sv[2][t] = oz;                    // averaged velocities of the first round
        ka[0][t] = 0.; ka[1][t] = 0.; ka[2][t] = 0.;
    }
    View sa = v;
    sa.vx = sv[0]; sa.vy = sv[1]; sa.vz = sv[2]; sa.ax = ka[0]; sa.ay = ka[1]; sa.az = ka[2];
    int it = 0;
    for (it = 0; it < 10; it++) {
        wave_sync();
        head_body<EXACT>(sa, prog, hc, status, nullptr, 0., sh, t);
        wave_sync();
        double d2 = 0., t2 = 0.;
        if (valid) {
            const double nx = axpy1<EXACT>(ox, dt, ka[0][t]), ny = axpy1<EXACT>(oy, dt, ka[1][t]), nz = axpy1<EXACT>(oz, dt, ka[2][t]);
            const double dx = nx - fx, dy = ny - fy, dz = nz - fz;
            d2 = dx * dx + dy * dy + dz * dz;
            t2 = nx * nx + ny * ny + nz * nz;
            fx = nx; fy = ny; fz = nz;
            sv[0][t] = 0.5 * (ox + nx); sv[1][t] = 0.5 * (oy + ny); sv[2][t] = 0.5 * (oz + nz);
            ka[0][t] = 0.; ka[1][t] = 0.; ka[2][t] = 0.;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_down(d2, off, 64); t2 += __shfl_down(t2, off, 64); }
        const double s = __shfl(d2, 0, 64), tt = __shfl(t2, 0, 64);
        if (s / tt < 2.220446049250313e-16 * 2.220446049250313e-16) break;
    }
    if (valid) { vx[t] = fx; vy[t] = fy; vz[t] = fz; }
    if (t == 0) {
        if (it == 10) atomicOr(status, WARN_IM_ITER);
        out[0] = it;
    }
}

// The head pass when the head is ONE body (a star in front of massless bodies) and no op works in
// barycentric coordinates: nothing acts on that body (every op has it as its source, and its
// Jacobi frame is empty), so the pass only fills the tables -- with the arithmetic of head_kernel
// for K = 1, rounding included -- and, in a Wisdom-Holman run, records and drifts the body.  A
// fraction of head_kernel's latency: no loops over ops with barriers, no LDS.
// lead != nullptr (first step of a Wisdom-Holman run): body 0 is first recorded in `lead` and drifted over h_lead
// -- the record the tail pass needs for its leading half drift -- and the tables are then filled for the drifted body.
// S: body 0 (the same in every lane), taken as it stands and left as the pass leaves it (drifted, in a stepping run), so that a
// wavefront can make one head pass after the other without going through memory; write_row: also store the new position.
__device__ __forceinline__ void head_one_core(const View& v, const Program& prog, HeadConsts* hc, double* star, double h, const int t,
                                              double* lead, double h_lead, Src& S, bool write_row) {
    if (lead != nullptr) {
#pragma clang fp contract(off)
        const double nx = S.x + h_lead * S.vx, ny = S.y + h_lead * S.vy, nz = S.z + h_lead * S.vz;
        if (t == 0) {
            lead[0] = S.x; lead[1] = S.y; lead[2] = S.z; lead[3] = S.vx; lead[4] = S.vy; lead[5] = S.vz;
            lead[6] = nx; lead[7] = ny; lead[8] = nz; lead[12] = prog.G * S.m;
        }
        S.x = nx; S.y = ny; S.z = nz;                   // (carried in registers: the columns are written once, below)
    }
    if (t < prog.nops) {
        int s = prog.ops[t].src;
        if (prog.ops[t].kind == REBXAMD_OP_GR) s = 0;
        if (s >= 0) hc->src[t] = S;
    }
    if (t == 0) {
        if (S.m != 0.) { hc->head[0][0] = S.x; hc->head[0][1] = S.y; hc->head[0][2] = S.z; hc->head[0][3] = S.m; }
        hc->nheadm = S.m != 0. ? 1 : 0;
        hc->head0 = S.m != 0. ? 0 : -1;
        // In a stepping run the acceleration of body 0 starts from zero whatever the column holds (the columns of
        // the bodies behind it are neither read nor written there).
        double a0x = star != nullptr ? 0. : v.ax[0], a0y = star != nullptr ? 0. : v.ay[0], a0z = star != nullptr ? 0. : v.az[0];
        if (prog.gr_op >= 0) {
#pragma clang fp contract(off)
            const double eta = S.m, ei = 1. / eta;
            hc->gr_c[0] = (eta * S.x) * ei;  hc->gr_c[1] = (eta * S.y) * ei;  hc->gr_c[2] = (eta * S.z) * ei;
            hc->gr_c[3] = (eta * S.vx) * ei; hc->gr_c[4] = (eta * S.vy) * ei; hc->gr_c[5] = (eta * S.vz) * ei;
            hc->gr_c[6] = (eta * 0.) * ei;   hc->gr_c[7] = (eta * 0.) * ei;   hc->gr_c[8] = (eta * 0.) * ei;
            hc->gr_mu = prog.G * S.m;
            // what the back-transformation of an empty Jacobi chain leaves on the body: zero (NaN for a massless one)
            const double na0 = ((0. * eta) * ei);
            a0x += na0; a0y += na0; a0z += na0;
            if (star == nullptr) { v.ax[0] = a0x; v.ay[0] = a0y; v.az[0] = a0z; }
        }
        if (prog.mofcom_op >= 0) {
            Com c = {0., 0., 0., 0., 0., 0., 0.};
            com_add(c, S);
            hc->com[0] = c.x; hc->com[1] = c.y; hc->com[2] = c.z;
            hc->com[3] = c.vx; hc->com[4] = c.vy; hc->com[5] = c.vz; hc->com[6] = c.m;
            hc->nbary = 0;
        }
        if (star != nullptr) {
            star[0] = S.x; star[1] = S.y; star[2] = S.z; star[3] = S.vx; star[4] = S.vy; star[5] = S.vz;
            star[9] = a0x; star[10] = a0y; star[11] = a0z; star[12] = prog.G * S.m;
        }
    }
    if (star != nullptr) {
#pragma clang fp contract(off)
        const double nx = S.x + h * S.vx, ny = S.y + h * S.vy, nz = S.z + h * S.vz;
        if (t == 0) {
            star[6] = nx; star[7] = ny; star[8] = nz;
            if (write_row) {
                v.ax[0] = 0.; v.ay[0] = 0.; v.az[0] = 0.;
                const_cast<double*>(v.x)[0] = nx; const_cast<double*>(v.y)[0] = ny; const_cast<double*>(v.z)[0] = nz;
            }
        }
        S.x = nx; S.y = ny; S.z = nz;
    }
}
__device__ __forceinline__ void head_one_body(const View& v, const Program& prog, HeadConsts* hc, double* star, double h, const int t,
                                              double* lead = nullptr, double h_lead = 0.) {
    Src S;
    load_src(v, 0, S);                                  // every lane reads the same body
    head_one_core(v, prog, hc, star, h, t, lead, h_lead, S, true);
}
__global__ __launch_bounds__(64) void head_one_kernel(View v, Program prog, HeadConsts* hc, double* star, double h, double* lead, double h_lead) {
    head_one_body(v, prog, hc, star, h, threadIdx.x, lead, h_lead);
}
// The one-body head passes of nsub consecutive steps of a Wisdom-Holman run, one after the other in one wavefront (body 0
// carried in registers): tables hc[j] and records star[16 j] for step j; body 0 drifts over h after every step but the
// last of the batch, over h_last after that one.  What a pass needs to carry nsub steps of every massless body.
__global__ __launch_bounds__(64) void head_one_batch_kernel(View v, Program prog, HeadConsts* hc, double* star, int nsub, double h, double h_last,
                                                            double* lead, double h_lead) {
    Src S;
    load_src(v, 0, S);
    for (int j = 0; j < nsub; j++)
        head_one_core(v, prog, hc + j, star + 16 * j, j + 1 < nsub ? h : h_last, threadIdx.x, j == 0 ? lead : nullptr, h_lead, S, j + 1 == nsub);
}

// ------------------------------------------------------------------ streaming kernel

typedef double d2v __attribute__((ext_vector_type(2)));

// Which optional columns a kernel instance reads.  Instances exist for the column sets of
// the common force stacks; any other stack runs on the all-columns instance.
enum { COL_VEL = 1, COL_BETA = 2, COL_TIDES = 4, COL_TAU = 8, COL_ALL = 15 };

// Read-once columns (positions, velocities, parameters) are loaded with the nontemporal hint
// so that they do not evict the accelerations, which are read and written every pass and (up
// to ~1e7 particles) stay resident in the 256 MiB Infinity Cache between passes.
__device__ __forceinline__ void ldpair(const double* col, long long i0, double& a, double& b) {
    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(col + i0));
    a = t.x; b = t.y;
}
__device__ __forceinline__ double ldone(const double* col, long long i) { return __builtin_nontemporal_load(col + i); }

__device__ __forceinline__ void part_defaults(Part& p) {
    const double nan = unset_value();
    p.x = p.y = p.z = p.vx = p.vy = p.vz = p.m = 0.;
    p.beta = p.Rt = p.k1 = p.ta = p.te = p.ti = nan;
}

// One force of the program on one particle behind the head (massless by the caller's
// assertion: it pushes on nothing and feels no tides).  `t` points at the device tables; every
// lane reads the same entries, so these become scalar loads.  `kind` is op.kind, passed
// separately so that instances built for a fixed stack see it as a constant.
template <bool EXACT>
__device__ __forceinline__ void tail_op(const int kind, const int o, const DevOp& op, const double G, const HeadConsts* __restrict__ t,
                                        const Part& p, V3& acc, Geo& g, unsigned* status) {
    if (kind == REBXAMD_OP_GR) {
        double nx = 0., ny = 0., nz = 0.;
        Jac j;
        V3 corr;
        int q;
        const int nheadm = t->nheadm;
        if (!EXACT && nheadm == 1) {
            // one massive body: its separation serves the Newtonian pull and (to an ulp of the
            // body's own coordinates) the Jacobi position
            const int h0 = t->head0;
            if (g.src != h0) geo_compute(p, t->src[o], h0, g);
            const double w = -(G * t->head[0][3]) * (g.ri * g.ri * g.ri);
            j.x = g.dx; j.y = g.dy; j.z = g.dz;
            j.vx = p.vx - t->gr_c[3]; j.vy = p.vy - t->gr_c[4]; j.vz = p.vz - t->gr_c[5];
            j.ax = w * g.dx - t->gr_c[6]; j.ay = w * g.dy - t->gr_c[7]; j.az = w * g.dz - t->gr_c[8];
            q = gr_term_fast(j, g.ri, t->gr_mu, op.d, op.iarg, corr);
        } else {
            // Newtonian pull of the massive head bodies (all of them precede this body)
            for (int u = 0; u < nheadm; u++) {
                if (EXACT) newton_pair_exact(p.x, p.y, p.z, t->head[u][0], t->head[u][1], t->head[u][2], t->head[u][3], true, G, nx, ny, nz);
                else newton_pair_fast(p.x, p.y, p.z, t->head[u][0], t->head[u][1], t->head[u][2], t->head[u][3], G, nx, ny, nz);
            }
            {
#pragma clang fp contract(off)
                j.x = p.x - t->gr_c[0]; j.y = p.y - t->gr_c[1]; j.z = p.z - t->gr_c[2];
                j.vx = p.vx - t->gr_c[3]; j.vy = p.vy - t->gr_c[4]; j.vz = p.vz - t->gr_c[5];
                j.ax = nx - t->gr_c[6]; j.ay = ny - t->gr_c[7]; j.az = nz - t->gr_c[8];
            }
            q = EXACT ? gr_term_exact(j, t->gr_mu, op.c, op.iarg, corr)
                      : gr_term_fast(j, fast_rsqrt(j.x * j.x + j.y * j.y + j.z * j.z), t->gr_mu, op.d, op.iarg, corr);
        }
        if (q == 10) atomicOr(status, WARN_GR_ITER);
        // a massless body's inertial correction equals its Jacobi correction
        acc.x += corr.x; acc.y += corr.y; acc.z += corr.z;
    } else if (kind == REBXAMD_OP_MOF) {
        Src ref;
        if (op.iarg == REBX_COORDINATES_PARTICLE) {
            ref = t->src[o];
        } else {
            ref.x = t->com[0]; ref.y = t->com[1]; ref.z = t->com[2];
            ref.vx = t->com[3]; ref.vy = t->com[4]; ref.vz = t->com[5]; ref.m = t->com[6];
        }
        V3 a;
        if (EXACT) mof_exact(p, ref, a); else mof_fast(p, ref, a);
        acc.x += a.x; acc.y += a.y; acc.z += a.z;
        if (op.iarg == REBX_COORDINATES_BARYCENTRIC) {
            const int nb = t->nbary;
            for (int i = 0; i < nb; i++) {
                acc.x -= t->bary[i][0]; acc.y -= t->bary[i][1]; acc.z -= t->bary[i][2];
            }
        }
    } else if (kind != REBXAMD_OP_TIDES) {
        V3 d, br;
        if (sourced_effect<EXACT>(kind, op, p, t->src[o], G, g, d, br)) {
#pragma clang fp contract(off)
            acc.x += d.x; acc.y += d.y; acc.z += d.z;
        }
    }
}

// A fixed stack of forces as a compile-time signature: the op kinds in execution order, four
// bits each, first op in the lowest bits (kinds are 1..15; 0 ends the list).  Kernel instances
// built for a signature run the ops as straight-line code; signature 0 walks the op list at run time.
typedef unsigned long long Sig;
constexpr Sig sig_of(int k0, int k1 = 0, int k2 = 0, int k3 = 0, int k4 = 0, int k5 = 0, int k6 = 0, int k7 = 0) {
    return (Sig)k0 | (Sig)k1 << 4 | (Sig)k2 << 8 | (Sig)k3 << 12 | (Sig)k4 << 16 | (Sig)k5 << 20 | (Sig)k6 << 24 | (Sig)k7 << 28;
}
constexpr int sig_len(Sig s) { int n = 0; for (; s; s >>= 4) n++; return n; }

// Optional columns a fixed stack reads.  Behind the head every body is massless, and massless
// bodies neither raise nor feel tides (reference src/tides_precession.c:84-87): the tail pass of
// a stack with tides_precession leaves R_tides and k1 unread.
constexpr int sig_cols(Sig s, bool tail) {
    int c = 0;
    for (; s; s >>= 4) {
        const int k = (int)(s & 15ull);
        if (k == REBXAMD_OP_RADIATION) c |= COL_VEL | COL_BETA;
        else if (k == REBXAMD_OP_GR) c |= COL_VEL;
        else if (k == REBXAMD_OP_TIDES) c |= tail ? 0 : COL_TIDES;
        else if (k == REBXAMD_OP_MOF) c |= COL_VEL | COL_TAU;
    }
    return c;
}

template <bool EXACT, Sig SIG, int O>
__device__ __forceinline__ void tail_fixed(const Program& prog, const HeadConsts* __restrict__ t, const Part& p, V3& acc, Geo& g, unsigned* status) {
    constexpr int kind = (int)((SIG >> (4 * O)) & 15ull);
    if constexpr (kind != 0) {
        tail_op<EXACT>(kind, O, prog.ops[O], prog.G, t, p, acc, g, status);
        tail_fixed<EXACT, SIG, O + 1>(prog, t, p, acc, g, status);
    }
}

// All forces of the program on one particle behind the head.
template <bool EXACT, Sig SIG>
__device__ __forceinline__ void tail_particle(const Program& prog, const HeadConsts* __restrict__ t, const Part& p, V3& acc, unsigned* status) {
    Geo g; g.src = -1; g.dx = g.dy = g.dz = g.ri = 0.;
    if constexpr (SIG != 0ull) {
        tail_fixed<EXACT, SIG, 0>(prog, t, p, acc, g, status);
    } else {
        for (int o = 0; o < prog.nops; o++) {
            const DevOp op = prog.ops[o];
            tail_op<EXACT>(op.kind, o, op, prog.G, t, p, acc, g, status);
        }
    }
}

// One op on one particle where every body may be massive: the push on the source is returned.
template <bool EXACT>
__device__ __forceinline__ void general_op(const int kind, const DevOp& op, const Src& S, double G, const Part& p, Geo& g, V3& acc, double& bx, double& by, double& bz) {
    V3 d, br;
    if (kind == REBXAMD_OP_MOF) {
        mof_particle<EXACT>(p, S, acc, br);
        bx += br.x; by += br.y; bz += br.z;
    } else if (sourced_effect<EXACT>(kind, op, p, S, G, g, d, br)) {
#pragma clang fp contract(off)
        acc.x += d.x; acc.y += d.y; acc.z += d.z;
        bx += br.x; by += br.y; bz += br.z;
    }
}

struct StreamShared { double part[WAVES][REBXAMD_MAX_OPS][3]; };
struct GeneralLane { long long i0; bool act0, act1, any_mass; int wave, lane; };

// One op on this thread's two particles in the any-mass-layout pass; the wavefront's summed push on
// the op's source goes to LDS.
template <bool EXACT>
__device__ __forceinline__ void general_step(const int kind, const int o, const DevOp& op, const double G, const HeadConsts* __restrict__ tab,
                                             const Part (&p)[2], V3 (&acc)[2], Geo (&g)[2], const GeneralLane& L, StreamShared& sh) {
    const Src S = tab->src[o];
    double bx = 0., by = 0., bz = 0.;
    if (L.act0 && L.i0 != op.src) general_op<EXACT>(kind, op, S, G, p[0], g[0], acc[0], bx, by, bz);
    if (L.act1 && L.i0 + 1 != op.src) general_op<EXACT>(kind, op, S, G, p[1], g[1], acc[1], bx, by, bz);
    if (op.slot >= 0) {
        // a wavefront whose bodies are all massless pushes on nothing
        if (L.any_mass) { bx = wave_sum(bx); by = wave_sum(by); bz = wave_sum(bz); }
        else { bx = by = bz = 0.; }
        if (L.lane == 0) { sh.part[L.wave][op.slot][0] = bx; sh.part[L.wave][op.slot][1] = by; sh.part[L.wave][op.slot][2] = bz; }
    }
}
template <bool EXACT, Sig SIG, int O>
__device__ __forceinline__ void general_fixed(const Program& prog, const HeadConsts* __restrict__ tab, const Part (&p)[2], V3 (&acc)[2], Geo (&g)[2],
                                              const GeneralLane& L, StreamShared& sh) {
    constexpr int kind = (int)((SIG >> (4 * O)) & 15ull);
    if constexpr (kind != 0) {
        general_step<EXACT>(kind, O, prog.ops[O], prog.G, tab, p, acc, g, L, sh);
        general_fixed<EXACT, SIG, O + 1>(prog, tab, p, acc, g, L, sh);
    }
}

// The remaining stages of a Runge-Kutta update of one body behind the head, given its first-stage
// accelerations k1: each stage evaluates the stack at the stage velocity against that stage's head
// tables.  Arithmetic of the separate passes (lincomb / rk2_final / rk4_mid / rk4_final kernels).
template <bool EXACT, Sig SIG, int EPI>
__device__ __forceinline__ void integrate_tail(const Program& prog, const HeadConsts* __restrict__ tab, Part& p, const V3& k1, const StepArgs& sa, unsigned* status) {
    const double vx = p.vx, vy = p.vy, vz = p.vz;
    if (EPI == EPI_EULER) {
        p.vx = axpy1<EXACT>(vx, sa.c0, k1.x); p.vy = axpy1<EXACT>(vy, sa.c0, k1.y); p.vz = axpy1<EXACT>(vz, sa.c0, k1.z);
    } else if (EPI == EPI_RK2) {
        V3 k2 = {0., 0., 0.};
        p.vx = axpy1<EXACT>(vx, sa.c0, k1.x); p.vy = axpy1<EXACT>(vy, sa.c0, k1.y); p.vz = axpy1<EXACT>(vz, sa.c0, k1.z);
        tail_particle<EXACT, SIG>(prog, tab + 1, p, k2, status);
        p.vx = rk2_combine<EXACT>(vx, sa.c1, k1.x, sa.c2, k2.x);
        p.vy = rk2_combine<EXACT>(vy, sa.c1, k1.y, sa.c2, k2.y);
        p.vz = rk2_combine<EXACT>(vz, sa.c1, k1.z, sa.c2, k2.z);
    } else {
        V3 k2 = {0., 0., 0.}, k3 = {0., 0., 0.}, k4 = {0., 0., 0.};
        p.vx = axpy1<EXACT>(vx, sa.c0, k1.x); p.vy = axpy1<EXACT>(vy, sa.c0, k1.y); p.vz = axpy1<EXACT>(vz, sa.c0, k1.z);
        tail_particle<EXACT, SIG>(prog, tab + 1, p, k2, status);
        p.vx = axpy1<EXACT>(vx, sa.c0, k2.x); p.vy = axpy1<EXACT>(vy, sa.c0, k2.y); p.vz = axpy1<EXACT>(vz, sa.c0, k2.z);
        tail_particle<EXACT, SIG>(prog, tab + 2, p, k3, status);
        p.vx = axpy1<EXACT>(vx, sa.c1, k3.x); p.vy = axpy1<EXACT>(vy, sa.c1, k3.y); p.vz = axpy1<EXACT>(vz, sa.c1, k3.z);
        const double sx = k3.x + k2.x, sy = k3.y + k2.y, sz = k3.z + k2.z;
        tail_particle<EXACT, SIG>(prog, tab + 3, p, k4, status);
        p.vx = rk4_combine<EXACT>(vx, sa.c2, k1.x, k4.x, sx);
        p.vy = rk4_combine<EXACT>(vy, sa.c2, k1.y, k4.y, sy);
        p.vz = rk4_combine<EXACT>(vz, sa.c2, k1.z, k4.z, sz);
    }
}

// Wisdom-Holman kick of a massless body behind a head of several massive bodies: the pull of every
// head body (direct gravity; with one head body it cancels against the Jacobi term and neither is
// formed) and the Jacobi term + G M q / |q|^3, q = position from the head's centre of mass (`frame`).
template <bool EXACT>
__device__ __forceinline__ void wh_head_pull(const HeadConsts* __restrict__ tab, double G, const double* __restrict__ frame, const Part& p, V3& acc) {
    const int nh = tab->nheadm;
    if (EXACT) {
#pragma clang fp contract(off)
        for (int c = 0; c < nh; c++) {
            const double dx = p.x - tab->head[c][0], dy = p.y - tab->head[c][1], dz = p.z - tab->head[c][2];
            const double r = sqrt(dx * dx + dy * dy + dz * dz);
            const double pre = -G * tab->head[c][3] / (r * r * r);
            acc.x += pre * dx; acc.y += pre * dy; acc.z += pre * dz;
        }
        const double qx = p.x - frame[0], qy = p.y - frame[1], qz = p.z - frame[2];
        const double r2i = 1. / (qx * qx + qy * qy + qz * qz);
        const double pre3 = sqrt(r2i) * r2i * frame[12];
        acc.x += pre3 * qx; acc.y += pre3 * qy; acc.z += pre3 * qz;
    } else {
        for (int c = 0; c < nh; c++) {
            const double dx = p.x - tab->head[c][0], dy = p.y - tab->head[c][1], dz = p.z - tab->head[c][2];
            const double ri = fast_rsqrt(dx * dx + dy * dy + dz * dz);
            const double pre = -(G * tab->head[c][3]) * (ri * ri * ri);
            acc.x = fma(pre, dx, acc.x); acc.y = fma(pre, dy, acc.y); acc.z = fma(pre, dz, acc.z);
        }
        const double qx = p.x - frame[0], qy = p.y - frame[1], qz = p.z - frame[2];
        const double ri = fast_rsqrt(qx * qx + qy * qy + qz * qz);
        const double pre3 = frame[12] * (ri * ri * ri);
        acc.x = fma(pre3, qx, acc.x); acc.y = fma(pre3, qy, acc.y); acc.z = fma(pre3, qz, acc.z);
    }
}

// EPI (tail pass only; see StepArgs).  EPI_WH: the pass carries a whole Wisdom-Holman step of the
// massless bodies -- the forces start from zero instead of the acceleration columns, the body is
// kicked with them and drifted on its Kepler orbit about body 0, and positions and velocities are
// written back; the acceleration columns are neither read nor written.  EPI_EULER/RK2/RK4: the
// pass carries a whole velocity update of a force integrator; velocities and the first-stage
// accelerations are written.
// tiles of 2*BLOCK bodies from `base` to the end of the view (what stream_block_count returns on the host)
__host__ __device__ __forceinline__ unsigned stream_tiles(long long base, long long n) {
    const long long pairs = (n - base + 1) / 2;
    return (unsigned)((pairs + BLOCK - 1) / BLOCK);
}
static constexpr unsigned XCD_RUN = 128;                    // consecutive tiles per XCD
static constexpr unsigned XCD_REMAP_MIN_TILES = 4096;       // grids below this keep the linear order (no padding of small launches)

// Workgroups per CU the register allocation of an instance is held to (= wavefronts per SIMD).  Measured choices:
//   * the force-only instances of the common stacks fit 5 with a handful of spilled registers and gain 6 % from the fifth
//     wavefront (C2: 160 -> 150 us at 1e7 grains); at 6 they spill ten times as much and lose a third;
//   * so does the pass that carries an Euler update (C5 stack: 238 -> 225 us); those of RK2 and of the implicit midpoint
//     rule would spill 30-90 registers there and take 1.6-2.4 times as long: they stay at 4, like RK4;
//   * the passes that carry a Wisdom-Holman step or a batch of them stay at 4, spills included (the batch pass, which is
//     bound by vector issue and spills 78 registers there: 14.7 us per step at 2 without spills, 12.5 at 3, 12.1 at 4, 14.1 at 5);
//   * the op-list walk over massive bodies needs ~180 registers (2).
__host__ __device__ constexpr int stream_min_blocks(bool exact, bool backreact, int cols, Sig sig, int epi) {
    if (backreact) return sig == 0ull ? 2 : ((!exact && epi == EPI_NONE) ? 5 : 3);
    if (!exact && cols != COL_ALL) return ((epi == EPI_NONE || epi == EPI_EULER) && sig != 0ull) ? 5 : 4;
    return 3;
}

template <bool EXACT, bool BACKREACT, bool VEC2, int COLS, Sig SIG = 0ull, int EPI = EPI_NONE>
__global__ __launch_bounds__(BLOCK, stream_min_blocks(EXACT, BACKREACT, COLS, SIG, EPI)) void stream_kernel(View v, Program prog, long long lo, long long base,
                                                       const HeadConsts* __restrict__ tab, double* partials, unsigned* status, StepArgs step) {
    static_assert(EPI == EPI_NONE || (EPI == EPI_APPLY ? BACKREACT && VEC2 && !EXACT : !BACKREACT && (COLS & COL_VEL)),
                  "a stepping pass streams massless bodies and needs their velocities; a pass that applies a prefix-sum force runs on any mass layout");
    constexpr bool STEP = EPI != EPI_NONE && EPI != EPI_APPLY;        // forces start from zero
    const int tid = threadIdx.x;
    if (EPI == EPI_IM && step.stop != nullptr && *step.stop != 0u) return;      // a round queued past convergence
    const bool chained = (EPI == EPI_WH || EPI == EPI_WHB) && step.next_hc != nullptr;
    if (EPI == EPI_WHB && chained && blockIdx.x == 0) {
        // the extra workgroup of a pass that carries a batch of steps: the head goes through the steps of the NEXT batch (it
        // reads and writes the head bodies, the head's Jacobi state and the other copy of the batch tables, none of which a
        // workgroup of this pass touches), one wavefront, one step after the other
        if constexpr (EPI == EPI_WHB) {
            if (tid >= 64) return;
            if (step.wh_head > 1) {
                __shared__ HeadShared hsh;
                __shared__ WhHeadShared wsh;
                for (int j = 0; j < step.next_nsub; j++) {
                    head_body<EXACT>(v, prog, step.next_hc + j, status, nullptr, 0., hsh, tid);
                    wh_head_step_body<EXACT>(v, step.wh_head, prog.G, step.dtk, j + 1 < step.next_nsub ? step.h_sub : step.next_h, step.jac,
                                             step.next_star + 16 * j, wsh, tid);
                    wave_sync();
                }
            } else {
                Src S;
                load_src(v, 0, S);
                for (int j = 0; j < step.next_nsub; j++)
                    head_one_core(v, prog, step.next_hc + j, step.next_star + 16 * j, j + 1 < step.next_nsub ? step.h_sub : step.next_h, tid, nullptr, 0., S,
                                  j + 1 == step.next_nsub);
            }
        }
        return;
    }
    if (chained && blockIdx.x == 0) {
        // the extra workgroup of a Wisdom-Holman pass: the head pass of the next step (it reads and
        // writes the head bodies and the other copy of the tables, which no workgroup of this pass touches).
        // It is the first workgroup, so that its latency hides behind the others.
        if constexpr (EPI == EPI_WH) {
            if (step.wh_head > 1) {
                // star + planets: the general head pass (additional forces on the head bodies, tables), then the head's own
                // step -- gravity among its bodies, Jacobi kick, drift to the force evaluation after the next -- by the same
                // wavefront; the other wavefronts of the workgroup leave first (the head code synchronises with wave_sync(), which
                // holds no workgroup barrier)
                __shared__ HeadShared hsh;
                __shared__ WhHeadShared wsh;
                if (tid >= 64) return;
                head_body<EXACT>(v, prog, step.next_hc, status, nullptr, 0., hsh, tid);
                wh_head_step_body<EXACT>(v, step.wh_head, prog.G, step.dtk, step.next_h, step.jac, step.next_star, wsh, tid, step.next_pre);
                return;
            }
        }
        if (tid < 64) head_one_body(v, prog, step.next_hc, step.next_star, step.next_h, tid, step.next_lead, step.next_h_lead);
        return;
    }
    // Which tile of 2*BLOCK bodies this workgroup takes.  Workgroups are dealt to the 8 XCDs round-robin; on a large grid each
    // XCD takes XCD_RUN consecutive tiles of every group of 8*XCD_RUN, so that it streams contiguous runs of every column
    // (3-4 % more bandwidth than neighbouring tiles on different XCDs, at every size).  The grid is padded to whole groups:
    // workgroups whose tile lies behind the end leave at once.  Everything below is indexed by the tile, not by blockIdx.
    const unsigned nb = stream_tiles(base, v.n);
    unsigned blk = blockIdx.x - (chained ? 1u : 0u);
    if (nb >= XCD_REMAP_MIN_TILES) {
        const unsigned xcd = blk & 7u, seq = blk >> 3;
        blk = ((seq / XCD_RUN) * 8u + xcd) * XCD_RUN + seq % XCD_RUN;
        if (blk >= nb) return;
    }
    // A pass that applies a prefix-sum force, one wavefront per part of a segment of the per-body pass: the reactions of the
    // bodies behind the part are asked for first, so that they arrive together with the columns
    double tab_behind[3] = {0., 0., 0.}, tab_mass = 0.;
    if constexpr (EPI == EPI_APPLY) {
        static_assert(WAVES == SSUB && 2 * BLOCK == 64 * ST, "one wavefront per part of a segment of the weighted-sum passes");
        const ScanTail& T = step.tail;
        if (T.all == nullptr) {
            // (blk = this workgroup's segment of the per-body pass; every address is known here, so the requests leave together)
            const int part = tid >> 6, w = (int)(blk % SW);
            const long long seg = blk, first = seg - w;          // first: segment 0 of the same workgroup of that pass
            const long long running = (seg * (SSUB - 1) + (part + 1 < SSUB ? part : SSUB - 2)) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double later[SW];
#pragma unroll
                for (int q = 1; q < SW; q++) later[q] = T.seg_totals[(first + q) * 3 + c];
                const double own = T.seg_totals[seg * 3 + c], done = T.seg_running[running + c];
                double behind = T.behind_blocks[(first / SW) * 3 + c];
#pragma unroll
                for (int q = SW - 1; q >= 1; q--) if (q > w) behind += later[q];
                // (the rest of this segment: its total less the running total at the end of this wavefront's part)
                if (part + 1 < SSUB) behind += own - done;
                tab_behind[c] = behind;
            }
            tab_mass = T.part_mass[seg * SSUB + part];
        }
    }
    const long long i0 = base + 2 * ((long long)blk * BLOCK + tid);
    const bool in0 = i0 < v.n, in1 = i0 + 1 < v.n;
    const bool act0 = in0 && i0 >= lo, act1 = in1 && i0 + 1 >= lo;
    const bool full = VEC2 && act0 && act1;      // the common case: both particles, 16-byte accesses

    Part p[2];
    V3 acc[2];
    part_defaults(p[0]); part_defaults(p[1]);
    acc[0].x = acc[0].y = acc[0].z = acc[1].x = acc[1].y = acc[1].z = 0.;
    if (full) {
        ldpair(v.x, i0, p[0].x, p[1].x); ldpair(v.y, i0, p[0].y, p[1].y); ldpair(v.z, i0, p[0].z, p[1].z);
        if (COLS & COL_VEL) { ldpair(v.vx, i0, p[0].vx, p[1].vx); ldpair(v.vy, i0, p[0].vy, p[1].vy); ldpair(v.vz, i0, p[0].vz, p[1].vz); }
        // behind the head every body is massless by the caller's assertion: no need to read m
        if (BACKREACT) ldpair(v.m, i0, p[0].m, p[1].m);
        if ((COLS & COL_BETA) && v.beta) ldpair(v.beta, i0, p[0].beta, p[1].beta);
        if ((COLS & COL_TIDES) && v.Rt) ldpair(v.Rt, i0, p[0].Rt, p[1].Rt);
        if ((COLS & COL_TIDES) && v.k1) ldpair(v.k1, i0, p[0].k1, p[1].k1);
        if ((COLS & COL_TAU) && v.ta) ldpair(v.ta, i0, p[0].ta, p[1].ta);
        if ((COLS & COL_TAU) && v.te) ldpair(v.te, i0, p[0].te, p[1].te);
        if ((COLS & COL_TAU) && v.ti) ldpair(v.ti, i0, p[0].ti, p[1].ti);
        if (!STEP) {
            const d2v ax = *reinterpret_cast<const d2v*>(v.ax + i0), ay = *reinterpret_cast<const d2v*>(v.ay + i0), az = *reinterpret_cast<const d2v*>(v.az + i0);
            acc[0].x = ax.x; acc[1].x = ax.y; acc[0].y = ay.x; acc[1].y = ay.y; acc[0].z = az.x; acc[1].z = az.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!(k ? act1 : act0)) continue;
            const long long i = i0 + k;
            p[k].x = ldone(v.x, i); p[k].y = ldone(v.y, i); p[k].z = ldone(v.z, i);
            if (COLS & COL_VEL) { p[k].vx = ldone(v.vx, i); p[k].vy = ldone(v.vy, i); p[k].vz = ldone(v.vz, i); }
            if (BACKREACT) p[k].m = ldone(v.m, i);
            if ((COLS & COL_BETA) && v.beta) p[k].beta = ldone(v.beta, i);
            if ((COLS & COL_TIDES) && v.Rt) p[k].Rt = ldone(v.Rt, i);
            if ((COLS & COL_TIDES) && v.k1) p[k].k1 = ldone(v.k1, i);
            if ((COLS & COL_TAU) && v.ta) p[k].ta = ldone(v.ta, i);
            if ((COLS & COL_TAU) && v.te) p[k].te = ldone(v.te, i);
            if ((COLS & COL_TAU) && v.ti) p[k].ti = ldone(v.ti, i);
            if (!STEP) { acc[k].x = v.ax[i]; acc[k].y = v.ay[i]; acc[k].z = v.az[i]; }
        }
    }

    if ((EPI == EPI_WH || EPI == EPI_WHB) && step.star0 != nullptr) {
        // first step of a run: the leading half drift, ahead of the forces
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!(k ? act1 : act0)) continue;
            double s[6] = {p[k].x, p[k].y, p[k].z, p[k].vx, p[k].vy, p[k].vz};
            wh_kick_drift<EXACT, false>(s, 0., 0., 0., step.h0, 0., step.star0);
            p[k].x = s[0]; p[k].y = s[1]; p[k].z = s[2]; p[k].vx = s[3]; p[k].vy = s[4]; p[k].vz = s[5];
        }
    }
    if constexpr (EPI == EPI_WHB && !BACKREACT) {
        {
            // A batch of steps: the bodies of this thread go through all of them in registers.
            for (int j = 0; j < step.nsub; j++) {
                const HeadConsts* __restrict__ tj = step.tabs + j;
                const double* __restrict__ sj = step.star + 16 * j;
                const double hj = j + 1 < step.nsub ? step.h_sub : step.h;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (!(k ? act1 : act0)) continue;
                    V3 a = {0., 0., 0.};
                    tail_particle<EXACT, SIG>(prog, tj, p[k], a, status);
                    if (step.wh_head > 1) wh_head_pull<EXACT>(tj, prog.G, sj, p[k], a);
                    double s[6] = {p[k].x, p[k].y, p[k].z, p[k].vx, p[k].vy, p[k].vz};
                    wh_kick_drift<EXACT, true>(s, a.x, a.y, a.z, hj, step.dtk, sj);
                    p[k].x = s[0]; p[k].y = s[1]; p[k].z = s[2]; p[k].vx = s[3]; p[k].vy = s[4]; p[k].vz = s[5];
                }
            }
            double* const px = const_cast<double*>(v.x); double* const py = const_cast<double*>(v.y); double* const pz = const_cast<double*>(v.z);
            double* const pvx = const_cast<double*>(v.vx); double* const pvy = const_cast<double*>(v.vy); double* const pvz = const_cast<double*>(v.vz);
            if (full) {
                d2v t;
                t.x = p[0].x; t.y = p[1].x; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(px + i0));
                t.x = p[0].y; t.y = p[1].y; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(py + i0));
                t.x = p[0].z; t.y = p[1].z; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pz + i0));
                t.x = p[0].vx; t.y = p[1].vx; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvx + i0));
                t.x = p[0].vy; t.y = p[1].vy; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvy + i0));
                t.x = p[0].vz; t.y = p[1].vz; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvz + i0));
            } else {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (!(k ? act1 : act0)) continue;
                    const long long i = i0 + k;
                    px[i] = p[k].x; py[i] = p[k].y; pz[i] = p[k].z; pvx[i] = p[k].vx; pvy[i] = p[k].vy; pvz[i] = p[k].vz;
                }
            }
            return;
        }
    }
    double orig[2][3], fin[2][3];           // implicit midpoint: velocities at the start of the step, current estimate of the final ones
    if (EPI == EPI_IM) {
        if (full) {
            ldpair(step.aux[0], i0, fin[0][0], fin[1][0]); ldpair(step.aux[1], i0, fin[0][1], fin[1][1]); ldpair(step.aux[2], i0, fin[0][2], fin[1][2]);
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                fin[k][0] = fin[k][1] = fin[k][2] = 0.;
                if (k ? act1 : act0) { fin[k][0] = step.aux[0][i0 + k]; fin[k][1] = step.aux[1][i0 + k]; fin[k][2] = step.aux[2][i0 + k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            orig[k][0] = p[k].vx; orig[k][1] = p[k].vy; orig[k][2] = p[k].vz;
            p[k].vx = 0.5 * (orig[k][0] + fin[k][0]); p[k].vy = 0.5 * (orig[k][1] + fin[k][1]); p[k].vz = 0.5 * (orig[k][2] + fin[k][2]);
        }
    }

    if (!BACKREACT) {
        // one particle after the other keeps the live register set small
        if (act0) tail_particle<EXACT, SIG>(prog, tab, p[0], acc[0], status);
        if (act1) tail_particle<EXACT, SIG>(prog, tab, p[1], acc[1], status);
    } else {
        __shared__ StreamShared sh;
        const int wave = tid >> 6, lane = tid & 63;
        Geo g[2];
        g[0].src = -1; g[0].dx = g[0].dy = g[0].dz = g[0].ri = 0.; g[1] = g[0];
        const bool any_mass = __ballot((act0 && p[0].m != 0.) || (act1 && p[1].m != 0.)) != 0ull;
        GeneralLane L = {i0, act0, act1, any_mass, wave, lane};
        if constexpr (SIG != 0ull) {
            general_fixed<EXACT, SIG, 0>(prog, tab, p, acc, g, L, sh);
        } else {
            for (int o = 0; o < prog.nops; o++) general_step<EXACT>(prog.ops[o].kind, o, prog.ops[o], prog.G, tab, p, acc, g, L, sh);
        }
        __syncthreads();
        if (tid < prog.nslots * 3) {
            const int s = tid / 3, c = tid % 3;
            double t = sh.part[0][s][c];
#pragma unroll
            for (int w = 1; w < WAVES; w++) t += sh.part[w][s][c];
            partials[((long long)s * nb + blk) * 3 + c] = t;
        }
    }

    if (EPI == EPI_IM) {
        __shared__ double red[2][WAVES];
        double d2 = 0., t2 = 0.;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!(k ? act1 : act0)) continue;
            const double fx = axpy1<EXACT>(orig[k][0], step.c0, acc[k].x), fy = axpy1<EXACT>(orig[k][1], step.c0, acc[k].y), fz = axpy1<EXACT>(orig[k][2], step.c0, acc[k].z);
            const double dx = fx - fin[k][0], dy = fy - fin[k][1], dz = fz - fin[k][2];
            d2 += dx * dx + dy * dy + dz * dz;
            t2 += fx * fx + fy * fy + fz * fz;
            fin[k][0] = fx; fin[k][1] = fy; fin[k][2] = fz;
        }
        if (full) {
            d2v t;
            t.x = fin[0][0]; t.y = fin[1][0]; *reinterpret_cast<d2v*>(step.aux[0] + i0) = t;
            t.x = fin[0][1]; t.y = fin[1][1]; *reinterpret_cast<d2v*>(step.aux[1] + i0) = t;
            t.x = fin[0][2]; t.y = fin[1][2]; *reinterpret_cast<d2v*>(step.aux[2] + i0) = t;
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (!(k ? act1 : act0)) continue;
                step.aux[0][i0 + k] = fin[k][0]; step.aux[1][i0 + k] = fin[k][1]; step.aux[2][i0 + k] = fin[k][2];
            }
        }
        d2 = wave_sum(d2); t2 = wave_sum(t2);
        if ((tid & 63) == 0) { red[0][tid >> 6] = d2; red[1][tid >> 6] = t2; }
        __syncthreads();
        if (tid == 0) {
            double a = 0., b = 0.;
#pragma unroll
            for (int w = 0; w < WAVES; w++) { a += red[0][w]; b += red[1][w]; }
            step.partials[2 * (long long)blk] = a;
            step.partials[2 * (long long)blk + 1] = b;
        }
        return;
    }
    if (EPI >= EPI_EULER && EPI <= EPI_RK4) {
        double* const pvx = const_cast<double*>(v.vx); double* const pvy = const_cast<double*>(v.vy); double* const pvz = const_cast<double*>(v.vz);
        if (act0) integrate_tail<EXACT, SIG, EPI>(prog, tab, p[0], acc[0], step, status);
        if (act1) integrate_tail<EXACT, SIG, EPI>(prog, tab, p[1], acc[1], step, status);
        if (full) {
            d2v t;
            t.x = p[0].vx; t.y = p[1].vx; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvx + i0));
            t.x = p[0].vy; t.y = p[1].vy; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvy + i0));
            t.x = p[0].vz; t.y = p[1].vz; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvz + i0));
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (!(k ? act1 : act0)) continue;
                pvx[i0 + k] = p[k].vx; pvy[i0 + k] = p[k].vy; pvz[i0 + k] = p[k].vz;
            }
        }
        // the first-stage accelerations go to the acceleration columns below
    }
    if (EPI == EPI_WH) {
        double* const px = const_cast<double*>(v.x); double* const py = const_cast<double*>(v.y); double* const pz = const_cast<double*>(v.z);
        double* const pvx = const_cast<double*>(v.vx); double* const pvy = const_cast<double*>(v.vy); double* const pvz = const_cast<double*>(v.vz);
        double s[2][6];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!(k ? act1 : act0)) continue;
            s[k][0] = p[k].x; s[k][1] = p[k].y; s[k][2] = p[k].z; s[k][3] = p[k].vx; s[k][4] = p[k].vy; s[k][5] = p[k].vz;
            if (step.wh_head > 1) wh_head_pull<EXACT>(tab, prog.G, step.star, p[k], acc[k]);
            // (h == 0: a run left unsynchronised between steps -- the drift waits for the next pass, or for the end of the run)
            if (step.h == 0.) wh_kick<EXACT>(s[k][3], s[k][4], s[k][5], acc[k].x, acc[k].y, acc[k].z, step.dtk, step.star);
            else wh_kick_drift<EXACT, true>(s[k], acc[k].x, acc[k].y, acc[k].z, step.h, step.dtk, step.star);
        }
        if (full) {
            d2v t;
            t.x = s[0][0]; t.y = s[1][0]; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(px + i0));
            t.x = s[0][1]; t.y = s[1][1]; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(py + i0));
            t.x = s[0][2]; t.y = s[1][2]; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pz + i0));
            t.x = s[0][3]; t.y = s[1][3]; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvx + i0));
            t.x = s[0][4]; t.y = s[1][4]; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvy + i0));
            t.x = s[0][5]; t.y = s[1][5]; __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(pvz + i0));
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (!(k ? act1 : act0)) continue;
                const long long i = i0 + k;
                px[i] = s[k][0]; py[i] = s[k][1]; pz[i] = s[k][2]; pvx[i] = s[k][3]; pvy[i] = s[k][4]; pvz[i] = s[k][5];
            }
        }
        return;
    }
    if (EPI == EPI_APPLY) {
        // The workgroup's 2*BLOCK bodies are one wavefront segment of the per-body pass (the launcher checks the sizes), each
        // of its wavefronts one part of it: the reactions behind the part come from that pass's table, those inside it from a
        // suffix sum over the wavefront.
        const ScanTail& T = step.tail;
        const int lane = tid & 63;
        double aj[2][3], f[2] = {0., 0.};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            aj[0][c] = aj[1][c] = 0.;
            if (full) ldpair(T.aj[c], i0, aj[0][c], aj[1][c]);
            else { if (act0) aj[0][c] = ldone(T.aj[c], i0); if (act1) aj[1][c] = ldone(T.aj[c], i0 + 1); }
        }
        double s[2][3];
        if (T.all != nullptr) {
#pragma unroll
            for (int c = 0; c < 3; c++) s[0][c] = s[1][c] = T.all[c];
        } else {
            {
                // the share of each body's force that falls back on the bodies before it: m / (eta + m), eta = their mass
                const double m0 = act0 ? p[0].m : 0., m1 = act1 ? p[1].m : 0.;
                const double pair = m0 + m1;
                const double eta0 = tab_mass + (wave_incl_scan(pair, lane) - pair), eta1 = eta0 + m0;
                f[0] = eta0 + m0 > 0. ? m0 / (eta0 + m0) : 0.;
                f[1] = eta1 + m1 > 0. ? m1 / (eta1 + m1) : 0.;
            }
            double w[2][3], excl[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                w[0][c] = act0 ? reaction(f[0], aj[0][c]) : 0.;
                w[1][c] = act1 ? reaction(f[1], aj[1][c]) : 0.;
                const double pair = w[0][c] + w[1][c];
                const double incl = wave_incl_scan_rev(pair, lane);
                excl[c] = incl - pair;                       // the lanes behind this one
                s[1][c] = w[1][c] + (excl[c] + tab_behind[c]);
                s[0][c] = w[0][c] + s[1][c];
            }
        }
        acc[0].x += aj[0][0] - s[0][0]; acc[0].y += aj[0][1] - s[0][1]; acc[0].z += aj[0][2] - s[0][2];
        acc[1].x += aj[1][0] - s[1][0]; acc[1].y += aj[1][1] - s[1][1]; acc[1].z += aj[1][2] - s[1][2];
    }
    if (full) {
        d2v ax, ay, az;
        ax.x = acc[0].x; ax.y = acc[1].x; ay.x = acc[0].y; ay.y = acc[1].y; az.x = acc[0].z; az.y = acc[1].z;
        if (STEP) {         // first-stage accelerations of a carried update: written once, not read back by this pass
            __builtin_nontemporal_store(ax, reinterpret_cast<d2v*>(v.ax + i0));
            __builtin_nontemporal_store(ay, reinterpret_cast<d2v*>(v.ay + i0));
            __builtin_nontemporal_store(az, reinterpret_cast<d2v*>(v.az + i0));
        } else {
            *reinterpret_cast<d2v*>(v.ax + i0) = ax;
            *reinterpret_cast<d2v*>(v.ay + i0) = ay;
            *reinterpret_cast<d2v*>(v.az + i0) = az;
        }
    } else {
        if (act0) { v.ax[i0] = acc[0].x; v.ay[i0] = acc[0].y; v.az[i0] = acc[0].z; }
        if (act1) { v.ax[i0 + 1] = acc[1].x; v.ay[i0 + 1] = acc[1].y; v.az[i0 + 1] = acc[1].z; }
    }
}

// Fills the source table when no head pass runs (any mass layout).
__global__ __launch_bounds__(64) void stage_kernel(View v, Program prog, HeadConsts* hc) {
    const int t = threadIdx.x;
    if (t < prog.nops) {
        int s = prog.ops[t].src;
        if (s >= 0) {
            Src S; load_src(v, s, S); hc->src[t] = S;
            hc->src_acc[t][0] = v.ax[s]; hc->src_acc[t][1] = v.ay[s]; hc->src_acc[t][2] = v.az[s];
        }
    }
}

// ------------------------------------------------------------------ finalize: partials -> sources
// Two fixed-shape stages so that the result does not depend on timing: FIN_CHUNKS workgroups per
// slot each fold a contiguous range of the per-workgroup partials, then one workgroup folds the
// chunk sums slot by slot (in op order) and adds each total to its source particle.
static constexpr int FIN_CHUNKS = 64;

__device__ __forceinline__ void block_sum3(double& x, double& y, double& z, double (*red)[3]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    x = wave_sum(x); y = wave_sum(y); z = wave_sum(z);
    __syncthreads();
    if (lane == 0) { red[wave][0] = x; red[wave][1] = y; red[wave][2] = z; }
    __syncthreads();
    x = red[0][0]; y = red[0][1]; z = red[0][2];
    for (int w = 1; w < WAVES; w++) { x += red[w][0]; y += red[w][1]; z += red[w][2]; }
}

__global__ __launch_bounds__(BLOCK) void finalize_stage1(const double* partials, int nblocks, double* chunk_sums) {
    __shared__ double red[WAVES][3];
    const int s = blockIdx.y, c = blockIdx.x;
    const int per = (nblocks + FIN_CHUNKS - 1) / FIN_CHUNKS;
    const int b0 = c * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
    double x = 0., y = 0., z = 0.;
    for (int b = b0 + threadIdx.x; b < b1; b += BLOCK) {
        const double* q = partials + ((long long)s * nblocks + b) * 3;
        x += q[0]; y += q[1]; z += q[2];
    }
    block_sum3(x, y, z, red);
    if (threadIdx.x == 0) {
        double* o = chunk_sums + ((long long)s * FIN_CHUNKS + c) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// sums_out == nullptr: add each op's total push to its source particle; otherwise store the
// totals (3 doubles per op, zero for ops that push on nothing) for the caller to combine
// across devices first.
__global__ __launch_bounds__(64) void finalize_stage2(View v, Program prog, const double* chunk_sums, double* sums_out) {
    const int lane = threadIdx.x;
    for (int o = 0; o < prog.nops; o++) {
        const int s = prog.ops[o].slot;
        double x = 0., y = 0., z = 0.;
        if (s >= 0) {
            const double* q = chunk_sums + ((long long)s * FIN_CHUNKS + lane) * 3;
            x = wave_sum(q[0]); y = wave_sum(q[1]); z = wave_sum(q[2]);
        }
        if (lane == 0) {
            if (sums_out) {
                sums_out[o * 3 + 0] = x; sums_out[o * 3 + 1] = y; sums_out[o * 3 + 2] = z;
            } else if (s >= 0) {
                const int src = prog.ops[o].src;
                v.ax[src] += x; v.ay[src] += y; v.az[src] += z;
            }
        }
    }
}

// ------------------------------------------------------------------ sources in the reference's order
// The reference walks a force's particles in index order, and every one of them pushes on the force's source as it is
// visited: the source's acceleration is a chain of additions, interleaved -- force by force -- with what the source gains
// as an ordinary particle of the other forces.  The streaming pass sums the pushes per wavefront and workgroup (another
// order: a few ulp away).  Here ONE workgroup per source rebuilds that chain from the acceleration the run of forces found
// (stage_kernel kept it): force after force, either the source's own update (one thread), or the pushes of all other
// bodies -- evaluated 256 at a time in parallel, then added by one thread in the order the force visits them.  The result replaces what the
// streaming pass left in the source's row.  A chain of n additions per force that pushes: bodies x ~10 clock cycles.
static constexpr int RT = 256;

__device__ __forceinline__ void load_part_all(const View& v, const Program& prog, long long i, Part& p) {
    part_defaults(p);
    p.x = v.x[i]; p.y = v.y[i]; p.z = v.z[i];
    if (prog.need_vel) { p.vx = v.vx[i]; p.vy = v.vy[i]; p.vz = v.vz[i]; }
    if (v.m) p.m = v.m[i];
    if (prog.need_beta && v.beta) p.beta = v.beta[i];
    if (prog.need_tides && v.Rt) p.Rt = v.Rt[i];
    if (prog.need_tides && v.k1) p.k1 = v.k1[i];
    if (prog.need_tau && v.ta) p.ta = v.ta[i];
    if (prog.need_tau && v.te) p.te = v.te[i];
    if (prog.need_tau && v.ti) p.ti = v.ti[i];
}

// nchain: the bodies that can push -- all of them, or those up to the last massive one (a massless body pushes with a zero)
__global__ __launch_bounds__(RT) void source_replay_kernel(View v, Program prog, const HeadConsts* __restrict__ tab, long long nchain) {
    __shared__ double push[3][RT];
    const int first = blockIdx.x, tid = threadIdx.x;
    const int s = prog.ops[first].src;
    if (prog.ops[first].slot < 0 || s < 0) return;
    for (int e = 0; e < first; e++)
        if (prog.ops[e].slot >= 0 && prog.ops[e].src == s) return;      // (the first force that pushes on a body rebuilds that body)
    V3 val = {tab->src_acc[first][0], tab->src_acc[first][1], tab->src_acc[first][2]};
    Part self;
    load_part_all(v, prog, s, self);
    for (int k = 0; k < prog.nops; k++) {
        const DevOp op = prog.ops[k];
        if (op.src < 0) continue;
        const Src S = tab->src[k];
        Geo g;
        g.src = -1; g.dx = g.dy = g.dz = g.ri = 0.;
        if (op.src != s) {
            // an ordinary particle of this force
            double bx = 0., by = 0., bz = 0.;
            if (tid == 0) general_op<true>(op.kind, op, S, prog.G, self, g, val, bx, by, bz);
            continue;
        }
        if (op.slot < 0) continue;              // a source that nothing pushes on (radiation)
        // (modify_orbits_forces visits its particles from the last to the first, src/rebxtools.c:71; the other forces ascend)
        const bool descending = op.kind == REBXAMD_OP_MOF;
        const long long ntiles = (nchain + RT - 1) / RT;
        for (long long t = 0; t < ntiles; t++) {
            const long long t0 = (descending ? ntiles - 1 - t : t) * RT;
            const long long i = t0 + tid;
            double bx = 0., by = 0., bz = 0.;
            if (i < nchain && i != s) {
                Part p;
                load_part_all(v, prog, i, p);
                V3 own = {0., 0., 0.};
                general_op<true>(op.kind, op, S, prog.G, p, g, own, bx, by, bz);
            }
            push[0][tid] = bx; push[1][tid] = by; push[2][tid] = bz;
            __syncthreads();
            if (tid == 0) {
                const int cnt = (int)((nchain - t0) < RT ? (nchain - t0) : RT);
                for (int jj = 0; jj < cnt; jj++) {
                    const int j = descending ? cnt - 1 - jj : jj;
                    if (t0 + j == s) continue;
                    val.x += push[0][j]; val.y += push[1][j]; val.z += push[2][j];
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) { v.ax[s] = val.x; v.ay[s] = val.y; v.az[s] = val.z; }
}

// adds externally combined pushes (3 doubles per op) to the ops' source particles, in op order
__global__ __launch_bounds__(64) void add_pushes_kernel(View v, Program prog, const double* sums) {
    if (threadIdx.x != 0) return;
    for (int o = 0; o < prog.nops; o++) {
        const int src = prog.ops[o].src;
        if (prog.ops[o].slot < 0 || src < 0) continue;
        v.ax[src] += sums[o * 3 + 0]; v.ay[src] += sums[o * 3 + 1]; v.az[src] += sums[o * 3 + 2];
    }
}

// ------------------------------------------------------------------ utilities
__global__ __launch_bounds__(BLOCK) void last_massive_kernel(const double* m, long long n, long long* out) {
    long long best = -1;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK)
        if (m[i] != 0.) best = i;       // ascending per thread: keeps the largest
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_down(best, off, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best >= 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)(best + 1));
}

__global__ __launch_bounds__(BLOCK) void reset_kernel(View v) {
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < v.n; i += (long long)gridDim.x * BLOCK) {
        v.ax[i] = 0.; v.ay[i] = 0.; v.az[i] = 0.;
    }
}

// ------------------------------------------------------------------ launchers
int stream_bodies_per_block() { return 2 * BLOCK; }
int stream_block_count(long long lo, long long n, bool) { return (int)stream_tiles(lo & ~1ll, n); }

hipError_t launch_head(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, bool exact, hipStream_t st, double* star, double h,
                       double* lead, double h_lead) {
    const long long K = v.n_head < v.n ? v.n_head : v.n;
    bool one = K == 1 && v.vx != nullptr && v.m != nullptr;
    for (int o = 0; o < prog.nops && one; o++)
        if (prog.ops[o].kind == REBXAMD_OP_MOF && prog.ops[o].iarg == REBX_COORDINATES_BARYCENTRIC) one = false;   // there body 0 acts on itself
    if (one) {
        hipLaunchKernelGGL(head_one_kernel, dim3(1), dim3(64), 0, st, v, prog, hc, star, h, lead, h_lead);
        return hipGetLastError();
    }
    if (lead != nullptr) return hipErrorInvalidValue;       // (the leading drift rides only in the one-body head pass)
    if (exact) hipLaunchKernelGGL(head_kernel<true>, dim3(1), dim3(64), 0, st, v, prog, hc, status, star, h);
    else hipLaunchKernelGGL(head_kernel<false>, dim3(1), dim3(64), 0, st, v, prog, hc, status, star, h);
    return hipGetLastError();
}

hipError_t launch_head_batch(const View& v, const Program& prog, HeadConsts* hc, double* star, int nsub, double h, double h_last, double* lead,
                             double h_lead, hipStream_t st) {
    if (nsub < 1 || v.n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_one_batch_kernel, dim3(1), dim3(64), 0, st, v, prog, hc, star, nsub, h, h_last, lead, h_lead);
    return hipGetLastError();
}

// Star + planets: the head passes and the head's own Wisdom-Holman steps of nsub consecutive steps, one after the other in one
// wavefront (lane = head body; every lane reads back only what it has stored itself): tables hc[j] and frame records
// star[16 j] of step j, the head's state after the batch in the columns and in `jac`.  The head drifts over h after every
// step but the last of the batch, over h_last after that one.  lead != nullptr (first step of a run): the start of the run
// first (wh_head_begin_body: record `lead`, leading drift over h_lead).
template <bool EXACT>
__global__ __launch_bounds__(64) void wh_head_batch_kernel(View v, Program prog, HeadConsts* hc, unsigned* status, int K, double dtk, double* jac,
                                                           double* star, int nsub, double h, double h_last, double* lead, double h_lead, double* pre) {
    __shared__ HeadShared hsh;
    __shared__ WhHeadShared wsh;
    const int t = threadIdx.x;
    if (lead != nullptr) {
        wh_head_begin_body<EXACT>(v, K, prog.G, h_lead, jac, lead, wsh, t);
        wave_sync();
    }
    for (int j = 0; j < nsub; j++) {
        head_body<EXACT>(v, prog, hc + j, status, nullptr, 0., hsh, t);
        wh_head_step_body<EXACT>(v, K, prog.G, dtk, j + 1 < nsub ? h : h_last, jac, star + 16 * j, wsh, t, pre);
        wave_sync();
    }
}

hipError_t launch_wh_head_batch(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, long long K, double dtk, double* jac, double* star,
                                int nsub, double h, double h_last, bool exact, hipStream_t st, double* lead, double h_lead, double* pre) {
    if (nsub < 1 || K < 2 || K > REBXAMD_MAX_HEAD || K > v.n) return hipErrorInvalidValue;
    if (exact) hipLaunchKernelGGL(wh_head_batch_kernel<true>, dim3(1), dim3(64), 0, st, v, prog, hc, status, (int)K, dtk, jac, star, nsub, h, h_last, lead, h_lead, pre);
    else hipLaunchKernelGGL(wh_head_batch_kernel<false>, dim3(1), dim3(64), 0, st, v, prog, hc, status, (int)K, dtk, jac, star, nsub, h, h_last, lead, h_lead, pre);
    return hipGetLastError();
}

template <bool EXACT, bool BACKREACT>
static void launch_stream_cols(dim3 g, dim3 b, hipStream_t st, bool vec2, bool fixed, int cols, const View& v, const Program& prog, long long lo,
                               long long base, const HeadConsts* hc, double* partials, unsigned* status, const StepArgs* step) {
    const StepArgs sa = step ? *step : StepArgs{0., 0., nullptr, EPI_NONE, 0., 0., 0., {nullptr, nullptr, nullptr}, nullptr, nullptr, 0.};
    Sig sig = 0ull;
    if (fixed && vec2 && prog.nops <= 8) for (int o = 0; o < prog.nops; o++) sig |= (Sig)(prog.ops[o].kind & 15) << (4 * o);
    if (step) {
        // a whole Wisdom-Holman step or force-integrator update in the pass (massless tail only)
        if constexpr (!BACKREACT) {
#define EPILOGUE(E) \
    { FIXED_EPI(E, REBXAMD_OP_RADIATION, REBXAMD_OP_J2, REBXAMD_OP_J4, REBXAMD_OP_GR) \
      FIXED_EPI(E, REBXAMD_OP_RADIATION, REBXAMD_OP_J2, REBXAMD_OP_J4, REBXAMD_OP_GR_POTENTIAL, REBXAMD_OP_TIDES, REBXAMD_OP_MOF) \
      FIXED_EPI(E, REBXAMD_OP_RADIATION) \
      if (vec2) hipLaunchKernelGGL((stream_kernel<EXACT, false, true, COL_ALL, 0ull, E>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa); \
      else hipLaunchKernelGGL((stream_kernel<EXACT, false, false, COL_ALL, 0ull, E>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa); \
      return; }
#define FIXED_EPI(E, ...) \
    { constexpr Sig S_ = sig_of(__VA_ARGS__); \
      if (sig == S_) { hipLaunchKernelGGL((stream_kernel<EXACT, false, true, sig_cols(S_, true) | COL_VEL, S_, E>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa); return; } }
            switch (sa.epi) {
            case EPI_WH: EPILOGUE(EPI_WH)
            case EPI_WHB: EPILOGUE(EPI_WHB)
            case EPI_EULER: EPILOGUE(EPI_EULER)
            case EPI_RK2: EPILOGUE(EPI_RK2)
            case EPI_RK4: EPILOGUE(EPI_RK4)
            case EPI_IM: EPILOGUE(EPI_IM)
            default: break;
            }
#undef FIXED_EPI
#undef EPILOGUE
        }
        if constexpr (BACKREACT && !EXACT) {
            // the pass also applies a prefix-sum force (launch_stream has checked the alignment)
#define FIXED_APPLY(...) \
    { constexpr Sig S_ = sig_of(__VA_ARGS__); \
      if (sig == S_) { hipLaunchKernelGGL((stream_kernel<false, true, true, sig_cols(S_, false), S_, EPI_APPLY>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa); return; } }
            FIXED_APPLY(REBXAMD_OP_TIDES, REBXAMD_OP_GR_POTENTIAL)
            FIXED_APPLY(REBXAMD_OP_J2, REBXAMD_OP_J4)
            FIXED_APPLY(REBXAMD_OP_GR_POTENTIAL)
            FIXED_APPLY(REBXAMD_OP_TIDES)
#undef FIXED_APPLY
            hipLaunchKernelGGL((stream_kernel<false, true, true, COL_ALL, 0ull, EPI_APPLY>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa);
        }
        return;
    }
#define LAUNCH(V, C) hipLaunchKernelGGL((stream_kernel<EXACT, BACKREACT, V, C>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa)
    if (!vec2) { LAUNCH(false, COL_ALL); return; }
    // stacks with an instance of their own
#define FIXED(...) \
    { constexpr Sig S_ = sig_of(__VA_ARGS__); \
      if (sig == S_) { hipLaunchKernelGGL((stream_kernel<EXACT, BACKREACT, true, sig_cols(S_, !BACKREACT), S_>), g, b, 0, st, v, prog, lo, base, hc, partials, status, sa); return; } }
    if constexpr (!BACKREACT) {
        FIXED(REBXAMD_OP_RADIATION, REBXAMD_OP_J2, REBXAMD_OP_J4, REBXAMD_OP_GR)
        FIXED(REBXAMD_OP_RADIATION, REBXAMD_OP_J2, REBXAMD_OP_J4, REBXAMD_OP_GR_POTENTIAL, REBXAMD_OP_TIDES, REBXAMD_OP_MOF)
        FIXED(REBXAMD_OP_RADIATION)
        FIXED(REBXAMD_OP_J2, REBXAMD_OP_J4)
        FIXED(REBXAMD_OP_GR)
        FIXED(REBXAMD_OP_GR_POTENTIAL)
        FIXED(REBXAMD_OP_MOF)
    } else {
        FIXED(REBXAMD_OP_TIDES, REBXAMD_OP_MOF, REBXAMD_OP_GR_POTENTIAL)
        FIXED(REBXAMD_OP_TIDES, REBXAMD_OP_GR_POTENTIAL)
        FIXED(REBXAMD_OP_J2, REBXAMD_OP_J4)
        FIXED(REBXAMD_OP_GR_POTENTIAL)
        FIXED(REBXAMD_OP_TIDES)
        FIXED(REBXAMD_OP_MOF)
    }
#undef FIXED
    if (cols == 0) LAUNCH(true, 0);
    else if ((cols & ~(COL_VEL | COL_BETA)) == 0) LAUNCH(true, COL_VEL | COL_BETA);
    else if ((cols & ~(COL_VEL | COL_TIDES | COL_TAU)) == 0) LAUNCH(true, COL_VEL | COL_TIDES | COL_TAU);
    else LAUNCH(true, COL_ALL);
#undef LAUNCH
}

hipError_t launch_stream(const View& v, const Program& prog, long long lo, const HeadConsts* hc, double* partials,
                         int* nblocks_out, bool backreact, bool exact, bool vec2, bool fixed, unsigned* status, hipStream_t st, const StepArgs* step) {
    if (step && step->epi == EPI_APPLY) {
        if (!backreact || exact || !vec2 || lo != 0 || step->tail.segment_bodies != 2 * BLOCK) return hipErrorInvalidValue;
        if (step->tail.all == nullptr && (step->tail.behind_blocks == nullptr || step->tail.seg_totals == nullptr || step->tail.seg_running == nullptr || step->tail.part_mass == nullptr))
            return hipErrorInvalidValue;
    } else if (step && backreact) return hipErrorInvalidValue;
    const long long base = lo & ~1ll;
    const int nb = stream_block_count(lo, v.n, vec2);
    if (nblocks_out) *nblocks_out = nb;
    if (nb <= 0) return hipSuccess;
    // (large grids are padded to whole groups of 8 * XCD_RUN tiles: see the tile mapping in stream_kernel)
    const unsigned padded = (unsigned)nb >= XCD_REMAP_MIN_TILES ? ((unsigned)nb + 8u * XCD_RUN - 1u) / (8u * XCD_RUN) * (8u * XCD_RUN) : (unsigned)nb;
    const dim3 g(padded + ((step && (step->epi == EPI_WH || step->epi == EPI_WHB) && step->next_hc != nullptr) ? 1 : 0)), b(BLOCK);
    const int cols = (prog.need_vel ? COL_VEL : 0) | (prog.need_beta ? COL_BETA : 0) | (prog.need_tides ? COL_TIDES : 0) | (prog.need_tau ? COL_TAU : 0);
    if (exact) {
        if (backreact) launch_stream_cols<true, true>(g, b, st, vec2, fixed, cols, v, prog, lo, base, hc, partials, status, step);
        else launch_stream_cols<true, false>(g, b, st, vec2, fixed, cols, v, prog, lo, base, hc, partials, status, step);
    } else {
        if (backreact) launch_stream_cols<false, true>(g, b, st, vec2, fixed, cols, v, prog, lo, base, hc, partials, status, step);
        else launch_stream_cols<false, false>(g, b, st, vec2, fixed, cols, v, prog, lo, base, hc, partials, status, step);
    }
    return hipGetLastError();
}

hipError_t launch_head_integrate(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, int scheme, double c0, double c1, double c2,
                                 double* w, long long stride, bool exact, hipStream_t st) {
    if (exact) hipLaunchKernelGGL(head_integrate_kernel<true>, dim3(1), dim3(64), 0, st, v, prog, hc, status, scheme, c0, c1, c2, w, stride);
    else hipLaunchKernelGGL(head_integrate_kernel<false>, dim3(1), dim3(64), 0, st, v, prog, hc, status, scheme, c0, c1, c2, w, stride);
    return hipGetLastError();
}

hipError_t launch_head_implicit_midpoint(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, double dt, double* w, long long stride, int* out,
                                         bool exact, hipStream_t st) {
    if (exact) hipLaunchKernelGGL(head_implicit_midpoint_kernel<true>, dim3(1), dim3(64), 0, st, v, prog, hc, status, dt, w, stride, out);
    else hipLaunchKernelGGL(head_implicit_midpoint_kernel<false>, dim3(1), dim3(64), 0, st, v, prog, hc, status, dt, w, stride, out);
    return hipGetLastError();
}

hipError_t launch_stage(const View& v, const Program& prog, HeadConsts* hc, hipStream_t st) {
    hipLaunchKernelGGL(stage_kernel, dim3(1), dim3(64), 0, st, v, prog, hc);
    return hipGetLastError();
}

hipError_t launch_finalize(const View& v, const Program& prog, const double* partials, int nblocks, double* chunk_sums, double* sums_out, hipStream_t st) {
    static_assert(FIN_CHUNKS == 64, "stage 2 folds the chunk sums with one wavefront");
    if (prog.nslots > 0) hipLaunchKernelGGL(finalize_stage1, dim3(FIN_CHUNKS, prog.nslots), dim3(BLOCK), 0, st, partials, nblocks, chunk_sums);
    hipLaunchKernelGGL(finalize_stage2, dim3(1), dim3(64), 0, st, v, prog, chunk_sums, sums_out);
    return hipGetLastError();
}

hipError_t launch_source_replay(const View& v, const Program& prog, const HeadConsts* hc, long long nchain, hipStream_t st) {
    hipLaunchKernelGGL(source_replay_kernel, dim3(prog.nops), dim3(RT), 0, st, v, prog, hc, nchain < v.n ? nchain : v.n);
    return hipGetLastError();
}

hipError_t launch_add_pushes(const View& v, const Program& prog, const double* sums, hipStream_t st) {
    hipLaunchKernelGGL(add_pushes_kernel, dim3(1), dim3(64), 0, st, v, prog, sums);
    return hipGetLastError();
}

hipError_t launch_last_massive(const double* m, long long n, long long* out_dev, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out_dev, 0, sizeof(long long), st);
    if (e != hipSuccess) return e;
    if (n <= 0) return hipSuccess;
    long long nb = (n + BLOCK - 1) / BLOCK;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(last_massive_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, m, n, out_dev);
    return hipGetLastError();
}

hipError_t launch_reset(const View& v, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    long long nb = (v.n + BLOCK - 1) / BLOCK;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(reset_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, v);
    return hipGetLastError();
}

}  // namespace rebxamd

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

namespace rebxamd {

static constexpr int BLOCK = 256;
static constexpr int WAVES = BLOCK / 64;

// ------------------------------------------------------------------ small helpers

__device__ __forceinline__ double ld_or_nan(const double* col, long long i) {
    return col ? col[i] : __builtin_nan("");
}

__device__ __forceinline__ void load_src(const View& v, int idx, Src& S) {
    S.x = v.x[idx]; S.y = v.y[idx]; S.z = v.z[idx];
    S.vx = v.vx ? v.vx[idx] : 0.; S.vy = v.vy ? v.vy[idx] : 0.; S.vz = v.vz ? v.vz[idx] : 0.;
    S.m = v.m ? v.m[idx] : 0.;
    const double rt = ld_or_nan(v.Rt, idx), k1 = ld_or_nan(v.k1, idx);
    S.Rt = is_set(rt) ? rt : 0.;
    S.k1 = is_set(k1) ? k1 : 0.;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// One force on one particle.  Returns true when the particle was acted on; `d` is its own
// update, `br` the push on the source (already signed).
template <bool EXACT>
__device__ __forceinline__ bool sourced_effect(const DevOp& op, const Part& p, const Src& S, double G, V3& d, V3& br) {
    br.x = br.y = br.z = 0.;
    switch (op.kind) {
    case REBXAMD_OP_RADIATION:
        if (!is_set(p.beta)) return false;
        if (EXACT) rad_exact(p, S, G, op.a, d); else rad_fast(p, S, G, op.c, d);
        return true;
    case REBXAMD_OP_J2:
        if (EXACT) j2_exact(p, S, G, op.a, op.b, d, br); else j2_fast(p, S, G, op.a, op.b, d, br);
        return true;
    case REBXAMD_OP_J4:
        if (EXACT) j4_exact(p, S, G, op.a, op.b, d, br); else j4_fast(p, S, G, op.a, op.b, d, br);
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
            grpot_fast(p, S, 6. * gm * gm * op.d, 1. / S.m, d, br);   // op.d = 1/c^2
        }
        return true;
    }
    case REBXAMD_OP_CENTRAL:
        if (EXACT) central_exact(p, S, op.a, op.b, d, br); else central_fast(p, S, op.a, op.b, d, br);
        return true;
    case REBXAMD_OP_TIDES:
        return EXACT ? tides_exact(p, S, G, d, br) : tides_fast(p, S, G, d, br);
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

template <bool EXACT>
__global__ __launch_bounds__(64) void head_kernel(View v, Program prog, HeadConsts* hc, unsigned* status) {
    __shared__ HeadShared sh;
    const int t = threadIdx.x;
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
    __syncthreads();

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
            __syncthreads();
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
            __syncthreads();
            // --- post-Newtonian term per Jacobi body
            if (valid && t >= 1) {
                Jac j;
                j.x = sh.pj[t][0]; j.y = sh.pj[t][1]; j.z = sh.pj[t][2];
                j.vx = sh.pj[t][3]; j.vy = sh.pj[t][4]; j.vz = sh.pj[t][5];
                j.ax = sh.pj[t][6]; j.ay = sh.pj[t][7]; j.az = sh.pj[t][8];
                V3 corr;
                const double mu = G * sh.hp[0].m;
                const int q = EXACT ? gr_term_exact(j, mu, op.c, op.iarg, corr) : gr_term_fast(j, mu, op.d, op.iarg, corr);
                if (q == 10) atomicOr(status, WARN_GR_ITER);
                sh.pj[t][6] = corr.x; sh.pj[t][7] = corr.y; sh.pj[t][8] = corr.z;
            }
            __syncthreads();
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
            __syncthreads();
            if (valid) { acc.x += sh.na[t][0]; acc.y += sh.na[t][1]; acc.z += sh.na[t][2]; }
            __syncthreads();
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
            __syncthreads();
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
            __syncthreads();
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
            __syncthreads();
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
                    acted = sourced_effect<EXACT>(op, p, S, G, d, br);
                    if (acted) {
#pragma clang fp contract(off)
                        acc.x += d.x; acc.y += d.y; acc.z += d.z;
                    }
                }
            }
            sh.br[t][0] = br.x; sh.br[t][1] = br.y; sh.br[t][2] = br.z;
            sh.brf[t] = acted ? 1 : 0;
            __syncthreads();
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
            __syncthreads();
        }
    }
    if (valid) { v.ax[t] = acc.x; v.ay[t] = acc.y; v.az[t] = acc.z; }
}

// ------------------------------------------------------------------ streaming kernel

struct StreamShared {
    Src src[REBXAMD_MAX_OPS];
    double head[REBXAMD_MAX_HEAD][4];           // x y z m of massive head bodies (gr tail)
    int nheadm;
    HeadConsts hc;
    double part[WAVES][REBXAMD_MAX_OPS][3];
};

template <bool VEC2>
__device__ __forceinline__ void ld2(const double* col, long long i0, bool two, double& a, double& b) {
    if (VEC2 && two) {
        const double2 t = *reinterpret_cast<const double2*>(col + i0);
        a = t.x; b = t.y;
    } else {
        a = col[i0];
        b = two ? col[i0 + 1] : 0.;
    }
}

template <bool EXACT, bool BACKREACT, bool VEC2>
__global__ __launch_bounds__(BLOCK) void stream_kernel(View v, Program prog, long long lo, long long base,
                                                       const HeadConsts* hcg, double* partials, unsigned* status) {
    __shared__ StreamShared sh;
    const int tid = threadIdx.x;
    const double G = prog.G;

    // ---- stage sources (and, for the gr tail, the massive head bodies) in LDS
    if (tid < prog.nops) {
        const int s = prog.ops[tid].src;
        if (s >= 0) load_src(v, s, sh.src[tid]);
    }
    const bool tail = !BACKREACT;
    if (tail && (prog.gr_op >= 0 || prog.mofcom_op >= 0)) {
        // HeadConsts is a few hundred doubles; copy cooperatively
        const double* g = reinterpret_cast<const double*>(hcg);
        double* l = reinterpret_cast<double*>(&sh.hc);
        for (int k = tid; k < (int)(sizeof(HeadConsts) / sizeof(double)); k += BLOCK) l[k] = g[k];
        if (prog.gr_op >= 0 && tid < 64) {      // wavefront 0 compacts the massive head bodies, keeping their order
            const int K = (int)(v.n_head < v.n ? v.n_head : v.n);
            const double mu = tid < K ? v.m[tid] : 0.;
            const unsigned long long mask = __ballot(mu != 0.);
            if (mu != 0.) {
                const int c = __popcll(mask & ((1ull << tid) - 1ull));
                sh.head[c][0] = v.x[tid]; sh.head[c][1] = v.y[tid]; sh.head[c][2] = v.z[tid]; sh.head[c][3] = mu;
            }
            if (tid == 0) sh.nheadm = __popcll(mask);
        }
    }
    __syncthreads();

    const long long i0 = base + 2 * ((long long)blockIdx.x * BLOCK + tid);
    const bool in0 = i0 < v.n, in1 = i0 + 1 < v.n;
    const bool act0 = in0 && i0 >= lo, act1 = in1 && i0 + 1 >= lo;

    Part p[2];
    V3 acc[2];
    if (in0) {
        ld2<VEC2>(v.x, i0, in1, p[0].x, p[1].x);
        ld2<VEC2>(v.y, i0, in1, p[0].y, p[1].y);
        ld2<VEC2>(v.z, i0, in1, p[0].z, p[1].z);
        if (prog.need_vel) {
            ld2<VEC2>(v.vx, i0, in1, p[0].vx, p[1].vx);
            ld2<VEC2>(v.vy, i0, in1, p[0].vy, p[1].vy);
            ld2<VEC2>(v.vz, i0, in1, p[0].vz, p[1].vz);
        } else { p[0].vx = p[0].vy = p[0].vz = p[1].vx = p[1].vy = p[1].vz = 0.; }
        // behind the head every body is massless by the caller's assertion: no need to read m
        if (BACKREACT) ld2<VEC2>(v.m, i0, in1, p[0].m, p[1].m);
        else { p[0].m = 0.; p[1].m = 0.; }
        const double nan = __builtin_nan("");
        if (prog.need_beta && v.beta) ld2<VEC2>(v.beta, i0, in1, p[0].beta, p[1].beta); else { p[0].beta = p[1].beta = nan; }
        if (prog.need_tides && v.Rt) ld2<VEC2>(v.Rt, i0, in1, p[0].Rt, p[1].Rt); else { p[0].Rt = p[1].Rt = nan; }
        if (prog.need_tides && v.k1) ld2<VEC2>(v.k1, i0, in1, p[0].k1, p[1].k1); else { p[0].k1 = p[1].k1 = nan; }
        if (prog.need_tau && v.ta) ld2<VEC2>(v.ta, i0, in1, p[0].ta, p[1].ta); else { p[0].ta = p[1].ta = nan; }
        if (prog.need_tau && v.te) ld2<VEC2>(v.te, i0, in1, p[0].te, p[1].te); else { p[0].te = p[1].te = nan; }
        if (prog.need_tau && v.ti) ld2<VEC2>(v.ti, i0, in1, p[0].ti, p[1].ti); else { p[0].ti = p[1].ti = nan; }
        ld2<VEC2>(v.ax, i0, in1, acc[0].x, acc[1].x);
        ld2<VEC2>(v.ay, i0, in1, acc[0].y, acc[1].y);
        ld2<VEC2>(v.az, i0, in1, acc[0].z, acc[1].z);
    }
    const bool act[2] = {act0, act1};
    const int wave = tid >> 6, lane = tid & 63;

    for (int o = 0; o < prog.nops; o++) {
        const DevOp op = prog.ops[o];
        if (op.kind == REBXAMD_OP_GR) {
            if (tail) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (!act[k]) continue;
                    // Newtonian pull of the massive head bodies (all of them precede this body)
                    double nx = 0., ny = 0., nz = 0.;
                    for (int u = 0; u < sh.nheadm; u++) {
                        if (EXACT) newton_pair_exact(p[k].x, p[k].y, p[k].z, sh.head[u][0], sh.head[u][1], sh.head[u][2], sh.head[u][3], true, G, nx, ny, nz);
                        else newton_pair_fast(p[k].x, p[k].y, p[k].z, sh.head[u][0], sh.head[u][1], sh.head[u][2], sh.head[u][3], G, nx, ny, nz);
                    }
                    Jac j;
                    {
#pragma clang fp contract(off)
                        j.x = p[k].x - sh.hc.gr_c[0]; j.y = p[k].y - sh.hc.gr_c[1]; j.z = p[k].z - sh.hc.gr_c[2];
                        j.vx = p[k].vx - sh.hc.gr_c[3]; j.vy = p[k].vy - sh.hc.gr_c[4]; j.vz = p[k].vz - sh.hc.gr_c[5];
                        j.ax = nx - sh.hc.gr_c[6]; j.ay = ny - sh.hc.gr_c[7]; j.az = nz - sh.hc.gr_c[8];
                    }
                    V3 corr;
                    const int q = EXACT ? gr_term_exact(j, sh.hc.gr_mu, op.c, op.iarg, corr) : gr_term_fast(j, sh.hc.gr_mu, op.d, op.iarg, corr);
                    if (q == 10) atomicOr(status, WARN_GR_ITER);
                    // a massless body's inertial correction equals its Jacobi correction
                    acc[k].x += corr.x; acc[k].y += corr.y; acc[k].z += corr.z;
                }
            }
            continue;
        }
        if (op.kind == REBXAMD_OP_MOF && op.iarg != REBX_COORDINATES_PARTICLE) {
            if (tail) {
                Src ref;
                ref.x = sh.hc.com[0]; ref.y = sh.hc.com[1]; ref.z = sh.hc.com[2];
                ref.vx = sh.hc.com[3]; ref.vy = sh.hc.com[4]; ref.vz = sh.hc.com[5]; ref.m = sh.hc.com[6];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (!act[k]) continue;
                    V3 a;
                    if (EXACT) mof_exact(p[k], ref, a); else mof_fast(p[k], ref, a);
                    acc[k].x += a.x; acc[k].y += a.y; acc[k].z += a.z;
                    for (int i = 0; i < sh.hc.nbary; i++) {
                        acc[k].x -= sh.hc.bary[i][0]; acc[k].y -= sh.hc.bary[i][1]; acc[k].z -= sh.hc.bary[i][2];
                    }
                }
            }
            continue;
        }
        const Src S = sh.src[o];
        double bx = 0., by = 0., bz = 0.;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!act[k] || i0 + k == op.src) continue;
            V3 d, br;
            if (op.kind == REBXAMD_OP_MOF) {
                if (BACKREACT) {
                    mof_particle<EXACT>(p[k], S, acc[k], br);
                    bx += br.x; by += br.y; bz += br.z;
                } else {
                    V3 a;
                    if (EXACT) mof_exact(p[k], S, a); else mof_fast(p[k], S, a);
                    acc[k].x += a.x; acc[k].y += a.y; acc[k].z += a.z;
                }
            } else if (sourced_effect<EXACT>(op, p[k], S, G, d, br)) {
#pragma clang fp contract(off)
                acc[k].x += d.x; acc[k].y += d.y; acc[k].z += d.z;
                if (BACKREACT) { bx += br.x; by += br.y; bz += br.z; }
            }
        }
        if (BACKREACT && op.slot >= 0) {
            // a wavefront whose bodies are all massless pushes on nothing
            const bool any_mass = __ballot((act0 && p[0].m != 0.) || (act1 && p[1].m != 0.)) != 0ull;
            if (any_mass) { bx = wave_sum(bx); by = wave_sum(by); bz = wave_sum(bz); }
            else { bx = by = bz = 0.; }
            if (lane == 0) { sh.part[wave][op.slot][0] = bx; sh.part[wave][op.slot][1] = by; sh.part[wave][op.slot][2] = bz; }
        }
    }

    if (in0) {
        if (VEC2 && act0 && act1) {
            *reinterpret_cast<double2*>(v.ax + i0) = make_double2(acc[0].x, acc[1].x);
            *reinterpret_cast<double2*>(v.ay + i0) = make_double2(acc[0].y, acc[1].y);
            *reinterpret_cast<double2*>(v.az + i0) = make_double2(acc[0].z, acc[1].z);
        } else {
            if (act0) { v.ax[i0] = acc[0].x; v.ay[i0] = acc[0].y; v.az[i0] = acc[0].z; }
            if (act1) { v.ax[i0 + 1] = acc[1].x; v.ay[i0 + 1] = acc[1].y; v.az[i0 + 1] = acc[1].z; }
        }
    }

    if (BACKREACT && prog.nslots > 0) {
        __syncthreads();
        if (tid < prog.nslots * 3) {
            const int s = tid / 3, c = tid % 3;
            double t = sh.part[0][s][c];
#pragma unroll
            for (int w = 1; w < WAVES; w++) t += sh.part[w][s][c];
            partials[((long long)s * gridDim.x + blockIdx.x) * 3 + c] = t;
        }
    }
}

// ------------------------------------------------------------------ finalize: partials -> sources
__global__ __launch_bounds__(BLOCK) void finalize_kernel(View v, Program prog, const double* partials, int nblocks) {
    __shared__ double red[WAVES][3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int o = 0; o < prog.nops; o++) {
        const int s = prog.ops[o].slot;
        if (s < 0) continue;
        double x = 0., y = 0., z = 0.;
        for (int b = tid; b < nblocks; b += BLOCK) {
            const double* q = partials + ((long long)s * nblocks + b) * 3;
            x += q[0]; y += q[1]; z += q[2];
        }
        x = wave_sum(x); y = wave_sum(y); z = wave_sum(z);
        if (lane == 0) { red[wave][0] = x; red[wave][1] = y; red[wave][2] = z; }
        __syncthreads();
        if (tid == 0) {
            double tx = red[0][0], ty = red[0][1], tz = red[0][2];
            for (int w = 1; w < WAVES; w++) { tx += red[w][0]; ty += red[w][1]; tz += red[w][2]; }
            const int src = prog.ops[o].src;
            v.ax[src] += tx; v.ay[src] += ty; v.az[src] += tz;
        }
        __syncthreads();
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
int stream_block_count(long long lo, long long n, bool) {
    const long long base = lo & ~1ll;
    const long long pairs = (n - base + 1) / 2;
    return (int)((pairs + BLOCK - 1) / BLOCK);
}

hipError_t launch_head(const View& v, const Program& prog, HeadConsts* hc, unsigned* status, bool exact, hipStream_t st) {
    if (exact) hipLaunchKernelGGL(head_kernel<true>, dim3(1), dim3(64), 0, st, v, prog, hc, status);
    else hipLaunchKernelGGL(head_kernel<false>, dim3(1), dim3(64), 0, st, v, prog, hc, status);
    return hipGetLastError();
}

hipError_t launch_stream(const View& v, const Program& prog, long long lo, const HeadConsts* hc, double* partials,
                         int* nblocks_out, bool backreact, bool exact, bool vec2, unsigned* status, hipStream_t st) {
    const long long base = lo & ~1ll;
    const int nb = stream_block_count(lo, v.n, vec2);
    if (nblocks_out) *nblocks_out = nb;
    if (nb <= 0) return hipSuccess;
    const dim3 g(nb), b(BLOCK);
#define LAUNCH(E, B, V) hipLaunchKernelGGL((stream_kernel<E, B, V>), g, b, 0, st, v, prog, lo, base, hc, partials, status)
    if (exact) {
        if (backreact) { if (vec2) LAUNCH(true, true, true); else LAUNCH(true, true, false); }
        else           { if (vec2) LAUNCH(true, false, true); else LAUNCH(true, false, false); }
    } else {
        if (backreact) { if (vec2) LAUNCH(false, true, true); else LAUNCH(false, true, false); }
        else           { if (vec2) LAUNCH(false, false, true); else LAUNCH(false, false, false); }
    }
#undef LAUNCH
    return hipGetLastError();
}

hipError_t launch_finalize(const View& v, const Program& prog, const double* partials, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(BLOCK), 0, st, v, prog, partials, nblocks);
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

// scan_kernels.hip -- gr and Jacobi/barycentric modify_orbits_forces for arbitrary mass layouts.
//
// With massive bodies anywhere in the array the reference's sequential recurrences cannot be
// folded into constants; they are prefix/suffix sums instead:
//   * Jacobi coordinates (REBOUND's inertial_to_jacobi, used by src/gr.c:108): body i is measured
//     from  S_i / eta_i  with  eta_i = sum_{k<i} m_k,  S_i = sum_{k<i} m_k q_k   (q = pos, vel, acc)
//     -> exclusive prefix sums of (m, m x, m v, m a).
//   * back to inertial accelerations (jacobi_to_inertial_acc, src/gr.c:162): unrolling the reverse
//     recurrence gives  a_i = aJ_i - sum_{k>=i} (m_k / eta_{k+1}) aJ_k  (aJ_0 = 0)
//     -> inclusive suffix sums of w_k = (m_k / (eta_k + m_k)) aJ_k.
//   * modify_orbits_forces in Jacobi coordinates (src/rebxtools.c:71-107): reference point of body
//     i = centre of mass of bodies 0..i-1 = S_i / eta_i; every body j <= i loses
//     (m_i / (eta_i + m_i)) a_i -> the same two scans.  Barycentric: the sums over all bodies.
//   * the Newtonian term of gr (src/gr.c:86-103) needs every massive body: they are compacted
//     (deterministically, by a prefix sum of the "is massive" flags) and each body loops over
//     the compact list in LDS tiles -> O(N * massive).
// Sums are evaluated as parallel scans, so results differ from the sequential reference at
// rounding level (the 1e-12 bar), in either arithmetic mode.
#include <type_traits>
#include "internal.h"
#include "effects.h"
#include "wave_ops.h"
#include "scan_ops.h"

namespace rebxamd {

enum ScanKind { SCAN_GR = 0, SCAN_MOF_JACOBI = 1, SCAN_MOF_BARY = 2 };

// ---- per-element quantities that get scanned
// read-once streams carry the nontemporal hint (they would only push useful lines out of the caches)
__device__ __forceinline__ double ldnt(const double* col, long long i) { return __builtin_nontemporal_load(col + i); }

template <int C>
__device__ __forceinline__ void weighted_state(const View& v, const double* nax, const double* nay, const double* naz, long long i, double (&q)[C]) {
    const double m = ldnt(v.m, i);
    q[0] = m;
    q[1] = m * ldnt(v.x, i); q[2] = m * ldnt(v.y, i); q[3] = m * ldnt(v.z, i);
    q[4] = m * ldnt(v.vx, i); q[5] = m * ldnt(v.vy, i); q[6] = m * ldnt(v.vz, i);
    if constexpr (C == 10) { q[7] = m * ldnt(nax, i); q[8] = m * ldnt(nay, i); q[9] = m * ldnt(naz, i); }
}

// Element order inside the weighted-sum passes: a workgroup covers SCHUNK consecutive bodies, each of
// its SW wavefronts a contiguous segment of 64*ST of them, tile by tile (64 consecutive bodies per
// tile).  Scans then never leave the wavefront: no LDS, no barriers inside the tile loop.
__device__ __forceinline__ long long segment_start(int wave) { return (long long)blockIdx.x * SCHUNK + (long long)wave * (64 * ST); }

// pass 1: totals of the weighted state per wavefront segment (wavesums, may be null) and per workgroup
template <int C>
__global__ __launch_bounds__(SB) void scan_block_totals(View v, const double* nax, const double* nay, const double* naz, double* blocksums,
                                                        double* wavesums) {
    __shared__ double lds[SW][C];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long seg = segment_start(wave);
    double acc[C], cur[C], nxt[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0.;
    // Loads carry no branch (rows behind the end read the last body and are masked afterwards) and
    // the next tile is requested before the current one is added up: a tile's loads then leave
    // together instead of one by one, each waited for.
    const long long last = v.n - 1;
    auto request = [&](int t, double (&raw)[C]) {
        const long long i = seg + t * 64 + lane, ic = i < v.n ? i : last;
        raw[0] = ldnt(v.m, ic);
        raw[1] = ldnt(v.x, ic); raw[2] = ldnt(v.y, ic); raw[3] = ldnt(v.z, ic);
        raw[4] = ldnt(v.vx, ic); raw[5] = ldnt(v.vy, ic); raw[6] = ldnt(v.vz, ic);
        if constexpr (C == 10) { raw[7] = ldnt(nax, ic); raw[8] = ldnt(nay, ic); raw[9] = ldnt(naz, ic); }
    };
    request(0, cur);
#pragma unroll
    for (int t = 0; t < ST; t++) {
        if (t + 1 < ST) request(t + 1, nxt);
        const double m = seg + t * 64 + lane < v.n ? cur[0] : 0.;
        acc[0] += m;
#pragma unroll
        for (int c = 1; c < C; c++) acc[c] += m * cur[c];
#pragma unroll
        for (int c = 0; c < C; c++) cur[c] = nxt[c];
    }
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = wave_total(acc[c], lane);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            lds[wave][c] = acc[c];
            if (wavesums != nullptr) wavesums[((long long)blockIdx.x * SW + wave) * C + c] = acc[c];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            double t = 0.;
            for (int w = 0; w < SW; w++) t += lds[w][c];
            blocksums[(long long)blockIdx.x * C + c] = t;
        }
    }
}


// What the per-body pass reads of one body (zeros / unset behind the end of the view).
template <bool WITH_ACC>
struct ScanRaw {
    double m, x, y, z, vx, vy, vz, ta, te, ti, nx, ny, nz;
    __device__ __forceinline__ void load(const View& v, const double* nax, const double* nay, const double* naz, long long i) {
        const double nan = unset_value();
        m = x = y = z = vx = vy = vz = nx = ny = nz = 0.;
        ta = te = ti = nan;
        if (i < v.n) {
            m = ldnt(v.m, i); x = ldnt(v.x, i); y = ldnt(v.y, i); z = ldnt(v.z, i); vx = ldnt(v.vx, i); vy = ldnt(v.vy, i); vz = ldnt(v.vz, i);
            if (v.ta) ta = ldnt(v.ta, i);
            if (v.te) te = ldnt(v.te, i);
            if (v.ti) ti = ldnt(v.ti, i);
            if (WITH_ACC) { nx = ldnt(nax, i); ny = ldnt(nay, i); nz = ldnt(naz, i); }
        }
    }
};

// pass 3: per-body force in its own frame (aJ) and its weighted reaction (w); totals of w per
// wavefront segment (wwave) and per workgroup (wsums)
template <int KIND, bool EXACT>
__global__ __launch_bounds__(SB) void scan_main(View v, DevOp op, double G, const double* nax, const double* nay, const double* naz,
                                                const double* blocksums, const double* wavesums, const double* totals,
                                                double* ajx, double* ajy, double* ajz, double* wf,
                                                double* wsums, double* wwave, unsigned* status, const double* pre, long long first_acting, const double* m0,
                                                double* wrun = nullptr, double* part_mass = nullptr) {
    // wrun (may be null): running totals of the reactions at the end of each of the first SSUB - 1 parts of every segment, for
    // a pass that applies the force one wavefront per part; part_mass (with wrun): the mass of every body before each part --
    // that pass then forms the share f = m / (eta + m) itself (wf is not written)
    constexpr int C = (KIND == SCAN_GR) ? 10 : 7;
    __shared__ double lds3[SW][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long seg = segment_start(wave);
    double base[C];             // sum over every body before the current tile
    double wacc[3] = {0., 0., 0.};
#pragma unroll
    for (int c = 0; c < C; c++) {
        base[c] = 0.;
        if (KIND != SCAN_MOF_BARY) {
            base[c] = blocksums[(long long)blockIdx.x * C + c];
            for (int w = 0; w < wave; w++) base[c] += wavesums[((long long)blockIdx.x * SW + w) * C + c];
            if (pre != nullptr) base[c] += pre[c];          // bodies that precede this view (other shards, the head)
        }
    }
    // the inputs of the next tile are requested before the current one is worked on
    ScanRaw<KIND == SCAN_GR> cur, nxt;
    cur.load(v, nax, nay, naz, seg + lane);
    nxt = cur;
    for (int t = 0; t < ST; t++) {
        const long long i = seg + t * 64 + lane;
        const bool live = i < v.n;
        if (t + 1 < ST) nxt.load(v, nax, nay, naz, i + 64);
        if (KIND != SCAN_MOF_BARY && part_mass != nullptr && t % (ST / SSUB) == 0 && lane == 0)
            part_mass[((long long)blockIdx.x * SW + wave) * SSUB + t / (ST / SSUB)] = base[0];
        double q[C], own[C];
        q[0] = cur.m;
        q[1] = cur.m * cur.x; q[2] = cur.m * cur.y; q[3] = cur.m * cur.z;
        q[4] = cur.m * cur.vx; q[5] = cur.m * cur.vy; q[6] = cur.m * cur.vz;
        if constexpr (C == 10) { q[7] = cur.m * cur.nx; q[8] = cur.m * cur.ny; q[9] = cur.m * cur.nz; }
#pragma unroll
        for (int c = 0; c < C; c++) { own[c] = q[c]; if (KIND != SCAN_MOF_BARY) q[c] = wave_incl_scan(q[c], lane); }
        if (live) {
            V3 a = {0., 0., 0.};
            double f = 0.;      // share of the body's force that falls back on the bodies before it
            double S[C];        // exclusive prefix (bodies before i), or the grand total for barycentric
#pragma unroll
            for (int c = 0; c < C; c++) S[c] = (KIND == SCAN_MOF_BARY) ? totals[c] : base[c] + (q[c] - own[c]);
            const double eta = S[0], mi = own[0];
            Part p;
            p.x = cur.x; p.y = cur.y; p.z = cur.z; p.vx = cur.vx; p.vy = cur.vy; p.vz = cur.vz; p.m = mi;
            p.beta = p.Rt = p.k1 = unset_value();
            p.ta = cur.ta; p.te = cur.te; p.ti = cur.ti;
            if constexpr (KIND == SCAN_GR) {
                if (i >= first_acting) {
                    const double ei = 1. / eta;
                    Jac j;
                    j.x = p.x - S[1] * ei; j.y = p.y - S[2] * ei; j.z = p.z - S[3] * ei;
                    j.vx = p.vx - S[4] * ei; j.vy = p.vy - S[5] * ei; j.vz = p.vz - S[6] * ei;
                    j.ax = cur.nx - S[7] * ei; j.ay = cur.ny - S[8] * ei; j.az = cur.nz - S[9] * ei;
                    const double mu = G * m0[0];       // body 0 of the whole system (the view may be a range behind it)
                    const int it = EXACT ? gr_term_exact(j, mu, op.c, op.iarg, a)
                                         : gr_term_fast(j, fast_rsqrt(j.x * j.x + j.y * j.y + j.z * j.z), mu, op.d, op.iarg, a);
                    if (it == 10) atomicOr(status, WARN_GR_ITER);
                    f = mi / (eta + mi);
                }
            } else {
                const bool acts = (KIND == SCAN_MOF_BARY) || i >= first_acting;
                if (acts) {
                    Src ref;
                    const double inv = eta > 0. ? 1. / eta : 1.;
                    ref.x = S[1] * inv; ref.y = S[2] * inv; ref.z = S[3] * inv; ref.vx = S[4] * inv; ref.vy = S[5] * inv; ref.vz = S[6] * inv; ref.m = eta;
                    if (EXACT) mof_exact(p, ref, a); else mof_fast(p, ref, a);
                    f = (KIND == SCAN_MOF_BARY) ? mi / eta : mi / (eta + mi);
                }
            }
            // the reaction itself (f times the force) is not stored: the last pass forms it again
            // from these four numbers, with the same rounding
            __builtin_nontemporal_store(a.x, ajx + i); __builtin_nontemporal_store(a.y, ajy + i); __builtin_nontemporal_store(a.z, ajz + i);
            if (KIND != SCAN_MOF_BARY && part_mass == nullptr) __builtin_nontemporal_store(f, wf + i);
            wacc[0] += reaction(f, a.x); wacc[1] += reaction(f, a.y); wacc[2] += reaction(f, a.z);
        }
        if (wrun != nullptr && (t + 1) % (ST / SSUB) == 0 && t + 1 < ST) {
            const int part = (t + 1) / (ST / SSUB) - 1;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double h = wave_total(wacc[k], lane);
                if (lane == 0) wrun[(((long long)blockIdx.x * SW + wave) * (SSUB - 1) + part) * 3 + k] = h;
            }
        }
        if (KIND != SCAN_MOF_BARY) {
#pragma unroll
            for (int c = 0; c < C; c++) base[c] += lane_value(q[c], 63);
        }
        cur = nxt;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) wacc[k] = wave_total(wacc[k], lane);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lds3[wave][k] = wacc[k]; wwave[((long long)blockIdx.x * SW + wave) * 3 + k] = wacc[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double t = 0.;
            for (int w = 0; w < SW; w++) t += lds3[w][k];
            wsums[(long long)blockIdx.x * 3 + k] = t;
        }
    }
}

// Pass 3 of modify_orbits_forces in Jacobi coordinates, default arithmetic, for a caller whose pass applies the force one
// wavefront per part of a segment (16-byte aligned columns, which that caller has checked): a part is one tile of 128
// consecutive bodies, two per lane -- 16-byte loads and stores and half the wavefront scans of the one-body-per-lane pass.
// A tile's 1 KiB of requests per stream and wavefront keeps the memory system busy without a second set of registers for
// the next tile (162 registers as it is: three wavefronts per SIMD; held to 128 it spills and loses 20 %).  Outputs as
// scan_main's with wrun and part_mass (aJ stored, the share not).
typedef double scan_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ldnt2(const double* col, long long i, long long n, double none, double& a, double& b) {
    a = b = none;
    if (col == nullptr) return;
    if (i + 1 < n) { const scan_d2 t = __builtin_nontemporal_load(reinterpret_cast<const scan_d2*>(col + i)); a = t.x; b = t.y; }
    else if (i < n) a = ldnt(col, i);
}
__device__ __forceinline__ void stnt2(double* col, long long i, long long n, double a, double b) {
    if (i + 1 < n) { scan_d2 t; t.x = a; t.y = b; __builtin_nontemporal_store(t, reinterpret_cast<scan_d2*>(col + i)); }
    else if (i < n) __builtin_nontemporal_store(a, col + i);
}

__global__ __launch_bounds__(SB) void mof_jacobi_pairs(View v, const double* blocksums, const double* wavesums, double* ajx, double* ajy, double* ajz,
                                                       double* wsums, double* wwave, double* wrun, double* part_mass, const double* pre, long long first_acting) {
    static_assert(ST / SSUB == 2, "a part is two tiles of 64 bodies = one tile of 128");
    __shared__ double lds3[SW][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long seg = segment_start(wave);
    const double nan = unset_value();
    double base[7], run[3] = {0., 0., 0.};
#pragma unroll
    for (int c = 0; c < 7; c++) {
        base[c] = blocksums[(long long)blockIdx.x * 7 + c];
        for (int w = 0; w < wave; w++) base[c] += wavesums[((long long)blockIdx.x * SW + w) * 7 + c];
        if (pre != nullptr) base[c] += pre[c];
    }
#pragma unroll 1
    for (int t = 0; t < SSUB; t++) {
        const long long i = seg + t * 128 + 2 * lane;
        double m[2], r[6][2], tau[3][2];
        ldnt2(v.m, i, v.n, 0., m[0], m[1]);
        ldnt2(v.x, i, v.n, 0., r[0][0], r[0][1]); ldnt2(v.y, i, v.n, 0., r[1][0], r[1][1]); ldnt2(v.z, i, v.n, 0., r[2][0], r[2][1]);
        ldnt2(v.vx, i, v.n, 0., r[3][0], r[3][1]); ldnt2(v.vy, i, v.n, 0., r[4][0], r[4][1]); ldnt2(v.vz, i, v.n, 0., r[5][0], r[5][1]);
        ldnt2(v.ta, i, v.n, nan, tau[0][0], tau[0][1]); ldnt2(v.te, i, v.n, nan, tau[1][0], tau[1][1]); ldnt2(v.ti, i, v.n, nan, tau[2][0], tau[2][1]);
        if (lane == 0) part_mass[((long long)blockIdx.x * SW + wave) * SSUB + t] = base[0];
        // S0: the sums over every body before this lane's first body; base moves on to the end of the tile
        double q0[7], S0[7];
        {
            q0[0] = m[0];
            const double pair = m[0] + m[1];
            const double incl = wave_incl_scan(pair, lane);
            S0[0] = base[0] + (incl - pair);
            base[0] += lane_value(incl, 63);
        }
#pragma unroll
        for (int c = 1; c < 7; c++) {
            q0[c] = m[0] * r[c - 1][0];
            const double pair = q0[c] + m[1] * r[c - 1][1];
            const double incl = wave_incl_scan(pair, lane);
            S0[c] = base[c] + (incl - pair);
            base[c] += lane_value(incl, 63);
        }
        double a2[3][2] = {{0., 0.}, {0., 0.}, {0., 0.}}, wl[3] = {0., 0., 0.};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (i + k >= v.n || i + k < first_acting) continue;
            double S[7];
#pragma unroll
            for (int c = 0; c < 7; c++) S[c] = S0[c] + (k ? q0[c] : 0.);
            const double eta = S[0], mi = m[k];
            Part p;
            p.x = r[0][k]; p.y = r[1][k]; p.z = r[2][k]; p.vx = r[3][k]; p.vy = r[4][k]; p.vz = r[5][k]; p.m = mi;
            p.beta = p.Rt = p.k1 = nan;
            p.ta = tau[0][k]; p.te = tau[1][k]; p.ti = tau[2][k];
            Src ref;
            const double inv = eta > 0. ? 1. / eta : 1.;
            ref.x = S[1] * inv; ref.y = S[2] * inv; ref.z = S[3] * inv; ref.vx = S[4] * inv; ref.vy = S[5] * inv; ref.vz = S[6] * inv; ref.m = eta;
            V3 a = {0., 0., 0.};
            mof_fast(p, ref, a);
            const double f = mi / (eta + mi);
            a2[0][k] = a.x; a2[1][k] = a.y; a2[2][k] = a.z;
            wl[0] += reaction(f, a.x); wl[1] += reaction(f, a.y); wl[2] += reaction(f, a.z);
        }
        stnt2(ajx, i, v.n, a2[0][0], a2[0][1]); stnt2(ajy, i, v.n, a2[1][0], a2[1][1]); stnt2(ajz, i, v.n, a2[2][0], a2[2][1]);
#pragma unroll
        for (int k = 0; k < 3; k++) run[k] += wave_total(wl[k], lane);
        if (t + 1 < SSUB && lane == 0) {
            const long long hrow = ((long long)blockIdx.x * SW + wave) * (SSUB - 1) + t;
#pragma unroll
            for (int k = 0; k < 3; k++) wrun[hrow * 3 + k] = run[k];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lds3[wave][k] = run[k]; wwave[((long long)blockIdx.x * SW + wave) * 3 + k] = run[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double t = 0.;
            for (int w = 0; w < SW; w++) t += lds3[w][k];
            wsums[(long long)blockIdx.x * 3 + k] = t;
        }
    }
}

// pass 5: a_i += aJ_i - (sum of the reactions of bodies k >= i)   [barycentric: of all bodies]
// A tile's loads carry no branch (rows behind the end read the last body and are masked) and the
// next tile is requested before the current one is worked on.
template <bool TOTAL>
__global__ __launch_bounds__(SB) void scan_apply(View v, const double* ajx, const double* ajy, const double* ajz, const double* wf,
                                                 const double* wsums_excl, const double* wwave, const double* wtotals, const double* suf) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long seg = segment_start(wave);
    double base[3] = {0., 0., 0.};          // reactions of every body behind the current tile
    if (!TOTAL) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            base[k] = wsums_excl[(long long)blockIdx.x * 3 + k];
            for (int w = SW - 1; w > wave; w--) base[k] += wwave[((long long)blockIdx.x * SW + w) * 3 + k];
            if (suf != nullptr) base[k] += suf[k];          // bodies that follow this view
        }
    }
    double tot[3] = {0., 0., 0.};
    if (TOTAL) { tot[0] = wtotals[0]; tot[1] = wtotals[1]; tot[2] = wtotals[2]; }
    const long long last = v.n - 1;
    double cur[7], nxt[7];
    auto request = [&](int t, double (&r)[7]) {
        const long long i = seg + t * 64 + lane, ic = i < v.n ? i : last;
        r[0] = ldnt(ajx, ic); r[1] = ldnt(ajy, ic); r[2] = ldnt(ajz, ic);
        // (rows behind the end never read the columns this pass writes)
        r[3] = *(i < v.n ? v.ax + i : ajx + last); r[4] = *(i < v.n ? v.ay + i : ajy + last); r[5] = *(i < v.n ? v.az + i : ajz + last);
        r[6] = TOTAL ? 0. : ldnt(wf, ic);
    };
    request(ST - 1, cur);
#pragma unroll
    for (int t = ST - 1; t >= 0; t--) {
        if (t > 0) request(t - 1, nxt);
        const long long i = seg + t * 64 + lane;
        const bool live = i < v.n;
        double w[3] = {0., 0., 0.};
        if (!TOTAL) {
#pragma unroll
            for (int k = 0; k < 3; k++) w[k] = wave_incl_scan_rev(live ? reaction(cur[6], cur[k]) : 0., lane);
        }
        if (live) {
            double s0, s1, s2;
            if (TOTAL) { s0 = tot[0]; s1 = tot[1]; s2 = tot[2]; }
            else { s0 = w[0] + base[0]; s1 = w[1] + base[1]; s2 = w[2] + base[2]; }
            v.ax[i] = cur[3] + (cur[0] - s0); v.ay[i] = cur[4] + (cur[1] - s1); v.az[i] = cur[5] + (cur[2] - s2);
        }
        if (!TOTAL) {
#pragma unroll
            for (int k = 0; k < 3; k++) base[k] += lane_value(w[k], 0);
        }
#pragma unroll
        for (int k = 0; k < 7; k++) cur[k] = nxt[k];
    }
}

// ---- Newtonian term of gr: compact the massive bodies, then every body sums over them
__global__ __launch_bounds__(SB) void massive_flag_totals(View v, double* blocksums) {
    __shared__ double lds[SW][1];
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    double q[1] = {(i < v.n && v.m[i] != 0.) ? 1. : 0.}, tot[1];
    block_scan<1, false>(q, tot, lds);
    if (threadIdx.x == 0) blocksums[blockIdx.x] = tot[0];
}
__global__ __launch_bounds__(SB) void massive_compact(View v, const double* blocksums, double* cx, double* cy, double* cz, double* cgm, long long* cidx, double G) {
    __shared__ double lds[SW][1];
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    const bool massive = i < v.n && v.m[i] != 0.;
    double q[1] = {massive ? 1. : 0.}, tot[1];
    block_scan<1, false>(q, tot, lds);
    if (massive) {
        const long long pos = (long long)(blocksums[blockIdx.x] + q[0] - 1.);      // counts are exact in fp64
        cx[pos] = v.x[i]; cy[pos] = v.y[i]; cz[pos] = v.z[i]; cgm[pos] = G * v.m[i]; cidx[pos] = i;
    }
}
template <bool EXACT>
__global__ __launch_bounds__(SB) void newton_from_compact(View v, const double* cx, const double* cy, const double* cz, const double* cgm,
                                                          const long long* cidx, const double* count, double* nax, double* nay, double* naz) {
    __shared__ double tx[SB], ty[SB], tz[SB], tg[SB];
    __shared__ long long ti[SB];
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    const bool live = i < v.n;
    const long long M = (long long)count[0];
    double xi = 0., yi = 0., zi = 0.;
    if (live) { xi = v.x[i]; yi = v.y[i]; zi = v.z[i]; }
    double ax = 0., ay = 0., az = 0.;
    for (long long j0 = 0; j0 < M; j0 += SB) {
        const long long j = j0 + threadIdx.x;
        __syncthreads();
        if (j < M) { tx[threadIdx.x] = cx[j]; ty[threadIdx.x] = cy[j]; tz[threadIdx.x] = cz[j]; tg[threadIdx.x] = cgm[j]; ti[threadIdx.x] = cidx[j]; }
        __syncthreads();
        const int cnt = (int)((M - j0) < SB ? (M - j0) : SB);
        for (int k = 0; k < cnt; k++) {
            if (ti[k] == i) continue;
            const double dx = tx[k] - xi, dy = ty[k] - yi, dz = tz[k] - zi;
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (EXACT) {
#pragma clang fp contract(off)
                const double w = tg[k] / (r2 * sqrt(r2));
                ax += w * dx; ay += w * dy; az += w * dz;
            } else {
                const double ri = fast_rsqrt(r2);
                const double w = tg[k] * (ri * ri * ri);
                ax = fma(w, dx, ax); ay = fma(w, dy, ay); az = fma(w, dz, az);
            }
        }
    }
    if (live) { nax[i] = ax; nay[i] = ay; naz[i] = az; }
}

// The same for particle shards: the massive bodies of a range as records {x, y, z, G m, id} (ids are
// global body numbers, exact in fp64), and the pull of a list of such records (gathered from every
// shard, in ascending id order) on the bodies of a range.
__global__ __launch_bounds__(SB) void massive_records(View v, const double* blocksums, double* rec, double G, double id0) {
    __shared__ double lds[SW][1];
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    const bool massive = i < v.n && v.m[i] != 0.;
    double q[1] = {massive ? 1. : 0.}, tot[1];
    block_scan<1, false>(q, tot, lds);
    if (massive) {
        double* r = rec + 5 * (long long)(blocksums[blockIdx.x] + q[0] - 1.);
        r[0] = v.x[i]; r[1] = v.y[i]; r[2] = v.z[i]; r[3] = G * v.m[i]; r[4] = id0 + (double)i;
    }
}
template <bool EXACT>
__global__ __launch_bounds__(SB) void newton_from_records(View v, const double* rec, long long M, double id0, double* nax, double* nay, double* naz) {
    __shared__ double tx[SB], ty[SB], tz[SB], tg[SB], ti[SB];
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    const bool live = i < v.n;
    const double own = id0 + (double)i;
    double xi = 0., yi = 0., zi = 0.;
    if (live) { xi = v.x[i]; yi = v.y[i]; zi = v.z[i]; }
    double ax = 0., ay = 0., az = 0.;
    for (long long j0 = 0; j0 < M; j0 += SB) {
        const long long j = j0 + threadIdx.x;
        __syncthreads();
        if (j < M) { const double* r = rec + 5 * j; tx[threadIdx.x] = r[0]; ty[threadIdx.x] = r[1]; tz[threadIdx.x] = r[2]; tg[threadIdx.x] = r[3]; ti[threadIdx.x] = r[4]; }
        __syncthreads();
        const int cnt = (int)((M - j0) < SB ? (M - j0) : SB);
        for (int k = 0; k < cnt; k++) {
            if (ti[k] == own) continue;
            const double dx = tx[k] - xi, dy = ty[k] - yi, dz = tz[k] - zi;
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (EXACT) {
#pragma clang fp contract(off)
                const double w = tg[k] / (r2 * sqrt(r2));
                ax += w * dx; ay += w * dy; az += w * dz;
            } else {
                const double ri = fast_rsqrt(r2);
                const double w = tg[k] * (ri * ri * ri);
                ax = fma(w, dx, ax); ay = fma(w, dy, ay); az = fma(w, dz, az);
            }
        }
    }
    if (live) { nax[i] = ax; nay[i] = ay; naz[i] = az; }
}

// ---- gr Hamiltonian (reference src/gr.c:190-246): Newtonian pair potential over the massive
// bodies, Jacobi masses (src/rebxtools.c:27-35), kinetic and post-Newtonian terms per body.
__global__ __launch_bounds__(SB) void newton_pot_from_compact(View v, const double* cx, const double* cy, const double* cz, const double* cgm,
                                                              const long long* cidx, const double* count, double* pot) {
    __shared__ double tx[SB], ty[SB], tz[SB], tg[SB];
    __shared__ long long ti[SB];
    const long long i = (long long)blockIdx.x * SB + threadIdx.x;
    const bool live = i < v.n;
    const long long M = (long long)count[0];
    double xi = 0., yi = 0., zi = 0.;
    if (live) { xi = v.x[i]; yi = v.y[i]; zi = v.z[i]; }
    double phi = 0.;
    for (long long j0 = 0; j0 < M; j0 += SB) {
        const long long j = j0 + threadIdx.x;
        __syncthreads();
        if (j < M) { tx[threadIdx.x] = cx[j]; ty[threadIdx.x] = cy[j]; tz[threadIdx.x] = cz[j]; tg[threadIdx.x] = cgm[j]; ti[threadIdx.x] = cidx[j]; }
        __syncthreads();
        const int cnt = (int)((M - j0) < SB ? (M - j0) : SB);
        for (int k = 0; k < cnt; k++) {
            if (ti[k] == i) continue;
            {
#pragma clang fp contract(off)
                const double dx = xi - tx[k], dy = yi - ty[k], dz = zi - tz[k];
                phi += tg[k] / sqrt(dx * dx + dy * dy + dz * dz);
            }
        }
    }
    if (live) pot[i] = phi;
}

// per-workgroup sums of (kinetic, Newtonian potential, post-Newtonian potential * c^2)
__global__ __launch_bounds__(SB) void gr_hamiltonian_main(View v, double G, double C2, const double* pot, const double* blocksums,
                                                         const double* totals, double* partials) {
    __shared__ double lds[SW][7];
    __shared__ double lds3[SW][3];
    double carry[7] = {0., 0., 0., 0., 0., 0., 0.};
    double e[3] = {0., 0., 0.};
    for (int t = 0; t < ST; t++) {
        const long long i = (long long)blockIdx.x * SCHUNK + t * SB + threadIdx.x;
        const bool live = i < v.n;
        double q[7], own[7], tot[7];
#pragma unroll
        for (int c = 0; c < 7; c++) q[c] = 0.;
        if (live) weighted_state<7>(v, nullptr, nullptr, nullptr, i, q);
#pragma unroll
        for (int c = 0; c < 7; c++) own[c] = q[c];
        block_scan<7, false>(q, tot, lds);
        if (live) {
#pragma clang fp contract(off)
            const double mi = own[0];
            e[1] += -0.5 * mi * pot[i];                    // each pair is seen from both ends
            if (i == 0) {
                // Jacobi "body" 0 is the centre of mass carrying the total mass (gr.c:219)
                const double M = totals[0], iM = 1. / M;
                const double vx = totals[4] * iM, vy = totals[5] * iM, vz = totals[6] * iM;
                e[0] += 0.5 * M * (vx * vx + vy * vy + vz * vz);
            } else {
                double S[7];
#pragma unroll
                for (int c = 0; c < 7; c++) S[c] = blocksums[(long long)blockIdx.x * 7 + c] + carry[c] + (q[c] - own[c]);
                const double eta = S[0], ei = 1. / eta;
                const double mj = mi * eta / (eta + mi);
                const double x = v.x[i] - S[1] * ei, y = v.y[i] - S[2] * ei, z = v.z[i] - S[3] * ei;
                const double vx = v.vx[i] - S[4] * ei, vy = v.vy[i] - S[5] * ei, vz = v.vz[i] - S[6] * ei;
                const double mu = G * v.m[0];
                const double rdoti2 = vx * vx + vy * vy + vz * vz;
                double vt2 = rdoti2;
                const double ri = sqrt(x * x + y * y + z * z);
                const double vscale2 = mu / ri;
                for (int it = 0; it < 10; it++) {
                    const double old = vt2;
                    const double A = (0.5 * vt2 + 3. * vscale2) / C2;
                    vt2 = rdoti2 / ((1. - A) * (1. - A));
                    if ((vt2 - old) / vt2 < 2.220446049250313e-16) break;
                }
                e[2] += mj * (0.5 * mu * mu / (ri * ri) - 0.125 * vt2 * vt2 - 1.5 * mu * vt2 / ri);
                e[0] += 0.5 * mj * vt2;
            }
        }
#pragma unroll
        for (int c = 0; c < 7; c++) carry[c] += tot[c];
    }
    block_total<3>(e, lds3);
    if (threadIdx.x == 0) {
        partials[3 * (long long)blockIdx.x + 0] = e[0];
        partials[3 * (long long)blockIdx.x + 1] = e[1];
        partials[3 * (long long)blockIdx.x + 2] = e[2];
    }
}

// ------------------------------------------------------------------ host driver
int scan_segment_bodies() { return 64 * ST; }
size_t scan_workspace_doubles(long long n) {
    const long long nb = (n + SB - 1) / SB;
    const long long nbs = (n + SCHUNK - 1) / SCHUNK;
    // 7 body columns (newton 3, aJ 3, reaction share 1) + compact list (4 doubles + 1 index) + workgroup sums (10 + 3 + 1) + totals
    // + wavefront sums of the weighted-sum passes, running totals inside the segments and the table per part
    return (size_t)(12 * ((n + 1) & ~1LL) + 14 * nb + 64 + (13 + 4 * SSUB) * SW * nbs);
}

hipError_t launch_scan_op(const View& v, const DevOp& op, double G, bool exact, double* ws, unsigned* status, hipStream_t st, ScanTail* tail) {
    const long long n = v.n;
    if (n <= 0) return hipSuccess;
    const long long nb = (n + SB - 1) / SB;
    const long long ns = (n + 1) & ~1LL;                    // every body column of the workspace starts 16-byte aligned
    double* nax = ws; double* nay = nax + ns; double* naz = nay + ns;
    double* ajx = naz + ns; double* ajy = ajx + ns; double* ajz = ajy + ns;
    double* wf = ajz + ns;
    double* cx = wf + ns; double* cy = cx + ns; double* cz = cy + ns; double* cgm = cz + ns;
    long long* cidx = reinterpret_cast<long long*>(cgm + ns);
    double* bs = reinterpret_cast<double*>(cidx + ns);     // nb * 10
    double* wsum = bs + 10 * nb;                            // nb * 3
    double* fl = wsum + 3 * nb;                             // nb
    double* totals = fl + nb;                               // 16
    double* wtot = totals + 16;                             // 3
    double* count = wtot + 3;                               // 1
    const long long nbs = (n + SCHUNK - 1) / SCHUNK;        // workgroups of the weighted-sum passes
    double* wavesums = count + 8;                           // nbs * SW * 10
    double* wwave = wavesums + 10 * SW * nbs;               // nbs * SW * 3
    double* wrun = tail != nullptr ? wwave + 3 * SW * nbs : nullptr;            // nbs * SW * (SSUB - 1) * 3
    double* part_mass = tail != nullptr ? wwave + 3 * SSUB * SW * nbs : nullptr;            // nbs * SW * SSUB
    const dim3 g((unsigned)nb), gs((unsigned)nbs), b(SB);
    int kind = SCAN_GR;
    if (op.kind == REBXAMD_OP_MOF) kind = (op.iarg == REBX_COORDINATES_BARYCENTRIC) ? SCAN_MOF_BARY : SCAN_MOF_JACOBI;
    if (kind == SCAN_GR) {
        hipLaunchKernelGGL(massive_flag_totals, g, b, 0, st, v, fl);
        hipLaunchKernelGGL((scan_of_totals<1, false>), dim3(1), dim3(SOT), 0, st, fl, nb, count);
        hipLaunchKernelGGL(massive_compact, g, b, 0, st, v, fl, cx, cy, cz, cgm, cidx, G);
        if (exact) hipLaunchKernelGGL(newton_from_compact<true>, g, b, 0, st, v, cx, cy, cz, cgm, cidx, count, nax, nay, naz);
        else hipLaunchKernelGGL(newton_from_compact<false>, g, b, 0, st, v, cx, cy, cz, cgm, cidx, count, nax, nay, naz);
        hipLaunchKernelGGL(scan_block_totals<10>, gs, b, 0, st, v, nax, nay, naz, bs, wavesums);
        hipLaunchKernelGGL((scan_of_totals<10, false>), dim3(10), dim3(SOT), 0, st, bs, nbs, totals);
        if (exact) hipLaunchKernelGGL((scan_main<SCAN_GR, true>), gs, b, 0, st, v, op, G, nax, nay, naz, bs, wavesums, totals, ajx, ajy, ajz, wf, wsum, wwave, status, (const double*)nullptr, 1LL, v.m, wrun, part_mass);
        else hipLaunchKernelGGL((scan_main<SCAN_GR, false>), gs, b, 0, st, v, op, G, nax, nay, naz, bs, wavesums, totals, ajx, ajy, ajz, wf, wsum, wwave, status, (const double*)nullptr, 1LL, v.m, wrun, part_mass);
    } else {
        hipLaunchKernelGGL(scan_block_totals<7>, gs, b, 0, st, v, nax, nay, naz, bs, wavesums);
        hipLaunchKernelGGL((scan_of_totals<7, false>), dim3(7), dim3(SOT), 0, st, bs, nbs, totals);
        if (kind == SCAN_MOF_JACOBI) {
            if (exact) hipLaunchKernelGGL((scan_main<SCAN_MOF_JACOBI, true>), gs, b, 0, st, v, op, G, nax, nay, naz, bs, wavesums, totals, ajx, ajy, ajz, wf, wsum, wwave, status, (const double*)nullptr, 1LL, v.m, wrun, part_mass);
            else if (tail != nullptr)
                hipLaunchKernelGGL(mof_jacobi_pairs, gs, b, 0, st, v, (const double*)bs, (const double*)wavesums, ajx, ajy, ajz, wsum, wwave, wrun, part_mass, (const double*)nullptr, 1LL);
            else hipLaunchKernelGGL((scan_main<SCAN_MOF_JACOBI, false>), gs, b, 0, st, v, op, G, nax, nay, naz, bs, wavesums, totals, ajx, ajy, ajz, wf, wsum, wwave, status, (const double*)nullptr, 1LL, v.m, wrun, part_mass);
        } else {
            if (exact) hipLaunchKernelGGL((scan_main<SCAN_MOF_BARY, true>), gs, b, 0, st, v, op, G, nax, nay, naz, bs, wavesums, totals, ajx, ajy, ajz, wf, wsum, wwave, status, (const double*)nullptr, 1LL, v.m, wrun, part_mass);
            else hipLaunchKernelGGL((scan_main<SCAN_MOF_BARY, false>), gs, b, 0, st, v, op, G, nax, nay, naz, bs, wavesums, totals, ajx, ajy, ajz, wf, wsum, wwave, status, (const double*)nullptr, 1LL, v.m, wrun, part_mass);
        }
    }
    hipLaunchKernelGGL((scan_of_totals<3, true>), dim3(3), dim3(SOT), 0, st, wsum, nbs, wtot);
    if (tail != nullptr) {
        // the caller's streaming pass applies the force
        *tail = ScanTail{{ajx, ajy, ajz}, part_mass, wsum, wwave, wrun, kind == SCAN_MOF_BARY ? wtot : nullptr, 64 * ST};
        return hipGetLastError();
    }
    if (kind == SCAN_MOF_BARY) hipLaunchKernelGGL(scan_apply<true>, gs, b, 0, st, v, ajx, ajy, ajz, wf, wsum, wwave, wtot, (const double*)nullptr);
    else hipLaunchKernelGGL(scan_apply<false>, gs, b, 0, st, v, ajx, ajy, ajz, wf, wsum, wwave, wtot, (const double*)nullptr);
    return hipGetLastError();
}

}  // namespace rebxamd

namespace rebxamd {

// ---- pieces of the Jacobi/barycentric modify_orbits_forces for particle shards.  `v` is the
// sub-view of one contiguous range (column pointers already offset, n = its length); `ws` is the
// workspace slice that belongs to the same range (6 columns of v.n doubles, then block sums).
static inline long long blocks_of(long long n) { return (n + SCHUNK - 1) / SCHUNK; }
// (gr: 9 columns; per weighted-sum workgroup 10 + 3 sums and SW times as many wavefront sums; one flag total per
// compaction workgroup)
size_t scan_range_doubles(long long n) { return (size_t)(7 * n + 13 * (1 + SW) * blocks_of(n) + (n + SB - 1) / SB + 32); }

// sums7 <- (m, m x, m y, m z, m vx, m vy, m vz) summed over the range
hipError_t launch_range_sums(const View& v, double* ws, double* sums7, hipStream_t st) {
    if (v.n <= 0) return hipMemsetAsync(sums7, 0, 7 * sizeof(double), st);
    const long long nb = blocks_of(v.n);
    double* bs = ws + 6 * v.n;
    hipLaunchKernelGGL(scan_block_totals<7>, dim3((unsigned)nb), dim3(SB), 0, st, v, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, bs,
                       (double*)nullptr);
    hipLaunchKernelGGL((scan_of_totals<7, false>), dim3(7), dim3(SOT), 0, st, bs, nb, sums7);
    return hipGetLastError();
}

// own forces and weighted reactions of the range (kept in ws); push3 <- sum of the reactions.
// before7: Jacobi = sums of every body that precedes the range; barycentric = sums of ALL bodies.
hipError_t launch_range_mof_forces(const View& v, const DevOp& op, double G, bool barycentric, bool starts_at_origin, const double* before7,
                                   double* ws, double* push3, hipStream_t st) {
    if (v.n <= 0) return hipMemsetAsync(push3, 0, 3 * sizeof(double), st);
    const long long n = v.n, nb = blocks_of(n);
    double* ajx = ws; double* ajy = ajx + n; double* ajz = ajy + n; double* wf = ajz + n;
    double* bs = wf + n;                    // nb * 7
    double* wsum = bs + 7 * nb;             // nb * 3
    double* scratch = wsum + 3 * nb;        // 16
    double* wavesums = scratch + 16;        // nb * SW * 7
    double* wwave = wavesums + 7 * SW * nb; // nb * SW * 3
    const dim3 g((unsigned)nb), b(SB);
    const double* none = nullptr;
    hipLaunchKernelGGL(scan_block_totals<7>, g, b, 0, st, v, none, none, none, bs, wavesums);
    hipLaunchKernelGGL((scan_of_totals<7, false>), dim3(7), dim3(SOT), 0, st, bs, nb, scratch);
    if (barycentric)
        hipLaunchKernelGGL((scan_main<SCAN_MOF_BARY, true>), g, b, 0, st, v, op, G, none, none, none, (const double*)bs, (const double*)wavesums, before7, ajx, ajy, ajz, wf, wsum,
                           wwave, (unsigned*)nullptr, none, 0LL, v.m);
    else
        hipLaunchKernelGGL((scan_main<SCAN_MOF_JACOBI, true>), g, b, 0, st, v, op, G, none, none, none, (const double*)bs, (const double*)wavesums, (const double*)scratch, ajx, ajy, ajz, wf, wsum,
                           wwave, (unsigned*)nullptr, before7, starts_at_origin ? 1LL : 0LL, v.m);
    hipLaunchKernelGGL((scan_of_totals<3, true>), dim3(3), dim3(SOT), 0, st, wsum, nb, push3);
    return hipGetLastError();
}

// a_i += own_i - (reactions of the bodies from i on).  after3: Jacobi = reactions of every body
// that follows the range; barycentric = reactions of ALL bodies.
hipError_t launch_range_mof_apply(const View& v, bool barycentric, const double* after3, double* ws, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    const long long n = v.n, nb = blocks_of(n);
    double* ajx = ws; double* ajy = ajx + n; double* ajz = ajy + n; double* wf = ajz + n;
    double* wsum = wf + n + 7 * nb;
    double* wwave = wsum + 3 * nb + 16 + 7 * SW * nb;
    const dim3 g((unsigned)nb), b(SB);
    if (barycentric) hipLaunchKernelGGL(scan_apply<true>, g, b, 0, st, v, ajx, ajy, ajz, wf, (const double*)wsum, (const double*)wwave, after3, (const double*)nullptr);
    else hipLaunchKernelGGL(scan_apply<false>, g, b, 0, st, v, ajx, ajy, ajz, wf, (const double*)wsum, (const double*)wwave, (const double*)nullptr, after3);
    return hipGetLastError();
}

// ---- pieces of gr for particle shards.  Range workspace: Newtonian acc (3 columns), own
// post-Newtonian acc (3), weighted reactions (3), workgroup sums (10 + 3 per workgroup), flag totals.
struct GrRange { double *nax, *nay, *naz, *ajx, *ajy, *ajz, *wf, *bs, *wsum, *fl, *scratch, *wavesums, *wwave; };
static GrRange gr_range(double* ws, long long n) {
    GrRange r;
    const long long nb = blocks_of(n);
    r.nax = ws; r.nay = r.nax + n; r.naz = r.nay + n; r.ajx = r.naz + n; r.ajy = r.ajx + n; r.ajz = r.ajy + n;
    r.wf = r.ajz + n;
    r.bs = r.wf + n; r.wsum = r.bs + 10 * nb; r.fl = r.wsum + 3 * nb; r.scratch = r.fl + (n + SB - 1) / SB;
    r.wavesums = r.scratch + 32; r.wwave = r.wavesums + 10 * SW * nb;
    return r;
}

// rec <- records {x, y, z, G m, id0 + index} of the massive bodies of the range, in index order; count <- how many
hipError_t launch_range_massive(const View& v, double G, double id0, double* ws, double* rec, double* count, hipStream_t st) {
    if (v.n <= 0) return hipMemsetAsync(count, 0, sizeof(double), st);
    const GrRange r = gr_range(ws, v.n);
    const long long nb = (v.n + SB - 1) / SB;
    hipLaunchKernelGGL(massive_flag_totals, dim3((unsigned)nb), dim3(SB), 0, st, v, r.fl);
    hipLaunchKernelGGL((scan_of_totals<1, false>), dim3(1), dim3(SOT), 0, st, r.fl, nb, count);
    hipLaunchKernelGGL(massive_records, dim3((unsigned)nb), dim3(SB), 0, st, v, (const double*)r.fl, rec, G, id0);
    return hipGetLastError();
}

// Newtonian pull of the M gathered records on the bodies of the range (kept in ws), then
// sums10 <- (m, m x, m v, m a_newton) summed over the range
hipError_t launch_range_gr_newton(const View& v, const double* rec, long long M, double id0, bool exact, double* ws, double* sums10, hipStream_t st) {
    if (v.n <= 0) return hipMemsetAsync(sums10, 0, 10 * sizeof(double), st);
    const GrRange r = gr_range(ws, v.n);
    const dim3 g((unsigned)((v.n + SB - 1) / SB)), b(SB);
    if (exact) hipLaunchKernelGGL(newton_from_records<true>, g, b, 0, st, v, rec, M, id0, r.nax, r.nay, r.naz);
    else hipLaunchKernelGGL(newton_from_records<false>, g, b, 0, st, v, rec, M, id0, r.nax, r.nay, r.naz);
    const long long nb = blocks_of(v.n);
    hipLaunchKernelGGL(scan_block_totals<10>, dim3((unsigned)nb), b, 0, st, v, (const double*)r.nax, (const double*)r.nay, (const double*)r.naz, r.bs, r.wavesums);
    hipLaunchKernelGGL((scan_of_totals<10, false>), dim3(10), dim3(SOT), 0, st, r.bs, nb, sums10);
    return hipGetLastError();
}

// post-Newtonian term of every body of the range in its Jacobi frame and its weighted reaction (kept
// in ws); push3 <- sum of the reactions.  before10: sums of every body that precedes the range
// (nullptr for the range that starts at body 0); m0: mass of body 0 of the whole system.
// Runs after launch_range_gr_newton, which leaves the exclusive workgroup sums in ws.
hipError_t launch_range_gr_forces(const View& v, const DevOp& op, double G, bool starts_at_origin, const double* before10, const double* m0,
                                  bool exact, double* ws, double* push3, unsigned* status, hipStream_t st) {
    if (v.n <= 0) return hipMemsetAsync(push3, 0, 3 * sizeof(double), st);
    const GrRange r = gr_range(ws, v.n);
    const long long nb = blocks_of(v.n);
    const dim3 g((unsigned)nb), b(SB);
    const long long first = starts_at_origin ? 1LL : 0LL;
    if (exact) hipLaunchKernelGGL((scan_main<SCAN_GR, true>), g, b, 0, st, v, op, G, (const double*)r.nax, (const double*)r.nay, (const double*)r.naz, (const double*)r.bs,
                                  (const double*)r.wavesums, (const double*)r.scratch, r.ajx, r.ajy, r.ajz, r.wf, r.wsum, r.wwave, status, before10, first, m0);
    else hipLaunchKernelGGL((scan_main<SCAN_GR, false>), g, b, 0, st, v, op, G, (const double*)r.nax, (const double*)r.nay, (const double*)r.naz, (const double*)r.bs,
                            (const double*)r.wavesums, (const double*)r.scratch, r.ajx, r.ajy, r.ajz, r.wf, r.wsum, r.wwave, status, before10, first, m0);
    hipLaunchKernelGGL((scan_of_totals<3, true>), dim3(3), dim3(SOT), 0, st, r.wsum, nb, push3);
    return hipGetLastError();
}

// a_i += own_i - (reactions of the bodies from i on); after3 = reactions of every body that follows the range
hipError_t launch_range_gr_apply(const View& v, const double* after3, double* ws, hipStream_t st) {
    if (v.n <= 0) return hipSuccess;
    const GrRange r = gr_range(ws, v.n);
    hipLaunchKernelGGL(scan_apply<false>, dim3((unsigned)blocks_of(v.n)), dim3(SB), 0, st, v, (const double*)r.ajx, (const double*)r.ajy, (const double*)r.ajz,
                       (const double*)r.wf, (const double*)r.wsum, (const double*)r.wwave, (const double*)nullptr, after3);
    return hipGetLastError();
}

// out (device, 3 doubles): kinetic, Newtonian potential, post-Newtonian potential times c^2
hipError_t launch_gr_hamiltonian(const View& v, double G, double c, double* ws, double* out, hipStream_t st) {
    const long long n = v.n;
    if (n <= 0) return hipSuccess;
    const long long nb = (n + SB - 1) / SB;
    double* pot = ws;
    double* cx = ws + 7 * n; double* cy = cx + n; double* cz = cy + n; double* cgm = cz + n;
    long long* cidx = reinterpret_cast<long long*>(cgm + n);
    double* bs = reinterpret_cast<double*>(cidx + n);
    double* part = bs + 10 * nb;                            // nb * 3
    double* fl = part + 3 * nb;
    double* totals = fl + nb;
    double* count = totals + 16 + 3;
    const long long nbs = (n + SCHUNK - 1) / SCHUNK;
    const dim3 g((unsigned)nb), gs((unsigned)nbs), b(SB);
    hipLaunchKernelGGL(massive_flag_totals, g, b, 0, st, v, fl);
    hipLaunchKernelGGL((scan_of_totals<1, false>), dim3(1), dim3(SOT), 0, st, fl, nb, count);
    hipLaunchKernelGGL(massive_compact, g, b, 0, st, v, fl, cx, cy, cz, cgm, cidx, G);
    hipLaunchKernelGGL(newton_pot_from_compact, g, b, 0, st, v, cx, cy, cz, cgm, cidx, count, pot);
    hipLaunchKernelGGL(scan_block_totals<7>, gs, b, 0, st, v, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, bs, (double*)nullptr);
    hipLaunchKernelGGL((scan_of_totals<7, false>), dim3(7), dim3(SOT), 0, st, bs, nbs, totals);
    hipLaunchKernelGGL(gr_hamiltonian_main, gs, b, 0, st, v, G, c * c, pot, bs, totals, part);
    hipLaunchKernelGGL((scan_of_totals<3, false>), dim3(3), dim3(SOT), 0, st, part, nbs, out);
    return hipGetLastError();
}

}  // namespace rebxamd

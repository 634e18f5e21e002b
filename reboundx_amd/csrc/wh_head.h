// wh_head.h -- one Wisdom-Holman step of a short massive head (star + planets) in Jacobi coordinates,
// as device functions for one wavefront: shared by the kernels of step_kernels.hip and by the pass over
// the massless bodies, whose extra workgroup runs the head's next step (stream_kernels.hip).
#pragma once
#include "wave_ops.h"
#include "internal.h"
#include "kepler.h"

namespace rebxamd {

// Star + planets (K <= 64 bodies, one wavefront, lane t = body t) in Jacobi coordinates, the
// democratic-free splitting of Rein & Tamayo (2015, MNRAS 452, 376) with REBOUND's defaults: body
// i >= 1 moves on a Kepler orbit about G*eta_i (eta_i = m_0 + ... + m_i), the centre of mass drifts
// linearly and is never kicked, the pair (0,1) is left out of the direct gravity sum (it is the
// Kepler problem of Jacobi body 1), bodies i >= 2 are kicked with the Jacobi term G eta_i x_i/|x_i|^3.
// The recurrences between inertial and Jacobi coordinates are sequential over the K bodies (lane 0).
// State kept between the steps of a run in `jac`: [6*i .. 6*i+5] Jacobi position and velocity of
// body i (slot 0: the centre of mass), [384 + i] eta_i.
struct WhHeadShared {
    double q[REBXAMD_MAX_HEAD][6];      // inertial state
    double j[REBXAMD_MAX_HEAD][6];      // Jacobi state
    double a[REBXAMD_MAX_HEAD][3];      // inertial accelerations, then Jacobi ones
    double m[REBXAMD_MAX_HEAD], eta[REBXAMD_MAX_HEAD];
};

__device__ __forceinline__ void wh_head_to_jacobi(WhHeadShared& sh, int K) {
#pragma clang fp contract(off)
    double eta = sh.m[0];
    double s[6];
    for (int c = 0; c < 6; c++) s[c] = eta * sh.q[0][c];
    sh.eta[0] = eta;
    for (int i = 1; i < K; i++) {
        const double ei = 1. / eta, mi = sh.m[i];
        for (int c = 0; c < 6; c++) sh.j[i][c] = sh.q[i][c] - s[c] * ei;
        eta += mi;
        sh.eta[i] = eta;
        for (int c = 0; c < 6; c++) s[c] += mi * sh.q[i][c];
    }
    const double Mi = 1. / eta;
    for (int c = 0; c < 6; c++) sh.j[0][c] = s[c] * Mi;
}

__device__ __forceinline__ void wh_head_to_inertial(WhHeadShared& sh, int K) {
#pragma clang fp contract(off)
    double eta = sh.eta[K - 1];
    double s[6];
    for (int c = 0; c < 6; c++) s[c] = sh.j[0][c] * eta;
    for (int i = K - 1; i > 0; i--) {
        const double ei = 1. / eta, mi = sh.m[i];
        for (int c = 0; c < 6; c++) {
            s[c] = (s[c] - mi * sh.j[i][c]) * ei;
            sh.q[i][c] = sh.j[i][c] + s[c];
        }
        eta -= mi;
        for (int c = 0; c < 6; c++) s[c] *= eta;
    }
    const double Mi = 1. / eta;
    for (int c = 0; c < 6; c++) sh.q[0][c] = s[c] * Mi;
}

// Kepler drift of the Jacobi bodies over h (lanes), linear drift of the centre of mass
template <bool EXACT>
__device__ __forceinline__ void wh_head_drift(WhHeadShared& sh, int K, double G, double h, int t) {
    if (t >= 1 && t < K) {
        double q[6];
        for (int c = 0; c < 6; c++) q[c] = sh.j[t][c];
        kepler_dev<EXACT>(q, G * sh.eta[t], h);
        for (int c = 0; c < 6; c++) sh.j[t][c] = q[c];
    } else if (t == 0) {
#pragma clang fp contract(off)
        for (int c = 0; c < 3; c++) sh.j[0][c] += h * sh.j[0][3 + c];
    }
}

__device__ __forceinline__ void wh_head_store(const View& v, WhHeadShared& sh, int K, int t, double* jac) {
    if (t >= K) return;
    const_cast<double*>(v.x)[t] = sh.q[t][0]; const_cast<double*>(v.y)[t] = sh.q[t][1]; const_cast<double*>(v.z)[t] = sh.q[t][2];
    const_cast<double*>(v.vx)[t] = sh.q[t][3]; const_cast<double*>(v.vy)[t] = sh.q[t][4]; const_cast<double*>(v.vz)[t] = sh.q[t][5];
    v.ax[t] = 0.; v.ay[t] = 0.; v.az[t] = 0.;
    for (int c = 0; c < 6; c++) jac[6 * t + c] = sh.j[t][c];
    jac[6 * REBXAMD_MAX_HEAD + t] = sh.eta[t];
}

// record of the frame of the massless bodies behind the head (see wh_kick_drift): the head's centre of mass
// before the drift over h, its position after it, its acceleration, G * (mass of the head)
__device__ __forceinline__ void wh_head_record(const WhHeadShared& sh, int K, double G, double h, const double acc[3], double* rec) {
#pragma clang fp contract(off)
    for (int c = 0; c < 6; c++) rec[c] = sh.j[0][c];
    for (int c = 0; c < 3; c++) { rec[6 + c] = sh.j[0][c] + h * sh.j[0][3 + c]; rec[9 + c] = acc[c]; }
    rec[12] = G * sh.eta[K - 1];
}

// Start of a run: the head's Jacobi state from the columns, the record of its centre of mass for the leading half drift of
// the massless bodies (`lead`), the head's own leading drift over h0, the drifted inertial state back into the columns and the
// Jacobi state into `jac`.  Called by the 64 lanes of one wavefront, t = lane.
template <bool EXACT>
__device__ __forceinline__ void wh_head_begin_body(const View& v, int K, double G, double h0, double* jac, double* lead, WhHeadShared& sh, int t) {
    if (t < K) {
        sh.q[t][0] = v.x[t]; sh.q[t][1] = v.y[t]; sh.q[t][2] = v.z[t]; sh.q[t][3] = v.vx[t]; sh.q[t][4] = v.vy[t]; sh.q[t][5] = v.vz[t];
        sh.m[t] = v.m[t];
    }
    wave_sync();
    if (t == 0) {
        wh_head_to_jacobi(sh, K);
        const double none[3] = {0., 0., 0.};
        wh_head_record(sh, K, G, h0, none, lead);
    }
    wave_sync();
    wh_head_drift<EXACT>(sh, K, G, h0, t);
    wave_sync();
    if (t == 0) wh_head_to_inertial(sh, K);
    wave_sync();
    wh_head_store(v, sh, K, t, jac);
}

// The head's step after the head pass has left the additional accelerations of the head bodies in their columns: gravity among
// the head bodies, Jacobi kick over dtk, record of the centre of mass in `rec`, drift over h, new inertial state into the columns
// (accelerations cleared), Jacobi state into `jac`.  Called by the 64 lanes of one wavefront, t = lane.
// pre != nullptr: the head's Jacobi state after the kick and before the drift is kept there (layout of `jac`): what a run
// left unsynchronised between steps needs to close with a half drift where this step made a whole one.
template <bool EXACT>
__device__ __forceinline__ void wh_head_step_body(const View& v, int K, double G, double dtk, double h, double* jac, double* rec, WhHeadShared& sh, int t,
                                                  double* pre = nullptr) {
    if (t < K) {
        sh.q[t][0] = v.x[t]; sh.q[t][1] = v.y[t]; sh.q[t][2] = v.z[t]; sh.q[t][3] = v.vx[t]; sh.q[t][4] = v.vy[t]; sh.q[t][5] = v.vz[t];
        sh.m[t] = v.m[t];
        sh.eta[t] = jac[6 * REBXAMD_MAX_HEAD + t];
        for (int c = 0; c < 6; c++) sh.j[t][c] = jac[6 * t + c];
    }
    wave_sync();
    if (t < K) {
#pragma clang fp contract(off)
        // direct gravity among the head bodies, partners ascending, without the pair (0,1); on top of it what the head
        // pass has left in the columns (the additional forces)
        double ax = 0., ay = 0., az = 0.;
        for (int u = 0; u < K; u++) {
            if (u == t || (t == 0 && u == 1) || (t == 1 && u == 0) || sh.m[u] == 0.) continue;
            const double dx = sh.q[t][0] - sh.q[u][0], dy = sh.q[t][1] - sh.q[u][1], dz = sh.q[t][2] - sh.q[u][2];
            const double r = sqrt(dx * dx + dy * dy + dz * dz);
            const double pre = -G * sh.m[u] / (r * r * r);
            ax += pre * dx; ay += pre * dy; az += pre * dz;
        }
        sh.a[t][0] = ax + v.ax[t]; sh.a[t][1] = ay + v.ay[t]; sh.a[t][2] = az + v.az[t];
    }
    wave_sync();
    if (t == 0) {
#pragma clang fp contract(off)
        // inertial -> Jacobi accelerations; slot 0: acceleration of the centre of mass (what the massless bodies subtract)
        double eta = sh.m[0];
        double s[3] = {eta * sh.a[0][0], eta * sh.a[0][1], eta * sh.a[0][2]};
        for (int i = 1; i < K; i++) {
            const double ei = 1. / eta, mi = sh.m[i];
            const double a0 = sh.a[i][0], a1 = sh.a[i][1], a2 = sh.a[i][2];
            sh.a[i][0] = a0 - s[0] * ei; sh.a[i][1] = a1 - s[1] * ei; sh.a[i][2] = a2 - s[2] * ei;
            eta += mi;
            s[0] += mi * a0; s[1] += mi * a1; s[2] += mi * a2;
        }
        const double Mi = 1. / eta;
        const double acom[3] = {s[0] * Mi, s[1] * Mi, s[2] * Mi};
        wh_head_record(sh, K, G, h, acom, rec);
    }
    wave_sync();
    if (t >= 1 && t < K) {
#pragma clang fp contract(off)
        double* j = sh.j[t];
        j[3] += dtk * sh.a[t][0]; j[4] += dtk * sh.a[t][1]; j[5] += dtk * sh.a[t][2];
        if (t > 1) {
            const double r2i = 1. / (j[0] * j[0] + j[1] * j[1] + j[2] * j[2]);
            const double pre3 = dtk * sqrt(r2i) * r2i * G * sh.eta[t];
            j[3] += pre3 * j[0]; j[4] += pre3 * j[1]; j[5] += pre3 * j[2];
        }
    }
    wave_sync();
    if (pre != nullptr && t < K) {
        for (int c = 0; c < 6; c++) pre[6 * t + c] = sh.j[t][c];
        pre[6 * REBXAMD_MAX_HEAD + t] = sh.eta[t];
    }
    wh_head_drift<EXACT>(sh, K, G, h, t);
    wave_sync();
    if (t == 0) wh_head_to_inertial(sh, K);
    wave_sync();
    wh_head_store(v, sh, K, t, jac);
}

// End of a run left unsynchronised between steps: the head goes back to its state after the kick of the last step (`pre`)
// and drifts over h (half a step) from there, as the last step of a closed run does; rec_close <- the frame record of that
// step (`rec_last`, made for a whole drift) with the frame's position after the half drift.
template <bool EXACT>
__device__ __forceinline__ void wh_head_close_body(const View& v, int K, double G, double h, const double* pre, double* jac, const double* rec_last,
                                                   double* rec_close, WhHeadShared& sh, int t) {
    if (t < K) {
        for (int c = 0; c < 6; c++) sh.j[t][c] = pre[6 * t + c];
        sh.eta[t] = pre[6 * REBXAMD_MAX_HEAD + t];
        sh.m[t] = v.m[t];
    }
    if (t == 0) {
#pragma clang fp contract(off)
        for (int c = 0; c < 6; c++) rec_close[c] = rec_last[c];
        for (int c = 0; c < 3; c++) { rec_close[6 + c] = rec_last[c] + h * rec_last[3 + c]; rec_close[9 + c] = rec_last[9 + c]; }
        rec_close[12] = rec_last[12];
    }
    wave_sync();
    wh_head_drift<EXACT>(sh, K, G, h, t);
    wave_sync();
    if (t == 0) wh_head_to_inertial(sh, K);
    wave_sync();
    wh_head_store(v, sh, K, t, jac);
}


}  // namespace rebxamd

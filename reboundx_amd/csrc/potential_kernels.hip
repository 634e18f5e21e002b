// potential_kernels.hip -- potential energies of the position-dependent forces, the diagnostics
// the reference's conservation tests are built on (reboundx/test/test_conservation.py:38-96):
//   gravitational_harmonics.c:142-231, central_force.c:97-141, tides_precession.c:132-202,
//   gr_potential.c:100-132.
// One pass over the particles: every thread forms its particle's term for each op in the
// reference's expression order (no FMA contraction), workgroups reduce per op, and a second
// kernel adds the workgroup sums in a fixed order.  The reference adds the terms sequentially;
// the results agree to rounding of a sum of that many terms.
#include "internal.h"
#include "wave_ops.h"
#include <float.h>

namespace rebxamd {

static constexpr int PB = 256;

// what one thread holds of its particle (parameter columns: the unset marker = not set)
struct PotBody { double x, y, z, m, Rt, k1; };

__device__ __forceinline__ double potential_term(const DevOp& op, const View& v, long long i, const PotBody& b, double G) {
#pragma clang fp contract(off)
    const int s = op.src;
    if (i == s) return 0.;
    if (op.kind == REBXAMD_OP_GR_POTENTIAL && i < 1) return 0.;
    const double dx = b.x - v.x[s], dy = b.y - v.y[s], dz = b.z - v.z[s];
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double mi = b.m, ms = v.m[s];
    switch (op.kind) {
    case REBXAMD_OP_J2: {
        const double r = sqrt(r2);
        const double costheta2 = dz * dz / r2;
        const double prefac = G * mi * ms * op.b * op.b / r2 / r * op.a;
        const double P2 = 0.5 * (3. * costheta2 - 1.);
        return prefac * P2;
    }
    case REBXAMD_OP_J4: {
        const double r = sqrt(r2);
        const double costheta2 = dz * dz / r2;
        const double prefac = G * mi * ms * op.b * op.b * op.b * op.b / r2 / r2 / r * op.a;
        const double P4 = (35. * costheta2 * costheta2 - 30. * costheta2 + 3.) / 8.;
        return prefac * P4;
    }
    case REBXAMD_OP_CENTRAL: {
        const double A = op.a, gamma = op.b;
        if (fabs(gamma + 1.) < DBL_EPSILON) return -(mi * A * log(sqrt(r2)));
        return -(mi * A * pow(r2, (gamma + 1.) / 2.) / (gamma + 1.));
    }
    case REBXAMD_OP_TIDES: {
        const double mratio = mi / ms;
        if (mratio < DBL_MIN) return 0.;
        const double R0 = (v.Rt && is_set(v.Rt[s])) ? v.Rt[s] : 0.;
        const double k10 = (v.k1 && is_set(v.k1[s])) ? v.k1[s] : 0.;
        const double fac0 = k10 * R0 * R0 * R0 * R0 * R0;
        double fac = 0.;
        fac += fac0 * mratio;
        const double Rp = is_set(b.Rt) ? b.Rt : 0.;
        const double k1p = is_set(b.k1) ? b.k1 : 0.;
        fac += k1p * Rp * Rp * Rp * Rp * Rp / mratio;
        return -3. / 6. * G * (ms + mi) * mi / (r2 * r2 * r2) * fac;
    }
    case REBXAMD_OP_GR_POTENTIAL: {
        const double mu = G * ms;
        const double prefac = 3. * mu * mu / op.c;      // op.c = c^2
        return -(prefac * mi / r2);
    }
    default:
        return 0.;
    }
}

__global__ __launch_bounds__(PB) void potential_kernel(View v, Program prog, double* partials) {
    __shared__ double red[REBXAMD_MAX_OPS][PB / 64];
    const long long i = (long long)blockIdx.x * PB + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool in = i < v.n;
    // the particle is read once for all ops (read-once columns: nontemporal)
    const double nan = unset_value();
    PotBody b = {0., 0., 0., 0., nan, nan};
    if (in) {
        b.x = __builtin_nontemporal_load(v.x + i); b.y = __builtin_nontemporal_load(v.y + i); b.z = __builtin_nontemporal_load(v.z + i);
        b.m = __builtin_nontemporal_load(v.m + i);
        if (prog.need_tides && v.Rt) b.Rt = __builtin_nontemporal_load(v.Rt + i);
        if (prog.need_tides && v.k1) b.k1 = __builtin_nontemporal_load(v.k1 + i);
    }
    for (int k = 0; k < prog.nops; k++) {
        const double t = wave_total(in ? potential_term(prog.ops[k], v, i, b, prog.G) : 0., lane);
        if (lane == 0) red[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < prog.nops) {
        double s = 0.;
        for (int w = 0; w < PB / 64; w++) s += red[threadIdx.x][w];
        partials[(long long)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(1024) void potential_sum_kernel(const double* partials, long long nb, double* out) {
    __shared__ double red[16];
    const double* p = partials + (long long)blockIdx.x * nb;
    double a = 0.;
    for (long long k = threadIdx.x; k < nb; k += 1024) a += p[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.;
        for (int w = 0; w < 16; w++) s += red[w];
        out[blockIdx.x] = s;
    }
}

long long potential_block_count(long long n) { return (n + PB - 1) / PB; }

// partials: nops * potential_block_count(n) doubles; out: nops doubles (device)
hipError_t launch_potentials(const View& v, const Program& prog, double* partials, double* out, hipStream_t st) {
    if (prog.nops <= 0) return hipSuccess;
    const long long nb = potential_block_count(v.n);
    if (nb > 0) hipLaunchKernelGGL(potential_kernel, dim3((unsigned)nb), dim3(PB), 0, st, v, prog, partials);
    hipLaunchKernelGGL(potential_sum_kernel, dim3((unsigned)prog.nops), dim3(1024), 0, st, partials, nb, out);
    return hipGetLastError();
}

}  // namespace rebxamd

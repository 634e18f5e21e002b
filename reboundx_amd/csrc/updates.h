// updates.h -- per-component velocity updates of the force integrators (device code), shared by the
// elementwise kernels and by the streaming pass that carries a whole update.
#pragma once

namespace rebxamd {

// v + c*a in the mode's arithmetic
template <bool EXACT>
__device__ __forceinline__ double axpy1(double v, double c, double a) {
    if (EXACT) {
#pragma clang fp contract(off)
        return v + c * a;
    } else {
        return fma(c, a, v);
    }
}
// integrator_rk2.c:61-66:  v + (b1*a1 + b2*a2)
template <bool EXACT>
__device__ __forceinline__ double rk2_combine(double v, double b1, double a1, double b2, double a2) {
    if (EXACT) {
#pragma clang fp contract(off)
        v += b1 * a1 + b2 * a2;
        return v;
    } else {
        v += b1 * a1 + b2 * a2;
        return v;
    }
}
// integrator_rk4.c:85-90:  v + dt6*(a1 + a4 + 2.*(k2+k3))
template <bool EXACT>
__device__ __forceinline__ double rk4_combine(double v, double dt6, double a1, double a4, double k23) {
    if (EXACT) {
#pragma clang fp contract(off)
        v += dt6 * (a1 + a4 + 2. * k23);
        return v;
    } else {
        v += dt6 * (a1 + a4 + 2. * k23);
        return v;
    }
}

}  // namespace rebxamd

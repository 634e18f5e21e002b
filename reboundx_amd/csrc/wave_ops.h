// wave_ops.h -- wavefront-wide scans and sums of doubles without the LDS crossbar (device code).
#pragma once

namespace rebxamd {

// Wavefront scans.  Within each row of 16 lanes: four data-parallel-primitive shifts (row_shr /
// row_shl by 1, 2, 4, 8; lanes the shift leaves empty read zero), which run at VALU rate instead
// of going through the LDS crossbar; across the four rows: row broadcasts (row_bcast:15, row_bcast:31)
// for the prefix scan, v_readlane of the row totals for the suffix scan (no broadcast runs towards
// lower rows).
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double dpp_shift(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROWS, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROWS, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
    v += dpp_shift<0x111>(v);       // row_shr:1
    v += dpp_shift<0x112>(v);       // row_shr:2
    v += dpp_shift<0x114>(v);       // row_shr:4
    v += dpp_shift<0x118>(v);       // row_shr:8
    // across the rows, forwards: the last lane of a row broadcast to the row behind it (rows 1 and
    // 3 take the totals of rows 0 and 2), then lane 31 to rows 2 and 3; rows outside the mask read zero
    (void)lane;
    v += dpp_shift<0x142, 0xa>(v);  // row_bcast:15 row_mask:0xa
    v += dpp_shift<0x143, 0xc>(v);  // row_bcast:31 row_mask:0xc
    return v;
}
__device__ __forceinline__ double wave_incl_scan_rev(double v, int lane) {
    v += dpp_shift<0x101>(v);       // row_shl:1
    v += dpp_shift<0x102>(v);       // row_shl:2
    v += dpp_shift<0x104>(v);       // row_shl:4
    v += dpp_shift<0x108>(v);       // row_shl:8
    const int row = lane >> 4;
    const double t1 = lane_value(v, 16), t3 = lane_value(v, 48);
    v += (row == 0) ? t1 : ((row == 2) ? t3 : 0.);
    const double t23 = lane_value(v, 32);
    v += (row <= 1) ? t23 : 0.;
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load
// and store the wavefront has in flight, which turns a loop with requests issued ahead into a chain
// of memory round trips; use this one when the barrier protects LDS contents alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Barrier of code that runs as ONE wavefront (the head passes: 64 lanes, lane = head body).  It orders the wavefront's
// LDS and global traffic exactly as __syncthreads() does -- a workgroup-scope fence, so every lane's writes have landed
// before any lane reads them -- but leaves out the workgroup barrier instruction.  One wavefront needs none (its lanes
// run in lockstep), and the head code also runs in the extra workgroup of a stepping pass after the other wavefronts of
// that workgroup have returned, where a workgroup barrier that not every thread reaches would be undefined behaviour.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Sum over the wavefront (the same value in every lane).
__device__ __forceinline__ double wave_total(double v, int lane) { return lane_value(wave_incl_scan(v, lane), 63); }

}  // namespace rebxamd

// scan_ops.h -- building blocks of the prefix/suffix-sum passes: workgroup scans and totals of C
// doubles per thread, and the single-workgroup scan of a table of workgroup totals.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include "wave_ops.h"

namespace rebxamd {

static constexpr int SB = 256;          // elements per scan workgroup (one per thread)
static constexpr int SW = SB / 64;
static constexpr int ST = 8;            // tiles of SB elements per workgroup in the weighted-sum passes: short lists of workgroup totals
static constexpr int SCHUNK = SB * ST;
static constexpr int SSUB = 4;          // parts of a wavefront segment at whose ends the per-body pass can leave running totals (ST / SSUB tiles each)
static_assert(ST % SSUB == 0, "whole tiles per part");

// Workgroup-wide scan of C values per thread.  REVERSE: suffix instead of prefix.
// On return v[c] holds the INCLUSIVE scan and tot[c] the workgroup total.
template <int C, bool REVERSE>
__device__ __forceinline__ void block_scan(double (&v)[C], double (&tot)[C], double (*lds)[C]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < C; c++) v[c] = REVERSE ? wave_incl_scan_rev(v[c], lane) : wave_incl_scan(v[c], lane);
    __syncthreads();
    if (lane == (REVERSE ? 0 : 63)) {
#pragma unroll
        for (int c = 0; c < C; c++) lds[wave][c] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++) {
        double before = 0., all = 0.;
        for (int w = 0; w < SW; w++) {
            const double t = lds[w][c];
            all += t;
            if (REVERSE ? (w > wave) : (w < wave)) before += t;
        }
        v[c] += before;
        tot[c] = all;
    }
}

// Workgroup total of C per-thread values (fixed order: lanes by shuffle tree, then waves in order); valid in thread 0.
template <int C>
__device__ __forceinline__ void block_total(double (&v)[C], double (*lds)[C]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; c++) lds[wave][c] = v[c];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < C; c++) { double t = 0.; for (int w = 0; w < SW; w++) t += lds[w][c]; v[c] = t; }
    }
}

static constexpr int SOT = 512;
// pass 2: exclusive scan of the workgroup totals (rows of C doubles), in place; totals[c] = grand total.
// One workgroup per component (grid = C): the scan of a column is a chain of rounds of SOT rows, and
// a round costs instructions in proportion to the components it carries.  A thread owns one row of
// the round; the next round's value is requested before the current one is scanned.
// Scan order: row 0 first, or the last row first when REVERSE.
// Whole rounds run without a branch around their load and store, and the first one stands outside
// the loop: every way into the loop then has the same memory operations behind it, and the wait for
// the requested value does not also wait for the stores issued after the request.
template <int C, bool REVERSE>
__global__ __launch_bounds__(SOT) void scan_of_totals(double* table, long long nb, double* totals) {
    __shared__ double wtot[2][SOT / 64];
    const int c = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long nfull = nb / SOT;
    const bool tail = nfull * SOT < nb;
    // address of this thread's row in round s; where that round has no row for it: its row in the
    // round before, which no other thread touches (the value read there is not used)
    auto at = [&](long long s) -> double* {
        long long r = s * SOT + tid;
        if (r > nb - 1) r -= SOT;
        if (r < 0) r = 0;
        return table + (REVERSE ? nb - 1 - r : r) * C + c;
    };
    double carry = 0., ahead = 0.;
    auto round = [&](long long s, auto whole) {
        const bool live = decltype(whole)::value || s * SOT + tid < nb;
        const double v = live ? ahead : 0.;
        double* const mine = at(s);
        if (decltype(whole)::value) ahead = *at(s + 1);     // (a partial round is the last one)
        const double incl = wave_incl_scan(v, lane);
        double* const wt = wtot[s & 1];                 // two copies: one barrier per round
        if (lane == 63) wt[wave] = incl;
        lds_barrier();
        double before = 0., all = 0.;
#pragma unroll
        for (int w = 0; w < SOT / 64; w++) { const double t = wt[w]; all += t; if (w < wave) before += t; }
        if (live) *mine = carry + before + (incl - v);
        carry += all;
    };
    if (tid < nb) ahead = *at(0);
    if (nfull > 0) round(0, std::true_type());
#pragma unroll 1
    for (long long s = 1; s < nfull; s++) round(s, std::true_type());
    if (tail) round(nfull, std::false_type());
    if (tid == 0) totals[c] = carry;
}

}  // namespace rebxamd

// membench4.hip -- workgroup-to-address mappings for the 7-in 3-rmw fp64 stream at sizes beyond the last-level cache
// (development aid): does it matter which XCD streams which part of a column?
//   linear      workgroup b takes bodies [512 b, 512 (b + 1))
//   xcd         workgroups are dealt round-robin to the 8 XCDs; XCD x streams the x-th eighth of every column
//   xcd-chunk   the same in chunks of C workgroups: XCD x streams C consecutive tiles, then the next group of 8 C tiles
// build: hipcc -O3 --offload-arch=gfx950 scripts/membench4.hip -o /tmp/membench4 ; run: /tmp/membench4 [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
struct Cols { const double* in[7]; double* a[3]; long long n; };
static constexpr int BLOCK = 256;

template <int NIN, bool NT, bool RMW>
__device__ __forceinline__ void tile(const Cols& c, long long i) {
    d2 in[7], a[3];
#pragma unroll
    for (int q = 0; q < NIN; q++) in[q] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(c.in[q] + i)) : *reinterpret_cast<const d2*>(c.in[q] + i);
    if (RMW) {
#pragma unroll
        for (int q = 0; q < 3; q++) a[q] = *reinterpret_cast<const d2*>(c.a[q] + i);
    } else {
        a[0] = a[1] = a[2] = d2{0., 0.};
    }
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int q = 0; q < NIN; q++) { s0 += in[q].x; s1 += in[q].y; }
#pragma unroll 8
    for (int r = 0; r < 100; r++) { s0 = fma(s0, 1.0000001, 1e-9); s1 = fma(s1, 1.0000001, 1e-9); }
#pragma unroll
    for (int q = 0; q < 3; q++) { d2 o; o.x = a[q].x + s0; o.y = a[q].y + s1; *reinterpret_cast<d2*>(c.a[q] + i) = o; }
}

// MAP 0: linear; 1: xcd (chunk = nb / 8); 2: xcd-chunk with `chunk` tiles
template <int NIN, bool NT, bool RMW, int MAP>
__global__ __launch_bounds__(BLOCK, 4) void stream(Cols c, unsigned nb, unsigned chunk) {
    unsigned b = blockIdx.x;
    if (MAP != 0) {
        const unsigned x = b & 7u, k = b >> 3;            // XCD, position in that XCD's sequence
        const unsigned group = k / chunk, within = k % chunk;
        b = (group * 8u + x) * chunk + within;
        if (b >= nb) return;
    }
    const long long i = 2 * ((long long)b * BLOCK + threadIdx.x);
    if (i + 1 >= c.n) return;
    tile<NIN, NT, RMW>(c, i);
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; i++) f();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 100000000;
    Cols c; c.n = n;
    for (int q = 0; q < 7; q++) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); c.in[q] = p; }
    for (int q = 0; q < 3; q++) { CK(hipMalloc(&c.a[q], n * 8)); CK(hipMemset(c.a[q], 0, n * 8)); }
    const unsigned nb = (unsigned)((n / 2 + BLOCK - 1) / BLOCK);
    const unsigned nb8 = (nb + 7) / 8 * 8;
#define RUN(NIN, NT, RMW, MAP, CHUNK, LABEL) { \
        const unsigned ch = (MAP == 1) ? nb8 / 8 : CHUNK; \
        const unsigned grid = (MAP == 0) ? nb : ((nb + 8 * ch - 1) / (8 * ch)) * 8 * ch; \
        float us = timeit([&] { hipLaunchKernelGGL((stream<NIN, NT, RMW, MAP>), dim3(grid), dim3(BLOCK), 0, 0, c, nb, ch); }, 10); \
        const double bytes = (8.0 * NIN + (RMW ? 48.0 : 24.0)) * n; \
        printf("%-46s %9.1f us %7.1f GB/s\n", LABEL, us, bytes / us / 1e3); }
    printf("n = %lld (%u tiles of 512 bodies)\n", n, nb);
    RUN(7, true, true, 0, 1, "7 in (nt) + 3 rmw, linear");
    RUN(7, false, true, 0, 1, "7 in + 3 rmw, linear");
    RUN(7, true, true, 1, 1, "7 in (nt) + 3 rmw, xcd eighths");
    RUN(7, true, true, 2, 1, "7 in (nt) + 3 rmw, xcd-chunk 1 (= linear order)");
    RUN(7, true, true, 2, 16, "7 in (nt) + 3 rmw, xcd-chunk 16");
    RUN(7, true, true, 2, 64, "7 in (nt) + 3 rmw, xcd-chunk 64");
    RUN(7, true, true, 2, 128, "7 in (nt) + 3 rmw, xcd-chunk 128");
    RUN(7, true, true, 2, 256, "7 in (nt) + 3 rmw, xcd-chunk 256");
    RUN(7, true, true, 2, 512, "7 in (nt) + 3 rmw, xcd-chunk 512");
    RUN(7, true, true, 2, 1024, "7 in (nt) + 3 rmw, xcd-chunk 1024");
    RUN(7, true, true, 2, 2048, "7 in (nt) + 3 rmw, xcd-chunk 2048");
    RUN(7, true, true, 2, 4096, "7 in (nt) + 3 rmw, xcd-chunk 4096");
    RUN(7, true, false, 0, 1, "7 in (nt) + 3 write-only, linear");
    RUN(7, true, false, 1, 1, "7 in (nt) + 3 write-only, xcd eighths");
    RUN(3, true, true, 0, 1, "3 in (nt) + 3 rmw, linear");
    RUN(3, true, true, 1, 1, "3 in (nt) + 3 rmw, xcd eighths");
    RUN(1, true, true, 0, 1, "1 in (nt) + 3 rmw, linear");
    return 0;
}

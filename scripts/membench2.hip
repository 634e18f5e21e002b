// membench2.hip -- nontemporal / width / occupancy variants of the 7-in 3-rmw fp64 stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
struct Cols { const double* in[7]; double* a[3]; long long n; };

// W = doubles per lane per column (1 or 2); NTI = nt loads of inputs; NTA = nt loads of acc; NTS = nt stores
template <int W, bool NTI, bool NTA, bool NTS, int VALU, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(Cols c) {
    extern __shared__ double dyn[];
    const long long i = W * ((long long)blockIdx.x * BLOCK + threadIdx.x);
    if (i + W - 1 >= c.n) return;
    double s[W];
    for (int w = 0; w < W; w++) s[w] = 0;
    double a[3][W];
    if (W == 2) {
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const d2* p = reinterpret_cast<const d2*>(c.in[q] + i);
            d2 t = NTI ? __builtin_nontemporal_load(p) : *p;
            s[0] += t.x; s[W - 1] += t.y;
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const d2* p = reinterpret_cast<const d2*>(c.a[q] + i);
            d2 t = NTA ? __builtin_nontemporal_load(p) : *p;
            a[q][0] = t.x; a[q][W - 1] = t.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 7; q++) s[0] += NTI ? __builtin_nontemporal_load(c.in[q] + i) : c.in[q][i];
#pragma unroll
        for (int q = 0; q < 3; q++) a[q][0] = NTA ? __builtin_nontemporal_load(c.a[q] + i) : c.a[q][i];
    }
#pragma unroll 8
    for (int r = 0; r < VALU; r++) for (int w = 0; w < W; w++) s[w] = fma(s[w], 1.0000001, 1e-9);
    if (W == 2) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            d2 t; t.x = a[q][0] + s[0]; t.y = a[q][W - 1] + s[W - 1];
            d2* p = reinterpret_cast<d2*>(c.a[q] + i);
            if (NTS) __builtin_nontemporal_store(t, p); else *p = t;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 3; q++) { if (NTS) __builtin_nontemporal_store(a[q][0] + s[0], c.a[q] + i); else c.a[q][i] = a[q][0] + s[0]; }
    }
    if (dyn[0] == 123.456) c.a[0][0] = 1;
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    const long long n = 10000000;
    Cols c; c.n = n;
    for (int q = 0; q < 7; q++) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); c.in[q] = p; }
    for (int q = 0; q < 3; q++) { CK(hipMalloc(&c.a[q], n * 8)); CK(hipMemset(c.a[q], 0, n * 8)); }
    const double bytes = 104.0 * n;
#define RUN(W, NTI, NTA, NTS, VALU, LDSKB) { float us = timeit([&] { hipLaunchKernelGGL((k<W, NTI, NTA, NTS, VALU, 256>), dim3((n / W + 255) / 256), dim3(256), LDSKB * 1024, 0, c); }, 20); \
    printf("W=%d nt(in=%d acc=%d st=%d) valu=%-3d lds=%2dKB  %7.1f us %7.1f GB/s\n", W, NTI, NTA, NTS, VALU, LDSKB, us, bytes / us / 1e3); }
    RUN(1, 0, 0, 0, 0, 0); RUN(1, 1, 0, 0, 0, 0); RUN(1, 1, 1, 0, 0, 0); RUN(1, 1, 1, 1, 0, 0); RUN(1, 0, 0, 1, 0, 0);
    RUN(2, 0, 0, 0, 0, 0); RUN(2, 1, 0, 0, 0, 0); RUN(2, 1, 1, 0, 0, 0); RUN(2, 1, 1, 1, 0, 0); RUN(2, 0, 0, 1, 0, 0);
    RUN(1, 1, 1, 0, 400, 0); RUN(2, 1, 1, 0, 400, 0); RUN(1, 1, 0, 0, 400, 0); RUN(2, 1, 0, 0, 400, 0);
    RUN(1, 1, 1, 0, 400, 40); RUN(2, 1, 1, 0, 400, 40); RUN(1, 1, 1, 0, 400, 80); RUN(2, 1, 1, 0, 400, 80);
    RUN(1, 1, 1, 0, 0, 80); RUN(2, 1, 1, 0, 0, 80); RUN(2, 1, 1, 0, 0, 40);
    return 0;
}

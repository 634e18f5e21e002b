// membench3.hip -- one-shot vs software-pipelined variants of the 7-in 3-rmw fp64 stream with
// arithmetic between the loads and the stores (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
struct Cols { const double* in[7]; double* a[3]; long long n; };
static constexpr int BLOCK = 256;

struct Tile { d2 in[7]; d2 a[3]; };

__device__ __forceinline__ void load_tile(const Cols& c, long long i, Tile& t) {
#pragma unroll
    for (int q = 0; q < 7; q++) t.in[q] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(c.in[q] + i));
#pragma unroll
    for (int q = 0; q < 3; q++) t.a[q] = *reinterpret_cast<const d2*>(c.a[q] + i);
}
template <int VALU>
__device__ __forceinline__ void work_store(const Cols& c, long long i, const Tile& t) {
    double s0 = 0, s1 = 0, u0 = 1, u1 = 1;
#pragma unroll
    for (int q = 0; q < 7; q++) { s0 += t.in[q].x; s1 += t.in[q].y; }
#pragma unroll 8
    for (int r = 0; r < VALU / 2; r++) { s0 = fma(s0, 1.0000001, 1e-9); u0 = fma(u0, 0.9999999, 1e-9); s1 = fma(s1, 1.0000001, 1e-9); u1 = fma(u1, 0.9999999, 1e-9); }
#pragma unroll
    for (int q = 0; q < 3; q++) { d2 o; o.x = t.a[q].x + s0 * u0; o.y = t.a[q].y + s1 * u1; *reinterpret_cast<d2*>(c.a[q] + i) = o; }
}

template <int VALU, int MINB>
__global__ __launch_bounds__(BLOCK, MINB) void oneshot(Cols c) {
    const long long i = 2 * ((long long)blockIdx.x * BLOCK + threadIdx.x);
    if (i + 1 >= c.n) return;
    Tile t; load_tile(c, i, t);
    work_store<VALU>(c, i, t);
}

// each block walks TILES consecutive tiles of 2*BLOCK particles; the next tile's loads are in flight
// while the current one is worked on
template <int VALU, int MINB, int TILES>
__global__ __launch_bounds__(BLOCK, MINB) void pipelined(Cols c) {
    long long i = 2 * ((long long)blockIdx.x * TILES * BLOCK + threadIdx.x);
    Tile cur, nxt;
    if (i + 1 < c.n) load_tile(c, i, cur);
#pragma unroll 1
    for (int k = 0; k < TILES; k++) {
        const long long in = i + 2 * BLOCK;
        const bool more = (k + 1 < TILES) && (in + 1 < c.n);
        if (more) load_tile(c, in, nxt);
        if (i + 1 < c.n) work_store<VALU>(c, i, cur);
        if (more) cur = nxt;
        i = in;
    }
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

int main(int argc, char** argv) {
    const long long n = argc > 1 ? (long long)atof(argv[1]) : 10000000;  // optional argument: number of particles
    Cols c; c.n = n;
    for (int q = 0; q < 7; q++) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); c.in[q] = p; }
    for (int q = 0; q < 3; q++) { CK(hipMalloc(&c.a[q], n * 8)); CK(hipMemset(c.a[q], 0, n * 8)); }
    const double bytes = 104.0 * n;
    const int nb = (int)((n / 2 + BLOCK - 1) / BLOCK);
#define ONE(VALU, MINB, LDSKB) { float us = timeit([&] { hipLaunchKernelGGL((oneshot<VALU, MINB>), dim3(nb), dim3(BLOCK), LDSKB * 1024, 0, c); }, 20); \
    printf("oneshot   valu=%-3d minb=%d lds=%2dKB           %7.1f us %7.1f GB/s\n", VALU, MINB, LDSKB, us, bytes / us / 1e3); }
#define PIPE(VALU, MINB, TILES, LDSKB) { float us = timeit([&] { hipLaunchKernelGGL((pipelined<VALU, MINB, TILES>), dim3((nb + TILES - 1) / TILES), dim3(BLOCK), LDSKB * 1024, 0, c); }, 20); \
    printf("pipelined valu=%-3d minb=%d lds=%2dKB tiles=%-3d %7.1f us %7.1f GB/s\n", VALU, MINB, LDSKB, TILES, us, bytes / us / 1e3); }
    ONE(0, 4, 0); ONE(0, 4, 40); ONE(200, 4, 0); ONE(200, 4, 40); ONE(200, 4, 52); ONE(400, 4, 0); ONE(400, 4, 40); ONE(400, 4, 52);
    PIPE(0, 4, 4, 0); PIPE(200, 4, 4, 40); PIPE(200, 4, 8, 40); PIPE(200, 4, 4, 52); PIPE(200, 4, 8, 52); PIPE(200, 4, 8, 80);
    PIPE(400, 4, 4, 40); PIPE(400, 4, 8, 40); PIPE(400, 4, 8, 52); PIPE(400, 4, 8, 80); PIPE(400, 4, 16, 52); PIPE(200, 4, 16, 52);
    return 0;
}

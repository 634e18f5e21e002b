// membench.hip -- what a 10-column-in / 3-column-out fp64 stream can reach on this GPU,
// for a few kernel shapes (development aid for sizing the force kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Cols { const double* in[7]; double* a[3]; long long n; };

template <int NIN, int VALU>
__device__ __forceinline__ void body(const Cols& c, long long i) {
    double2 v[NIN];
#pragma unroll
    for (int k = 0; k < NIN; k++) v[k] = *reinterpret_cast<const double2*>(c.in[k] + i);
    double2 a0 = *reinterpret_cast<const double2*>(c.a[0] + i);
    double2 a1 = *reinterpret_cast<const double2*>(c.a[1] + i);
    double2 a2 = *reinterpret_cast<const double2*>(c.a[2] + i);
    double sx = 0, sy = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) { sx += v[k].x; sy += v[k].y; }
#pragma unroll 8
    for (int r = 0; r < VALU; r++) { sx = fma(sx, 1.0000001, 1e-9); sy = fma(sy, 1.0000001, 1e-9); }
    a0.x += sx; a0.y += sy; a1.x += sx; a1.y += sy; a2.x += sx; a2.y += sy;
    *reinterpret_cast<double2*>(c.a[0] + i) = a0;
    *reinterpret_cast<double2*>(c.a[1] + i) = a1;
    *reinterpret_cast<double2*>(c.a[2] + i) = a2;
}

template <int NIN, int VALU, int BLOCK>
__global__ __launch_bounds__(BLOCK) void oneshot(Cols c) {
    const long long i = 2 * ((long long)blockIdx.x * BLOCK + threadIdx.x);
    if (i + 1 < c.n) body<NIN, VALU>(c, i);
}
template <int NIN, int VALU, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gridstride(Cols c) {
    for (long long i = 2 * ((long long)blockIdx.x * BLOCK + threadIdx.x); i + 1 < c.n; i += 2ll * gridDim.x * BLOCK) body<NIN, VALU>(c, i);
}
// same, with an LDS stage + barrier at the start (as the force kernel does for its sources)
template <int NIN, int VALU, int BLOCK>
__global__ __launch_bounds__(BLOCK) void oneshot_lds(Cols c) {
    __shared__ double s[64];
    if (threadIdx.x < 8) s[threadIdx.x] = c.in[0][threadIdx.x];
    __syncthreads();
    const long long i = 2 * ((long long)blockIdx.x * BLOCK + threadIdx.x);
    if (i + 1 < c.n) body<NIN, VALU>(c, i);
    if (s[3] == 123.456) c.a[0][0] = 1;
}

// 8 bytes per lane
template <int NIN, int VALU, int BLOCK>
__global__ __launch_bounds__(BLOCK) void oneshot8(Cols c) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    double v[NIN];
#pragma unroll
    for (int k = 0; k < NIN; k++) v[k] = c.in[k][i];
    double a0 = c.a[0][i], a1 = c.a[1][i], a2 = c.a[2][i];
    double sx = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) sx += v[k];
#pragma unroll 8
    for (int r = 0; r < VALU; r++) sx = fma(sx, 1.0000001, 1e-9);
    c.a[0][i] = a0 + sx; c.a[1][i] = a1 + sx; c.a[2][i] = a2 + sx;
}
// 16 bytes per lane, occupancy limited by dynamic LDS
template <int NIN, int VALU, int BLOCK>
__global__ __launch_bounds__(BLOCK) void oneshot_occ(Cols c) {
    extern __shared__ double dyn[];
    const long long i = 2 * ((long long)blockIdx.x * BLOCK + threadIdx.x);
    if (i + 1 < c.n) body<NIN, VALU>(c, i);
    if (dyn[0] == 123.456) c.a[0][0] = 1;
}
// nontemporal accesses
template <int NIN, int VALU, int BLOCK, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void oneshot_nt(Cols c) {
    const long long i = 2 * ((long long)blockIdx.x * BLOCK + threadIdx.x);
    if (i + 1 >= c.n) return;
    double vx[NIN], vy[NIN];
#pragma unroll
    for (int k = 0; k < NIN; k++) {
        if (NTL) { vx[k] = __builtin_nontemporal_load(c.in[k] + i); vy[k] = __builtin_nontemporal_load(c.in[k] + i + 1); }
        else { double2 t = *reinterpret_cast<const double2*>(c.in[k] + i); vx[k] = t.x; vy[k] = t.y; }
    }
    double ax[3], ay[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (NTL) { ax[k] = __builtin_nontemporal_load(c.a[k] + i); ay[k] = __builtin_nontemporal_load(c.a[k] + i + 1); }
        else { double2 t = *reinterpret_cast<const double2*>(c.a[k] + i); ax[k] = t.x; ay[k] = t.y; }
    }
    double sx = 0, sy = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) { sx += vx[k]; sy += vy[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (NTS) { __builtin_nontemporal_store(ax[k] + sx, c.a[k] + i); __builtin_nontemporal_store(ay[k] + sy, c.a[k] + i + 1); }
        else *reinterpret_cast<double2*>(c.a[k] + i) = make_double2(ax[k] + sx, ay[k] + sy);
    }
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const long long n = 10000000;
    Cols c; c.n = n;
    for (int k = 0; k < 7; k++) { double* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); c.in[k] = p; }
    for (int k = 0; k < 3; k++) { CK(hipMalloc(&c.a[k], n * 8)); CK(hipMemset(c.a[k], 0, n * 8)); }
    const int reps = 20;
    auto report = [&](const char* name, float us, int nin) {
        const double bytes = (nin * 8.0 + 48.0) * n;
        printf("%-44s %8.1f us  %7.1f GB/s\n", name, us, bytes / us / 1e3);
    };
#define ONES(NIN, VALU, BLOCK) report("oneshot nin=" #NIN " valu=" #VALU " block=" #BLOCK, timeit([&] { hipLaunchKernelGGL((oneshot<NIN, VALU, BLOCK>), dim3((n / 2 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, c); }, reps), NIN)
#define ONEL(NIN, VALU, BLOCK) report("oneshot+lds nin=" #NIN " valu=" #VALU " block=" #BLOCK, timeit([&] { hipLaunchKernelGGL((oneshot_lds<NIN, VALU, BLOCK>), dim3((n / 2 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, c); }, reps), NIN)
#define GRID(NIN, VALU, BLOCK, NB) report("gridstride nin=" #NIN " valu=" #VALU " block=" #BLOCK " nb=" #NB, timeit([&] { hipLaunchKernelGGL((gridstride<NIN, VALU, BLOCK>), dim3(NB), dim3(BLOCK), 0, 0, c); }, reps), NIN)
    ONES(7, 0, 256); ONES(7, 0, 512); ONES(7, 0, 1024); ONES(3, 0, 256);
    ONEL(7, 0, 256);
    GRID(7, 0, 256, 2048); GRID(7, 0, 256, 4096); GRID(7, 0, 512, 1024); GRID(7, 0, 256, 8192);
    ONES(7, 100, 256); ONES(7, 200, 256); ONES(7, 400, 256);
    GRID(7, 200, 256, 2048); GRID(7, 200, 256, 4096); GRID(7, 400, 256, 4096);
#define ONE8(NIN, VALU, BLOCK) report("oneshot8B nin=" #NIN " valu=" #VALU " block=" #BLOCK, timeit([&] { hipLaunchKernelGGL((oneshot8<NIN, VALU, BLOCK>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, c); }, reps), NIN)
    ONE8(7, 0, 256); ONE8(7, 200, 256); ONE8(7, 0, 512);
#define OCC(NIN, VALU, BLOCK, LDSKB) report("oneshot occ nin=" #NIN " valu=" #VALU " lds=" #LDSKB "KB", timeit([&] { hipLaunchKernelGGL((oneshot_occ<NIN, VALU, BLOCK>), dim3((n / 2 + BLOCK - 1) / BLOCK), dim3(BLOCK), LDSKB * 1024, 0, c); }, reps), NIN)
    OCC(7, 0, 256, 16); OCC(7, 0, 256, 32); OCC(7, 0, 256, 40); OCC(7, 0, 256, 52); OCC(7, 0, 256, 64); OCC(7, 0, 256, 80);
    OCC(7, 400, 256, 32); OCC(7, 400, 256, 40); OCC(7, 400, 256, 52); OCC(7, 400, 256, 80);
#define NT(NTL, NTS) report("oneshot nt load=" #NTL " store=" #NTS, timeit([&] { hipLaunchKernelGGL((oneshot_nt<7, 0, 256, NTL, NTS>), dim3((n / 2 + 255) / 256), dim3(256), 0, 0, c); }, reps), 7)
    NT(false, false); NT(true, false); NT(false, true); NT(true, true);
    return 0;
}

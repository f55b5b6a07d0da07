// tools/store_alloc_probe.hip — does the history kernel's store rate depend on WHERE its two output arrays lie?  (diagnostic,
// not part of the product.)  The store pattern of k_trace<..., HIST> (tools/store_ceiling.hip, k_hist) on (a) eight separately
// hipMalloc'ed pairs of [12][9437184] Float64 arrays and (b) pairs cut from ONE allocation with a chosen gap between the arrays.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_alloc_probe tools/store_alloc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_hist(double* xv, double* yv, long n, int S, long ld)
{
    const long col = (long)blockIdx.x * 512 + threadIdx.x * 2;
    if (col >= n) return;
    double a = (double)col, b = a + 1.0;
    for (int r = 0; r < S; ++r) {
        d2 vx; vx.x = a; vx.y = b; d2 vy; vy.x = b; vy.y = a;
        __builtin_nontemporal_store(vx, (d2*)(xv + (long)r * ld + col)); __builtin_nontemporal_store(vy, (d2*)(yv + (long)r * ld + col));
    }
}

static double rate(double* xv, double* yv, long n, int S, double seconds)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const dim3 g((unsigned)(n / 512)), blk(256);
    hipLaunchKernelGGL(k_hist, g, blk, 0, 0, xv, yv, n, S, n); CK(hipDeviceSynchronize());
    long cnt = 0; auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(a));
    do { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_hist, g, blk, 0, 0, xv, yv, n, S, n); cnt += 100; CK(hipDeviceSynchronize()); }
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 16.0 * n * S / (ms / cnt * 1e-3) / 1e9;
}

int main()
{
    const long n = 9437184; const int S = 12; const size_t bytes = sizeof(double) * n * S;
    // warm the clocks
    { double *x, *y; CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); rate(x, y, n, S, 1.0); CK(hipFree(x)); CK(hipFree(y)); }
    printf("(a) separately allocated pairs, GB/s (x address, y - x in bytes):\n");
    double* px[8]; double* py[8];
    for (int k = 0; k < 8; ++k) { CK(hipMalloc(&px[k], bytes)); CK(hipMalloc(&py[k], bytes)); }
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < 8; ++k)
            printf("  pair %d rep %d: %7.1f   x = %p  y - x = %lld\n", k, rep, rate(px[k], py[k], n, S, 0.4), (void*)px[k], (long long)((char*)py[k] - (char*)px[k]));
    for (int k = 0; k < 8; ++k) { CK(hipFree(px[k])); CK(hipFree(py[k])); }
    printf("(b) both arrays in ONE allocation, y = x + bytes + gap:\n");
    const long gaps[] = {0, 256, 4096, 65536, 262144, 1 << 20, (1 << 21), (1 << 21) + 4096, (1 << 21) + 65536, 3 << 20, 1 << 24, (1 << 24) + 8192};
    char* big; CK(hipMalloc(&big, 2 * bytes + (1 << 25)));
    for (long gap : gaps)
        printf("  gap %9ld: %7.1f\n", gap, rate((double*)big, (double*)(big + bytes + gap), n, S, 0.4));
    CK(hipFree(big));
    // (d) ONE arena of 48 GiB, the pair cut from it every GiB: is a place a region of the address space?
    {
        const size_t GiB = 1ull << 30;
        char* arena; CK(hipMalloc(&arena, 48 * GiB));
        printf("(d) one 48-GiB allocation, x at offset k GiB, y right behind it (+2 MiB), GB/s by k:\n ");
        for (int k = 0; k + 2 < 48; ++k) {
            double* x = (double*)(arena + k * GiB); double* y = (double*)(arena + k * GiB + bytes + (2 << 20));
            printf(" %d:%.0f", k, rate(x, y, n, S, 0.15)); fflush(stdout);
        }
        printf("\n");
        CK(hipFree(arena));
    }
    // (e) the DISTANCE between the two arrays inside one 72-GiB arena: x at offset X GiB, y at X + D GiB
    {
        const size_t GiB = 1ull << 30;
        char* arena; CK(hipMalloc(&arena, 72 * GiB));
        printf("(e) one 72-GiB allocation, x at X GiB, y at (X + D) GiB, GB/s:\n");
        const int Xs[] = {0, 1, 5, 16};
        const int Ds[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 31, 32, 33, 40, 48};
        for (int X : Xs) {
            printf("  X = %2d:", X);
            for (int D : Ds) {
                if (X + D + 1 > 72) { printf("  D%d:-", D); continue; }
                printf("  D%d:%.0f", D, rate((double*)(arena + X * GiB), (double*)(arena + (X + D) * GiB), n, S, 0.12)); fflush(stdout);
            }
            printf("\n");
        }
        // x alone and y alone (one array, the other pointer the same array's second half is not possible: write x twice)
        CK(hipFree(arena));
    }
    // (c) the ROW STRIDE: 24 rows are written at once per tile, ld * 8 B apart (72 MiB at ld = n); does a padded leading
    // dimension (the ABI's ld >= n) spread them better over the channels, whatever the place?
    printf("(c) six pairs, rows padded by `pad` elements (ld = n + pad), GB/s:\n");
    const long pads[] = {0, 32, 512, 544, 8192, 8192 + 512, 131072 + 512, 1 << 20};
    const long maxpad = 1 << 20;
    const size_t pbytes = sizeof(double) * (n + maxpad) * S;
    double* qx[6]; double* qy[6];
    for (int k = 0; k < 6; ++k) { CK(hipMalloc(&qx[k], pbytes)); CK(hipMalloc(&qy[k], pbytes)); }
    for (long pad : pads) {
        printf("  pad %8ld:", pad);
        for (int k = 0; k < 6; ++k) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const dim3 g((unsigned)(n / 512)), blk(256);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_hist, g, blk, 0, 0, qx[k], qy[k], n, S, n + pad);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_hist, g, blk, 0, 0, qx[k], qy[k], n, S, n + pad);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %7.1f", 16.0 * n * S / (ms / 400 * 1e-3) / 1e9);
        }
        printf("\n");
    }
    return 0;
}

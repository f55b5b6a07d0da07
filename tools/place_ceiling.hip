// tools/place_ceiling.hip — what rate can ANY kernel with k_ft_place's traffic MIX reach on this box?
// (not part of the product; grounds the roofline figure of the full_trace placement pass.)  Per element: 32 B read
// (four arrays, 16-byte loads) and 64 B written (eight streams, 16-byte non-temporal stores), everything aligned, no
// compaction offsets, no LDS, no arithmetic — config 3's survivor count.  Each variant runs >= 1 s back to back.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/place_ceiling tools/place_ceiling.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// one workgroup per tile of 512 elements, as k_ft_place; the second half of every output lies m elements on
__global__ __launch_bounds__(256) void k_mix(const double* a0, const double* a1, const double* a2, const double* a3,
                                              double* o0, double* o1, double* o2, double* o3, long m)
{
    const long j = (long)blockIdx.x * 512 + threadIdx.x * 2;
    if (j >= m) return;
    const d2 v0 = *(const d2*)(a0 + j), v1 = *(const d2*)(a1 + j), v2 = *(const d2*)(a2 + j), v3 = *(const d2*)(a3 + j);
    __builtin_nontemporal_store(v0, (d2*)(o0 + j)); __builtin_nontemporal_store(v1, (d2*)(o1 + j));
    __builtin_nontemporal_store(v2, (d2*)(o2 + j)); __builtin_nontemporal_store(v3, (d2*)(o3 + j));
    __builtin_nontemporal_store(-v0, (d2*)(o0 + m + j)); __builtin_nontemporal_store(v1, (d2*)(o1 + m + j));
    __builtin_nontemporal_store(v2, (d2*)(o2 + m + j)); __builtin_nontemporal_store(v3, (d2*)(o3 + m + j));
}
// the same bytes as a grid-stride loop over a chip-sized grid (no per-tile workgroup start-up)
__global__ __launch_bounds__(256) void k_mix_stride(const double* a0, const double* a1, const double* a2, const double* a3,
                                                     double* o0, double* o1, double* o2, double* o3, long m)
{
    for (long j = ((long)blockIdx.x * 256 + threadIdx.x) * 2; j < m; j += (long)gridDim.x * 512) {
        const d2 v0 = *(const d2*)(a0 + j), v1 = *(const d2*)(a1 + j), v2 = *(const d2*)(a2 + j), v3 = *(const d2*)(a3 + j);
        __builtin_nontemporal_store(v0, (d2*)(o0 + j)); __builtin_nontemporal_store(v1, (d2*)(o1 + j));
        __builtin_nontemporal_store(v2, (d2*)(o2 + j)); __builtin_nontemporal_store(v3, (d2*)(o3 + j));
        __builtin_nontemporal_store(-v0, (d2*)(o0 + m + j)); __builtin_nontemporal_store(v1, (d2*)(o1 + m + j));
        __builtin_nontemporal_store(v2, (d2*)(o2 + m + j)); __builtin_nontemporal_store(v3, (d2*)(o3 + m + j));
    }
}

template <typename F> static double sustain(F launch, double seconds)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    long n = 0; auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(a));
    do { for (int i = 0; i < 100; ++i) launch(); n += 100; CK(hipDeviceSynchronize()); }
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / n;
}

int main()
{
    const long m = 27881472;                  // config 3's survivors (27.9 M), a multiple of 512
    double *in[4], *out[4];
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&in[i], sizeof(double) * m)); CK(hipMemset(in[i], 0, sizeof(double) * m)); CK(hipMalloc(&out[i], sizeof(double) * 2 * m)); }
    const double bytes = 96.0 * m;
    const double t_tile = sustain([&] { hipLaunchKernelGGL(k_mix, dim3((unsigned)(m / 512)), dim3(256), 0, 0, in[0], in[1], in[2], in[3], out[0], out[1], out[2], out[3], m); }, 1.0);
    const double t_stride = sustain([&] { hipLaunchKernelGGL(k_mix_stride, dim3(256 * 8), dim3(256), 0, 0, in[0], in[1], in[2], in[3], out[0], out[1], out[2], out[3], m); }, 1.0);
    printf("{\"elements\": %ld, \"bytes\": %.0f, \"per_tile_ms\": %.4f, \"per_tile_TBps\": %.3f, \"grid_stride_ms\": %.4f, \"grid_stride_TBps\": %.3f}\n",
           m, bytes, t_tile, bytes / t_tile / 1e9, t_stride, bytes / t_stride / 1e9);
    return 0;
}

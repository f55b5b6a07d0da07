// tools/ubench_f32.hip — Float32 issue rates on gfx950 (not part of the product): v_fma_f32 against the packed v_pk_fma_f32
// (two floats per lane and instruction) and v_rsq_f32.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f32 tools/ubench_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float F2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    const float b = 1.0000001f, c = 1e-9f;
    if (OP == 0) {
        float a[8]; for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 1e-3f + 0.1f * j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(a[j], b, c);
        float s = 0; for (int j = 0; j < 8; ++j) s += a[j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if (OP == 1) {
        F2 a[8]; for (int j = 0; j < 8; ++j) a[j] = F2(seed + threadIdx.x * 1e-3f + 0.1f * j);
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_elementwise_fma(a[j], F2(b), F2(c));
        F2 s = F2(0); for (int j = 0; j < 8; ++j) s += a[j];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else {
        float a[8]; for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 1e-3f + 0.1f * j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_rsqf(a[j]);
        float s = 0; for (int j = 0; j < 8; ++j) s += a[j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = 256 * 8, iters = 8192;
    float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_rsq_f32"};
    for (int op = 0; op < 3; ++op) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            for (int r = 0; r < 5; ++r) {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        double instr = (double)blocks * 256 * 8.0 * iters;       // wave-level instructions x 64 lanes
        printf("%-14s %8.3f ms  %.2f lane-instr/clk/CU (x2 flops%s)\n", names[op], ms,
               instr / (ms * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3), op == 1 ? ", x2 floats" : "");
    }
    return 0;
}

// tools/ubench_cmp.hip — what one "rare event" test per FP64 value costs in VALU issue slots on gfx950 (not part of the
// product): the near-branch tests of MATH_FAST (ort_device.hpp, near_zero) sit on every surface of every ray.
//   OP 0: 8 independent v_fma_f64 chains (the baseline slot)          OP 1: + |x| < lim   as v_cmp_lt_f64
//   OP 2: + v_cmp_class_f64                                           OP 3: + |hi32(x)| < hi32(lim) as v_cmp_lt_f32
//   OP 4: + v_and_b32 / v_cmp_lt_u32 on the high dword                OP 5: + v_min_f64 into a running minimum
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench_cmp tools/ubench_cmp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed, double lim)
{
    double a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 1e-3 + 0.1 * j;
    const double b = 1.0000001, c = 1e-9;
    bool odd = false;
    double mn = 1e300;
    const float limf = __int_as_float(__double2hiint(lim));
    const unsigned limu = (unsigned)__double2hiint(lim);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = __builtin_fma(a[j], b, c);
            if (OP == 1) odd = odd || (__builtin_fabs(a[j]) < lim);
            if (OP == 2) odd = odd || __builtin_amdgcn_class(a[j], 0x01c);
            if (OP == 3) odd = odd || (__builtin_fabsf(__int_as_float(__double2hiint(a[j]))) < limf);
            if (OP == 4) odd = odd || (((unsigned)__double2hiint(a[j]) & 0x7fffffffu) < limu);
            if (OP == 5) mn = __builtin_fmin(mn, a[j]);
        }
    }
    double s = mn;
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s + (odd ? 1.0 : 0.0);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = 256 * 8, iters = 4096;
    double* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    const char* names[] = {"fma only", "+ v_cmp_lt_f64 |x|", "+ v_cmp_class_f64", "+ v_cmp_lt_f32 |hi32|", "+ v_and + v_cmp_lt_u32", "+ v_min_f64"};
    float base = 0;
    for (int op = 0; op < 6; ++op) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto launch = [&]() {
            switch (op) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); break;
            }
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        if (op == 0) base = ms;
        printf("%-26s %8.3f ms   extra per value: %.2f fma-slots\n", names[op], ms, (ms - base) / base);
    }
    return 0;
}

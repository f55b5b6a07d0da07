// tools/ubench.hip — micro-measurements that ground the kernel design (not part of the product):
//   1. FP64 VALU issue rates on gfx950: v_fma_f64, v_mul_f64, v_rcp_f64, v_rsq_f64, v_cndmask
//   2. accuracy of the v_rcp_f64 / v_rsq_f64 seeds (how many Newton steps the fast policy needs)
//   3. streaming-store ceilings for the history layout: 8 B vs 16 B per lane, plain vs nontemporal
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_alu(double* out, int iters, double seed)
{
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
    double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                             \
        if (OP == 0) x = __builtin_fma(x, b, c);                            \
        else if (OP == 1) x = x * b;                                        \
        else if (OP == 2) x = __builtin_amdgcn_rcp(x);                      \
        else if (OP == 3) x = __builtin_amdgcn_rsq(x);                      \
        else if (OP == 4) x = x + c;                                        \
        else if (OP == 5) x = __builtin_sqrt(x);                            \
        else if (OP == 6) x = b / x;
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_seed(const double* x, double* rcp, double* rsq, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rcp[i] = __builtin_amdgcn_rcp(x[i]); rsq[i] = __builtin_amdgcn_rsq(x[i]); }
}

typedef double d2 __attribute__((ext_vector_type(2)));
template <int W, bool NT>
__global__ __launch_bounds__(256) void k_store(double* out, long n, int rows, long ld)
{
    // history-like pattern: each block owns 512 consecutive columns and writes `rows` rows
    long col = (long)blockIdx.x * 512 + threadIdx.x * 2;
    if (col >= n) return;
    for (int r = 0; r < rows; ++r) {
        double v = (double)r + col;
        double* p = out + (long)r * ld + col;
        if (W == 16) { d2 x; x.x = v; x.y = v + 1; if (NT) __builtin_nontemporal_store(x, (d2*)p); else *(d2*)p = x; }
        else { if (NT) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v + 1, p + 1); } else { p[0] = v; p[1] = v + 1; } }
    }
}

static float timeit(void (*launch)(void*), void* arg, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(arg); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch(arg);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct AluArg { double* out; int op; int iters; int blocks; };
static void launch_alu(void* p)
{
    AluArg* a = (AluArg*)p;
    switch (a->op) {
    case 0: hipLaunchKernelGGL(k_alu<0>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    case 1: hipLaunchKernelGGL(k_alu<1>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    case 2: hipLaunchKernelGGL(k_alu<2>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    case 3: hipLaunchKernelGGL(k_alu<3>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    case 4: hipLaunchKernelGGL(k_alu<4>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    case 5: hipLaunchKernelGGL(k_alu<5>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    case 6: hipLaunchKernelGGL(k_alu<6>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0); break;
    }
}

struct StArg { double* out; long n; int rows; int variant; };
static void launch_store(void* p)
{
    StArg* a = (StArg*)p;
    dim3 g((unsigned)((a->n + 511) / 512)), b(256);
    switch (a->variant) {
    case 0: hipLaunchKernelGGL((k_store<8, false>), g, b, 0, 0, a->out, a->n, a->rows, a->n); break;
    case 1: hipLaunchKernelGGL((k_store<8, true>), g, b, 0, 0, a->out, a->n, a->rows, a->n); break;
    case 2: hipLaunchKernelGGL((k_store<16, false>), g, b, 0, 0, a->out, a->n, a->rows, a->n); break;
    case 3: hipLaunchKernelGGL((k_store<16, true>), g, b, 0, 0, a->out, a->n, a->rows, a->n); break;
    }
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    // 1. ALU rates: 256 CUs x 8 blocks, 256 threads, 8 independent chains per lane
    const int blocks = 256 * 8, iters = 4096;
    double* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_add_f64", "sqrt(f64) ieee", "div(f64) ieee"};
    for (int op = 0; op < 7; ++op) {
        AluArg a{out, op, op >= 5 ? iters / 8 : iters, blocks};
        float ms = timeit(launch_alu, &a, 5);
        double ops = (double)blocks * 256 * 8.0 * a.iters;
        printf("%-16s %8.3f ms  %9.3f Gop/s  -> %.2f lane-ops/clk/CU at %d MHz\n", names[op], ms, ops / ms / 1e6,
               ops / (ms * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3), prop.clockRate / 1000);
    }
    // 2. seed accuracy
    const int n = 1 << 20;
    std::vector<double> hx(n), hr(n), hs(n);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = std::ldexp(1.0 + (double)rand() / RAND_MAX, rand() % 40 - 20);
    double *dx, *dr, *ds; CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dr, n * 8)); CK(hipMalloc(&ds, n * 8));
    CK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_seed, dim3(n / 256), dim3(256), 0, 0, dx, dr, ds, n);
    CK(hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), ds, n * 8, hipMemcpyDeviceToHost));
    double er = 0, es = 0;
    for (int i = 0; i < n; ++i) {
        er = fmax(er, fabs(hr[i] * hx[i] - 1.0));
        es = fmax(es, fabs(hs[i] * std::sqrt(hx[i]) - 1.0));
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f)   v_rsq_f64 max rel err %.3e (2^%.1f)\n", er, std::log2(er), es, std::log2(es));
    // 3. store ceilings: 24 rows x 9.4M columns x 8 B = 1.81 GB (the bench's history footprint)
    const long cols = 9437184; const int rows = 24;
    double* big; CK(hipMalloc(&big, (size_t)cols * rows * 8));
    const char* sn[] = {"8B plain", "8B nontemporal", "16B plain", "16B nontemporal"};
    for (int v = 0; v < 4; ++v) {
        StArg a{big, cols, rows, v};
        float ms = timeit(launch_store, &a, 10);
        printf("store %-16s %7.3f ms  %8.1f GB/s\n", sn[v], ms, (double)cols * rows * 8 / ms / 1e6);
    }
    return 0;
}

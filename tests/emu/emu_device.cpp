// emu_device.cpp — the device's per-surface step functions (opticalraytracing.jl_amd/csrc/ort_device.hpp) compiled
// for the HOST with stand-ins for the gfx950 builtins (stub/hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY: the CPU
// suite uses it to check the MATH_FAST arms, the near-branch `odd` logic and the polynomial forms against the
// oracle with no GPU present.  One ray at a time, the loop of k_trace's trace_surfaces (ort_kernels.hpp).
//   g++ -O2 -ffp-contract=off -std=c++17 -shared -fPIC -Itests/emu/stub -o tests/emu/libemu_device.so tests/emu/emu_device.cpp
#include "../../opticalraytracing.jl_amd/csrc/ort_device.hpp"

#include <cstring>
#include <vector>

using namespace ort;

template <int MATH, int ARMS>
static bool trace_one(int S, const SurfRec<double>* rec, const double* polys, double y, double x, double u, double v,
                      double* xv, double* yv)
{
    Ray<double> ray[1];
    ray_init<double, MATH>(ray[0], y, x, u, v);
    bool odd = false;
    if (MATH == MATH_FAST)
        odd = t_class(ray[0].x, kClassNonFinite) || t_class(ray[0].y, kClassNonFinite) || t_class(ray[0].k0 + ray[0].k1, kClassNonFinite);
    for (int i = 0; i < S; ++i) {
        const int cls = rec[i].cls;
        const bool even = MATH == MATH_FAST && (cls & (CLS_PEVEN | CLS_FINITE)) == (CLS_PEVEN | CLS_FINITE);
        const double* pl = polys ? polys + (size_t)i * kPolyRec + (even ? 24 : 0) : nullptr;
        surface_step_n<double, MATH, 1, ARMS>(ray, rec[i], pl, cls, i == S - 1, odd);
        xv[i] = ray[0].x; yv[i] = ray[0].y;
    }
    return odd;
}

// rows x {R, t, n, K (or null)}, coef [rows][ncoef] (or null); rays y, x, u = tan U, v = tan V; xv, yv : [S][n];
// odd : [n] (FAST: the ray left the fast forms' domain or came near a branch -> the kernel would retrace its wave)
extern "C" int emu_trace(int fast, int rows, const double* R, const double* t, const double* n, const double* K,
                         const double* coef, int ncoef, long nrays, const double* y, const double* x, const double* u,
                         const double* v, double* xv, double* yv, int* odd, int* arms_out)
{
    const int S = rows - 1;
    int needs = 0;
    std::vector<SurfRec<double>> rec(S);
    std::vector<double> polys;
    const bool hasp = coef && ncoef > 0;
    if (hasp) polys.assign((size_t)S * kPolyRec, 0.0);
    for (int i = 0; i < S; ++i) {
        int nc = 0, pcls = 0;
        if (hasp) pcls = make_poly_rec<double>(polys.data() + (size_t)i * kPolyRec, coef + (size_t)(i + 1) * ncoef, ncoef, &nc);
        memset(&rec[i], 0, sizeof rec[i]);
        needs |= make_rec<double>(rec[i], t[i], R[i + 1], n[i], n[i + 1], K ? K[i + 1] : 0.0, nc, pcls);
    }
    // the build the host would launch for this system (ort_hip.hip: build_records -> arms_of_needs); the reference
    // sequence has one polynomial build
    const int arms = arms_of_needs(needs);
    if (arms_out) *arms_out = arms;
    std::vector<double> bx(S), by(S);
    for (long r = 0; r < nrays; ++r) {
        const double* pp = hasp ? polys.data() : nullptr;
        bool o;
        if (!fast)                    o = trace_one<MATH_IEEE, ARMS_POLY>(S, rec.data(), pp, y[r], x[r], u[r], v[r], bx.data(), by.data());
        else if (arms == ARMS_BASIC)  o = trace_one<MATH_FAST, ARMS_BASIC>(S, rec.data(), pp, y[r], x[r], u[r], v[r], bx.data(), by.data());
        else if (arms == ARMS_GENERAL) o = trace_one<MATH_FAST, ARMS_GENERAL>(S, rec.data(), pp, y[r], x[r], u[r], v[r], bx.data(), by.data());
        else if (arms == ARMS_EVEN)   o = trace_one<MATH_FAST, ARMS_EVEN>(S, rec.data(), pp, y[r], x[r], u[r], v[r], bx.data(), by.data());
        else                          o = trace_one<MATH_FAST, ARMS_POLY>(S, rec.data(), pp, y[r], x[r], u[r], v[r], bx.data(), by.data());
        for (int i = 0; i < S; ++i) { xv[(size_t)i * nrays + r] = bx[i]; yv[(size_t)i * nrays + r] = by[i]; }
        if (odd) odd[r] = o ? 1 : 0;
    }
    return 0;
}

// theta of the full_trace output under MATH_FAST (ort_device.hpp: fast_atan2), element-wise
extern "C" void emu_fast_atan2(long n, const double* y, const double* x, double* out)
{
    for (long i = 0; i < n; ++i) out[i] = ort::fast_atan2(y[i], x[i]);
}

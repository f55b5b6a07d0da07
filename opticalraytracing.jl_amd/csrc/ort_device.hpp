// ort_device.hpp — device-side ray state, surface records and the per-surface step for
// the gfx950 batched ray-trace engine.
//
// What is computed follows the reference loop (Sagnac/OpticalRayTracing.jl,
// src/PupilSampling.jl:34-65 with sag :1-14, tilt :16-19, refract! :21-32); how it is
// computed is MI355X-first: one ray per lane slot (RPT rays per lane for 16-byte stores
// and ILP across the FP64 div/sqrt chains), everything wave-uniform hoisted into a
// per-surface record that a workgroup stages once into LDS (or reads through scalar
// loads), no per-ray arrays, no allocation, branches reduced to selects.
//
// Two arithmetic policies:
//   MATH_IEEE  the exact operation sequence of the reference loop, one IEEE operation per
//              reference operation, no contraction (this file is compiled with
//              -ffp-contract=off), correctly rounded / and sqrt.
//   MATH_FAST  algebraically merged: reciprocal square roots shared between tilt and
//              normalisation, one reciprocal for both slopes, explicit FMAs.  Differs from
//              MATH_IEEE by a few ulp per surface (tested to <= 1e-12 relative).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ort {

enum { MATH_IEEE = 0, MATH_FAST = 1 };

// One record per loop iteration i (0-based): the transfer through row i followed by the
// refraction at row i+1.  Every derived field is the same IEEE operation the reference
// performs per ray, done once per system on the host.
template <typename T>
struct alignas(16) SurfRec {
    T t;       // t[i]                        PupilSampling.jl:46-47
    T R;       // R[i+1]                      :48
    T R2;      // R^2                         :17
    T sgn;     // sign(R)                     :7,18
    T opk;     // 1 + K[i+1]                  :5,17
    T eta;     // n[i] / n[i+1]               :22
    T eta2;    // eta^2                       :24
    T K;       // K[i+1]
    T invR;    // 1 / R (0 for a flat row)    MATH_FAST only
    T pad_;
    int32_t finite;   // isfinite(R)          :2
    int32_t ncoef;    // coefficients in use for this row (0 -> p = zero)
    int32_t pad2_[2];
};

template <typename T>
struct Ray {
    T x, y;        // current transverse position
    T u, v;        // slopes dy/dz, dx/dz     (:38-39,59-60)
    T k0, k1, k2;  // direction cosines, k = [v, u, 1] normalised (Q5)
    T sprev;       // sag of the previous surface: ts[i] = t[i] - s_{i-1} (Q9)
};

template <typename T> __device__ __forceinline__ T t_sqrt(T a);
template <> __device__ __forceinline__ double t_sqrt<double>(double a) { return __builtin_sqrt(a); }
template <> __device__ __forceinline__ float t_sqrt<float>(float a) { return __builtin_sqrtf(a); }

template <typename T> __device__ __forceinline__ T t_fma(T a, T b, T c);
template <> __device__ __forceinline__ double t_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float t_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> __device__ __forceinline__ T t_nan();
template <> __device__ __forceinline__ double t_nan<double>() { return __builtin_nan(""); }
template <> __device__ __forceinline__ float t_nan<float>() { return __builtin_nanf(""); }

template <typename T> __device__ __forceinline__ bool t_isnan(T a) { return a != a; }

// fast reciprocal / reciprocal square root: hardware seed + Newton steps (MATH_FAST).
__device__ __forceinline__ double fast_rcp(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float a)
{
    float r = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ double fast_rsqrt(double a)
{
    double r = __builtin_amdgcn_rsq(a);
    // two Newton steps: r <- r + r*(0.5 - 0.5*a*r*r) ... written with the half-residual
    double h = 0.5 * r;
    double g = a * r;
    double e = __builtin_fma(-h, g, 0.5);
    r = __builtin_fma(r, e, r);
    h = 0.5 * r;
    g = a * r;
    e = __builtin_fma(-h, g, 0.5);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ float fast_rsqrt(float a)
{
    float r = __builtin_amdgcn_rsqf(a);
    float e = __builtin_fmaf(-0.5f * a * r, r, 0.5f);
    return __builtin_fmaf(r, e, r);
}
// sqrt(a) = a * rsqrt(a), with one correction step; a = 0 handled.
__device__ __forceinline__ double fast_sqrt(double a)
{
    double r = fast_rsqrt(a);
    double g = a * r;
    double d = __builtin_fma(-g, g, a);
    g = __builtin_fma(0.5 * r, d, g);
    return a == 0.0 ? 0.0 : g;
}
__device__ __forceinline__ float fast_sqrt(float a)
{
    float r = fast_rsqrt(a);
    float g = a * r;
    float d = __builtin_fmaf(-g, g, a);
    g = __builtin_fmaf(0.5f * r, d, g);
    return a == 0.0f ? 0.0f : g;
}

// p(y), Horner (Types.jl:21-27 restricted to a power series).
template <typename T>
__device__ __forceinline__ T poly_eval(const T* __restrict__ c, int nc, T y)
{
    T acc = c[nc - 1];
    for (int j = nc - 2; j >= 0; --j) acc = acc * y + c[j];
    return acc;
}
// p'(y): analytic derivative.  The reference takes a complex step with eps = 2^-26
// (RayTracing.jl:103); for a polynomial that equals p' up to O(eps^2) relative.
template <typename T>
__device__ __forceinline__ T poly_deriv(const T* __restrict__ c, int nc, T y)
{
    if (nc < 2) return T(0);
    T acc = T(nc - 1) * c[nc - 1];
    for (int j = nc - 2; j >= 1; --j) acc = acc * y + T(j) * c[j];
    return acc;
}

// Launch direction cosines from slopes: k = normalize([v, u, 1])  (:40-41, Q6).
template <typename T, int MATH>
__device__ __forceinline__ void ray_init(Ray<T>& r, T y, T x, T u, T v)
{
    r.y = y; r.x = x; r.u = u; r.v = v; r.sprev = T(0);
    if (MATH == MATH_IEEE) {
        T nrm = t_sqrt<T>((v * v + u * u) + T(1));
        T inv = T(1) / nrm;
        r.k0 = v * inv; r.k1 = u * inv; r.k2 = inv;
    } else {
        T inv = fast_rsqrt(t_fma<T>(v, v, t_fma<T>(u, u, T(1))));
        r.k0 = v * inv; r.k1 = u * inv; r.k2 = inv;
    }
}

// One loop iteration of src/PupilSampling.jl:45-63 in the reference's operation order.
template <typename T>
__device__ __forceinline__ void surface_step_ieee(Ray<T>& r, const SurfRec<T>& s,
                                                  const T* __restrict__ coef)
{
    const T tcur = s.t - r.sprev;                    // ts[i] (:54-55 of the previous pass)
    r.y = r.y + r.u * tcur;                          // :46
    r.x = r.x + r.v * tcur;                          // :47
    T sg;
    if (s.finite) {                                  // wave-uniform (:2)
        const T beta = (s.R - r.y * r.u) - r.x * r.v;            // :3
        const T r2 = r.x * r.x + r.y * r.y;                      // :4
        const T D = beta * beta - r2 * ((s.opk + r.u * r.u) + r.v * r.v);   // :5
        sg = r2 / (beta + s.sgn * t_sqrt<T>(D));                 // :7
        if (s.ncoef > 0) sg = sg + poly_eval<T>(coef, s.ncoef, r.y);
        else             sg = sg + T(0);
        sg = (D >= T(0)) ? sg : t_nan<T>();                      // :6,9
    } else {
        sg = T(0);                                               // :12
    }
    r.y = r.y + sg * r.u;                            // :52
    r.x = r.x + sg * r.v;                            // :53
    r.sprev = sg;
    // tilt (:16-19), normal (:56-57)
    const T Dt = s.R2 - (r.x * r.x + r.y * r.y) * s.opk;
    const T sq = t_sqrt<T>(Dt);
    T tx = s.sgn * r.x / sq;
    T ty = s.sgn * r.y / sq;
    if (s.ncoef > 0) {
        tx = tx + poly_deriv<T>(coef, s.ncoef, r.x);             // Q2: p'(x) on the x slope
        ty = ty + poly_deriv<T>(coef, s.ncoef, r.y);
    } else {
        tx = tx + T(0);
        ty = ty + T(0);
    }
    const T nrm = t_sqrt<T>((tx * tx + ty * ty) + T(1));
    const T inv = T(1) / nrm;
    const T m0 = tx * inv, m1 = ty * inv, m2 = -inv;
    // refract! (:21-32)
    const T g = -((r.k0 * m0 + r.k1 * m1) + r.k2 * m2);
    const T D2 = T(1) - s.eta2 * (T(1) - g * g);
    const T cf = s.eta * g - t_sqrt<T>(D2);
    const bool ok = D2 >= T(0);                      // TIR / NaN: k untouched (Q1)
    const T n0 = s.eta * r.k0 + cf * m0;
    const T n1 = s.eta * r.k1 + cf * m1;
    const T n2 = s.eta * r.k2 + cf * m2;
    r.k0 = ok ? n0 : r.k0;
    r.k1 = ok ? n1 : r.k1;
    r.k2 = ok ? n2 : r.k2;
    r.u = r.k1 / r.k2;                               // :59
    r.v = r.k0 / r.k2;                               // :60
}

// The same iteration, algebraically merged (MATH_FAST):
//   normal: with Dt = R^2 - r^2(1+K), |(tx,ty,-1)|^2 = (R^2 - K r^2)/Dt, hence
//           m = (sgn x, sgn y, -sqrt(Dt)) / sqrt(R^2 - K r^2)      (no polynomial)
//   slopes: one reciprocal of k2;  sag: one reciprocal;  FMAs throughout.
template <typename T>
__device__ __forceinline__ void surface_step_fast(Ray<T>& r, const SurfRec<T>& s,
                                                  const T* __restrict__ coef)
{
    const T tcur = s.t - r.sprev;
    r.y = t_fma<T>(r.u, tcur, r.y);
    r.x = t_fma<T>(r.v, tcur, r.x);
    T m0, m1, m2;
    if (s.finite) {
        const T beta = t_fma<T>(-r.x, r.v, t_fma<T>(-r.y, r.u, s.R));
        const T r2 = t_fma<T>(r.x, r.x, r.y * r.y);
        const T A = t_fma<T>(r.v, r.v, t_fma<T>(r.u, r.u, s.opk));
        const T D = t_fma<T>(beta, beta, -(r2 * A));
        T sg = r2 * fast_rcp(beta + s.sgn * fast_sqrt(D));
        if (s.ncoef > 0) sg = sg + poly_eval<T>(coef, s.ncoef, r.y);
        sg = (D >= T(0)) ? sg : t_nan<T>();
        r.y = t_fma<T>(sg, r.u, r.y);
        r.x = t_fma<T>(sg, r.v, r.x);
        r.sprev = sg;
        const T rr = t_fma<T>(r.x, r.x, r.y * r.y);
        const T Dt = t_fma<T>(-rr, s.opk, s.R2);
        if (s.ncoef > 0) {
            const T is = s.sgn * fast_rsqrt(Dt);
            const T tx = t_fma<T>(r.x, is, poly_deriv<T>(coef, s.ncoef, r.x));
            const T ty = t_fma<T>(r.y, is, poly_deriv<T>(coef, s.ncoef, r.y));
            const T inv = fast_rsqrt(t_fma<T>(tx, tx, t_fma<T>(ty, ty, T(1))));
            m0 = tx * inv; m1 = ty * inv; m2 = -inv;
        } else {
            const T q = t_fma<T>(-s.K, rr, s.R2);        // R^2 - K r^2
            const T inv = fast_rsqrt(q);
            const T si = s.sgn * inv;
            m0 = r.x * si; m1 = r.y * si; m2 = -(fast_sqrt(Dt) * inv);
            // Dt < 0 (beyond the conic's rim): reference normal is NaN -> ray undeviated
            m2 = (Dt >= T(0)) ? m2 : t_nan<T>();
        }
    } else {
        r.sprev = T(0);
        m0 = T(0); m1 = T(0); m2 = T(-1);
    }
    const T g = -t_fma<T>(r.k2, m2, t_fma<T>(r.k1, m1, r.k0 * m0));
    const T D2 = t_fma<T>(-s.eta2, t_fma<T>(-g, g, T(1)), T(1));
    const T cf = t_fma<T>(s.eta, g, -fast_sqrt(D2));
    const bool ok = D2 >= T(0);
    const T n0 = t_fma<T>(s.eta, r.k0, cf * m0);
    const T n1 = t_fma<T>(s.eta, r.k1, cf * m1);
    const T n2 = t_fma<T>(s.eta, r.k2, cf * m2);
    r.k0 = ok ? n0 : r.k0;
    r.k1 = ok ? n1 : r.k1;
    r.k2 = ok ? n2 : r.k2;
    const T ik = fast_rcp(r.k2);
    r.u = r.k1 * ik;
    r.v = r.k0 * ik;
}

template <typename T, int MATH>
__device__ __forceinline__ void surface_step(Ray<T>& r, const SurfRec<T>& s,
                                             const T* __restrict__ coef)
{
    if (MATH == MATH_IEEE) surface_step_ieee<T>(r, s, coef);
    else                   surface_step_fast<T>(r, s, coef);
}

}  // namespace ort

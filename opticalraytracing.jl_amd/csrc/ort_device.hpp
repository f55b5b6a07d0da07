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
//   MATH_FAST  the same geometry in direction-cosine form (no slopes): for a conic row the
//              path length is d = F / (G + sqrt(G^2 - a F)), the discriminant's root IS the
//              cosine of incidence on a sphere, and the sphere's normal needs no square root:
//              2 rsq + 1 rcp seeds per ray-surface instead of 4 IEEE sqrt + 6 IEEE divisions.
//              Seeds (v_rsq_f64 / v_rcp_f64, 2^-24 accurate, measured: tools/ubench.hip) are
//              refined by ONE Newton step to ~2^-49.  Rows carrying a
//              polynomial term keep the reference's slope form (its additive p(y) is a quirk,
//              not a geometric intersection, Q2).  Differs from MATH_IEEE by rounding only
//              (tested <= 1e-12 relative, bar 1e-10).  The forms hold for the rays a lens passes; where the
//              reference's formulas stop being the geometry — a hit beyond a sphere's equator (it keeps the
//              vertex-side slope, PupilSampling.jl:16-19), a direction refracted backward (it keeps tracing the
//              line by its slopes), a polynomial row met outside its conic's radius (NaN tilt, k untouched) —
//              the step raises `odd` and the kernel retraces that wave with the MATH_IEEE sequence.
//              The same happens to a ray that comes within kNear (normalised) of one of the reference's BRANCHES —
//              the sag discriminant's sign (:6), the refraction discriminant's (:25), the equator, the stop filter
//              (:132) — so every branch outcome (status, TIR, survivor count) is decided by the reference sequence
//              itself wherever the two arithmetics could disagree: MATH_FAST is status-exact by construction.
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
    T K;       // K[i+1]; for KIND_SPHERE_C rows: t + R (centre form)
    T invR;    // c = 1 / R (0 for a flat row)   MATH_FAST only
    T ome2;    // 1 - eta^2                   MATH_FAST only
    T e2c2;    // eta^2 c^2                   MATH_FAST centre form
    T ec;      // eta |c|                     MATH_FAST centre form
    T dlim;    // kNear R^2: |sag discriminant| below this -> retrace (centre form, polynomial rows)   MATH_FAST only
    T zlim;    // |R| (1 - sqrt(kNear)): |sag| beyond this is at / past the equator -> retrace          MATH_FAST centre form
    int32_t finite;   // isfinite(R)          :2
    int32_t ncoef;    // coefficients in use for this row (0 -> p = zero)
    int32_t spare;
    int32_t cls;      // packed wave-uniform class bits (CLS_*), read once per surface
};

// Near-branch thresholds of MATH_FAST (see the notes at the top).  The fast forms differ from the reference sequence by
// <= ~1e-11 relative (measured 5e-12, bar 1e-10), so a branch quantity farther than this from its boundary has the same
// sign in both; Float32: eps is 2^29 times larger.
template <typename T> struct Near;
template <> struct Near<double> { static constexpr double thr = 1e-9, root = 3.1622776601683795e-5; };
template <> struct Near<float> { static constexpr float thr = 1e-4f, root = 1e-2f; };

// Polynomial rows: one record of kPolyRec values per loop iteration, built once per system (host or k_build_tables):
//   [0,12)  pc[j] = c_j                       value coefficients, zero padded          (Types.jl:21-27)
//   [12,24) dc[j] = (j+1) c_{j+1}             coefficients of p' (one IEEE product each, as the per-ray T(j) * c[j] was)
//   [24,30) ev[k] = c_{2k}                    even form, rows whose odd coefficients are all zero:  p(y)  = E(y^2)
//   [30,36) qd[k] = (2k+2) c_{2k+2}                                                                 p'(y) = y Q(y^2)
// The kernel stages kPolyLds values per row in LDS: pc | dc, or ev | qd (MATH_FAST, even rows).
constexpr int kPolyRec = 36, kPolyLds = 24, kPolyMax = 12;

enum { KIND_SPHERE = 0, KIND_FLAT = 1, KIND_CONIC = 2, KIND_POLY = 3, KIND_SPHERE_C = 4 };
// packed wave-uniform class bits of a row (SurfRec::cls)
enum { CLS_FINITE = 1, CLS_HASP = 2, CLS_REFR = 4, CLS_TIR = 8, CLS_KIND_SHIFT = 4,
       CLS_PEVEN = 1 << 8,    // polynomial row whose odd coefficients are all zero (even form, ev | qd)
       CLS_PBIG = 1 << 9 };   // ... of more than 4 even terms / more than 8 coefficients: the long unrolled chains


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

// v_cmp_class: does `a` belong to one of the classes of `mask` (bit 2/3/4 = -inf/-normal/-denormal, 7/8/9 = +denormal/
// +normal/+inf; NaNs are bits 0-1)?  The mask is a runtime operand.
__device__ __forceinline__ bool t_class(double a, int mask) { return __builtin_amdgcn_class(a, mask); }
__device__ __forceinline__ bool t_class(float a, int mask) { return __builtin_amdgcn_classf(a, mask); }
constexpr int kClassPositive = 0x380, kClassNegative = 0x01c, kClassNonFinite = 0x207;   // 0x207: NaNs and both infinities

__device__ __forceinline__ double t_abs(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float t_abs(float a) { return __builtin_fabsf(a); }

// Fast reciprocal / reciprocal square root (MATH_FAST): hardware seed (relative error
// 2^-24.4 / 2^-24.2 on gfx950, tools/ubench.hip) + ONE Newton step:
//   1/a      = r0 (1 + e),      e = 1 - a r0        -> error e^2      ~ 2^-48.8
//   1/sqrt a = r0 (1 + e/2),    e = 1 - a r0^2      -> error 3/8 e^2  ~ 2^-49.8
// (the accuracy sqrt_core below always had; the cubically convergent step of rounds 1-2 bought 2^-71 for one more FMA per
// seed — measured on configs 2 and 3: worst deviation from the reference sequence 3.7e-13 / 2.9e-13 with it, 3.7e-13 /
// 1.9e-13 without, bar 1e-10.)
__device__ __forceinline__ double fast_rcp(double a)
{
    const double r0 = __builtin_amdgcn_rcp(a);
    const double e = __builtin_fma(-a, r0, 1.0);
    return __builtin_fma(r0, e, r0);
}
// Float32: v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 are 1-ulp instructions — the Float32 policy's own rounding level — and are
// used as they come (a Newton step on top bought half an ulp for 2-4 more operations per seed; without it ort_spot_batch_f32
// on config 5 runs 3 % faster and its hits sit as close to the Float64 trace as before: rms 7.6e-6 mm)
__device__ __forceinline__ float fast_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ double fast_rsqrt(double a)
{
    const double r0 = __builtin_amdgcn_rsq(a);
    const double t = a * r0, h = 0.5 * r0;
    const double e = __builtin_fma(-t, r0, 1.0);
    return __builtin_fma(h, e, r0);
}
__device__ __forceinline__ float fast_rsqrt(float a) { return __builtin_amdgcn_rsqf(a); }

// MATH_FAST: atan(y, x) (theta of the full_trace output, src/PupilSampling.jl:133) without ocml's ~120 instructions per
// ray — a fifth of everything the full_trace kernel issues per ray on a 12-surface system.  One division:
// t = min / max of (|x|, |y|) directly, or (min - max) / (min + max) = (t - 1) / (t + 1) when t > tan(pi/8), which
// brings the argument to |u| <= tan(pi/8); atan(u) = u + u w P(w), w = u^2, P of degree 9 (interpolated at Chebyshev
// nodes in 50-digit arithmetic: approximation error 5e-17); then the octant is unfolded.  Measured against libm on 4e5
// points: <= 4.4e-16 absolute.  atan(0, 0) = 0 for either sign of x's zero (libm: pi for -0); NaN in, NaN out.
__constant__ double kAtanP[10] = {-3.33333333333332482e-01, 1.99999999998984074e-01, -1.42857142660966191e-01, 1.11111096365343609e-01,
                                        -9.09085255717604901e-02, 7.69105515839314940e-02, -6.64961369529166874e-02, 5.73633216590764272e-02,
                                        -4.48333462227288593e-02, 2.27505269933616708e-02};
__device__ __forceinline__ double fast_atan2(double y, double x)
{
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
    const bool big = mn > 0.41421356237309503 * mx;
    const double num = big ? mn - mx : mn;
    double den = big ? mn + mx : mx;
    den = den == 0.0 ? 1.0 : den;
    const double r = fast_rcp(den);
    double u = num * r;
    u = __builtin_fma(__builtin_fma(-u, den, num), r, u);
    const double w = u * u;
    // the coefficients come from constant memory (scalar loads -> SGPR operands of v_fma): as literals the compiler moves
    // each one into a VGPR pair first (two v_mov per FMA)
    double p = kAtanP[9];
#pragma unroll
    for (int j = 8; j >= 0; --j) p = __builtin_fma(p, w, kAtanP[j]);
    double a = __builtin_fma(u * w, p, u);
    a = big ? 0.7853981633974483 + a : a;
    a = ay > ax ? 1.5707963267948966 - a : a;
    a = x < 0.0 ? 3.141592653589793 - a : a;
    a = __builtin_isunordered(x, y) ? __builtin_nan("") : a;      // (max / min drop a NaN operand)
    return __builtin_copysign(a, y);
}
__device__ __forceinline__ float fast_atan2(float y, float x) { return ::atan2f(y, x); }
// sqrt(a) = a * rsqrt(a) for a > 0; a < 0 -> NaN (a ray that misses, :9).
// Float64: g = a r0 refined by ONE Newton step, g (1 + e/2) with e = 1 - g r0: error 3/8 e^2 ~ 2^-50
// — 5 instructions instead of the 7 of a * fast_rsqrt(a), and it is the sqrt, not the reciprocal root,
// that the sphere rows need twice per intersection.  Measured against the IEEE policy on configs 2 and 3:
// worst deviation 3.7e-13 relative (1.5e-13 with the cubic root), bar 1e-10; -5 % kernel time.
__device__ __forceinline__ double sqrt_core(double a)
{
    const double r0 = __builtin_amdgcn_rsq(a);
    const double g = a * r0, h = 0.5 * r0;
    const double e = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, e, g);
}
__device__ __forceinline__ float sqrt_core(float a) { return __builtin_amdgcn_sqrtf(a); }   // NaN for a < 0 (a miss, :9), 0 for 0
// (a = 0 gives 0 * inf = NaN, a < 0 NaN: every radicand of the fast arms that can reach zero is a BRANCH quantity of the
// reference — sag discriminant, refraction discriminant — whose neighbourhood raises `odd`, so the value is never used there)

// p(y), Horner (Types.jl:21-27 restricted to a power series): acc = c[nc-1]; acc = acc * y + c[j], j = nc-2 .. 0 — the
// loop of the reference restatement, unrolled: a wave-uniform switch enters the chain at the table's width nc, so the
// operations and their order are the loop's (no zero padding: 0 * inf would differ), the coefficient reads have fixed
// LDS addresses and nothing waits per term.  pc = the row's value coefficients (record layout: kPolyRec above).
#define ORT_HORNER_CASE(n, c) case n: acc = acc * y + (c)[n - 2]; [[fallthrough]];
template <typename T>
__device__ __forceinline__ T poly_eval(const T* __restrict__ pc, int nc, T y)
{
    T acc = pc[nc - 1];
    switch (nc) {
    ORT_HORNER_CASE(12, pc) ORT_HORNER_CASE(11, pc) ORT_HORNER_CASE(10, pc) ORT_HORNER_CASE(9, pc) ORT_HORNER_CASE(8, pc)
    ORT_HORNER_CASE(7, pc) ORT_HORNER_CASE(6, pc) ORT_HORNER_CASE(5, pc) ORT_HORNER_CASE(4, pc) ORT_HORNER_CASE(3, pc)
    case 2: acc = acc * y + pc[0]; [[fallthrough]];
    default: break;
    }
    return acc;
}
// p'(y): analytic derivative.  The reference takes a complex step with eps = 2^-26 (RayTracing.jl:103); for a polynomial
// that equals p' up to O(eps^2) relative.  dc[j] = (j+1) c_{j+1}: acc = dc[nc-2]; acc = acc * y + dc[j], j = nc-3 .. 0.
template <typename T>
__device__ __forceinline__ T poly_deriv(const T* __restrict__ dc, int nc, T y)
{
    if (nc < 2) return T(0);
    T acc = dc[nc - 2];
    switch (nc - 1) {
    ORT_HORNER_CASE(11, dc) ORT_HORNER_CASE(10, dc) ORT_HORNER_CASE(9, dc) ORT_HORNER_CASE(8, dc) ORT_HORNER_CASE(7, dc)
    ORT_HORNER_CASE(6, dc) ORT_HORNER_CASE(5, dc) ORT_HORNER_CASE(4, dc) ORT_HORNER_CASE(3, dc)
    case 2: acc = acc * y + dc[0]; [[fallthrough]];
    default: break;
    }
    return acc;
}
#undef ORT_HORNER_CASE

// Loop forms over a raw coefficient row c[0 .. nc-1] (meridional kernels: a handful of rays per launch).
template <typename T>
__device__ __forceinline__ T poly_eval_loop(const T* __restrict__ c, int nc, T y)
{
    T acc = c[nc - 1];
    for (int j = nc - 2; j >= 0; --j) acc = acc * y + c[j];
    return acc;
}
template <typename T>
__device__ __forceinline__ T poly_deriv_loop(const T* __restrict__ c, int nc, T y)
{
    if (nc < 2) return T(0);
    T acc = T(nc - 1) * c[nc - 1];
    for (int j = nc - 2; j >= 1; --j) acc = acc * y + T(j) * c[j];
    return acc;
}

// MATH_FAST forms: fused, fixed length (zero padded).  FORM 0 / 1: every odd coefficient of the row is zero (the
// usual even asphere): p(y) = E(y^2) with NE = 4 / 6 terms, p'(y) = y Q(y^2); x^2 and y^2 are shared with the conic's
// r^2.  FORM 2 / 3: general series of <= 8 / 12 coefficients.  pl = the row's staged block: ev | qd, or pc | dc.
template <typename T, int N>
__device__ __forceinline__ T horner_fast(const T* __restrict__ c, T t)
{
    T acc = c[N - 1];
#pragma unroll
    for (int j = N - 2; j >= 0; --j) acc = t_fma<T>(acc, t, c[j]);
    return acc;
}
template <typename T, int FORM>
__device__ __forceinline__ T poly_value_fast(const T* __restrict__ pl, T y)
{
    if (FORM == 0) return horner_fast<T, 4>(pl, y * y);
    if (FORM == 1) return horner_fast<T, 6>(pl, y * y);
    if (FORM == 2) return horner_fast<T, 8>(pl, y);
    return horner_fast<T, 12>(pl, y);
}
// tilt slopes (:18-19, Q2): tx = x is + p'(x), ty = y is + p'(y), `is` = sign(R) / sqrt(R^2 - r^2 (1+K)) (0 on a flat row)
template <typename T, int FORM>
__device__ __forceinline__ void poly_tilt_fast(const T* __restrict__ pl, T x, T y, T xx, T yy, T is, T& tx, T& ty)
{
    if (FORM == 0 || FORM == 1) {
        constexpr int NQ = FORM == 0 ? 3 : 5;
        tx = x * (is + horner_fast<T, NQ>(pl + 6, xx));
        ty = y * (is + horner_fast<T, NQ>(pl + 6, yy));
    } else {
        constexpr int ND = FORM == 2 ? 7 : 11;
        tx = t_fma<T>(x, is, horner_fast<T, ND>(pl + kPolyMax, x));
        ty = t_fma<T>(y, is, horner_fast<T, ND>(pl + kPolyMax, y));
    }
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

// ---- correctly rounded FP64 division and square root for MATH_IEEE --------------------------
// hipcc expands `a / b` to  div_scale x2, rcp, two Newton steps on the reciprocal, q0 = a r,
// rem = fma(-b, q0, a), div_fmas(rem, r, q0), div_fixup  and `sqrt(x)` to a range-scaling
// ldexp pair around  rsq + one Goldschmidt step + two residual corrections.  The scaling only acts
// for |exponents| beyond 2^±767 / quotients near the range limits, never reached by lens
// geometry in millimetres, so the same sequences without it give the SAME bits; div_fixup and
// the class test keep 0, inf and NaN operands exact.  What this buys: the two quotients that share a
// denominator (tilt: sgn x / sqrt(D), sgn y / sqrt(D); slopes: k1 / k3, k0 / k3) share ONE
// refined reciprocal.  Bit-identity with the CPU oracle is asserted by tests/test_gpu_parity.py.
__device__ __forceinline__ double ieee_rcp_refined(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double ieee_div_with(double a, double b, double r)
{
    const double q0 = a * r;
    const double rem = __builtin_fma(-b, q0, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q0), b, a);
}
__device__ __forceinline__ void ieee_div2(double a1, double a2, double b, double& q1, double& q2)
{
    const double r = ieee_rcp_refined(b);
    q1 = ieee_div_with(a1, b, r);
    q2 = ieee_div_with(a2, b, r);
}
__device__ __forceinline__ void ieee_div2(float a1, float a2, float b, float& q1, float& q2) { q1 = a1 / b; q2 = a2 / b; }
__device__ __forceinline__ double ieee_div(double a, double b) { return ieee_div_with(a, b, ieee_rcp_refined(b)); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }
__device__ __forceinline__ double ieee_sqrt_any(double x)      // every IEEE special case: +-0 and +inf return x, x < 0 -> NaN
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;
}
__device__ __forceinline__ float ieee_sqrt_any(float x) { return __builtin_sqrtf(x); }
// The same for a FINITE radicand (the sag and refraction discriminants, :5,24):
__device__ __forceinline__ double ieee_sqrt(double x)
{
    // x = 0: the seed of rsq(0) is +inf and 0 * inf = NaN; seeding from x + 1e-300 instead (one add where a class
    // test and two selects would be) gives 0 * 1e150 = 0 and every later step keeps the 0 exactly.  x + 1e-300 == x
    // for every x > 1e-284, and a non-zero radicand of this loop (a sum / difference of doubles of the size of 1 or
    // R^2, never a product of tiny factors) is far above that; x < 0 is still NaN through rsq.  +inf would give NaN
    // instead of +inf: the one radicand that can be infinite (the tilt's, on a flat row with a polynomial) uses ieee_sqrt_any.
    const double y = __builtin_amdgcn_rsq(x + 1e-300);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return g;
}
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }

// The same square root for a radicand that only ever feeds a DIVISOR (tilt :17, normalize! :57), together with the
// reciprocal of the result.  (1) No special-case select: x = 0, inf or < 0 gives NaN here where IEEE gives 0 /
// inf / NaN, and every one of those ends in a NaN normal in the reference too (x / 0 = inf -> norm inf -> inf * 0),
// i.e. in "k untouched" (Q1) — no output can tell.  (2) The iteration's h = 1 / (2 sqrt x) (~2^-49) seeds the
// reciprocal of g = RN(sqrt x): ONE Newton step replaces v_rcp_f64 + two (the quotients built on r are then
// corrected by their exact remainder as in ieee_div_with: same bits as the correctly rounded division, asserted
// against the CPU oracle by the bit-exact suites of tests/test_gpu_parity.py).
__device__ __forceinline__ void ieee_sqrt_rcp(double x, double& g, double& r)
{
    const double y = __builtin_amdgcn_rsq(x);
    g = x * y;
    double h = y * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    const double r0 = h + h;
    const double f = __builtin_fma(-g, r0, 1.0);
    r = __builtin_fma(r0, f, r0);
}
// a / b with r ~ 1 / b given, no special-case fixup (see ieee_sqrt_rcp for when that is safe)
__device__ __forceinline__ double ieee_div_nofix(double a, double b, double r)
{
    const double q0 = a * r;
    const double rem = __builtin_fma(-b, q0, a);
    return __builtin_fma(rem, r, q0);
}

// One loop iteration of src/PupilSampling.jl:45-63 in the reference's operation order.
// FINITE = isfinite(R) (:2), HASP = the row carries polynomial coefficients: both are
// wave-uniform, so the kernel branches on them ONCE per surface and runs this straight-line
// body for all of a lane's rays (the scheduler interleaves their div/sqrt chains).
// LAST = the final loop iteration: xv[i], yv[i] (:61-62) are fixed once the sag is added, and nothing the
// reference computes afterwards in that iteration (tilt, refract!, slopes) reaches an output — skipped.
// Rows without a polynomial drop the reference's `+ zero(y)` / `+ dp_dy(zero, .)` additions of 0.0: a + 0.0 == a
// bit for bit except for a = -0.0 (-> +0.0), and a zero's sign reaches no output (no division by it).
template <typename T, bool FINITE, bool HASP, bool LAST>
__device__ __forceinline__ void surface_step_ieee(Ray<T>& r, const SurfRec<T>& s,
                                                  const T* __restrict__ pl)      // the row's pc | dc block (HASP rows)
{
    const int nc = HASP ? __builtin_amdgcn_readfirstlane(s.ncoef) : 0;           // wave-uniform: the Horner switch is a scalar branch
    const T tcur = s.t - r.sprev;                    // ts[i] (:54-55 of the previous pass)
    r.y = r.y + r.u * tcur;                          // :46
    r.x = r.x + r.v * tcur;                          // :47
    T sg;
    if (FINITE) {                                    // :2
        const T beta = (s.R - r.y * r.u) - r.x * r.v;            // :3
        const T r2 = r.x * r.x + r.y * r.y;                      // :4
        const T D = beta * beta - r2 * ((s.opk + r.u * r.u) + r.v * r.v);   // :5
        // :7; sign(R) sqrt(D) is exact, so the fused beta + sign(R) sqrt(D) rounds once, like the reference's sum
        sg = ieee_div(r2, t_fma<T>(s.sgn, ieee_sqrt(D), beta));
        if (HASP) sg = sg + poly_eval<T>(pl, nc, r.y);
        // :6,9 — D < 0 or NaN: sqrt(D) is NaN and so is sg, no select needed
    } else {
        sg = T(0);                                               // :12
    }
    r.y = r.y + sg * r.u;                            // :52
    r.x = r.x + sg * r.v;                            // :53
    r.sprev = sg;
    if (LAST) return;
    // tilt (:16-19), normal (:56-57)
    const T Dt = s.R2 - (r.x * r.x + r.y * r.y) * s.opk;
    T m0, m1, m2;
    if (sizeof(T) == 8 && !HASP) {
        // sign(R) x / sqrt(Dt), sign(R) y / sqrt(Dt): the sign is applied to the quotient (exact either way); the
        // reciprocals of sqrt(Dt) and of the norm come out of their own square-root iterations (ieee_sqrt_rcp)
        double sq, rq, nrm, inv0;
        ieee_sqrt_rcp((double)Dt, sq, rq);
        const double tx = ieee_div_nofix((double)r.x, sq, rq), ty = ieee_div_nofix((double)r.y, sq, rq);
        ieee_sqrt_rcp((tx * tx + ty * ty) + 1.0, nrm, inv0);
        const double inv = ieee_div_nofix(1.0, nrm, inv0);
        const double si = (double)s.sgn * inv;                   // exact: (sign(R) t) inv == t (sign(R) inv)
        m0 = (T)(tx * si); m1 = (T)(ty * si); m2 = (T)(-inv);
    } else {
        const T sq = ieee_sqrt_any(Dt);                          // inf on a flat row (with a polynomial)
        T tx, ty;
        ieee_div2(s.sgn * r.x, s.sgn * r.y, sq, tx, ty);         // one refined reciprocal, two quotients
        if (HASP) {
            tx = tx + poly_deriv<T>(pl + kPolyMax, nc, r.x);     // Q2: p'(x) on the x slope
            ty = ty + poly_deriv<T>(pl + kPolyMax, nc, r.y);
        }
        const T nrm = ieee_sqrt((tx * tx + ty * ty) + T(1));
        const T inv = ieee_div(T(1), nrm);
        m0 = tx * inv; m1 = ty * inv; m2 = -inv;
    }
    // refract! (:21-32)
    const T g = -((r.k0 * m0 + r.k1 * m1) + r.k2 * m2);
    const T D2 = T(1) - s.eta2 * (T(1) - g * g);
    const T cf = s.eta * g - ieee_sqrt(D2);
    const bool ok = D2 >= T(0);                      // TIR / NaN: k untouched (Q1)
    const T n0 = s.eta * r.k0 + cf * m0;
    const T n1 = s.eta * r.k1 + cf * m1;
    const T n2 = s.eta * r.k2 + cf * m2;
    r.k0 = ok ? n0 : r.k0;
    r.k1 = ok ? n1 : r.k1;
    r.k2 = ok ? n2 : r.k2;
    ieee_div2(r.k1, r.k0, r.k2, r.u, r.v);           // :59-60
}

// The same iteration for a FLAT row without a polynomial (R = Inf: stop and image planes, plane windows), with
// the reference's arithmetic evaluated symbolically where its operands are exact:
//   tilt: sign(Inf) x / sqrt(Inf - r^2 (1+K)) = x / Inf = (+-)0 for every finite x, likewise y   (:16-19)
//   normalize!: sqrt(0 + 0 + 1) = 1, inv = 1  ->  m = (0, 0, -1) exactly                          (:56-57)
//   g = -((k0 0 + k1 0) + k2 (-1)) = k2;   eta k0 + cf 0 = eta k0;   eta k2 + cf (-1) = eta k2 - cf  (:23-31)
// so one sqrt and one shared reciprocal remain of the 3 sqrt + 4 divisions of the general body, and every x, y
// the reference produces is reproduced bit for bit.  (A ray whose x or y is already NaN keeps them NaN on every
// later surface whatever its direction, so the reference's "k untouched on NaN" is not observable there.)
template <typename T>
__device__ __forceinline__ void surface_step_ieee_flat(Ray<T>& r, const SurfRec<T>& s)
{
    const T tcur = s.t - r.sprev;
    r.y = r.y + r.u * tcur;                          // :46
    r.x = r.x + r.v * tcur;                          // :47
    r.y = r.y + T(0) * r.u;                          // :52 with s = 0 (:12): NaN for an infinite slope, as there
    r.x = r.x + T(0) * r.v;                          // :53
    r.sprev = T(0);
    const T g = r.k2;
    const T D2 = T(1) - s.eta2 * (T(1) - g * g);     // :24
    const T cf = s.eta * g - ieee_sqrt(D2);          // :26
    const bool ok = D2 >= T(0);
    const T n0 = s.eta * r.k0, n1 = s.eta * r.k1, n2 = s.eta * r.k2 - cf;
    r.k0 = ok ? n0 : r.k0;
    r.k1 = ok ? n1 : r.k1;
    r.k2 = ok ? n2 : r.k2;
    ieee_div2(r.k1, r.k0, r.k2, r.u, r.v);           // :59-60
}

// |a| < lim as ONE compare (source modifier); false for NaN: a ray that missed outright is no near-branch case.
template <typename T> __device__ __forceinline__ bool near_zero(T a, T lim) { return t_abs(a) < lim; }

// MATH_FAST, row with a polynomial term: the reference's slope form (sag :1-14, tilt :16-19)
// with fused arithmetic.  r.sprev is the z offset of the ray point from the current vertex.
// FORM: how the row's polynomial is evaluated (poly_value_fast).  `odd` (see the notes at the top): the sag
// discriminant within kNear R^2 of zero (:6), the point within that of the conic's radius or outside it (:17: NaN tilt in
// the reference, k untouched), the refraction discriminant within kNear of zero (:25), a refracted direction (nearly) backward.
// FINITE = isfinite(R), TIR as for the conic rows (below): both wave-uniform class bits, so the body is straight-line.
template <typename T, int FORM, bool FINITE, bool TIR>
__device__ __forceinline__ void surface_step_fast_poly(Ray<T>& r, const SurfRec<T>& s,
                                                       const T* __restrict__ pl, bool& odd)
{
    const T ik = fast_rcp(r.k2);
    const T u = r.k1 * ik, v = r.k0 * ik;                        // :59-60
    const T tcur = s.t - r.sprev;
    r.y = t_fma<T>(u, tcur, r.y);
    r.x = t_fma<T>(v, tcur, r.x);
    T sg = T(0), is = T(0);
    if (FINITE) {                                    // (:2)
        const T pv = poly_value_fast<T, FORM>(pl, r.y);          // p(y) at the vertex plane (:7, Q2)
        const T beta = t_fma<T>(-r.x, v, t_fma<T>(-r.y, u, s.R));
        const T r2 = t_fma<T>(r.x, r.x, r.y * r.y);
        const T A = t_fma<T>(v, v, t_fma<T>(u, u, s.opk));
        const T D = t_fma<T>(beta, beta, -(r2 * A));
        odd = odd || near_zero<T>(D, s.dlim);
        // D < 0: the root, and with it sg, is NaN (:9) — no select; D = 0 (0 * inf) is a near-branch case, retraced
        sg = t_fma<T>(r2, fast_rcp(t_fma<T>(s.sgn, sqrt_core(D), beta)), pv);
        r.y = t_fma<T>(sg, u, r.y);
        r.x = t_fma<T>(sg, v, r.x);
    }                                                // flat row: sag = 0 without p(y) (:12), tilt = p' only (:18)
    const T xx = r.x * r.x, yy = r.y * r.y;
    if (FINITE) {
        const T rad = t_fma<T>(-(xx + yy), s.opk, s.R2);
        odd = odd || (rad < s.dlim);
        is = s.sgn * fast_rsqrt(rad);
    }
    r.sprev = sg;
    T tx, ty;
    poly_tilt_fast<T, FORM>(pl, r.x, r.y, xx, yy, is, tx, ty);   // Q2: p'(x) on the x slope, p'(y) on the y slope
    // un-normalised normal N = (tx, ty, -1), |N|^2 = n2, gu = -k.N:  with g = gu / |N| the reference's
    //   k' = eta k + (eta g - sqrt(1 - eta^2 (1 - g^2))) N / |N|  =  eta k + (eta gu - sqrt(W)) N / n2,
    //   W = (1 - eta^2) n2 + eta^2 gu^2 = n2 (1 - eta^2 (1 - g^2)):  the root and the reciprocal are independent of each
    // other (one refined reciprocal root and its two products less, and the two seeds issue back to back)
    const T n2 = t_fma<T>(tx, tx, t_fma<T>(ty, ty, T(1)));
    const T gu = t_fma<T>(-r.k1, ty, t_fma<T>(-r.k0, tx, r.k2));
    const T W = t_fma<T>(s.eta2 * gu, gu, s.ome2 * n2);
    // TIR rows: total internal reflection (the reference leaves k untouched, Q1) and its neighbourhood (:25) are left to
    // the reference sequence: 1 - eta^2 (1 - g^2) < thr  <=>  W < thr n2
    if (TIR) odd = odd || (W < Near<T>::thr * n2);
    const T cf = t_fma<T>(s.eta, gu, -sqrt_core(W)) * fast_rcp(n2);
    const T ee = s.eta;
    // product on the OLD component first, then accumulate into it: the two-address v_fmac then updates k in
    // place (the other association lands in a temporary and costs a v_mov per component)
    r.k0 = t_fma<T>(cf, tx, ee * r.k0);
    r.k1 = t_fma<T>(cf, ty, ee * r.k1);
    r.k2 = t_fma<T>(ee, r.k2, -cf);
    odd = odd || (r.k2 < Near<T>::root);
}

// The same row for the TWO rays of a lane in lockstep (A/B build, ORT_POLY_INTERLEAVE): the arm is one dependent chain per ray
// through five hardware seeds (v_rcp / v_rsq: the longest latencies of the loop), and the compiler, left alone, lays ray 0's
// whole chain ahead of ray 1's (register pressure decides).  Here the chain is cut right after every seed issue and the
// stages alternate between the rays — stage k of ray 1 runs while the seed of ray 0's stage k is in flight — with a
// scheduling barrier between stages so they stay where they are put.  Same operations on the same values as
// surface_step_fast_poly (finite rows): same bits.
template <typename T> __device__ __forceinline__ T seed_rcp(T a);
template <> __device__ __forceinline__ double seed_rcp<double>(double a) { return __builtin_amdgcn_rcp(a); }
template <> __device__ __forceinline__ float seed_rcp<float>(float a) { return __builtin_amdgcn_rcpf(a); }
template <typename T> __device__ __forceinline__ T seed_rsq(T a);
template <> __device__ __forceinline__ double seed_rsq<double>(double a) { return __builtin_amdgcn_rsq(a); }
template <> __device__ __forceinline__ float seed_rsq<float>(float a) { return __builtin_amdgcn_rsqf(a); }
__device__ __forceinline__ double refine_rcp(double a, double r0) { return __builtin_fma(r0, __builtin_fma(-a, r0, 1.0), r0); }
__device__ __forceinline__ float refine_rcp(float, float r0) { return r0; }
__device__ __forceinline__ double refine_sqrt(double a, double r0)          // sqrt_core from its seed
{
    const double g = a * r0, h = 0.5 * r0;
    return __builtin_fma(g, __builtin_fma(-h, g, 0.5), g);
}
__device__ __forceinline__ float refine_sqrt(float a, float) { return __builtin_amdgcn_sqrtf(a); }
__device__ __forceinline__ double refine_rsqrt(double a, double r0)         // fast_rsqrt from its seed
{
    const double t = a * r0, h = 0.5 * r0;
    return __builtin_fma(h, __builtin_fma(-t, r0, 1.0), r0);
}
__device__ __forceinline__ float refine_rsqrt(float, float r0) { return r0; }

template <typename T, int FORM, bool TIR>
__device__ __forceinline__ void surface_step_fast_poly2(Ray<T> (&r)[2], const SurfRec<T>& s, const T* __restrict__ pl, bool& odd)
{
#define ORT_STAGE(...) { _Pragma("unroll") for (int q = 0; q < 2; ++q) { __VA_ARGS__ } __builtin_amdgcn_sched_barrier(0); }
    T sd[2], u[2], v[2], pv[2], beta[2], r2[2], D[2], den[2], rad[2], tx[2], ty[2], n2[2], gu[2], W[2], sd2[2];
    ORT_STAGE(sd[q] = seed_rcp<T>(r[q].k2);)
    ORT_STAGE(
        const T ik = refine_rcp(r[q].k2, sd[q]);
        u[q] = r[q].k1 * ik; v[q] = r[q].k0 * ik;                  // :59-60
        const T tcur = s.t - r[q].sprev;
        r[q].y = t_fma<T>(u[q], tcur, r[q].y);
        r[q].x = t_fma<T>(v[q], tcur, r[q].x);
        pv[q] = poly_value_fast<T, FORM>(pl, r[q].y);
        beta[q] = t_fma<T>(-r[q].x, v[q], t_fma<T>(-r[q].y, u[q], s.R));
        r2[q] = t_fma<T>(r[q].x, r[q].x, r[q].y * r[q].y);
        const T A = t_fma<T>(v[q], v[q], t_fma<T>(u[q], u[q], s.opk));
        D[q] = t_fma<T>(beta[q], beta[q], -(r2[q] * A));
        odd = odd || near_zero<T>(D[q], s.dlim);
        sd[q] = seed_rsq<T>(D[q]);)
    ORT_STAGE(
        den[q] = t_fma<T>(s.sgn, refine_sqrt(D[q], sd[q]), beta[q]);
        sd[q] = seed_rcp<T>(den[q]);)
    ORT_STAGE(
        const T sg = t_fma<T>(r2[q], refine_rcp(den[q], sd[q]), pv[q]);
        r[q].y = t_fma<T>(sg, u[q], r[q].y);
        r[q].x = t_fma<T>(sg, v[q], r[q].x);
        r[q].sprev = sg;
        const T xx = r[q].x * r[q].x, yy = r[q].y * r[q].y;
        rad[q] = t_fma<T>(-(xx + yy), s.opk, s.R2);
        odd = odd || (rad[q] < s.dlim);
        sd[q] = seed_rsq<T>(rad[q]);)
    ORT_STAGE(
        const T is = s.sgn * refine_rsqrt(rad[q], sd[q]);
        poly_tilt_fast<T, FORM>(pl, r[q].x, r[q].y, r[q].x * r[q].x, r[q].y * r[q].y, is, tx[q], ty[q]);
        n2[q] = t_fma<T>(tx[q], tx[q], t_fma<T>(ty[q], ty[q], T(1)));
        gu[q] = t_fma<T>(-r[q].k1, ty[q], t_fma<T>(-r[q].k0, tx[q], r[q].k2));
        W[q] = t_fma<T>(s.eta2 * gu[q], gu[q], s.ome2 * n2[q]);
        if (TIR) odd = odd || (W[q] < Near<T>::thr * n2[q]);
        sd[q] = seed_rsq<T>(W[q]); sd2[q] = seed_rcp<T>(n2[q]);)
    ORT_STAGE(
        const T cf = t_fma<T>(s.eta, gu[q], -refine_sqrt(W[q], sd[q])) * refine_rcp(n2[q], sd2[q]);
        r[q].k0 = t_fma<T>(cf, tx[q], s.eta * r[q].k0);
        r[q].k1 = t_fma<T>(cf, ty[q], s.eta * r[q].k1);
        r[q].k2 = t_fma<T>(s.eta, r[q].k2, -cf);
        odd = odd || (r[q].k2 < Near<T>::root);)
#undef ORT_STAGE
}

// MATH_FAST, conic row (sphere, flat, conic) in direction-cosine form.  With the ray point
// P0 = (x, y, z0) relative to the row's vertex and unit direction k (k0 <-> x, k1 <-> y,
// k2 <-> z), the conic  c (x^2 + y^2 + (1+K) z^2) - 2 z = 0  is met at path length
//     d = F / (G + sqrt(G^2 - a F)),   F = c (x^2 + y^2 + (1+K) z0^2) - 2 z0,
//     G = k2 - c (x k0 + y k1 + (1+K) z0 k2),   a = c (1 + K k2^2)
// (the root that tends to -z0/k2 as c -> 0: the vertex-side sheet the reference picks with
// sign(R), PupilSampling.jl:7).  The unnormalised normal is (-c x, -c y, 1 - c (1+K) z); on
// a sphere it is already unit and k.n equals the square root above.
// REFR: the row refracts (eta != 1).  TIR: the row is NOT in 0 < eta <= 1 — eta > 1, where total internal
// reflection is possible and the reference's "k untouched" rule (Q1) needs a select, or a mirror's eta < 0; for
// 0 < eta <= 1 the radicand 1 - eta^2 (1 - cos^2 I) >= 1 - eta^2 >= 0, no select is emitted, and the refracted
// direction cannot point backward.
template <typename T> struct ConicHit { T cn, n2, cosi, cos2; };   // n = (-cn x, -cn y, n2), cos I = k.n

// Transfer to the row and intersection.  Returns false when the row does not refract (flat, eta == 1).
// `odd` (curved rows): the discriminant E2 (= cos^2 of incidence, dimensionless) within kNear of zero — the miss branch
// (:6) —, and the hit at or beyond the equator / the conic's radius (axial component of the normal below sqrt(kNear)).
template <typename T, int KIND, bool REFR>
__device__ __forceinline__ bool fast_conic_hit(Ray<T>& r, const SurfRec<T>& s, ConicHit<T>& h, bool& odd)
{
    const T z0 = r.sprev - s.t;
    const T c = s.invR;
    if (KIND == KIND_FLAT) {
        const T d = -z0 * fast_rcp(r.k2);
        r.x = t_fma<T>(d, r.k0, r.x);
        r.y = t_fma<T>(d, r.k1, r.y);
        r.sprev = T(0);
        if (!REFR) return false;
        h.cn = T(0); h.n2 = T(1); h.cosi = r.k2; h.cos2 = r.k2 * r.k2;
    } else if (KIND == KIND_SPHERE) {
        const T Pk = t_fma<T>(z0, r.k2, t_fma<T>(r.y, r.k1, r.x * r.k0));
        const T P2 = t_fma<T>(z0, z0, t_fma<T>(r.y, r.y, r.x * r.x));
        const T F = t_fma<T>(c, P2, T(-2) * z0);
        const T G = t_fma<T>(-c, Pk, r.k2);
        const T E2 = t_fma<T>(G, G, -(c * F));
        odd = odd || near_zero<T>(E2, Near<T>::thr);
        const T E = sqrt_core(E2);                            // NaN when the ray misses (:9); E2 = 0 is retraced
        const T d = F * fast_rcp(G + E);
        r.x = t_fma<T>(d, r.k0, r.x);
        r.y = t_fma<T>(d, r.k1, r.y);
        const T z = t_fma<T>(d, r.k2, z0);
        r.sprev = z;
        h.cn = c; h.n2 = t_fma<T>(-c, z, T(1));                  // unit GEOMETRIC normal (-c x, -c y, 1 - c z)
        odd = odd || (h.n2 < Near<T>::root);                     // at / beyond the equator: the reference keeps the vertex-side slope
        h.cosi = E; h.cos2 = E2;                                 // k.n = sqrt(G^2 - c F)
    } else {
        const T zk = s.opk * z0;
        const T Pk = t_fma<T>(zk, r.k2, t_fma<T>(r.y, r.k1, r.x * r.k0));
        const T P2 = t_fma<T>(zk, z0, t_fma<T>(r.y, r.y, r.x * r.x));
        const T F = t_fma<T>(c, P2, T(-2) * z0);
        const T G = t_fma<T>(-c, Pk, r.k2);
        const T a = c * t_fma<T>(s.K * r.k2, r.k2, T(1));
        const T E2 = t_fma<T>(G, G, -(a * F));
        odd = odd || near_zero<T>(E2, Near<T>::thr);
        const T E = sqrt_core(E2);
        const T d = F * fast_rcp(G + E);
        r.x = t_fma<T>(d, r.k0, r.x);
        r.y = t_fma<T>(d, r.k1, r.y);
        const T z = t_fma<T>(d, r.k2, z0);
        r.sprev = z;
        // |N2|: the reference's tilt is the slope of the VERTEX-side sheet (PupilSampling.jl:16-19), whose normal
        // has a positive axial component also where the chosen root lies on the far sheet (free source modifier);
        // N2^2 = (R^2 - r^2 (1+K)) / R^2, the tilt's radicand (:17): near zero the slope blows up -> retraced
        const T N0 = c * r.x, N1 = c * r.y, N2 = t_abs(t_fma<T>(-c * s.opk, z, T(1)));
        odd = odd || (N2 < Near<T>::root);
        const T inv = fast_rsqrt(t_fma<T>(N2, N2, t_fma<T>(N1, N1, N0 * N0)));
        h.cn = c * inv; h.n2 = N2 * inv;
        h.cosi = t_fma<T>(r.k2, h.n2, -h.cn * t_fma<T>(r.k1, r.y, r.k0 * r.x));
        h.cos2 = h.cosi * h.cosi;
    }
    return true;
}

// vector Snell (:21-32) with n = -m:  k' = eta k + (cos I' - eta cos I) n
//   1 - eta^2 (1 - cos^2 I) = (1 - eta^2) + eta^2 cos^2 I
// TIR rows: a radicand within kNear of zero (:25) raises `odd`.
template <typename T, bool TIR>
__device__ __forceinline__ void fast_snell(Ray<T>& r, const SurfRec<T>& s, const ConicHit<T>& h, bool& odd)
{
    const T D2 = t_fma<T>(s.eta2, h.cos2, s.ome2);
    // TIR rows: D2 < 0 (k untouched in the reference, Q1) and the neighbourhood of the branch (:25) raise `odd` — the
    // wave retraces with the reference sequence, so no select here; elsewhere D2 >= 1 - eta^2 > 0.  NaN passes through.
    if (TIR) odd = odd || (D2 < Near<T>::thr);
    const T gam = t_fma<T>(-s.eta, h.cosi, sqrt_core(D2));
    const T gc = gam * h.cn;
    r.k0 = t_fma<T>(-gc, r.x, s.eta * r.k0);                     // in-place form, see surface_step_fast_poly
    r.k1 = t_fma<T>(-gc, r.y, s.eta * r.k1);
    r.k2 = t_fma<T>(gam, h.n2, s.eta * r.k2);
}

// `odd`: see the MATH_FAST notes at the top.  Flat rows raise it only through the TIR radicand (k2' = cos I' >= 0);
// curved rows also test the refracted k2 (one compare).
template <typename T, int KIND, bool REFR, bool TIR>
__device__ __forceinline__ void surface_step_fast_conic(Ray<T>& r, const SurfRec<T>& s, bool& odd)
{
    ConicHit<T> h;
    if (!fast_conic_hit<T, KIND, REFR>(r, s, h, odd)) return;
    fast_snell<T, TIR>(r, s, h, odd);
    // backward direction: impossible for 0 < eta <= 1 (gam >= 0 and the normal's axial component is positive once
    // far-cap hits are out), so only the rows of the TIR class (eta > 1, or a mirror's eta < 0) are tested
    if (KIND != KIND_FLAT && TIR) odd = odd || (r.k2 < Near<T>::root);
}

// MATH_FAST, strongly curved sphere (|R| <= kCentreFormMaxR) in CENTRE form: with Q = P - C
// (C = centre of curvature) the sphere is |Q|^2 = R^2, the vertex-side root of the quadratic is
//     d = -b - sign(R) sqrt(b^2 - |Q0|^2 + R^2),   b = Q0 . k,
// no reciprocal at all; the unit normal is -Q/R and cos I = |c| sqrt(...).  d is a difference of
// two numbers of size |R|, so its absolute error is ~eps |R|: kept to rows where that is
// << 1e-10 x 1 mm (measured <= 2e-13 relative on the Tessar's R = -275.7 row).
// s.K holds t + R for these rows.
constexpr double kCentreFormMaxR = 1.0e3;
// Float32: eps is 2^29 times larger, so the form is kept to |R| <= 200 mm, where it costs nothing measurable — image-plane
// hits of BASELINE config 5's systems against the Float64 trace: rms 8.2e-6 mm (7.7e-6 in the vertex form, 7.2e-6 for the
// Float32 reference sequence: the rounding of the inputs dominates); 1.5e-5 mm with the Float64 limit
// (profiles/r03_ab_config5_f32.log; -4.5 % on ort_spot_batch_f32)
constexpr double kCentreFormMaxR32 = 200.0;

template <typename T>
__device__ __forceinline__ void fast_sphere_c_hit(Ray<T>& r, const SurfRec<T>& s, T& sq, T& disc, T& Qz)
{
    const T Qz0 = r.sprev - s.K;
    const T b = t_fma<T>(Qz0, r.k2, t_fma<T>(r.y, r.k1, r.x * r.k0));
    const T q = t_fma<T>(Qz0, Qz0, t_fma<T>(r.y, r.y, r.x * r.x));
    disc = t_fma<T>(b, b, s.R2 - q);
    sq = sqrt_core(disc);                                        // NaN when the ray misses (:9); disc = 0 is retraced
    const T d = -t_fma<T>(s.sgn, sq, b);
    r.x = t_fma<T>(d, r.k0, r.x);
    r.y = t_fma<T>(d, r.k1, r.y);
    Qz = t_fma<T>(d, r.k2, Qz0);
    r.sprev = Qz + s.R;
}

// Snell with the geometric normal -Q/R (every hit on the vertex-side cap): D2 = radicand of cos I',
// gam = cos I' - eta cos I,  k' = eta k + gam n.
template <typename T>
__device__ __forceinline__ void fast_sphere_c_refract(Ray<T>& r, const SurfRec<T>& s, T sq, T Qz, T D2)
{
    const T gam = t_fma<T>(-s.ec, sq, sqrt_core(D2));            // cos I = |c| sq
    const T gc = gam * s.invR;
    r.k0 = t_fma<T>(-gc, r.x, s.eta * r.k0);                     // in-place form, see surface_step_fast_poly
    r.k1 = t_fma<T>(-gc, r.y, s.eta * r.k1);
    r.k2 = t_fma<T>(-gc, Qz, s.eta * r.k2);
}

// Centre-form row.  `odd` (see the MATH_FAST notes at the top), one compare each: the discriminant within kNear R^2 of
// zero (the miss branch, :6); the sag |z| = |Qz + R| beyond |R| (1 - sqrt(kNear)) — the hit at or past the equator
// (z / R runs from 0 at the vertex to 1 at the equator and 2 at the far pole), NaN-safe; TIR rows: the refraction
// radicand below kNear — total internal reflection itself and the neighbourhood of its branch (:25) — and a refracted
// k2 that is not clearly forward.
template <typename T, bool TIR>
__device__ __forceinline__ void surface_step_fast_sphere_c(Ray<T>& r, const SurfRec<T>& s, bool& odd)
{
    T sq, disc, Qz;
    fast_sphere_c_hit<T>(r, s, sq, disc, Qz);
    odd = odd || near_zero<T>(disc, s.dlim);
    odd = odd || (t_abs(r.sprev) > s.zlim);
    const T D2 = t_fma<T>(s.e2c2, disc, s.ome2);                 // (1 - eta^2) + eta^2 cos^2 I
    if (TIR) odd = odd || (D2 < Near<T>::thr);                   // TIR and its neighbourhood: see fast_snell
    fast_sphere_c_refract<T>(r, s, sq, Qz, D2);
    if (TIR) odd = odd || (r.k2 < Near<T>::root);                // see surface_step_fast_conic
}

// ---- table construction, shared by the host (ort_system_create) and the device (k_build_tables) -----------------
// One record from row i+1 of the prescription as loop iteration i sees it.  nc = coefficients in use (0: p = zero),
// pcls = the CLS_P* bits of the row's polynomial record.  Every derived field is ONE IEEE operation on the row's data,
// the same on both sides.
// Returns the row's NEED_* bits (below): which arms of the surface loop it takes.
template <typename T>
__host__ __device__ inline int make_rec(SurfRec<T>& r, T t, T Rv, T n1, T n2, T Kv, int nc, int pcls)
{
    r.t = t; r.R = Rv; r.R2 = Rv * Rv;
    r.sgn = Rv > T(0) ? T(1) : (Rv < T(0) ? T(-1) : Rv);
    r.opk = T(1) + Kv; r.eta = n1 / n2; r.eta2 = r.eta * r.eta; r.K = Kv;
    r.finite = __builtin_isfinite(Rv) ? 1 : 0;
    r.invR = r.finite ? T(1) / Rv : T(0);
    const T absc = r.invR < T(0) ? -r.invR : r.invR, absR = Rv < T(0) ? -Rv : Rv;
    r.ome2 = T(1) - r.eta2; r.e2c2 = r.eta2 * (r.invR * r.invR); r.ec = r.eta * absc;
    r.dlim = r.finite ? (T)Near<T>::thr * r.R2 : T(0);
    r.zlim = r.finite ? absR * (T(1) - (T)Near<T>::root) : T(0);
    r.ncoef = nc; r.spare = 0;
    int kind = nc > 0 ? KIND_POLY : (!r.finite ? KIND_FLAT : (Kv != T(0) ? KIND_CONIC : KIND_SPHERE));
    if (kind == KIND_SPHERE && (double)absR <= (sizeof(T) == 8 ? kCentreFormMaxR : kCentreFormMaxR32)) {
        kind = KIND_SPHERE_C;                  // MATH_FAST centre form; K (== 0 here) carries t + R
        r.K = r.t + Rv;
    }
    r.cls = (r.finite ? CLS_FINITE : 0) | (nc > 0 ? CLS_HASP : 0) | ((r.eta != T(1)) ? CLS_REFR : 0) |
            (!(r.eta > T(0) && r.eta <= T(1)) ? CLS_TIR : 0) | (kind << CLS_KIND_SHIFT) | (nc > 0 ? pcls : 0);
    // what this row asks of the kernel build (NEED_* below): a polynomial row in even form on a curved base has its own arms
    if (kind == KIND_POLY) return (r.finite && (pcls & CLS_PEVEN)) ? 2 /* NEED_EVEN */ : 4 /* NEED_POLY */;
    return (kind == KIND_CONIC || (kind == KIND_SPHERE && sizeof(T) == 8)) ? 1 /* NEED_GENERAL */ : 0;
}

// One polynomial record (layout: kPolyRec above) from a raw coefficient row c[0 .. ncoef); returns its CLS_P* bits.
// `nc_out`: the table width when any coefficient is non-zero AFTER the cast to T, else 0 (an all-zero row is the
// reference's `zero` polynomial).
template <typename T>
__host__ __device__ inline int make_poly_rec(T* rec, const double* c, int ncoef, int* nc_out)
{
    for (int j = 0; j < kPolyRec; ++j) rec[j] = T(0);
    int top = -1; bool odd_nz = false;
    for (int j = 0; j < ncoef && j < kPolyMax; ++j) {
        const T v = c ? (T)c[j] : T(0);
        rec[j] = v;
        if (v != T(0)) { top = j; odd_nz = odd_nz || (j & 1); }
    }
    for (int j = 0; j + 1 < kPolyMax; ++j) rec[kPolyMax + j] = T(j + 1) * rec[j + 1];
    for (int k = 0; k < 6; ++k) rec[24 + k] = rec[2 * k];
    for (int k = 0; k < 5; ++k) rec[30 + k] = T(2 * k + 2) * rec[2 * k + 2];
    *nc_out = top >= 0 ? ncoef : 0;
    int cls = odd_nz ? 0 : CLS_PEVEN;
    if ((!odd_nz && top > 6) || (odd_nz && top > 7)) cls |= CLS_PBIG;
    return cls;
}

// All N rays of a lane through one surface.  `cls` packs the row's wave-uniform class bits
// (scalar register): the branch is taken once per surface, the bodies are straight-line.
// odd (MATH_FAST only): raised when a ray of this lane leaves the domain of the fast forms (see the notes at the top).
// last (MATH_IEEE only, wave-uniform): this is the final loop iteration, see surface_step_ieee.
// ARMS: which row classes the kernel build carries (make_rec's `arms` level of a row; a batch runs the build of its
// highest row).  Every arm that shares the surface loop costs the others register copies at the loop's merge points
// (the allocator gives the loop-carried ray state different homes in different arms), so the loop of a plain
// spherical system holds the centre-form sphere and flat arms and NOTHING else:
//   ARMS_BASIC    centre-form spheres (Float32: + general-form spheres) + flat rows
//   ARMS_GENERAL  + general-form spheres (Float64 rows with |R| > kCentreFormMaxR) and conics
//   ARMS_EVEN     ARMS_BASIC + even aspheres on a curved base (polynomial rows whose odd coefficients are all zero, finite
//                 R: p(y) = E(y^2) of <= 4 / <= 6 terms), as independent arms — the usual aspheric lens; without the
//                 general arm in the loop its sphere / flat arms carry no merge copies (7 % fewer instructions on config 3)
//   ARMS_POLY     everything: + general-form spheres, conics and general polynomial rows (one grouped arm).  Polynomial
//                 rows: two interleaved ~100-instruction chains + the row's coefficients need the 128-VGPR budget
//                 (k_trace's launch bounds); at 96 they park tens of values per row in scratch
enum { ARMS_BASIC = 0, ARMS_GENERAL = 1, ARMS_EVEN = 2, ARMS_POLY = 3 };
// what a row needs (make_rec's return value, OR-ed over a batch's rows) -> the build that runs the batch
enum { NEED_GENERAL = 1, NEED_EVEN = 2, NEED_POLY = 4 };
__host__ __device__ inline int arms_of_needs(int m)
{
    if ((m & NEED_POLY) || ((m & NEED_EVEN) && (m & NEED_GENERAL))) return ARMS_POLY;
    return (m & NEED_EVEN) ? ARMS_EVEN : (m & NEED_GENERAL) ? ARMS_GENERAL : ARMS_BASIC;
}
template <typename T, int MATH, int N, int ARMS>
__device__ __forceinline__ void surface_step_n(Ray<T> (&r)[N], const SurfRec<T>& s,
                                               const T* __restrict__ pl, int cls, bool last, bool& odd)
{
#define ORT_ALL_RAYS(call) _Pragma("unroll") for (int q = 0; q < N; ++q) { call; }
    if (MATH == MATH_IEEE) {
        // hot arms first (spherical / conic rows and flat rows without a polynomial), as independent ifs; rows
        // with a polynomial and the final iteration share one grouped arm
        const bool fin = cls & CLS_FINITE, hasp = ARMS >= ARMS_EVEN && (cls & CLS_HASP);
        if (fin && !hasp && !last)  { ORT_ALL_RAYS((surface_step_ieee<T, true, false, false>(r[q], s, pl))) }
        if (!fin && !hasp && !last) { ORT_ALL_RAYS((surface_step_ieee_flat<T>(r[q], s))) }
        if (hasp || last) {
            if (last) {
                if (fin) {
                    if (hasp) { ORT_ALL_RAYS((surface_step_ieee<T, true, true, true>(r[q], s, pl))) }
                    else      { ORT_ALL_RAYS((surface_step_ieee<T, true, false, true>(r[q], s, pl))) }
                } else        { ORT_ALL_RAYS((surface_step_ieee<T, false, false, true>(r[q], s, pl))) }
            } else {
                if (fin) { ORT_ALL_RAYS((surface_step_ieee<T, true, true, false>(r[q], s, pl))) }
                else     { ORT_ALL_RAYS((surface_step_ieee<T, false, true, false>(r[q], s, pl))) }
            }
        }
    } else {
        // INDEPENDENT ifs on scalar conditions, not an else-if chain: each arm merges only with its own skip
        // path, which the register coalescer joins with the loop-carried state (updates in place); a multi-arm
        // merge costs a v_mov_b64 per state component per surface.
        const int kind = (cls >> CLS_KIND_SHIFT) & 7;
        const bool tir = cls & CLS_TIR;
        const bool refr = cls & CLS_REFR;
        if (kind == KIND_SPHERE_C && !tir) { ORT_ALL_RAYS((surface_step_fast_sphere_c<T, false>(r[q], s, odd))) }
        if (kind == KIND_SPHERE_C && tir)  { ORT_ALL_RAYS((surface_step_fast_sphere_c<T, true>(r[q], s, odd))) }
        if (kind == KIND_FLAT && !refr)    { ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_FLAT, false, false>(r[q], s, odd))) }
        if (kind == KIND_FLAT && refr)     { ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_FLAT, true, true>(r[q], s, odd))) }
        // Float32 keeps the centre form to |R| <= kCentreFormMaxR32 (cancellation), so the general sphere arms are hot arms too
        constexpr bool kF32 = sizeof(T) == 4;
        if (kF32 && kind == KIND_SPHERE && !tir) { ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_SPHERE, true, false>(r[q], s, odd))) }
        if (kF32 && kind == KIND_SPHERE && tir)  { ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_SPHERE, true, true>(r[q], s, odd))) }
        if (ARMS == ARMS_EVEN) {
            // even aspheres on a curved base (the usual case): four independent arms, nothing else in the loop
            const bool big = cls & CLS_PBIG;
#if defined(ORT_POLY_INTERLEAVE) && ORT_POLY_INTERLEAVE
            if constexpr (N == 2) {                              // A/B build: the two rays' seed chains in lockstep
                if (kind == KIND_POLY && !big && !tir) surface_step_fast_poly2<T, 0, false>(r, s, pl, odd);
                if (kind == KIND_POLY && !big && tir)  surface_step_fast_poly2<T, 0, true>(r, s, pl, odd);
                if (kind == KIND_POLY && big && !tir)  surface_step_fast_poly2<T, 1, false>(r, s, pl, odd);
                if (kind == KIND_POLY && big && tir)   surface_step_fast_poly2<T, 1, true>(r, s, pl, odd);
            } else
#endif
            {
            if (kind == KIND_POLY && !big && !tir) { ORT_ALL_RAYS((surface_step_fast_poly<T, 0, true, false>(r[q], s, pl, odd))) }
            if (kind == KIND_POLY && !big && tir)  { ORT_ALL_RAYS((surface_step_fast_poly<T, 0, true, true>(r[q], s, pl, odd))) }
            if (kind == KIND_POLY && big && !tir)  { ORT_ALL_RAYS((surface_step_fast_poly<T, 1, true, false>(r[q], s, pl, odd))) }
            if (kind == KIND_POLY && big && tir)   { ORT_ALL_RAYS((surface_step_fast_poly<T, 1, true, true>(r[q], s, pl, odd))) }
            }
        }
        if ((ARMS == ARMS_GENERAL || ARMS == ARMS_POLY) &&
            ((!kF32 && kind == KIND_SPHERE) || kind == KIND_CONIC || (ARMS == ARMS_POLY && kind == KIND_POLY)))
        {   // the general forms share ONE arm:
            // as independent arms they drag their merge copies back onto the path of the sphere / flat rows (measured)
            if (!kF32 && kind == KIND_SPHERE) {
                if (tir) { ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_SPHERE, true, true>(r[q], s, odd))) }
                else     { ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_SPHERE, true, false>(r[q], s, odd))) }
            } else if (kind == KIND_CONIC) {
                ORT_ALL_RAYS((surface_step_fast_conic<T, KIND_CONIC, true, true>(r[q], s, odd)))
            } else if (ARMS == ARMS_POLY) {
                // polynomial rows: FORM = how p is evaluated (even form needs a finite R: the staged block is ev | qd only then)
                const bool fin = cls & CLS_FINITE, big = cls & CLS_PBIG, even = (cls & CLS_PEVEN) && fin;
#define ORT_POLY_ARM(FORM, FIN) { if (tir) { ORT_ALL_RAYS((surface_step_fast_poly<T, FORM, FIN, true>(r[q], s, pl, odd))) } \
                                  else     { ORT_ALL_RAYS((surface_step_fast_poly<T, FORM, FIN, false>(r[q], s, pl, odd))) } }
                if (even && !big)             ORT_POLY_ARM(0, true)
                else if (even)                ORT_POLY_ARM(1, true)
                else if (fin && !big)         ORT_POLY_ARM(2, true)
                else if (fin)                 ORT_POLY_ARM(3, true)
                else                          ORT_POLY_ARM(3, false)
#undef ORT_POLY_ARM
            }
        }
    }
#undef ORT_ALL_RAYS
}

}  // namespace ort

/*
 * ort_oracle.h — CPU oracle for the batched ray-trace hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (Sagnac/OpticalRayTracing.jl v1.0.0), followed line by line with
 * every quirk kept.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (opticalraytracing.jl_amd/) never does.
 *
 * Parity pin: the reference is Julia source and no Julia runtime exists in the build
 * container, so the restatement is pinned by the reference's own known-answer tests
 * (test/runtests.jl, see tests/test_oracle_reference_vectors.py) and by an independent
 * 50-digit mpmath restatement (oracle/mp_model.py -> tests/golden/).  Items the
 * reference's tests do not pin (Julia Base `range`, BLAS ddot rounding, Optim BFGS
 * end points) are marked "parity unpinned" where they are restated.
 *
 * Build: gcc -O2 -ffp-contract=off (Julia never contracts a*b+c into an fma).
 * All citations are into /root/reference/.
 */
#ifndef ORT_ORACLE_H
#define ORT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Surface prescription, column form of the reference's `surfaces` matrix
 * (columns R, t, n; src/PupilSampling.jl:36) plus conic constants K[rows] and an
 * additive polynomial p_i(y) = sum_j coef[i*ncoef + j] * y^j  (Horner order).
 * K == NULL  -> zeros;  coef == NULL or ncoef == 0 -> p = zero.                      */

/* src/PupilSampling.jl:34-65 — one skew ray.  xv, yv have rows-1 entries.           */
void orc_trace_skew(int rows, const double *R, const double *t, const double *n,
                    const double *K, const double *coef, int ncoef,
                    double y, double x, double U, double V,
                    double *xv, double *yv);

/* Same loop with slopes u = tan U, v = tan V supplied by the caller (the tangent is
 * the only libm call in the loop; lets a test remove libm differences).             */
void orc_trace_skew_slopes(int rows, const double *R, const double *t, const double *n,
                           const double *K, const double *coef, int ncoef,
                           double y, double x, double u, double v,
                           double *xv, double *yv, int *tir_count);

/* Conditioning probe (see ort_oracle_skew.inc, skew_margins): marg[4*r + 0..3] = minimum normalised sag
 * discriminant, refraction discriminant, tilt radicand along ray r, and its count of far-cap hits.  Slopes in. */
void orc_skew_margins(int rows, const double *R, const double *t, const double *n,
                      const double *K, const double *coef, int ncoef,
                      int64_t nrays, const double *y, const double *x, const double *u, const double *v,
                      double *marg);

/* Derived per-ray status: 1-based loop index of the first surface whose
 * x or y is NaN, or rows (= S+1) when the ray reached the last row.                  */
int orc_status(int S, const double *xv, const double *yv);

/* Many rays, explicit lists; history written surface-major: xv[s*ld + r].           */
void orc_trace_skew_batch(int rows, const double *R, const double *t, const double *n,
                          const double *K, const double *coef, int ncoef,
                          int64_t nrays, const double *y, const double *x,
                          const double *U, const double *V,
                          double *xv, double *yv, int64_t ld, int32_t *status,
                          int nthreads);

/* Pupil-grid bundle (src/PupilSampling.jl:121-128): y outer, x inner, shared angles.
 * Returns the number of ray-surface intersections executed.  Any output may be NULL. */
int64_t orc_trace_skew_grid(int rows, const double *R, const double *t, const double *n,
                            const double *K, const double *coef, int ncoef,
                            int ny, const double *yaxis, int nx, const double *xaxis,
                            double U, double V,
                            double *xv, double *yv, int64_t ld, int32_t *status,
                            int nthreads);

/* src/RayTracing.jl:145-169 — one meridional ray.  y_out, U_out, ts_out: rows entries
 * (row 0 = input ray; ts_out = per-ray distances, z = cumsum(ts) per Types.jl:61-63).
 * layout_mode != 0 reproduces Layout input (p never ≡ zero -> always atan, Q16).     */
/* Returns 0, or the 1-based row of the first asin(y / R) with |y / R| > 1, where Base.asin throws a DomainError
 * (:162) and this restatement continues with NaN.                                                           */
int orc_trace_meridional(int rows, const double *R, const double *t, const double *n,
                         const double *K, const double *coef, int ncoef,
                         int layout_mode, double y, double U,
                         double *y_out, double *U_out, double *ts_out);

/* src/RayTracing.jl:38-53 — Lens(surfaces).  MUTATES t[0] like the reference (Q19).
 * tau, phi need `rows` entries; returns k = number of lens rows kept.                */
int orc_lens_from_surfaces(int rows, const double *R, double *t, const double *n,
                           double *tau, double *phi);

/* solve(surfaces, a, h′) + aberrations(surfaces, system, λ, δn) + incidences(surfaces, system):
 * src/RayTracing.jl:208-221, 246-263, 302-323, 338-353 and src/SeidelAberrations.jl:6-53.                */
typedef struct orc_system_t {
    double f, EBFD, EFFD, N, FOV, EP_D, EP_t, XP_D, XP_t, H, PN;
    double W040, W131, W222, W220, W311, W020, W111, W220P, W220M, W220T;     /* waves at lambda */
    int stop, k;
} orc_system_t;
enum { ORC_SURF_SPHERICAL = 0, ORC_SURF_COMA, ORC_SURF_ASTIGMATISM, ORC_SURF_SAGITTAL, ORC_SURF_DISTORTION,
       ORC_SURF_AXIAL, ORC_SURF_LATERAL, ORC_SURF_PETZVAL, ORC_SURF_MEDIAL, ORC_SURF_TANGENTIAL, ORC_SURF_COUNT };
int orc_solve_aberrations(int rows, const double *R, const double *t, const double *n,
                          const double *a, const double *dn, double hprime, double lambda,
                          orc_system_t *out, double *surf /* [ORC_SURF_COUNT][rows-1] */,
                          double *inc /* [4][rows-1]: ni, nibar, i, ibar */,
                          double *marg_y, double *marg_nu, double *chief_y, double *chief_nu /* [k+2] each */);

/* src/RayTracing.jl:127-143 — paraxial y-nu trace; rt_y, rt_w have k+1 entries.
 * a may be NULL (fill(Inf)).                                                         */
void orc_trace_paraxial(int k, const double *tau, const double *phi,
                        double y, double w, const double *a, int clip,
                        double *rt_y, double *rt_w);

/* src/TransferMatrix.jl:1-6 — ABCD product, M row-major {A,B,C,D}.                   */
void orc_abcd(int k, const double *tau, const double *phi, double *M);
/* src/TransferMatrix.jl:8 */
void orc_extend(const double *M, double tau, double tau_p, double *out);
/* src/TransferMatrix.jl:10 */
void orc_transfer(const double *M, const double *v, double tau, double tau_p, double *out);
/* src/TransferMatrix.jl:13 — `\` = LU with partial pivoting on the 2x2.              */
void orc_reverse_transfer(const double *M, const double *v, double tau_p, double tau,
                          double *out);

/* src/PupilSampling.jl:121-146,169-173 — grid, trace, stop filter, append, mirror,
 * rho/theta, sigma; aiming scalars are inputs (SURVEY §7 "hard parts").
 * rows = rows of the EXTENDED system (image row appended by the caller, :111-114).
 * raybasis != 0 reproduces :124-127 (U = (ybar - y_i)/z0, V = -x_i/z0, Q8).
 * Outputs ex, ey, rho, theta need 2*ny*nx entries.  Returns 2*survivors.             */
int64_t orc_full_trace_grid(int rows, const double *R, const double *t, const double *n,
                            const double *K, const double *coef, int ncoef,
                            int ny, const double *yaxis, int nx, const double *xaxis,
                            double U, double V, int raybasis, double ybar, double z0,
                            int stop, double a_stop, double hprime,
                            double *ex, double *ey, double *rho, double *theta,
                            double *rms, int64_t *traced);

/* src/PupilSampling.jl:169-173 */
double orc_sigma(int64_t n, const double *ex, const double *ey);

/* Julia Base `range(a, b, n)` element i (0-based), restated as the correctly rounded
 * value of a + i*(b-a)/(n-1) (binary128 arithmetic), end points exact.  Julia's
 * TwicePrecision range is built to give that value; no reference test pins it at the
 * last ulp: parity unpinned.                                                        */
double orc_linrange(double a, double b, int n, int i);

/* Float32 build of the skew grid (BASELINE config 5 is a build extension, Q21).      */
int64_t orc_trace_skew_grid_f32(int rows, const float *R, const float *t, const float *n,
                                const float *K, const float *coef, int ncoef,
                                int ny, const float *yaxis, int nx, const float *xaxis,
                                float u, float v,
                                float *xv, float *yv, int64_t ld, int32_t *status,
                                int nthreads);
/* ... and of the explicit ray list (slopes u = tan U, v = tan V given).                */
void orc_trace_skew_batch_f32(int rows, const float *R, const float *t, const float *n,
                              const float *K, const float *coef, int ncoef,
                              int64_t nrays, const float *y, const float *x, const float *u, const float *v,
                              float *xv, float *yv, int64_t ld, int32_t *status, int nthreads);

#ifdef __cplusplus
}
#endif
#endif

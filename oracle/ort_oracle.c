/*
 * ort_oracle.c — CPU oracle (plain C restatement of the reference hot path).
 * TEST INFRASTRUCTURE ONLY — see ort_oracle.h for the scope and the parity pin.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * Citations are into /root/reference/.
 */
#include "ort_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_ROWS 256

/* ---- skew loop, double and float instantiations ---------------------------------- */
#define REAL double
#define FN(x) x##_f64
#define SQRT sqrt
#include "ort_oracle_skew.inc"
#undef REAL
#undef FN
#undef SQRT

#define REAL float
#define FN(x) x##_f32
#define SQRT sqrtf
#include "ort_oracle_skew.inc"
#undef REAL
#undef FN
#undef SQRT

void orc_trace_skew_slopes(int rows, const double *R, const double *t, const double *n,
                           const double *K, const double *coef, int ncoef,
                           double y, double x, double u, double v,
                           double *xv, double *yv, int *tir_count)
{
    skew_f64(rows, R, t, n, K, coef, ncoef, y, x, u, v, xv, yv, tir_count);
}

/* src/PupilSampling.jl:34-65: u = tan(U), v = tan(V) (:38-39). */
void orc_trace_skew(int rows, const double *R, const double *t, const double *n,
                    const double *K, const double *coef, int ncoef,
                    double y, double x, double U, double V, double *xv, double *yv)
{
    skew_f64(rows, R, t, n, K, coef, ncoef, y, x, tan(U), tan(V), xv, yv, 0);
}

int orc_status(int S, const double *xv, const double *yv)
{
    for (int i = 0; i < S; ++i)
        if (isnan(xv[i]) || isnan(yv[i])) return i + 1;
    return S + 1;
}

void orc_trace_skew_batch(int rows, const double *R, const double *t, const double *n,
                          const double *K, const double *coef, int ncoef,
                          int64_t nrays, const double *y, const double *x,
                          const double *U, const double *V,
                          double *xv, double *yv, int64_t ld, int32_t *status,
                          int nthreads)
{
    const int S = rows - 1;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t r = 0; r < nrays; ++r) {
        double bx[ORC_MAX_ROWS], by[ORC_MAX_ROWS];
        skew_f64(rows, R, t, n, K, coef, ncoef, y[r], x[r], tan(U[r]), tan(V[r]),
                 bx, by, 0);
        if (xv && yv)
            for (int s = 0; s < S; ++s) { xv[s * ld + r] = bx[s]; yv[s * ld + r] = by[s]; }
        if (status) status[r] = orc_status(S, bx, by);
    }
}

int64_t orc_trace_skew_grid(int rows, const double *R, const double *t, const double *n,
                            const double *K, const double *coef, int ncoef,
                            int ny, const double *yaxis, int nx, const double *xaxis,
                            double U, double V,
                            double *xv, double *yv, int64_t ld, int32_t *status,
                            int nthreads)
{
    const int S = rows - 1;
    const double u = tan(U), v = tan(V);
    const int64_t nrays = (int64_t)ny * nx;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t r = 0; r < nrays; ++r) {      /* y outer, x inner: :123 */
        double bx[ORC_MAX_ROWS], by[ORC_MAX_ROWS];
        skew_f64(rows, R, t, n, K, coef, ncoef, yaxis[r / nx], xaxis[r % nx], u, v,
                 bx, by, 0);
        if (xv && yv)
            for (int s = 0; s < S; ++s) { xv[s * ld + r] = bx[s]; yv[s * ld + r] = by[s]; }
        if (status) status[r] = orc_status(S, bx, by);
    }
    return nrays * S;
}

int64_t orc_trace_skew_grid_f32(int rows, const float *R, const float *t, const float *n,
                                const float *K, const float *coef, int ncoef,
                                int ny, const float *yaxis, int nx, const float *xaxis,
                                float u, float v,
                                float *xv, float *yv, int64_t ld, int32_t *status,
                                int nthreads)
{
    const int S = rows - 1;
    const int64_t nrays = (int64_t)ny * nx;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t r = 0; r < nrays; ++r) {
        float bx[ORC_MAX_ROWS], by[ORC_MAX_ROWS];
        skew_f32(rows, R, t, n, K, coef, ncoef, yaxis[r / nx], xaxis[r % nx], u, v,
                 bx, by, 0);
        if (xv && yv)
            for (int s = 0; s < S; ++s) { xv[s * ld + r] = bx[s]; yv[s * ld + r] = by[s]; }
        if (status) {
            int st = S + 1;
            for (int s = 0; s < S; ++s)
                if (isnan(bx[s]) || isnan(by[s])) { st = s + 1; break; }
            status[r] = st;
        }
    }
    return nrays * S;
}

/* ---- meridional trace ------------------------------------------------------------ */

/* src/RayTracing.jl:75-88 */
static double sag2(double y, double U, double R, double K, const double *c, int ncoef)
{
    if (isfinite(R)) {
        double beta = R - y * tan(U);
        double y2 = y * y;
        double sec = 1.0 / cos(U);                 /* Base.sec(x) = inv(cos(x)) */
        double D = beta * beta - y2 * (sec * sec + K);
        if (D >= 0.0)
            return y2 / (beta + sgn_f64(R) * sqrt(D)) + poly_f64(c, ncoef, y);
        return NAN;
    }
    return 0.0;
}

/* src/RayTracing.jl:98 */
static double tilt2(double y, double R, double K, const double *c, int ncoef)
{
    return sgn_f64(R) * y / sqrt(R * R - y * y * (1.0 + K)) + dpoly_f64(c, ncoef, y);
}

/* src/RayTracing.jl:145-169 */
void orc_trace_meridional(int rows, const double *R, const double *t, const double *n,
                          const double *K, const double *coef, int ncoef,
                          int layout_mode, double y, double U,
                          double *y_out, double *U_out, double *ts_out)
{
    double ts[ORC_MAX_ROWS];
    for (int i = 0; i < rows; ++i) ts[i] = t[i];              /* :148 */
    y_out[0] = y; U_out[0] = U;                               /* :150 */
    for (int i = 0; i < rows - 1; ++i) {                      /* :151 */
        y += tan(U) * ts[i];                                  /* :152 */
        double Rs = R[i + 1];
        double Ks = K ? K[i + 1] : 0.0;
        const double *ps = rowcoef_f64(coef, ncoef, i + 1);
        double s = sag2(y, U, Rs, Ks, ps, ncoef);             /* :156 */
        y += s * tan(U);                                      /* :158 */
        ts[i] += s;                                           /* :160 */
        ts[i + 1] -= s;                                       /* :161 */
        /* :162 — asin(y/R) iff K == 0 and p is the function `zero` itself (Q16).
         * Base.asin throws DomainError for |arg| > 1; C returns NaN (not reproduced). */
        double theta;
        if (Ks == 0.0 && !layout_mode && !ps) theta = asin(y / Rs);
        else                                  theta = atan(tilt2(y, Rs, Ks, ps, ncoef));
        double sin_ip = n[i] * sin(U + theta) / n[i + 1];     /* :163 */
        U = fabs(sin_ip) <= 1.0 ? asin(sin_ip) - theta : NAN; /* :164 */
        y_out[i + 1] = y;                                     /* :165 */
        U_out[i + 1] = U;                                     /* :166 */
    }
    if (ts_out) for (int i = 0; i < rows; ++i) ts_out[i] = ts[i];
}

/* ---- paraxial ----------------------------------------------------------------------- */

/* src/RayTracing.jl:38-53 */
int orc_lens_from_surfaces(int rows, const double *R, double *t, const double *n,
                           double *tau, double *phi)
{
    if (!isfinite(t[0])) t[0] = 0.0;   /* :42  t[1] *= isfinite(t[1]); Julia's `false` is a strong zero */
    for (int i = 0; i < rows; ++i) tau[i] = t[i] / n[i];      /* :43 */
    for (int i = 0; i < rows - 1; ++i) phi[i] = (n[i + 1] - n[i]) / R[i + 1]; /* :45 */
    if (t[rows - 1] == 0.0 || !isfinite(t[rows - 1])) return rows - 1;        /* :47-48 */
    phi[rows - 1] = 0.0;                                      /* :50 */
    return rows;
}

/* src/RayTracing.jl:55-69,127-143 */
void orc_trace_paraxial(int k, const double *tau, const double *phi,
                        double y, double w, const double *a, int clip,
                        double *rt_y, double *rt_w)
{
    rt_y[0] = y; rt_w[0] = w;                                 /* :132 */
    for (int i = 0; i < k; ++i) {                             /* :133 */
        double yp = isfinite(tau[i]) ? y + w * tau[i] : y;    /* :61-64 */
        double wp = w - yp * phi[i];                          /* :66-69 */
        y = yp; w = wp;
        double ai = a ? a[i] : INFINITY;
        if (clip && fabs(y) - ai > 1e-13) {                   /* :135 */
            for (int j = i + 1; j <= k; ++j) { rt_y[j] = NAN; rt_w[j] = NAN; } /* :136 */
            break;
        }
        rt_y[i + 1] = y;                                      /* :139 */
        rt_w[i + 1] = w;                                      /* :140 */
    }
}

/* 2x2 row-major product C = A*B, generic matmul accumulation order a*b + c*d. */
static void mm2(const double *A, const double *B, double *C)
{
    double c0 = A[0] * B[0] + A[1] * B[2];
    double c1 = A[0] * B[1] + A[1] * B[3];
    double c2 = A[2] * B[0] + A[3] * B[2];
    double c3 = A[2] * B[1] + A[3] * B[3];
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = c3;
}

/* src/TransferMatrix.jl:4 — prod over reversed rows, left to right (Q20). */
void orc_abcd(int k, const double *tau, const double *phi, double *M)
{
    double acc[4] = { 1.0, 0.0, 0.0, 1.0 };
    int first = 1;
    for (int i = k - 1; i >= 0; --i) {
        double Mi[4] = { 1.0, tau[i], -phi[i], 1.0 - tau[i] * phi[i] };
        if (first) { memcpy(acc, Mi, sizeof acc); first = 0; }
        else       { double tmp[4]; mm2(acc, Mi, tmp); memcpy(acc, tmp, sizeof acc); }
    }
    memcpy(M, acc, sizeof acc);
}

/* src/TransferMatrix.jl:8 — [1 τ′; 0 1] * M * [1 τ; 0 1], left to right. */
void orc_extend(const double *M, double tau, double tau_p, double *out)
{
    double L[4] = { 1.0, tau_p, 0.0, 1.0 };
    double Rm[4] = { 1.0, tau, 0.0, 1.0 };
    double tmp[4];
    mm2(L, M, tmp);
    mm2(tmp, Rm, out);
}

/* src/TransferMatrix.jl:10 */
void orc_transfer(const double *M, const double *v, double tau, double tau_p, double *out)
{
    double E[4];
    orc_extend(M, tau, tau_p, E);
    out[0] = E[0] * v[0] + E[1] * v[1];
    out[1] = E[2] * v[0] + E[3] * v[1];
}

/* src/TransferMatrix.jl:13 — extend(M, τ, τ′) \ v; dense `\` = LU, partial pivoting. */
void orc_reverse_transfer(const double *M, const double *v, double tau_p, double tau,
                          double *out)
{
    double E[4];
    orc_extend(M, tau, tau_p, E);
    double a = E[0], b = E[1], c = E[2], d = E[3], r0 = v[0], r1 = v[1];
    if (fabs(c) > fabs(a)) {
        double tt;
        tt = a; a = c; c = tt; tt = b; b = d; d = tt; tt = r0; r0 = r1; r1 = tt;
    }
    double l = c / a;
    double d2 = d - l * b;
    double y1 = r1 - l * r0;
    double x1 = y1 / d2;
    double x0 = (r0 - b * x1) / a;
    out[0] = x0; out[1] = x1;
}

/* ---- full_trace grid --------------------------------------------------------------- */

/* Base.sum on a Vector: pairwise above 1024 elements, sequential blocks below (the
 * in-block order is @simd-reassociable in Julia: parity unpinned at the last ulps).   */
static double psum(const double *v, int64_t lo, int64_t hi)
{
    if (hi - lo <= 1024) {
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) s += v[i];
        return s;
    }
    int64_t mid = lo + ((hi - lo) >> 1);
    return psum(v, lo, mid) + psum(v, mid, hi);
}
static double psum_sqdev(const double *v, double mu, int64_t lo, int64_t hi)
{
    if (hi - lo <= 1024) {
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) { double d = v[i] - mu; s += d * d; }
        return s;
    }
    int64_t mid = lo + ((hi - lo) >> 1);
    return psum_sqdev(v, mu, lo, mid) + psum_sqdev(v, mu, mid, hi);
}

/* src/PupilSampling.jl:169-173 */
double orc_sigma(int64_t n, const double *ex, const double *ey)
{
    double mux = psum(ex, 0, n) / (double)n;
    double muy = psum(ey, 0, n) / (double)n;
    return sqrt((psum_sqdev(ex, mux, 0, n) + psum_sqdev(ey, muy, 0, n)) / (double)n);
}

/* src/PupilSampling.jl:121-146 */
int64_t orc_full_trace_grid(int rows, const double *R, const double *t, const double *n,
                            const double *K, const double *coef, int ncoef,
                            int ny, const double *yaxis, int nx, const double *xaxis,
                            double U, double V, int raybasis, double ybar, double z0,
                            int stop, double a_stop, double hprime,
                            double *ex, double *ey, double *rho, double *theta,
                            double *rms, int64_t *traced)
{
    const int S = rows - 1;
    int64_t m = 0, ntr = 0;
    double xv[ORC_MAX_ROWS], yv[ORC_MAX_ROWS];
    double rmax = -INFINITY;
    for (int iy = 0; iy < ny; ++iy) {                /* :123, y outer */
        for (int ix = 0; ix < nx; ++ix) {            /*        x inner */
            double yi = yaxis[iy], xi = xaxis[ix];
            if (raybasis) {                          /* :124-127 (Q8) */
                U = (ybar - yi) / z0;
                V = -xi / z0;
            }
            orc_trace_skew(rows, R, t, n, K, coef, ncoef, yi, xi, U, V, xv, yv); /* :128 */
            ++ntr;
            double xf = xv[S - 1], yf = yv[S - 1];   /* :129-130 */
            double ri = hypot(xv[stop - 1], yv[stop - 1]);               /* :131 */
            if (ri > a_stop || isnan(xf) || isnan(yf)) continue;         /* :132 */
            double th = atan2(yv[stop - 1], xv[stop - 1]);               /* :133 */
            ey[m] = yf - hprime;                     /* :134 */
            ex[m] = xf;                              /* :135 */
            rho[m] = ri;                             /* :136 */
            theta[m] = th;                           /* :137 */
            if (ri > rmax) rmax = ri;
            ++m;
        }
    }
    if (traced) *traced = ntr;
    if (m == 0) { if (rms) *rms = NAN; return 0; }   /* maximum(r) of empty throws */
    for (int64_t i = 0; i < m; ++i) {                /* :140-144 */
        ey[m + i] = ey[i];
        ex[m + i] = -ex[i];
        rho[i] = rho[i] / rmax;
        rho[m + i] = rho[i];
        theta[m + i] = M_PI - theta[i];
    }
    if (rms) *rms = orc_sigma(2 * m, ex, ey);        /* :146 */
    return 2 * m;
}

/* ---- range ------------------------------------------------------------------------- */
double orc_linrange(double a, double b, int n, int i)
{
    if (n <= 1) return a;
    if (i <= 0) return a;
    if (i >= n - 1) return b;
    __float128 A = a, B = b;
    __float128 r = A + ((B - A) * (__float128)i) / (__float128)(n - 1);
    return (double)r;
}

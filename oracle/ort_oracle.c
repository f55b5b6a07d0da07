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

void orc_skew_margins(int rows, const double *R, const double *t, const double *n,
                      const double *K, const double *coef, int ncoef,
                      int64_t nrays, const double *y, const double *x, const double *u, const double *v,
                      double *marg)
{
    for (int64_t r = 0; r < nrays; ++r)
        skew_margins_f64(rows, R, t, n, K, coef, ncoef, y[r], x[r], u[r], v[r], marg + 4 * r);
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

/* The list form of the same Float32 build (ort_trace_skew_f32): ray r = (y[r], x[r]) launched with the SLOPES
 * u[r] = tan U, v[r] = tan V (taken by the caller: no libm call on either side).                         */
void orc_trace_skew_batch_f32(int rows, const float *R, const float *t, const float *n,
                              const float *K, const float *coef, int ncoef,
                              int64_t nrays, const float *y, const float *x, const float *u, const float *v,
                              float *xv, float *yv, int64_t ld, int32_t *status, int nthreads)
{
    const int S = rows - 1;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t r = 0; r < nrays; ++r) {
        float bx[ORC_MAX_ROWS], by[ORC_MAX_ROWS];
        skew_f32(rows, R, t, n, K, coef, ncoef, y[r], x[r], u[r], v[r], bx, by, 0);
        if (xv && yv)
            for (int s = 0; s < S; ++s) { xv[s * ld + r] = bx[s]; yv[s * ld + r] = by[s]; }
        if (status) {
            int st = S + 1;
            for (int s = 0; s < S; ++s)
                if (isnan(bx[s]) || isnan(by[s])) { st = s + 1; break; }
            status[r] = st;
        }
    }
}

/* src/RayTracing.jl:145-169 */
int orc_trace_meridional(int rows, const double *R, const double *t, const double *n,
                         const double *K, const double *coef, int ncoef,
                         int layout_mode, double y, double U,
                         double *y_out, double *U_out, double *ts_out)
{
    int domain = 0;   /* 1-based row of the first asin(x), |x| > 1: Base.asin throws a DomainError there (:162) */
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
         * Base.asin throws DomainError for |arg| > 1; C returns NaN: the row is reported through the return value. */
        double theta;
        if (Ks == 0.0 && !layout_mode && !ps) { if (fabs(y / Rs) > 1.0 && !domain) domain = i + 1; theta = asin(y / Rs); }
        else                                  theta = atan(tilt2(y, Rs, Ks, ps, ncoef));
        double sin_ip = n[i] * sin(U + theta) / n[i + 1];     /* :163 */
        U = fabs(sin_ip) <= 1.0 ? asin(sin_ip) - theta : NAN; /* :164 */
        y_out[i + 1] = y;                                     /* :165 */
        U_out[i + 1] = U;                                     /* :166 */
    }
    if (ts_out) for (int i = 0; i < rows; ++i) ts_out[i] = ts[i];
    return domain;
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

/* ---- first-order solve, Seidel sums, paraxial incidences -------------------------------- */

/* Base.sum over a Vector{Float64} of fewer than 16 elements is a plain left fold (Base._mapreduce); longer
 * vectors go through a @simd-reassociable block: parity unpinned at the last ulps.                         */
static double vsum(const double *v, int n)
{
    double s = 0.0;
    if (n > 0) { s = v[0]; for (int i = 1; i < n; ++i) s += v[i]; }
    return s;
}

/* solve(surfaces, a, h′) (src/RayTracing.jl:325-327 -> _solve :302-323) followed by
 * aberrations(surfaces, system, λ, δn) (src/SeidelAberrations.jl:6-53) and incidences(surfaces, system)
 * (src/RayTracing.jl:338-353).  surf: 10 per-surface vectors of rows-1 entries each, in the order of
 * ORC_SURF_*; inc: 4 columns [ni nī i ī] of rows-1 entries.  marg / chief: y and nu of the paraxial marginal
 * and chief rays, k+2 entries each (ParaxialRay.ynu, src/Types.jl:29-51), or NULL.  Returns 0, or -1 when
 * Lens() keeps the last row (k == rows) — the reference then needs `rows` semi-diameters (:215).           */
int orc_solve_aberrations(int rows, const double *R, const double *t_in, const double *n,
                          const double *a, const double *dn, double hprime, double lambda,
                          orc_system_t *out, double *surf, double *inc,
                          double *marg_y, double *marg_nu, double *chief_y, double *chief_nu)
{
    double t[ORC_MAX_ROWS], tau[ORC_MAX_ROWS], phi[ORC_MAX_ROWS];
    for (int i = 0; i < rows; ++i) t[i] = t_in[i];
    const int k = orc_lens_from_surfaces(rows, R, t, n, tau, phi);              /* Lens(surfaces)  :38-53 */
    if (k != rows - 1) return -1;
    /* lens.n = surfaces[:,3] (:52); ParaxialRay's n = [n; n[end]] (Types.jl:39) */
    double nn[ORC_MAX_ROWS + 2];
    for (int i = 0; i < rows; ++i) nn[i] = n[i];
    nn[rows] = n[rows - 1];
    /* trace_marginal_ray(lens, a)  :208-221 */
    double my[ORC_MAX_ROWS + 2], mw[ORC_MAX_ROWS + 2];
    orc_trace_paraxial(k, tau, phi, 1.0, 0.0, 0, 0, my, mw);                   /* :209 */
    const double f = -(1.0 / mw[k]);                                            /* :213  -inv(ω[end]) */
    const double EBFD = my[k] * f;                                              /* :214 */
    int stop = 0; double s = a[0] / my[1];                                      /* :215-216 findmin: first minimum */
    for (int i = 1; i < k; ++i) { double sv = a[i] / my[i + 1]; if (sv < s) { s = sv; stop = i; } }
    stop += 1;                                                                  /* 1-based */
    for (int i = 0; i <= k; ++i) { my[i] *= s; mw[i] *= s; }                    /* :217 */
    mw[k + 1] = mw[k]; my[k + 1] = (mw[k] == 0.0) ? my[k] : 0.0;                /* extend  :202-206 */
    /* trace_chief_ray(lens, stop, marginal, h′)  :246-263 */
    double cy2[ORC_MAX_ROWS + 2], cw2[ORC_MAX_ROWS + 2], cy[ORC_MAX_ROWS + 2], cw[ORC_MAX_ROWS + 2];
    orc_trace_paraxial(k, tau, phi, 0.0, 1.0, 0, 0, cy2, cw2);                 /* :252 */
    const double y_stop = my[stop], y2_stop = cy2[stop];                        /* surface_ray(y)[stop], y2[stop] */
    const double nub = -mw[k + 1] * hprime / my[1];                             /* :256 */
    for (int i = 1; i <= k; ++i) {                                              /* :258 */
        cy[i] = nub * (cy2[i] - my[i] * y2_stop / y_stop);
        cw[i] = nub * (cw2[i] - mw[i] * y2_stop / y_stop);
    }
    cy[0] = 0.0; cw[0] = nub;                                                   /* :259 */
    cy[k + 1] = hprime; cw[k + 1] = cw[k];                                      /* :260 */
    /* _solve  :302-323 */
    const double ybar = cy[1], nubp = cw[k + 1], ym = my[0], ybpb = cy[k];
    const double delta = (hprime - nubp * f - ybar) / nub;                      /* :312 */
    const double H = nub * ym;                                                  /* :316 */
    if (out) {
        out->f = f; out->EBFD = EBFD; out->EFFD = delta - f;                    /* :313 */
        out->PN = (n[rows - 1] - n[0]) * f;                                     /* :314 */
        out->EP_D = fabs(ym) * 2.0; out->EP_t = -ybar / nub;                    /* :315 */
        out->H = H;
        out->XP_D = fabs(2.0 * H / nubp); out->XP_t = -ybpb / nubp;             /* :317 */
        out->N = fabs(f / out->EP_D);                                           /* :318 */
        out->FOV = 2.0 * (atan(fabs(cw[0] / nn[0])) * (180.0 / M_PI));          /* :319  2atand(|ū[1]|) */
        out->stop = stop; out->k = k;
    }
    for (int i = 0; i <= k + 1; ++i) {
        if (marg_y) marg_y[i] = my[i];
        if (marg_nu) marg_nu[i] = mw[i];
        if (chief_y) chief_y[i] = cy[i];
        if (chief_nu) chief_nu[i] = cw[i];
    }
    /* aberrations  SeidelAberrations.jl:10-35;  surface_ray(x) = x[2:end-1] */
    const int S = rows - 1;
    double u[ORC_MAX_ROWS + 2];
    for (int i = 0; i <= k + 1; ++i) u[i] = mw[i] / nn[i];                      /* Types.jl:40 */
    double sph[ORC_MAX_ROWS], coma[ORC_MAX_ROWS], ast[ORC_MAX_ROWS], ptz[ORC_MAX_ROWS], dist[ORC_MAX_ROWS],
           axial[ORC_MAX_ROWS], lateral[ORC_MAX_ROWS];
    for (int i = 0; i < S; ++i) {
        const double Ri = R[i + 1], yi = my[i + 1], ybi = cy[i + 1];
        const double A = mw[i] + nn[i] * yi / Ri;                               /* :18 */
        const double Ab = (H + A * ybi) / yi;                                   /* :19 */
        const double yD = yi * (u[i + 1] / nn[i + 1] - u[i] / nn[i]);           /* :20, Δ :4 */
        const double yd = dn ? yi * (dn[i + 1] / nn[i + 1] - dn[i] / nn[i]) : yi * (0.0 / nn[i + 1] - 0.0 / nn[i]);   /* :21 */
        const double i1 = 1.0 / nn[i + 1], i0 = 1.0 / nn[i];
        const double Dn2 = i1 * i1 - i0 * i0;                                   /* :22 */
        const double P = (i1 - i0) / Ri;                                        /* :23 */
        sph[i] = -(A * A) * yD / (8.0 * lambda);                                /* :25 */
        coma[i] = -A * Ab * yD / (2.0 * lambda);                                /* :26 */
        ast[i] = -(Ab * Ab) * yD / (2.0 * lambda);                              /* :27 */
        ptz[i] = -(H * H) * P / (4.0 * lambda);                                 /* :28 */
        dist[i] = -Ab * ((Ab * Ab) * yi * Dn2 - (H + Ab * yi) * ybi * P) / (2.0 * lambda);   /* :30 */
        axial[i] = A * yd / (2.0 * lambda);                                     /* :31 */
        lateral[i] = Ab * yd / lambda;                                          /* :32 */
        if (surf) {
            surf[ORC_SURF_SPHERICAL * S + i] = sph[i]; surf[ORC_SURF_COMA * S + i] = coma[i];
            surf[ORC_SURF_ASTIGMATISM * S + i] = ast[i]; surf[ORC_SURF_PETZVAL * S + i] = ptz[i];
            surf[ORC_SURF_SAGITTAL * S + i] = ptz[i] + ast[i] / 2.0;            /* :29 */
            surf[ORC_SURF_DISTORTION * S + i] = dist[i];
            surf[ORC_SURF_AXIAL * S + i] = axial[i]; surf[ORC_SURF_LATERAL * S + i] = lateral[i];
            surf[ORC_SURF_MEDIAL * S + i] = ptz[i] + ast[i];                    /* :33 */
            surf[ORC_SURF_TANGENTIAL * S + i] = ptz[i] + 1.5 * ast[i];          /* :34 */
        }
        if (inc) {                                                              /* RayTracing.jl:338-353 */
            const double ni = mw[i] + nn[i] * yi / Ri, nib = cw[i] + nn[i] * ybi / Ri;
            inc[0 * S + i] = ni; inc[1 * S + i] = nib;
            inc[2 * S + i] = ni / nn[i]; inc[3 * S + i] = nib / nn[i];
        }
    }
    if (out) {
        out->W040 = vsum(sph, S); out->W131 = vsum(coma, S); out->W222 = vsum(ast, S);   /* :37-40 */
        out->W311 = vsum(dist, S);
        out->W220P = vsum(ptz, S);                                              /* :42 */
        out->W220 = out->W220P + 0.5 * out->W222;                               /* :43 */
        out->W220M = out->W220P + out->W222; out->W220T = out->W220P + 1.5 * out->W222;  /* :44-45 */
        out->W020 = vsum(axial, S); out->W111 = vsum(lateral, S);               /* :48,50 */
    }
    return 0;
}

/* Plain-C caller of the drop-in boundary (include/ort.h): the reference's own test case — the Cooke triplet of
 * test/runtests.jl:19-35 — through ort_full_trace_batch_f64, i.e. full_trace(solve(surfaces, a, h'), H, 64)
 * for H = 0 and H = 1 in one call.  No Python, no torch: host arrays in, host arrays out.
 *
 *   gcc -O2 -Iinclude examples/cooke_full_trace.c -o build/cooke_full_trace \
 *       -Lopticalraytracing.jl_amd/csrc -lort_hip -Wl,-rpath,$PWD/opticalraytracing.jl_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ort.h"

#define ROWS 8

static int cmp_double(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

/* `cooke_full_trace --time [H] [fast]`: wall time of the reference's own call, full_trace(system, H, 64) (test/runtests.jl:364-372
 * shape: one system, one field, 64 x 32 half pupil), through the C ABI — error vectors back in host memory — 300 times, in the
 * default arithmetic policy (the reference's operation sequence, aiming loops through the meridional trig sequence) or, with
 * `fast`, under ORT_FAST_MATH (direction-cosine forms; the aiming loops trace this plain prescription without trigonometric
 * calls: their serial latency is most of a call this small). */
int main(int argc, char **argv)
{
    /* surfaces = [R t n] (test/runtests.jl:19-29), clear semi-diameters a (:31-33), image height h' (:35) */
    const double R[ROWS] = { INFINITY, 37.40, -341.48, -42.65, 36.40, INFINITY, 204.52, -37.05 };
    const double t[ROWS] = { 0.0, 5.90, 12.93, 2.50, 2.00, 9.85, 5.90, 0.0 };
    const double n[ROWS] = { 1.0, 1.61272, 1.0, 1.64769, 1.0, 1.0, 1.61272, 1.0 };
    const double a[ROWS - 1] = { 14.7, 14.7, 10.8, 10.8, 10.3, 11.6, 11.6 };
    const double hprime = 21.248, fields[2] = { 0.0, 1.0 };
    enum { K = 64, CAP = 2 * K * (K / 2) };
    ort_ctx *ctx = NULL;
    if (ort_ctx_create(0, NULL, &ctx) != ORT_OK) { fprintf(stderr, "ort_ctx_create: %s\n", ort_last_error()); return 2; }
    double *ex = malloc(sizeof(double) * 2 * CAP), *ey = malloc(sizeof(double) * 2 * CAP);
    double *rho = malloc(sizeof(double) * 2 * CAP), *theta = malloc(sizeof(double) * 2 * CAP);
    int64_t count[2]; double rms[2]; ort_first_order fo;
    int rc = ort_full_trace_batch_f64(ctx, 1, ROWS, R, t, n, a, &hprime, 2, fields, K, &fo, ex, ey, rho, theta, count, rms, 0);
    if (rc != ORT_OK) { fprintf(stderr, "ort_full_trace_batch_f64: %s\n", ort_last_error()); return 3; }
    printf("f = %.6f  stop = %d  BFD = %.6f\n", fo.f, fo.stop, fo.BFD);
    for (int b = 0; b < 2; ++b) {
        double sx = 0, sy = 0, q = 0;
        for (int64_t i = 0; i < count[b]; ++i) { sx += ex[b * CAP + i]; sy += ey[b * CAP + i]; }
        sx /= (double)count[b]; sy /= (double)count[b];
        for (int64_t i = 0; i < count[b]; ++i) { const double dx = ex[b * CAP + i] - sx, dy = ey[b * CAP + i] - sy; q += dx * dx + dy * dy; }
        printf("H = %.1f  rays = %lld  RMS = %.9f  (recomputed from the vectors: %.9f)\n", fields[b], (long long)count[b], rms[b],
               sqrt(q / (double)count[b]));
    }
    if (argc > 1 && strcmp(argv[1], "--time") == 0) {
        const double H = argc > 2 ? atof(argv[2]) : 1.0;
        const unsigned fl = (argc > 3 && strcmp(argv[3], "fast") == 0) ? ORT_FAST_MATH : 0u;
        enum { REPS = 300 };
        static double us[REPS];
        for (int full = 1; full >= 0; --full) {
            for (int i = -20; i < REPS; ++i) {
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                rc = full ? ort_full_trace_batch_f64(ctx, 1, ROWS, R, t, n, a, &hprime, 1, &H, K, &fo, ex, ey, rho, theta, count, rms, fl)
                          : ort_spot_batch_f64(ctx, 1, ROWS, R, t, n, a, &hprime, 1, &H, K, &fo, count, rms, fl);
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (rc != ORT_OK) { fprintf(stderr, "timed call: %s\n", ort_last_error()); return 3; }
                if (i >= 0) us[i] = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
            }
            qsort(us, REPS, sizeof(double), cmp_double);
            printf("{\"policy\": \"%s\", \"call\": \"%s\", \"H\": %.2f, \"rays\": %lld, \"rms\": %.9f, \"wall_us_median\": %.1f, \"wall_us_min\": %.1f, \"wall_us_p90\": %.1f}\n",
                   fl ? "fast" : "reference sequence (default)", full ? "ort_full_trace_batch_f64 (vectors back)" : "ort_spot_batch_f64 (count, RMS)", H, (long long)count[0], rms[0],
                   us[REPS / 2], us[0], us[REPS * 9 / 10]);
        }
    }
    /* the reference's own known answers for this prescription: f = 101.181, stop == 5 (test/runtests.jl:53-60) */
    const int ok = fabs(fo.f - 101.181) < 1e-3 && fo.stop == 5 && count[0] > 0 && count[1] > 0;
    free(ex); free(ey); free(rho); free(theta);
    ort_ctx_destroy(ctx);
    return ok ? 0 : 1;
}

/*
 * ort.h — C ABI of libort_hip.so, the MI355X (gfx950) batched ray-trace engine.
 *
 * This is the drop-in boundary for ONE path of Sagnac/OpticalRayTracing.jl: the per-ray
 * surface-by-surface propagation (skew, meridional, paraxial y-nu) and the ABCD product,
 * plus the pupil-grid driver of `full_trace`.  The reference has no FFI of its own (it is
 * pure Julia); every entry point below names the reference method (file:line under
 * /root/reference) whose batch form it is, and INTEGRATION.md shows the `ccall` stubs.
 *
 * Conventions
 *  - plain C: opaque handles, pointers, sizes.  No C++/torch types cross the boundary.
 *  - every function returns 0 on success, a negative ORT_E* code on failure;
 *    ort_last_error() gives the message of the calling thread's last failure.
 *  - `surfaces` columns follow the reference matrix (R, t, n; src/PupilSampling.jl:36):
 *    each column is a contiguous double[rows]; a Julia `Matrix` passes
 *    pointer(M), pointer(M)+rows, pointer(M)+2rows.
 *  - per-ray arrays are struct-of-arrays.  Per-surface history is surface-major:
 *    out[s*ld + ray], s = 0 .. rows-2  (the reference's xv[i], yv[i], Q10).
 *  - buffers are HOST pointers unless ORT_DEVICE_PTRS is set in `flags`, in which case
 *    every ray/axis/output pointer of that call is a device pointer on the context's GPU
 *    and the call is asynchronous on the context's stream.
 *  - NaN sentinels are the reference's: surface miss -> NaN coordinates from that surface
 *    on (src/PupilSampling.jl:9); skew TIR leaves the ray undeviated (Q1, :25-31);
 *    meridional TIR -> U = NaN (src/RayTracing.jl:164); paraxial clip -> NaN rows (:136).
 *  - status[ray] = 1-based loop index of the first surface whose x or y is NaN, or
 *    rows (= S+1) when the ray reached the last row; bit ORT_STATUS_STOPPED is or-ed in
 *    when the stop filter of src/PupilSampling.jl:131-132 rejected it.
 *  - there is NO CPU fallback: without a usable gfx950 device every compute call fails.
 */
#ifndef ORT_H
#define ORT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORT_VERSION 401            /* 0.4.1: ORT_FT_FUSED (full_trace's second pass inside the trace launch; bit-identical results) and its
                                      testing aid ort_ctx_test_fused_no_scan.  0.4.0: no new entry point; statistics-only full_trace walks tiles per workgroup (results within 1e-12 of
                                      0.3.0's); the device-side aiming loops trace without trigonometric calls only under ORT_FAST_MATH;
                                      the one-call pipelines report look-back faults to host callers; a context may be destroyed ahead of
                                      its communicators (0.3.0: ort_wavegrad_f64; ORT_NO_LDS dropped; look-back faults are reported) */
#define ORT_MAX_ROWS 64            /* surface-matrix rows per system, object row included */
#define ORT_MAX_NCOEF 12           /* polynomial coefficients per surface */

/* error codes */
#define ORT_OK 0
#define ORT_EINVAL (-1)            /* bad argument (shape, NULL, range) */
#define ORT_EDOMAIN (-2)           /* the reference would throw DomainError */
#define ORT_EHIP (-3)              /* HIP runtime failure (no device, launch error, OOM) */
#define ORT_ENOMEM (-4)

/* flags */
#define ORT_DEVICE_PTRS   (1u << 0) /* ray/axis/output pointers are device pointers */
#define ORT_INPUT_SLOPES  (1u << 1) /* U, V arrays hold tan U, tan V already */
#define ORT_RAYBASIS      (1u << 2) /* per-ray U=(ybar-y)/z0, V=-x/z0 (PupilSampling.jl:124-127) */
#define ORT_LAYOUT_INPUT  (1u << 3) /* meridional: input was a Layout -> always atan (Q16) */
#define ORT_CLIP          (1u << 4) /* paraxial: clip = true (RayTracing.jl:135) */
#define ORT_FAST_MATH     (1u << 5) /* direction-cosine / fused arithmetic: coordinates within 1e-10 relative of the
                                       reference sequence (measured <= 5e-12 on well-conditioned paths).  Status: a wave
                                       holding a ray within 1e-9 (normalised) of a branch of the reference loop (surface
                                       miss, total reflection, a sphere's equator, the stop filter's or a clear aperture's edge),
                                       a totally reflected ray, or a ray on which the reference's formulas are not the geometry
                                       (hit beyond the equator, direction refracted backward, polynomial row outside its conic)
                                       retraces with the reference sequence and is bit-identical there.  That margin is 100 x
                                       the deviation measured on ordinary paths; an ill-conditioned path could in principle
                                       amplify a rounding difference past it at a LATER surface, so identical status is an
                                       observation, not a theorem: no flip on any ray of the parity suites and soaks (> 1e7 rays
                                       of adversarial random systems, profiles/r0*_soak_*; every ray of configs 2 and 3).
                                       Prescriptions deeper than ORT_FAST_MAX_SURFACES are traced with the reference sequence
                                       whatever the flag (below).  Default: the op-for-op IEEE sequence of the reference loop,
                                       bit-identical to a non-fused CPU evaluation.  Both policies run the same entry points
                                       (tests/test_gpu_parity.py) */
#define ORT_FAST_MAX_SURFACES 48    /* ORT_FAST_MATH applies to prescriptions of at most this many loop iterations (rows - 1, the appended
                                       image plane included); deeper ones — where the path amplifies the fast forms' rounding differences
                                       towards the 1e-10 bar: worst 3.5e-11 at 47, a few rays of 60,000 past 1e-10 from 54 on,
                                       profiles/r04_fast_depth.log — are traced with the reference sequence under either flag */
#define ORT_NO_SMALL_PATH (1u << 8) /* testing aid: small problems (<= 256 (system, field) pairs; full_trace bundles of <= 32 tiles)
                                       normally run their setup and their finish as ONE launch each (k_small_prepare,
                                       k_ft_small_finish); this takes the general multi-launch route instead — same device
                                       functions, bit-identical results (tests/test_gpu_parity.py) */
/* bit 6 reserved (was ORT_NO_LDS, the scalar-load variant of the surface table: measured, not faster, dropped) */
#define ORT_AIM_EDGE_AS_FOUND (1u << 9) /* ort_aim_f64, diagnostic: leave the two edge-ray searches where the FD-Newton ends (either side
                                        of the stop's edge) instead of applying the fitted "end inside the edge" rule described at
                                        ort_aim_f64 */
#define ORT_FT_FUSED      (1u << 10) /* full_trace, two or more bundles: the second pass (offsets, placement in both halves, squared
                                        deviations) runs INSIDE the trace launch — a workgroup that has traced its tile places a tile of
                                        an earlier, complete bundle — so the HBM-bound pass overlaps the issue-bound one.  Same device
                                        functions: bit-identical results (BASELINE config 3: 1.50 instead of 1.67-1.73 ms).  Bundles
                                        of <= 32 tiles and single-bundle calls take the default route under this flag too.  Opt-in: the
                                        workgroups hand data to each other inside the launch (write-through stores, cache-bypassing
                                        loads, no fence) and wait only for workgroups with lower indices — properties observed on
                                        gfx950 / ROCm 7.2, not promised by HIP; a wait that outlasts its poll cap fails the call
                                        (ORT_EHIP; device-pointer callers: count = -1, rms = NaN), it never misplaces survivors */
#define ORT_FT_LOOKBACK   (1u << 7) /* full_trace: the trace kernel writes the survivors' first half at its final place
                                       (decoupled look-back over the bundle's tiles) instead of staging compacted tiles in
                                       a workspace: 82 instead of 100 B/ray of HBM traffic, but tiles wait for their
                                       predecessors' counts — slower whenever the trace is VALU-bound */

#define ORT_STATUS_STOPPED (1 << 16)
#define ORT_STATUS_VIGNETTED (1 << 17)                /* ort_system_set_apertures: outside a clear aperture */
#define ORT_STATUS_VIGNETTE_SURFACE(s) ((((s) >> 20) & 0xff) + 1) /* first such surface row, 1-based */
#define ORT_STATUS_INDEX(s) ((s) & 0xffff)

typedef struct ort_ctx ort_ctx;       /* one per (host thread, GPU) */
typedef struct ort_system ort_system; /* device-resident batch of prescriptions */

/* ---- context ---------------------------------------------------------------------- */
int ort_version(void);
const char *ort_last_error(void);
/* stream: a hipStream_t to launch on, or NULL for a stream owned by the context (created non-blocking: it does NOT wait
 * for work on the legacy default stream).  With ORT_DEVICE_PTRS the calls are asynchronous on that stream and read / write the
 * caller's device buffers there: a caller that fills or reads those buffers on another stream (an array library's own) orders
 * the two itself — hand its stream to ort_ctx_create / ort_ctx_set_stream, or synchronise it before the call and
 * ort_ctx_synchronize after (tests/test_gpu_parity.py does the latter around torch's fills). */
int ort_ctx_create(int device, void *stream, ort_ctx **out);
int ort_ctx_destroy(ort_ctx *ctx);
int ort_ctx_set_stream(ort_ctx *ctx, void *stream);
int ort_ctx_synchronize(ort_ctx *ctx);
/* hipEvent pair on the context's stream; stop returns elapsed milliseconds. */
int ort_ctx_timer_start(ort_ctx *ctx);
int ort_ctx_timer_stop(ort_ctx *ctx, float *ms);
/* device properties the bench reports (name, CU count, clock MHz, total bytes) */
int ort_ctx_device_info(ort_ctx *ctx, char *name, int name_len, int *cus, int *clock_mhz,
                        int64_t *mem_bytes);

/* ---- device buffers ---------------------------------------------------------------- 
 * For hosts without a GPU array library of their own (the Julia shim, a plain C caller): allocate on the
 * context's GPU, pass the pointers with ORT_DEVICE_PTRS, download only what is read on the host — a config-2
 * history is 1.8 GB, 30 ms over PCIe against 0.3 ms of kernel.  upload / download block until the copy is
 * done; free waits for the context's stream first.                                                      */
int ort_device_malloc(ort_ctx *ctx, size_t bytes, void **out);
/* waits for the context's stream AND for the collectives in flight on its communicators' own streams (ort_comm_*). */
int ort_device_free(ort_ctx *ctx, void *p);
int ort_device_upload(ort_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int ort_device_download(ort_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);

/* ---- systems ---------------------------------------------------------------------- */
/* Upload nsys prescriptions of `rows` rows each (host pointers, always).
 * R, t, n : [nsys][rows]   (reference `surfaces` columns; Types.jl:82-112 Layout.R/t/n)
 * K       : [nsys][rows] or NULL (zeros)          (Layout.K)
 * coef    : [nsys][rows][ncoef] or NULL           (Layout.p restricted to a power series,
 *           p_i(y) = sum_j coef[j] y^j, Horner; an arbitrary Julia closure cannot cross
 *           a C ABI — SURVEY §7)
 * Both a Float64 and a Float32 table are built.                                         */
int ort_system_create(ort_ctx *ctx, int nsys, int rows,
                      const double *R, const double *t, const double *n,
                      const double *K, const double *coef, int ncoef,
                      ort_system **out);
int ort_system_destroy(ort_system *sys);
/* EXTENSION (no reference counterpart: the reference filters real rays at the stop only, Q12, and
 * treats the other semi-diameters paraxially, src/Vignetting.jl): clear semi-diameters a : [nsys][rows-1]
 * for the real-ray trace, one per loop iteration (for a full_trace table: system.a, then Inf for the
 * appended image plane).  A ray with x*x + y*y > a*a on any surface gets ORT_STATUS_VIGNETTED and the
 * index of the first such surface in its status, and is dropped by the full_trace filter and compaction
 * like a ray outside the stop; coordinates and history are unaffected.  NULL clears.  Off by default. */
int ort_system_set_apertures(ort_system *sys, const double *a);
int ort_system_rows(const ort_system *sys);
int ort_system_count(const ort_system *sys);

/* ---- skew real-ray trace: raytrace(surfaces, y, x, U, V, Vector{RealRay}; K, p) ------
 * src/PupilSampling.jl:34-65, batch form over explicit ray lists through system `isys`.
 * y, x, U, V : [nrays]; xv, yv : [rows-1][ld] or NULL; status : [nrays] or NULL.        */
int ort_trace_skew_f64(ort_ctx *ctx, const ort_system *sys, int isys, int64_t nrays,
                       const double *y, const double *x, const double *U, const double *V,
                       double *xv, double *yv, int64_t ld, int32_t *status,
                       unsigned flags);
int ort_trace_skew_f32(ort_ctx *ctx, const ort_system *sys, int isys, int64_t nrays,
                       const float *y, const float *x, const float *U, const float *V,
                       float *xv, float *yv, int64_t ld, int32_t *status,
                       unsigned flags);

/* ---- pupil-grid bundles: the hot loop of full_trace, src/PupilSampling.jl:121-138 ----
 * A bundle is one (system, field) pair: rays are the cartesian grid yaxis (outer) x xaxis
 * (inner), all sharing the field angles U, V (or the RayBasis rule).  Ray r of bundle b
 * has global index b*ny*nx + r with r = iy*nx + ix.                                      */
typedef struct ort_bundle {
    int32_t system;      /* index into the ort_system batch */
    int32_t stop;        /* 1-based loop index tested by the stop filter (system.stop); 0 = none */
    double U, V;         /* field angles; tan() is taken on the host (PupilSampling.jl:38-39) */
    double a_stop;       /* |system.a[stop]|  (:91) */
    double hprime;       /* h' subtracted from y_f (:102-109,134) */
    double ybar, z0;     /* ORT_RAYBASIS only (:106-108) */
    int64_t yaxis_off;   /* element offsets of this bundle's axes inside `axes` */
    int64_t xaxis_off;
} ort_bundle;

/* Outputs of a grid trace; any pointer may be NULL.  N = nb*ny*nx.
 * xv, yv  : [rows-1][ld]  per-surface history (ld >= N)
 * xf, yf  : [N] image-row hit (xv[end], yv[end], :129-130)
 * xs, ys  : [N] hit on the stop row (:131), requires bundle.stop > 0
 * status  : [N]                                                                          */
typedef struct ort_grid_out_f64 {
    double *xv, *yv; int64_t ld;
    double *xf, *yf, *xs, *ys;
    int32_t *status;
} ort_grid_out_f64;
typedef struct ort_grid_out_f32 {
    float *xv, *yv; int64_t ld;
    float *xf, *yf, *xs, *ys;
    int32_t *status;
} ort_grid_out_f32;

/* bundles: host array [nb] (always host).  axes: [..] doubles, host or device per flags. */
int ort_trace_grid_f64(ort_ctx *ctx, const ort_system *sys, int nb, const ort_bundle *bundles,
                       const double *axes, int64_t axes_len, int ny, int nx,
                       const ort_grid_out_f64 *out, unsigned flags);
int ort_trace_grid_f32(ort_ctx *ctx, const ort_system *sys, int nb, const ort_bundle *bundles,
                       const float *axes, int64_t axes_len, int ny, int nx,
                       const ort_grid_out_f32 *out, unsigned flags);

/* ---- pupil axes on the device: range(y1, y2, ny), range(x1, x2, nx) per bundle ---------------
 * (src/PupilSampling.jl:121-122).  ends : [nb][4] = {y_first, y_last, x_first, x_last};
 * axes : [nb][ny + nx] (bundle b: y axis at b*(ny+nx), x axis at b*(ny+nx)+ny).  Double-double
 * evaluation of a + i (b - a)/(n - 1): correctly rounded, ties to even, end points exact — what
 * Julia's TwicePrecision `range` is built to give (not pinned at the last ulp by any reference
 * test; a Julia host that needs bit-identical grids passes collect(range(...)) instead).        */
int ort_make_axes_f64(ort_ctx *ctx, int nb, int ny, int nx, const double *ends, double *axes,
                      unsigned flags);

/* ---- full_trace: src/PupilSampling.jl:121-146 + sigma :169-173 ------------------------
 * grid -> trace -> stop filter -> order-preserving append -> mirror -> rho, theta -> RMS,
 * one RealRayError (Types.jl:184-192) per bundle.  Aiming scalars (y1, y2, y_EP through
 * the axes; U, h', stop, a_stop through the bundle) are inputs.
 * ex, ey, rho, theta : [nb][2*ny*nx] (bundle b starts at b*2*ny*nx; count[b] entries valid)
 * count : [nb] = 2*survivors;  rms : [nb].
 * ex = ey = rho = theta = NULL: spot statistics only (count, rms) — nothing but 16 B per bundle
 * leaves the device (the all-reduce-of-moments alternative to gathering hits).  The RMS
 * there is the reference's sigma (:169-173) from merged (n, mean, M2) partials, within 1e-12 relative
 * of its two-pass form; a bundle's (count, rms) depend on the bundle alone — never on what else is in
 * the call, nor on how the launch is cut into workgroups.                                         */
int ort_full_trace_f64(ort_ctx *ctx, const ort_system *sys, int nb, const ort_bundle *bundles,
                       const double *axes, int64_t axes_len, int ny, int nx,
                       double *ex, double *ey, double *rho, double *theta,
                       int64_t *count, double *rms, unsigned flags);
/* Float32 rays (BASELINE config 5 names Float32): the trace, the stop test and the error vectors are
 * binary32; the centroid and RMS are still accumulated in binary64.                              */
int ort_full_trace_f32(ort_ctx *ctx, const ort_system *sys, int nb, const ort_bundle *bundles,
                       const float *axes, int64_t axes_len, int ny, int nx,
                       float *ex, float *ey, float *rho, float *theta,
                       int64_t *count, double *rms, unsigned flags);

/* wavegrad(eps, lambda) = (eps.x nu / lambda, eps.y nu / lambda), src/PupilSampling.jl:165-167: the transverse errors of
 * nb full_trace results in waves, without bringing them to the host first.  ex, ey, gx, gy : [nb][cap] (cap = 2*ny*nx,
 * the slab stride of ort_full_trace_f64); count : [nb] valid entries per slab; nu : [nb] = system.marginal.nu[end].
 * Out of place, like the reference's map; every pointer host or device per ORT_DEVICE_PTRS. */
int ort_wavegrad_f64(ort_ctx *ctx, int nb, int64_t cap, const int64_t *count, const double *nu, double lambda,
                     const double *ex, const double *ey, double *gx, double *gy, unsigned flags);
/* Testing aid: shift the context's look-back ticket base against the device counter, the bookkeeping fault that
 * ort_full_trace_* with ORT_FT_LOOKBACK must report (ORT_EHIP; device-pointer callers see count = -1, rms = NaN). */
int ort_ctx_test_skew_tickets(ort_ctx *ctx, int64_t delta);
/* Testing aid: with on != 0 an ORT_FT_FUSED launch names no workgroup for the bundles' scans and polls briefly, so every
 * placement waits in vain — the in-launch hand-off fault that ort_full_trace_* must report the same way. */
int ort_ctx_test_fused_no_scan(ort_ctx *ctx, int on);

/* ---- meridional real-ray trace: raytrace(surfaces, y, U, RealRay; K, p) ---------------
 * src/RayTracing.jl:145-169.  y_out, U_out, ts_out : [rows][ld] (row 0 = input ray;
 * ts_out = per-ray distances whose cumsum is RealRay.z, Types.jl:61-63; may be NULL).    */
int ort_trace_meridional_f64(ort_ctx *ctx, const ort_system *sys, int isys, int64_t nrays,
                             const double *y, const double *U,
                             double *y_out, double *U_out, double *ts_out, int64_t ld,
                             unsigned flags);
/* Base.asin's DomainError (src/RayTracing.jl:162: asin(y / R) with |y / R| > 1, reachable only by rounding at a
 * sphere's equator): the reference's raytrace of THAT ray throws.  The batch form traces every ray — the
 * offending ones are NaN from that surface on — and reports it: with host pointers the call returns ORT_EDOMAIN
 * (outputs written, message names the first ray and surface); with ORT_DEVICE_PTRS query the last launch of
 * the context after it has run: ray = -1 when there was none (blocks on the context's stream).             */
int ort_ctx_domain_error(ort_ctx *ctx, int64_t *ray, int *surface, int64_t *count);

/* ---- batched real-ray aiming: trace_chief_ray / trace_marginal_ray / trace_edge_rays -----
 * One (system, field) pair per entry: the FD-Newton drivers of src/RayTracing.jl:223-240 (real
 * marginal) and :265-296 (real chief, traced through the REVERSED prescription, which the
 * caller uploads as `rev`, built as :267-277 incl. quirk Q17) and the edge-ray search of
 * src/PupilSampling.jl:67-83 (Optim.BFGS in the reference — a third-party optimiser, parity unpinned; restated as
 * the same FD-Newton), giving the aiming scalars of src/PupilSampling.jl:94-103 that ort_full_trace_f64 consumes.
 * Every trace inside the loops is the reference's meridional sequence (tan / asin / atan, src/RayTracing.jl:151-167).
 * With ORT_FAST_MATH a prescription of spheres and planes only is traced there in direction-vector form without a
 * trigonometric call — the same function of the launch data to rounding; since the loops stop at |residual| <=
 * sqrt(eps) (RayTracing.jl:1) the two forms may end on iterates a few 1e-9 apart (U, y1, y2, y_EP, and with them the
 * one-call pipelines' outputs to ~1e-7; the latency of a single small call is what it buys: DESIGN.md section 7).
 * FITTED TO ONE DOCS FIGURE, not a restatement: an edge-ray search that ends outside the stop's edge takes one more
 * step to atol inside, so y1 / y2 are biased inward by <= atol and the grid's two edge rays pass the filter
 * r > a_stop of :132.  The reference's published Tessar spot size (0.11975, real_spot_diagram.png) comes out with
 * this rule and 0.64 % low without it; on other systems the side the reference's BFGS ends on is unknown, so the
 * survivor count can differ from the reference's by the x = 0 rays of the first and last pupil row (at most 2 rays,
 * 4 with their mirror images).  ORT_AIM_EDGE_AS_FOUND turns the rule off.
 * `in`, `out`: host arrays [n] (device arrays with ORT_DEVICE_PTRS).                        */
typedef struct ort_aim_in {
    int32_t system;        /* index into fwd and rev */
    int32_t stop;          /* system.stop */
    int32_t layout_fwd;    /* 1 = the prescription is a Layout{Aspheric} (Q16) */
    int32_t layout_rev;    /* 1 = the reversed prescription was built from a Layout (:272-274) */
    double H;              /* normalised field, |H| <= 1 (checked: ORT_EDOMAIN) */
    double y_marg;         /* system.marginal.y[1] */
    double a_stop;         /* system.a[stop] */
    double chief_y_end;    /* system.chief.y[end] */
    double chief_u_end;    /* system.chief.u[end] */
    double f;              /* system.f */
    double atol;           /* sqrt(eps()) by default in the reference */
} ort_aim_in;
typedef struct ort_aim_out {
    double U, y1, y2, y_EP, hprime, EP_t, Ubar;
    double XP_t;           /* real_chief.z[end] - real_chief.z[end-1] (src/RayTracing.jl:294): exit pupil from the last vertex */
    int32_t iters;
    int32_t ok;            /* 0 = a Newton loop hit its iteration cap */
} ort_aim_out;
int ort_aim_f64(ort_ctx *ctx, const ort_system *fwd, const ort_system *rev, int n,
                const ort_aim_in *in, ort_aim_out *out, unsigned flags);

/* ---- meridional fans: TSA(surfaces, system, k_rays) and the caustic ray set ----------------------
 * src/SeidelAberrations.jl:116-137: for every request the k_rays rays y = range(y_m / k, y_m, k), U = 0 through
 * system `system` of the batch in one launch, each extended to the exit pupil and to the focal plane:
 *   y_XP[i] = ray.y[end] + tan(ray.u[end]) XP_t,   eps[i] = ray.y[end] + tan(ray.u[end]) (BFD - sag(ray))
 * (the last ray is the real marginal ray itself, :127-128: its sag is taken against the paraxial vertex, s_last without the
 * last thickness, :125-127 — the same number whenever the prescription ends in image space).  descending != 0 walks the range backwards,
 * range(y_m, y_m / k, k): with BFD = the paraxial or the marginal back focal distance this is the ray set
 * and the image-space end points of the caustic plot (ext/MakieExtension.jl:364-381; its per-surface
 * polylines are ort_trace_meridional_f64's history).  y_m, XP_t come from ort_aim_f64 (y_EP, XP_t).
 * in : [n] host (device with ORT_DEVICE_PTRS); y_XP, eps : [n][k_rays].                              */
typedef struct ort_fan_in {
    int32_t system;        /* index into the ort_system batch */
    int32_t layout_mode;   /* 1 = the prescription is a Layout{Aspheric} (Q16), as ORT_LAYOUT_INPUT */
    double y_marg;         /* real_marginal.y[1] */
    double XP_t;           /* real_chief.z[end] - real_chief.z[end-1] */
    double BFD;            /* back focal distance the rays are extended to, from the last vertex */
} ort_fan_in;
int ort_fan_f64(ort_ctx *ctx, const ort_system *sys, int n, const ort_fan_in *in, int k_rays, int descending,
                double *y_XP, double *eps, unsigned flags);

/* ---- batched first-order solve + Seidel sums ------------------------------------------------
 * solve(surfaces, a, h′) (src/RayTracing.jl:302-335: Lens(), the two paraxial traces, stop
 * selection, marginal / chief construction) and aberrations(surfaces, system, λ, δn)
 * (src/SeidelAberrations.jl:6-53), one thread per system; for Monte-Carlo runs over perturbed
 * instances.  R, t, n : [nsys][rows]; a : [nsys][rows-1]; dn : [nsys][rows] or NULL; hprime : [nsys]. */
typedef struct ort_first_order {
    double f, EBFD, EFFD, N, FOV, EP_D, EP_t, XP_D, XP_t, H;
    double y_marg, chief_y_end, chief_u_end, nu_end, BFD, PN;
    double W040, W131, W222, W220, W311, W020, W111, W220P;     /* waves at lambda */
    int32_t stop, k;
} ort_first_order;
/* The last thickness of every prescription must be 0 or infinite (image space): with a finite non-zero one
 * Lens() keeps the last row and the reference needs `rows` semi-diameters (DimensionMismatch for rows-1):
 * ORT_EINVAL (host pointers; with ORT_DEVICE_PTRS the stop search is bounded to the rows-1 given).        */
int ort_first_order_f64(ort_ctx *ctx, int nsys, int rows, const double *R, const double *t, const double *n,
                        const double *a, const double *dn, const double *hprime, double lambda,
                        ort_first_order *out, unsigned flags);

/* The same solve with the PER-SURFACE third-order contributions of `aberrations` (the vectors of the
 * Aberration struct, src/SeidelAberrations.jl:25-34, src/Types.jl:143-167) and the paraxial incidence table of
 * `incidences(surfaces, system)` = [ni nī i ī] (src/RayTracing.jl:338-353), for every system of the batch:
 *   surf : [ORT_SURF_COUNT][nsys][rows-1]   component-major; system s, surface j at surf[(c*nsys + s)*(rows-1) + j]
 *   inc  : [4][nsys][rows-1]                 columns ni, nī, i, ī in that order
 * either may be NULL.  `out` as in ort_first_order_f64 (W220M = W220P + W222, W220T = W220P + 1.5 W222).   */
enum { ORT_SURF_SPHERICAL = 0, ORT_SURF_COMA, ORT_SURF_ASTIGMATISM, ORT_SURF_SAGITTAL, ORT_SURF_DISTORTION,
       ORT_SURF_AXIAL, ORT_SURF_LATERAL, ORT_SURF_PETZVAL, ORT_SURF_MEDIAL, ORT_SURF_TANGENTIAL, ORT_SURF_COUNT };
int ort_aberrations_f64(ort_ctx *ctx, int nsys, int rows, const double *R, const double *t, const double *n,
                        const double *a, const double *dn, const double *hprime, double lambda,
                        ort_first_order *out, double *surf, double *inc, unsigned flags);

/* ---- device-resident spot pipeline: full_trace(solve(surfaces, a, h′), H, k_rays).RMS -------
 * for nsys spherical prescriptions x nfields fields in ONE call with no host round trip between
 * the stages: first-order solve (src/RayTracing.jl:302-323), real-ray aiming (:223-296,
 * src/PupilSampling.jl:67-83), pupil axes (:121-122), grid trace + stop filter + mirrored RMS
 * (:123-146,169-173; statistics-only route).  R, t, n : [nsys][rows]; a : [nsys][rows-1];
 * hprime : [nsys]; fields : [nfields]; count, rms : [nsys][nfields]; fo_out : [nsys] or NULL.
 * The last thickness of every prescription must be 0 (image space), as in the reference's tests. */
int ort_spot_batch_f64(ort_ctx *ctx, int nsys, int rows, const double *R, const double *t, const double *n,
                       const double *a, const double *hprime, int nfields, const double *fields, int k_rays,
                       ort_first_order *fo_out, int64_t *count, double *rms, unsigned flags);
/* The same pipeline returning the error vectors too — `full_trace(solve(surfaces, a, h′), H, k_rays)` of
 * src/PupilSampling.jl:159-163 for every (system, field) in one call: ex, ey, rho, theta :
 * [nsys*nfields][2*k_rays*(k_rays/2)], count[b] entries valid in slab b (= RealRayError.x/.y/.r/.t,
 * src/Types.jl:184-192), rms[b] = RealRayError.RMS. */
int ort_full_trace_batch_f64(ort_ctx *ctx, int nsys, int rows, const double *R, const double *t, const double *n,
                             const double *a, const double *hprime, int nfields, const double *fields, int k_rays,
                             ort_first_order *fo_out, double *ex, double *ey, double *rho, double *theta,
                             int64_t *count, double *rms, unsigned flags);
/* The same for aspheric prescriptions — `full_trace(solve(layout, a, h′), H, k_rays)` with
 * layout = Layout(R, t, n, K, p) (src/Types.jl:82-112): K : [nsys][rows] conic constants or NULL,
 * coef : [nsys][rows][ncoef] power-series coefficients of p (or NULL / ncoef = 0).  The real chief ray is
 * aimed through the reference's reversed Layout, K and p plainly reversed (src/RayTracing.jl:272-274).
 * ex = ey = rho = theta = NULL: statistics only. */
int ort_full_trace_layout_batch_f64(ort_ctx *ctx, int nsys, int rows, const double *R, const double *t, const double *n,
                                    const double *K, const double *coef, int ncoef,
                                    const double *a, const double *hprime, int nfields, const double *fields, int k_rays,
                                    ort_first_order *fo_out, double *ex, double *ey, double *rho, double *theta,
                                    int64_t *count, double *rms, unsigned flags);
/* same call with the pupil-grid trace in Float32 (solve and aiming stay Float64: they are O(rows)
 * per system and decide the grid end points).                                                     */
int ort_spot_batch_f32(ort_ctx *ctx, int nsys, int rows, const double *R, const double *t, const double *n,
                       const double *a, const double *hprime, int nfields, const double *fields, int k_rays,
                       ort_first_order *fo_out, int64_t *count, double *rms, unsigned flags);

/* ---- paraxial y-nu trace: raytrace(lens, y, ω, a; clip) --------------------------------
 * src/RayTracing.jl:127-143 (+ transfer/refract :55-69).  nlens lenses of k rows each
 * (Lens.M columns τ, ϕ: [nlens][k]); a: [nlens][k] or NULL (fill(Inf)); rays_per_lens rays
 * per lens: y, w : [nlens*rays_per_lens]; rt_y, rt_w : [k+1][ld], ld >= nlens*rays_per_lens. */
int ort_trace_paraxial_f64(ort_ctx *ctx, int nlens, int k,
                           const double *tau, const double *phi, const double *a,
                           int64_t rays_per_lens, const double *y, const double *w,
                           double *rt_y, double *rt_w, int64_t ld, unsigned flags);

/* ---- ABCD: TransferMatrix(lens), src/TransferMatrix.jl:1-6 ----------------------------
 * M : [nlens][4] row-major {A, B, C, D}.                                                  */
int ort_abcd_f64(ort_ctx *ctx, int nlens, int k, const double *tau, const double *phi,
                 double *M, unsigned flags);
/* transfer(M, v, τ, τ′) (:8-10): nv vectors through ONE matrix; v, out : [nv][2].
 * tau, tau_p : [nv] object / image space reduced distances.                              */
int ort_abcd_transfer_f64(ort_ctx *ctx, const double *M, int64_t nv, const double *v,
                          const double *tau, const double *tau_p, double *out,
                          unsigned flags);
/* reverse_transfer(M, v, τ′, τ) (:13): extend(M, τ, τ′) \ v.                              */
int ort_abcd_reverse_transfer_f64(ort_ctx *ctx, const double *M, int64_t nv, const double *v,
                                  const double *tau_p, const double *tau, double *out,
                                  unsigned flags);

/* ---- multi-GPU reassembly: ONE all-gather of image-plane hits over RCCL / xGMI -------------
 * One process per GPU.  The path shards over independent units (bundles, pupil rows: no data-path
 * collective); the only exchange is the reassembly of per-rank hit slabs in rank order — which, with
 * contiguous rank-ordered shards, reproduces the reference's append order
 * (src/PupilSampling.jl:134-137).  librccl.so is loaded on first use (dlopen), so the library has
 * no link-time dependency on it.
 *   rank 0: ort_comm_unique_id(id);  ship the 128 bytes to every rank (file, socket, MPI, ...)
 *   all   : ort_comm_create(ctx, nranks, rank, id, &comm)
 *   all   : ort_trace_grid_f64(... xf = hits, yf = hits + count ...)      the trace writes the packed slab
 *           ort_allgather_hits_packed_f64(comm, hits, count, gathered)    ONE ncclAllGather
 *           ... trace the next shard meanwhile ...                         (the collective runs on the
 *           ort_comm_wait(comm) / ort_comm_synchronize(comm)               communicator's own stream)
 * Every collective below is ordered after the work already queued on the context's stream and runs on the
 * communicator's own stream: it overlaps whatever the context's stream is given next.  ort_comm_wait makes the
 * context's stream wait for the collectives issued so far (no host block); ort_comm_synchronize blocks the host.
 * All data pointers are device pointers.                                                            */
typedef struct ort_comm ort_comm;
#define ORT_UNIQUE_ID_BYTES 128
int ort_comm_unique_id(void *id128);
int ort_comm_create(ort_ctx *ctx, int nranks, int rank, const void *id128, ort_comm **out);
int ort_comm_destroy(ort_comm *comm);
int ort_comm_size(const ort_comm *comm);
int ort_comm_rank(const ort_comm *comm);
int ort_comm_wait(ort_comm *comm);
/* the context's stream waits for the collective issued `lag` calls before the latest (0 = the latest, lag < 8):
 * with two hit buffers, ort_comm_wait_lag(comm, 1) before re-tracing into a buffer orders the trace after the
 * all-gather that last read it while the latest all-gather still overlaps the trace.                     */
int ort_comm_wait_lag(ort_comm *comm, int lag);
int ort_comm_synchronize(ort_comm *comm);
/* hits : [2][count] (x at +0, y at +count: pass xf = hits, yf = hits + count to the trace);
 * gathered : [nranks][2][count], rank r's x slab at r*2*count, its y slab at r*2*count + count.       */
int ort_allgather_hits_packed_f64(ort_comm *comm, const double *hits, int64_t count, double *gathered);
int ort_allgather_hits_packed_f32(ort_comm *comm, const float *hits, int64_t count, float *gathered);
/* separate x / y arrays: gx, gy : [nranks*count], rank r's slab at r*count (the two slabs go out fused
 * into one RCCL launch).                                                                              */
int ort_allgather_hits_f64(ort_comm *comm, const double *xf, const double *yf, int64_t count,
                           double *gx, double *gy);
/* ragged slabs (compacted survivors, uneven shards): counts are exchanged first (8 bytes per rank, read by the
 * host: this call blocks on that), then rank r's `count` values land at the exclusive offset of the counts.
 * gathered : [capacity] device; counts : [nranks] HOST, out (may be NULL).                            */
int ort_allgather_ragged_f64(ort_comm *comm, const double *values, int64_t count, double *gathered,
                             int64_t capacity, int64_t *counts);

#ifdef __cplusplus
}
#endif
#endif /* ORT_H */

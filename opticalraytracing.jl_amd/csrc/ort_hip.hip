// ort_hip.hip — C ABI (include/ort.h) over the gfx950 kernels of ort_kernels.hpp.
//
// Host side only marshals: it derives the per-surface records (the wave-uniform part of
// the reference loop, src/PupilSampling.jl:1-32, hoisted out of the per-ray work), takes
// tan() of bundle angles (:38-39), sizes the launch and owns device scratch.  There is no
// CPU compute path: every entry point ends in a kernel launch or fails.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared   (see build.py)
#include "../../include/ort.h"
#include "ort_kernels.hpp"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

using namespace ort;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ORT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                \
    } while (0)

// growable device scratch slot
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return ORT_OK;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(ORT_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return ORT_OK;
    }
    void release() { if (p) { hipError_t e = hipFree(p); (void)e; } p = nullptr; cap = 0; }
};

constexpr int kSmallTiles = 32;     // full_trace bundles of at most this many 512-ray tiles finish in one launch (k_ft_small_finish)
constexpr int kSmallGridTiles = 256;  // full_trace launches of at most this many tiles (of <= kSmallTiles per bundle) trace one ray per lane:
                                      // 8 waves per tile on 1024 SIMDs — beyond two waves per SIMD the two-ray form's lower instruction count wins
constexpr size_t kZeroCopyBytes = (size_t)1 << 20;   // spot pipelines whose packed block [inputs | results] is at most this large run on the
                                                     // pinned host block directly (no copy dispatches)
#ifndef ORT_WALK_TARGET
#define ORT_WALK_TARGET 16384
#endif
#ifndef ORT_FUSE_MARGIN
#define ORT_FUSE_MARGIN 4096
#endif
#ifndef ORT_FUSE_SCAN_LAG
#define ORT_FUSE_SCAN_LAG 1024
#endif
constexpr int kFuseMargin = ORT_FUSE_MARGIN;     // fused full_trace: workgroups between a bundle's last tile and the first placement of its tiles
constexpr int kFuseScanLag = ORT_FUSE_SCAN_LAG;  // ... and the workgroup that runs the bundle's scan (every tile's workgroup has long been dispatched by then)
static_assert(kFuseScanLag <= kFuseMargin, "the scan runs ahead of the placements");
constexpr int64_t kWalkTargetGroups = ORT_WALK_TARGET;   // statistics-only walk route: workgroups a launch is cut into when it has that many spans
                                               // (~8 per resident slot of the chip); fewer spans: one workgroup per span
constexpr int kSmallPairs = 256;    // spot pipelines of at most this many (system, field) pairs prepare in one launch (k_small_prepare)
constexpr size_t kPackedVectorBytes = 32u << 20;   // host callers: error vectors up to this size come back in ONE copy

enum { SL_IN0 = 0, SL_IN1, SL_IN2, SL_IN3, SL_OUT0, SL_OUT1, SL_OUT2, SL_OUT3, SL_OUT4,
       SL_BUNDLES, SL_AXES, SL_WEX, SL_WEY, SL_WR, SL_WTH, SL_TCNT, SL_TSX, SL_TSY, SL_TRM,
       SL_TOFF, SL_TSQ, SL_AGG, SL_RES0, SL_RES1, SL_TAB0, SL_TAB1, SL_TAB2, SL_TAB3,
       SL_SB_FO, SL_SB_REC, SL_SB_MF, SL_SB_MR, SL_SB_TLF, SL_SB_TLR, SL_SB_AIN, SL_SB_AOUT, SL_SB_ENDS, SL_SB_FLAG,
       SL_SB_FIELDS, SL_SB_A, SL_SB_HP, SL_SB_PACK, SL_SB_CEXT, SL_SB_CREV, SL_FTSTATE, SL_FTTICKET, SL_FTERR, SL_FTDONE, SL_FTREADY, SL_DOMAIN, SL_RBSLOPES, SL_COUNT };

}  // namespace

struct ort_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Scratch slot[SL_COUNT];
    std::vector<unsigned char> bundle_cache;   // last uploaded DevBundle bytes
    void* pin = nullptr;                       // page-locked staging for the packed small transfers
    size_t pin_cap = 0;
    // full_trace look-back (k_trace<FT_FULL>): words of earlier launches are told apart by the epoch, tickets by the base
    size_t ft_state_cap = 0;                   // capacity of SL_FTSTATE the zero fill was done for
    size_t ft_ready_cap = 0;                   // the same for SL_FTREADY (fused route)
    bool test_fused_no_scan = false;           // ort_ctx_test_fused_no_scan
    unsigned ft_epoch = 0;
    unsigned long long ft_ticket_base = 0;
    std::vector<struct ort_comm*> comms;       // live communicators of this context (their streams may hold buffers in flight)
    int pinned(size_t bytes, unsigned char** out)
    {
        if (bytes > pin_cap) {
            if (pin) { hipError_t e = hipHostFree(pin); (void)e; pin = nullptr; pin_cap = 0; }
            const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
            hipError_t e = hipHostMalloc(&pin, want, hipHostMallocDefault);
            if (e != hipSuccess) { pin = nullptr; return fail(ORT_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
            pin_cap = want;
        }
        *out = static_cast<unsigned char*>(pin);
        return ORT_OK;
    }
};

struct ort_system {
    ort_ctx* ctx = nullptr;
    int nsys = 0, rows = 0, ncoef = 0;
    SurfRec<double>* rec64 = nullptr;
    SurfRec<float>* rec32 = nullptr;
    double* coef64 = nullptr;      // [nsys][rows][ncoef] raw coefficients (meridional kernels)
    double* poly64 = nullptr;      // [nsys][S][kPolyRec] polynomial records of the skew kernels (ort_device.hpp)
    float* poly32 = nullptr;
    MerSurf* mer = nullptr;        // [nsys][S]
    std::vector<double> t_last;    // t[rows-1] per system (meridional ts tail)
    double* d_tlast = nullptr;     // the same on the device (aiming kernel)
    void* slab = nullptr;          // the one device allocation all of the above point into
    double* ap64 = nullptr;        // [nsys][S] squared clear semi-diameters (ort_system_set_apertures), or null
    float* ap32 = nullptr;
    int arms64 = 0, arms32 = 0;    // ARMS level of the batch (arms_of_needs over its rows): which kernel build runs it
};

namespace {

// Per-surface records (+ polynomial records when the batch carries coefficients) of nsys prescriptions; the same
// make_rec / make_poly_rec the device-side table builder runs (ort_device.hpp).
template <typename T>
int build_records(int nsys, int rows, int ncoef, const double* R, const double* t, const double* n,
                  const double* K, const double* coef, std::vector<SurfRec<T>>& out, std::vector<T>& pout)
{
    const int S = rows - 1;
    int arms = 0;
    out.resize((size_t)nsys * S);
    const bool hasp = coef && ncoef > 0;
    if (hasp) pout.assign((size_t)nsys * S * kPolyRec, T(0));
    for (int s = 0; s < nsys; ++s) {
        const double* Rs = R + (size_t)s * rows;
        const double* ts = t + (size_t)s * rows;
        const double* ns = n + (size_t)s * rows;
        const double* Ks = K ? K + (size_t)s * rows : nullptr;
        for (int i = 0; i < S; ++i) {
            SurfRec<T> r;
            memset(&r, 0, sizeof r);
            int nc = 0, pcls = 0;
            if (hasp)
                pcls = make_poly_rec<T>(pout.data() + ((size_t)s * S + i) * kPolyRec, coef + ((size_t)s * rows + (i + 1)) * ncoef, ncoef, &nc);
            arms |= make_rec<T>(r, (T)ts[i], (T)Rs[i + 1], (T)ns[i], (T)ns[i + 1], Ks ? (T)Ks[i + 1] : T(0), nc, pcls);   // NEED_* bits
            out[(size_t)s * S + i] = r;
        }
    }
    return arms_of_needs(arms);
}

template <typename T> struct Sel;
template <> struct Sel<double> {
    static const SurfRec<double>* rec(const ort_system* s) { return s->rec64; }
    static const double* poly(const ort_system* s) { return s->poly64; }
    static const double* ap2(const ort_system* s) { return s->ap64; }
    static int arms(const ort_system* s) { return s->arms64; }
};
template <> struct Sel<float> {
    static const SurfRec<float>* rec(const ort_system* s) { return s->rec32; }
    static const float* poly(const ort_system* s) { return s->poly32; }
    static const float* ap2(const ort_system* s) { return s->ap32; }
    static int arms(const ort_system* s) { return s->arms32; }
};

// pick the kernel instantiation
template <typename T, bool GRID, bool HIST, bool SUMM, int FT, int RPT = kRPT>
int launch_trace(ort_ctx* ctx, const TraceParams<T>& p, int64_t blocks, unsigned flags)
{
    if (blocks <= 0) return ORT_OK;
    if (blocks > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large: %lld workgroups", (long long)blocks);
    // ORT_FAST_MATH promises <= 1e-10 and the reference's status.  Its rounding differences (hardware seeds + one Newton step:
    // ~2^-49 per root / reciprocal, ~10 x an IEEE operation's) grow with the DEPTH of the prescription as the path amplifies
    // them: measured on relay chains (scripts/fast_depth.py, profiles/r04_fast_depth.log) worst 3.5e-11 at 47 loop iterations, a
    // few rays of 60,000 past 1e-10 from 54 on, one or two of them well-conditioned at 63.  Beyond ORT_FAST_MAX_SURFACES the
    // promise is kept by the reference sequence itself: such systems run MATH_IEEE under either flag.
    const bool fast = (flags & ORT_FAST_MATH) && p.S <= ORT_FAST_MAX_SURFACES;
    dim3 g((unsigned)blocks), b(kTile / RPT);
    // the build that carries the arms this batch's rows need (p.arms: ort_system::arms64 / arms32, surface_step_n)
#define ORT_LAUNCH(M, A) hipLaunchKernelGGL((k_trace<T, M, A, GRID, HIST, SUMM, FT, RPT>), g, b, 0, ctx->stream, p)
    if (p.arms <= ARMS_BASIC)        { if (fast) ORT_LAUNCH(MATH_FAST, ARMS_BASIC);   else ORT_LAUNCH(MATH_IEEE, ARMS_BASIC); }
    else if (p.arms == ARMS_GENERAL) { if (fast) ORT_LAUNCH(MATH_FAST, ARMS_GENERAL); else ORT_LAUNCH(MATH_IEEE, ARMS_GENERAL); }
    else if (p.arms == ARMS_EVEN && fast) ORT_LAUNCH(MATH_FAST, ARMS_EVEN);          // (the reference sequence has one polynomial build)
    else                             { if (fast) ORT_LAUNCH(MATH_FAST, ARMS_POLY);    else ORT_LAUNCH(MATH_IEEE, ARMS_POLY); }
#undef ORT_LAUNCH
    HIP_TRY(hipGetLastError());
    return ORT_OK;
}

template <typename T, bool GRID>
int launch_trace_modes(ort_ctx* ctx, const TraceParams<T>& p0, int64_t blocks, bool hist, bool summ, unsigned flags)
{
    TraceParams<T> p = p0;
    if (kSummaryWalks<T> && GRID && summ && !hist) {               // summary only: walk_group tiles of a bundle per workgroup (k_trace, SWALK)
        p.walk_group = (int)std::min<int64_t>(p.tiles_per_bundle, std::max<int64_t>(1, blocks / kWalkTargetGroups));
        blocks = (blocks / p.tiles_per_bundle) * ((p.tiles_per_bundle + p.walk_group - 1) / p.walk_group);
    }
    if (hist && summ)  return launch_trace<T, GRID, true, true, FT_NONE>(ctx, p, blocks, flags);
    if (hist && !summ) return launch_trace<T, GRID, true, false, FT_NONE>(ctx, p, blocks, flags);
    if (!hist && summ) return launch_trace<T, GRID, false, true, FT_NONE>(ctx, p, blocks, flags);
    return fail(ORT_EINVAL, "no output requested");
}

// finite-conjugate launch rule (ORT_RAYBASIS): the launch slopes per pupil row and column, one tiny launch ahead of the trace
// (k_make_slope_axes); needs p.bundles, p.axes, p.ny, p.nx
template <typename T>
int make_rb_slopes(ort_ctx* ctx, TraceParams<T>& p, int nb);

int check_ctx(ort_ctx* ctx)
{
    if (!ctx) return fail(ORT_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return ORT_OK;
}

int check_sys(ort_ctx* ctx, const ort_system* sys)
{
    if (!sys) return fail(ORT_EINVAL, "null system");
    if (sys->ctx != ctx) return fail(ORT_EINVAL, "system belongs to another context");
    return ORT_OK;
}

int sync_comm_streams(ort_ctx* ctx);    // (defined beside ort_comm, below)
void detach_comm(struct ort_comm* c);

// copy a host array into a scratch slot (synchronous with respect to the host buffer)
template <typename T>
int to_device(ort_ctx* ctx, int slot, const T* host, size_t count, const T** dev)
{
    int rc = ctx->slot[slot].ensure(count * sizeof(T));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->slot[slot].p, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *dev = static_cast<const T*>(ctx->slot[slot].p);
    return ORT_OK;
}

template <typename T>
int dev_out(ort_ctx* ctx, int slot, size_t count, T** dev)
{
    int rc = ctx->slot[slot].ensure(count * sizeof(T));
    if (rc) return rc;
    *dev = static_cast<T*>(ctx->slot[slot].p);
    return ORT_OK;
}

template <typename T>
int from_device(ort_ctx* ctx, T* host, const T* dev, size_t count)
{
    HIP_TRY(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return ORT_OK;
}

template <typename T>
int make_rb_slopes(ort_ctx* ctx, TraceParams<T>& p, int nb)
{
    if (!p.raybasis) return ORT_OK;
    const size_t cnt = (size_t)nb * (size_t)(p.ny + p.nx);
    T* d = nullptr;
    int rc = dev_out<T>(ctx, SL_RBSLOPES, cnt, &d); if (rc) return rc;
    hipLaunchKernelGGL((k_make_slope_axes<T>), dim3((unsigned)((cnt + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       nb, p.ny, p.nx, p.bundles, p.axes, d);
    HIP_TRY(hipGetLastError());
    p.rb_slopes = d;
    return ORT_OK;
}

template <typename T>
int upload_bundles(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles, int ny, int nx,
                   int64_t axes_len, unsigned flags, const DevBundle<T>** dev)
{
    std::vector<DevBundle<T>> hb((size_t)nb);
    const int S = sys->rows - 1;
    for (int b = 0; b < nb; ++b) {
        const ort_bundle& s = bundles[b];
        if (s.system < 0 || s.system >= sys->nsys) return fail(ORT_EINVAL, "bundle %d: system %d out of range", b, s.system);
        if (s.stop < 0 || s.stop > S) return fail(ORT_EINVAL, "bundle %d: stop %d out of range 0..%d", b, s.stop, S);
        if (s.yaxis_off < 0 || s.yaxis_off + ny > axes_len || s.xaxis_off < 0 || s.xaxis_off + nx > axes_len)
            return fail(ORT_EINVAL, "bundle %d: axis offsets outside the axes buffer", b);
        DevBundle<T> d;
        memset(&d, 0, sizeof d);
        d.system = s.system;
        d.stop = s.stop - 1;
        d.u = (T)std::tan(s.U);           // PupilSampling.jl:38
        d.v = (T)std::tan(s.V);           // :39
        {   // k = normalize([v, u, 1])  (:40-41, Q6): sqrt of the sequential sum of squares, times inv(norm)
            const T nrm = std::sqrt((d.v * d.v + d.u * d.u) + T(1));
            const T inv = T(1) / nrm;
            d.k0 = d.v * inv; d.k1 = d.u * inv; d.k2 = inv;
        }
        d.a_stop = (T)s.a_stop;
        d.hprime = (T)s.hprime;
        d.ybar = (T)s.ybar;
        d.z0 = (T)s.z0;
        d.yoff = s.yaxis_off;
        d.xoff = s.xaxis_off;
        hb[b] = d;
    }
    (void)flags;
    const size_t bytes = hb.size() * sizeof(DevBundle<T>);
    const unsigned char* raw = reinterpret_cast<const unsigned char*>(hb.data());
    const bool same = ctx->bundle_cache.size() == bytes && ctx->slot[SL_BUNDLES].p &&
                      memcmp(ctx->bundle_cache.data(), raw, bytes) == 0;
    if (!same) {
        int rc = ctx->slot[SL_BUNDLES].ensure(bytes);
        if (rc) return rc;
        // the staging vector dies with this frame: finish the copy before returning
        HIP_TRY(hipMemcpyAsync(ctx->slot[SL_BUNDLES].p, raw, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->bundle_cache.assign(raw, raw + bytes);
    }
    *dev = static_cast<const DevBundle<T>*>(ctx->slot[SL_BUNDLES].p);
    return ORT_OK;
}

template <typename T, typename OUT>
int trace_grid_impl(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles,
                    const T* axes, int64_t axes_len, int ny, int nx, const OUT* out, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    rc = check_sys(ctx, sys); if (rc) return rc;
    if (nb <= 0 || !bundles || !axes || ny <= 0 || nx <= 0 || !out) return fail(ORT_EINVAL, "bad grid arguments");
    const int S = sys->rows - 1;
    const int64_t rpb = (int64_t)ny * nx;
    if (rpb > 0x7fffffffLL - kTile) return fail(ORT_EINVAL, "bundle of %lld rays is too large (max 2^31)", (long long)rpb);
    const int64_t N = rpb * nb;
    const bool hist = out->xv && out->yv;
    const bool summ = out->xf || out->yf || out->xs || out->ys || out->status;
    if (!hist && !summ) return fail(ORT_EINVAL, "no output requested");
    if (hist && out->ld < N) return fail(ORT_EINVAL, "ld %lld < rays %lld", (long long)out->ld, (long long)N);
    if ((out->xs || out->ys)) for (int b = 0; b < nb; ++b) if (bundles[b].stop <= 0) return fail(ORT_EINVAL, "xs/ys requested but bundle %d has no stop", b);
    const bool devp = flags & ORT_DEVICE_PTRS;

    TraceParams<T> p;
    memset(&p, 0, sizeof p);
    p.recs = Sel<T>::rec(sys); p.polys = Sel<T>::poly(sys); p.arms = Sel<T>::arms(sys); p.S = S; p.apert2 = Sel<T>::ap2(sys);
    rc = upload_bundles<T>(ctx, sys, nb, bundles, ny, nx, axes_len, flags, &p.bundles); if (rc) return rc;
    p.ny = ny; p.nx = nx; p.rpb = rpb; p.tiles_per_bundle = (int)((rpb + kTile - 1) / kTile);
    p.raybasis = (flags & ORT_RAYBASIS) ? 1 : 0;
    const int64_t blocks = (int64_t)nb * p.tiles_per_bundle;

    if (devp) {
        p.axes = axes;
        rc = make_rb_slopes<T>(ctx, p, nb); if (rc) return rc;
        p.xv = out->xv; p.yv = out->yv; p.ld = out->ld;
        p.xf = out->xf; p.yf = out->yf; p.xs = out->xs; p.ys = out->ys; p.status = out->status;
        return launch_trace_modes<T, true>(ctx, p, blocks, hist, summ, flags);
    }
    // host buffers: stage through context scratch
    rc = to_device<T>(ctx, SL_AXES, axes, (size_t)axes_len, &p.axes); if (rc) return rc;
    rc = make_rb_slopes<T>(ctx, p, nb); if (rc) return rc;
    T *dxv = nullptr, *dyv = nullptr;
    if (hist) {
        rc = dev_out<T>(ctx, SL_OUT0, (size_t)S * N, &dxv); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_OUT1, (size_t)S * N, &dyv); if (rc) return rc;
        p.xv = dxv; p.yv = dyv; p.ld = N;
    }
    T* dsum = nullptr; int32_t* dst = nullptr;
    if (summ) {
        rc = dev_out<T>(ctx, SL_OUT2, (size_t)4 * N, &dsum); if (rc) return rc;
        rc = dev_out<int32_t>(ctx, SL_OUT3, (size_t)N, &dst); if (rc) return rc;
        p.xf = dsum; p.yf = dsum + N;
        bool have_stop = true;
        for (int b = 0; b < nb; ++b) have_stop = have_stop && bundles[b].stop > 0;
        if (have_stop) { p.xs = dsum + 2 * N; p.ys = dsum + 3 * N; }
        p.status = dst;
    }
    rc = launch_trace_modes<T, true>(ctx, p, blocks, hist, summ, flags); if (rc) return rc;
    if (hist) {
        if (out->ld == N) {
            rc = from_device<T>(ctx, out->xv, dxv, (size_t)S * N); if (rc) return rc;
            rc = from_device<T>(ctx, out->yv, dyv, (size_t)S * N); if (rc) return rc;
        } else {
            HIP_TRY(hipMemcpy2DAsync(out->xv, out->ld * sizeof(T), dxv, N * sizeof(T), N * sizeof(T), S, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpy2DAsync(out->yv, out->ld * sizeof(T), dyv, N * sizeof(T), N * sizeof(T), S, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (summ) {
        if (out->xf) { rc = from_device<T>(ctx, out->xf, p.xf, (size_t)N); if (rc) return rc; }
        if (out->yf) { rc = from_device<T>(ctx, out->yf, p.yf, (size_t)N); if (rc) return rc; }
        if (out->xs && p.xs) { rc = from_device<T>(ctx, out->xs, p.xs, (size_t)N); if (rc) return rc; }
        if (out->ys && p.ys) { rc = from_device<T>(ctx, out->ys, p.ys, (size_t)N); if (rc) return rc; }
        if (out->status) { rc = from_device<int32_t>(ctx, out->status, dst, (size_t)N); if (rc) return rc; }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ORT_OK;
}

template <typename T>
int trace_list_impl(ort_ctx* ctx, const ort_system* sys, int isys, int64_t nrays,
                    const T* y, const T* x, const T* U, const T* V,
                    T* xv, T* yv, int64_t ld, int32_t* status, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    rc = check_sys(ctx, sys); if (rc) return rc;
    if (isys < 0 || isys >= sys->nsys) return fail(ORT_EINVAL, "system index %d out of range", isys);
    if (nrays < 0 || !y || !x || !U || !V) return fail(ORT_EINVAL, "bad ray list arguments");
    if (nrays == 0) return ORT_OK;
    const int S = sys->rows - 1;
    const bool hist = xv && yv;
    const bool summ = status != nullptr;
    if (!hist && !summ) return fail(ORT_EINVAL, "no output requested");
    if (hist && ld < nrays) return fail(ORT_EINVAL, "ld %lld < rays %lld", (long long)ld, (long long)nrays);
    TraceParams<T> p;
    memset(&p, 0, sizeof p);
    p.recs = Sel<T>::rec(sys); p.polys = Sel<T>::poly(sys); p.arms = Sel<T>::arms(sys); p.S = S; p.apert2 = Sel<T>::ap2(sys);
    p.nrays = nrays; p.isys = isys; p.slopes_given = (flags & ORT_INPUT_SLOPES) ? 1 : 0;
    const int64_t blocks = (nrays + kTile - 1) / kTile;
    if (flags & ORT_DEVICE_PTRS) {
        p.ly = y; p.lx = x; p.lU = U; p.lV = V;
        p.xv = xv; p.yv = yv; p.ld = ld; p.status = status;
        return launch_trace_modes<T, false>(ctx, p, blocks, hist, summ, flags);
    }
    rc = to_device<T>(ctx, SL_IN0, y, (size_t)nrays, &p.ly); if (rc) return rc;
    rc = to_device<T>(ctx, SL_IN1, x, (size_t)nrays, &p.lx); if (rc) return rc;
    rc = to_device<T>(ctx, SL_IN2, U, (size_t)nrays, &p.lU); if (rc) return rc;
    rc = to_device<T>(ctx, SL_IN3, V, (size_t)nrays, &p.lV); if (rc) return rc;
    T *dxv = nullptr, *dyv = nullptr; int32_t* dst = nullptr;
    if (hist) {
        rc = dev_out<T>(ctx, SL_OUT0, (size_t)S * nrays, &dxv); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_OUT1, (size_t)S * nrays, &dyv); if (rc) return rc;
        p.xv = dxv; p.yv = dyv; p.ld = nrays;
    }
    if (summ) { rc = dev_out<int32_t>(ctx, SL_OUT3, (size_t)nrays, &dst); if (rc) return rc; p.status = dst; }
    rc = launch_trace_modes<T, false>(ctx, p, blocks, hist, summ, flags); if (rc) return rc;
    if (hist) {
        HIP_TRY(hipMemcpy2DAsync(xv, ld * sizeof(T), dxv, nrays * sizeof(T), nrays * sizeof(T), S, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpy2DAsync(yv, ld * sizeof(T), dyv, nrays * sizeof(T), nrays * sizeof(T), S, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (summ) { rc = from_device<int32_t>(ctx, status, dst, (size_t)nrays); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ORT_OK;
}

// The launch number that tells this launch's look-back / ready words from older ones (30 bits).  When it wraps, the words are
// zero-filled again before the launch (the capacities the fills were done for are forgotten).
static unsigned next_ft_epoch(ort_ctx* ctx)
{
    ctx->ft_epoch = (ctx->ft_epoch + 1) & 0x3fffffffu;
    if (ctx->ft_epoch == 0) { ctx->ft_epoch = 1; ctx->ft_state_cap = 0; ctx->ft_ready_cap = 0; }
    return ctx->ft_epoch;
}

// The full_trace stages on a prepared TraceParams (recs / coefs / bundles / axes / grid shape set by the
// caller): tile buffers, trace with the FT epilogue, scan + scatter + finalize (or the statistics-only
// merge), results to the caller.  ex == NULL selects statistics only.
template <typename T>
int run_full_trace(ort_ctx* ctx, TraceParams<T>& p, int nb, T* ex, T* ey, T* rho, T* theta,
                   int64_t* count, double* rms, unsigned flags)
{
    const bool devp = flags & ORT_DEVICE_PTRS;
    const bool stats_only = !ex;
    const int64_t rpb = p.rpb, N = rpb * nb;
    const int64_t tiles = (int64_t)nb * p.tiles_per_bundle;
    if (tiles > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large");
#ifndef ORT_POLY_RPT1
#define ORT_POLY_RPT1 0
#endif
    const bool small_f64 = sizeof(T) == 8 && ((p.tiles_per_bundle <= kSmallTiles && tiles <= kSmallGridTiles && !(flags & ORT_NO_SMALL_PATH)) ||
                                              (ORT_POLY_RPT1 && p.arms >= ARMS_EVEN));
    int rc;
    // per-tile aggregates — or, on the statistics-only walk route, four partials per span (of one tile at the least)
    const size_t npart = (size_t)tiles * (stats_only ? kBlock / 64 : 1);
    rc = dev_out<int32_t>(ctx, SL_TCNT, npart, &p.tile_cnt); if (rc) return rc;
    rc = dev_out<double>(ctx, SL_TSX, npart, &p.tile_sx); if (rc) return rc;
    rc = dev_out<double>(ctx, SL_TSY, npart, &p.tile_sy); if (rc) return rc;
    rc = dev_out<double>(ctx, SL_TRM, npart, &p.tile_rmax); if (rc) return rc;
    int64_t* dcount = count; double* drms = rms;
    if (!devp) {
        rc = dev_out<int64_t>(ctx, SL_RES0, (size_t)nb, &dcount); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_RES1, (size_t)nb, &drms); if (rc) return rc;
    }
    if (stats_only) {
        // one pass: trace + stop filter + per-tile (n, mean, M2), merged per bundle — no ray-sized buffer at all
        rc = dev_out<double>(ctx, SL_TOFF, npart, &p.tile_m2x); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_TSQ, npart, &p.tile_m2y); if (rc) return rc;
        // ONE route (k_trace, FT_WALK): a workgroup walks spans of tiles of its bundle, four partials (one per wave) per span.
        // The span length is a function of the BUNDLE's shape alone — one tile for bundles of a few tiles (the reference's
        // own call), kWalkTiles beyond — so a bundle's statistics never depend on what else is in the launch; how many
        // consecutive spans a workgroup walks (walk_group) is a launch-shape choice that no result depends on: as many as
        // leave the launch ~8 workgroups per resident slot, down to ONE tile per workgroup for a call of a few tiles
        // (latency: 4 workgroups for the reference's own 2,048 rays)
        const int span = p.tiles_per_bundle <= kSmallTiles ? 1 : kWalkTiles;
        p.walk_spans = (p.tiles_per_bundle + span - 1) / span;
        const int parts = p.walk_spans * (kBlock / 64);             // (n, mean, M2) partials per bundle
        const int64_t spans = (int64_t)nb * p.walk_spans;
        p.walk_group = (int)std::min<int64_t>(p.walk_spans, std::max<int64_t>(1, spans / kWalkTargetGroups));
        const int groups = (p.walk_spans + p.walk_group - 1) / p.walk_group;
        rc = span == 1 ? launch_trace<T, true, false, false, FT_WALK1>(ctx, p, (int64_t)nb * groups, flags)
                       : launch_trace<T, true, false, false, FT_WALK>(ctx, p, (int64_t)nb * groups, flags);
        if (rc) return rc;
        if (parts <= 64 && !(flags & ORT_NO_SMALL_PATH))           // one wave per bundle: the same merges in the same order (bit-identical)
            hipLaunchKernelGGL(k_ft_stats_reduce_wave, dim3((unsigned)((nb + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, ctx->stream,
                               p.tile_cnt, p.tile_sx, p.tile_sy, p.tile_m2x, p.tile_m2y, p.tile_rmax, parts, nb, dcount, drms);
        else
            hipLaunchKernelGGL(k_ft_stats_reduce, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream,
                               p.tile_cnt, p.tile_sx, p.tile_sy, p.tile_m2x, p.tile_m2y, p.tile_rmax, parts, dcount, drms);
        HIP_TRY(hipGetLastError());
        if (!devp) {
            rc = from_device<int64_t>(ctx, count, dcount, (size_t)nb); if (rc) return rc;
            rc = from_device<double>(ctx, rms, drms, (size_t)nb); if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        return ORT_OK;
    }
    double* chunk_sq; FtBundleAgg* agg; int* ft_err = nullptr; bool fused_finish = false;
    rc = dev_out<double>(ctx, SL_TSQ, (size_t)tiles * (kBlock / 64), &chunk_sq); if (rc) return rc;   // (fused route: a partial per wave)
    rc = dev_out<FtBundleAgg>(ctx, SL_AGG, (size_t)nb, &agg); if (rc) return rc;
    T *dex = ex, *dey = ey, *drho = rho, *dth = theta;
    if (!devp) {
        rc = dev_out<T>(ctx, SL_OUT0, (size_t)2 * N, &dex); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_OUT1, (size_t)2 * N, &dey); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_OUT2, (size_t)2 * N, &drho); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_OUT4, (size_t)2 * N, &dth); if (rc) return rc;
    }
    if (!(flags & ORT_FT_LOOKBACK)) {
        // default route: tile-local compaction into a workspace | tile offsets + bundle aggregates | placement of the
        // survivors in both halves + squared deviations | sigma.  No workgroup waits for another.
        T *wex, *wey, *wr, *wth; int64_t* tile_off;
        const size_t nw = (size_t)tiles * kTile;                 // a slot of kTile entries per tile
        rc = dev_out<T>(ctx, SL_WEX, nw, &wex); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_WEY, nw, &wey); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_WR, nw, &wr); if (rc) return rc;
        rc = dev_out<T>(ctx, SL_WTH, nw, &wth); if (rc) return rc;
        rc = dev_out<int64_t>(ctx, SL_TOFF, (size_t)tiles, &tile_off); if (rc) return rc;
        p.out_ex = wex; p.out_ey = wey; p.out_r = wr; p.out_th = wth;
        const bool fused = (flags & ORT_FT_FUSED) && nb >= 2 && p.tiles_per_bundle > kSmallTiles && !small_f64;
        if (fused) {
            // the second pass inside the trace launch (k_trace, FT_FUSED): workgroup i traces tile i and places tile i - lag
            rc = dev_out<int>(ctx, SL_FTDONE, (size_t)nb + 4, &p.ft_done); if (rc) return rc;
            HIP_TRY(hipMemsetAsync(p.ft_done, 0, ((size_t)nb * sizeof(int) + 15) & ~(size_t)15, ctx->stream));   // arrivals are counted within the call
            p.ft_epoch = next_ft_epoch(ctx);
            rc = dev_out<unsigned>(ctx, SL_FTREADY, (size_t)nb, &p.ft_ready); if (rc) return rc;
            if (ctx->slot[SL_FTREADY].cap != ctx->ft_ready_cap) {
                HIP_TRY(hipMemsetAsync(p.ft_ready, 0, ctx->slot[SL_FTREADY].cap, ctx->stream));
                ctx->ft_ready_cap = ctx->slot[SL_FTREADY].cap;
            }
            rc = dev_out<int>(ctx, SL_FTERR, 1, &p.ft_err); if (rc) return rc;
            HIP_TRY(hipMemsetAsync(p.ft_err, 0, sizeof(int), ctx->stream));
            ft_err = p.ft_err;
            p.tile_off = tile_off; p.agg = agg; p.tile_sq = chunk_sq;
            p.fin_ex = dex; p.fin_ey = dey; p.fin_rho = drho; p.fin_th = dth;
            p.fuse_ntiles = (int)tiles;
            p.fuse_lag = p.tiles_per_bundle + kFuseMargin;
            p.fuse_scan_lag = kFuseScanLag;
            p.fuse_spin_cap = 1 << 22;
            if (ctx->test_fused_no_scan) {                           // testing aid: nobody runs the scans, the waits give up early
                p.fuse_scan_lag = 0x3fffffff; p.fuse_spin_cap = 1 << 10;
            }
            if (tiles + p.fuse_lag > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large");
            rc = launch_trace<T, true, false, false, FT_FUSED>(ctx, p, tiles + p.fuse_lag, flags); if (rc) return rc;
        } else
        if (small_f64) rc = launch_trace<T, true, false, false, FT_FULL, sizeof(T) == 8 ? 1 : kRPT>(ctx, p, tiles, flags);
        else rc = launch_trace<T, true, false, false, FT_FULL>(ctx, p, tiles, flags);
        if (rc) return rc;
        if (fused) {
            hipLaunchKernelGGL((k_ft_finalize<kBlock / 64>), dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream,
                               chunk_sq, p.tiles_per_bundle, agg, dcount, drms, (const int*)ft_err);
            HIP_TRY(hipGetLastError());
            fused_finish = true;
        } else
        if (p.tiles_per_bundle <= kSmallTiles && !(flags & ORT_NO_SMALL_PATH)) {
            // bundles of a few tiles (the reference's own call: 4): offsets, placement and sigma by one workgroup per bundle
            // in ONE launch, through the same bodies (k_ft_small_finish)
            hipLaunchKernelGGL((k_ft_small_finish<T>), dim3((unsigned)nb), dim3(kBlock * kFinishGroups), 0, ctx->stream,
                               wex, wey, wr, wth, rpb, p.tiles_per_bundle, p.tile_cnt, p.tile_sx, p.tile_sy, p.tile_rmax, tile_off, agg,
                               dex, dey, drho, dth, chunk_sq, dcount, drms);
            HIP_TRY(hipGetLastError());
            fused_finish = true;
        } else {
            hipLaunchKernelGGL(k_ft_scan, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream,
                               p.tile_cnt, p.tile_sx, p.tile_sy, p.tile_rmax, p.tiles_per_bundle, tile_off, agg);
            HIP_TRY(hipGetLastError());
            const int64_t pgroups = (int64_t)nb * ((p.tiles_per_bundle + kPlaceTiles - 1) / kPlaceTiles);   // kPlaceTiles tiles per workgroup
            hipLaunchKernelGGL((k_ft_place<T>), dim3((unsigned)pgroups), dim3(kBlock), 0, ctx->stream,
                               wex, wey, wr, wth, rpb, p.tiles_per_bundle, p.tile_cnt, tile_off, agg, dex, dey, drho, dth, chunk_sq);
            HIP_TRY(hipGetLastError());
        }
    } else {
        // ORT_FT_LOOKBACK: the trace kernel writes the first half itself.  Look-back state: one 8-byte word per tile +
        // the ticket counter; zero-filled when (re)allocated only — the epoch tells the words of this launch from
        // older ones, the base does the same for tickets
        p.ft_epoch = next_ft_epoch(ctx);
        rc = dev_out<unsigned long long>(ctx, SL_FTSTATE, (size_t)tiles, &p.ft_state); if (rc) return rc;
        if (ctx->slot[SL_FTSTATE].cap != ctx->ft_state_cap) {
            HIP_TRY(hipMemsetAsync(p.ft_state, 0, ctx->slot[SL_FTSTATE].cap, ctx->stream));
            ctx->ft_state_cap = ctx->slot[SL_FTSTATE].cap;
        }
        if (!ctx->slot[SL_FTTICKET].p) {
            rc = dev_out<unsigned long long>(ctx, SL_FTTICKET, 1, &p.ft_ticket); if (rc) return rc;
            HIP_TRY(hipMemsetAsync(p.ft_ticket, 0, sizeof(unsigned long long), ctx->stream));
            ctx->ft_ticket_base = 0;
        }
        p.ft_ticket = static_cast<unsigned long long*>(ctx->slot[SL_FTTICKET].p);
        rc = dev_out<int>(ctx, SL_FTERR, 1, &p.ft_err); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(p.ft_err, 0, sizeof(int), ctx->stream));
        ft_err = p.ft_err;
        p.ft_ticket_base = ctx->ft_ticket_base;
        p.out_ex = dex; p.out_ey = dey; p.out_r = drho; p.out_th = dth;
        rc = launch_trace<T, true, false, false, FT_LOOKBACK>(ctx, p, tiles, flags); if (rc) return rc;
        ctx->ft_ticket_base += (unsigned long long)tiles;        // only a launch that went out has taken its tickets
        hipLaunchKernelGGL(k_ft_scan, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream,
                           p.tile_cnt, p.tile_sx, p.tile_sy, p.tile_rmax, p.tiles_per_bundle, (int64_t*)nullptr, agg);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL((k_ft_mirror<T>), dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream,
                           rpb, p.tiles_per_bundle, agg, dex, dey, drho, dth, chunk_sq, (const int*)ft_err);
        HIP_TRY(hipGetLastError());
    }
    if (!fused_finish) {
        hipLaunchKernelGGL((k_ft_finalize<1>), dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream,
                           chunk_sq, p.tiles_per_bundle, agg, dcount, drms, (const int*)ft_err);
        HIP_TRY(hipGetLastError());
    }
    if (!devp) {
        int herr = 0;
        rc = from_device<int64_t>(ctx, count, dcount, (size_t)nb); if (rc) return rc;
        rc = from_device<double>(ctx, rms, drms, (size_t)nb); if (rc) return rc;
        if (ft_err) { rc = from_device<int>(ctx, &herr, ft_err, 1); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (herr)
            return fail(ORT_EHIP, "full_trace in-launch hand-off fault (%s): the survivors' offsets are not trustworthy, nothing was returned",
                        (herr & 2) ? "tile tickets outside the launch: the context's ticket base and the device counter disagree"
                                   : "a workgroup waited for its predecessors beyond the poll cap");
        for (int b = 0; b < nb; ++b) {
            const size_t off = (size_t)b * 2 * rpb, cnt = (size_t)count[b];
            if (!cnt) continue;
            rc = from_device<T>(ctx, ex + off, dex + off, cnt); if (rc) return rc;
            rc = from_device<T>(ctx, ey + off, dey + off, cnt); if (rc) return rc;
            rc = from_device<T>(ctx, rho + off, drho + off, cnt); if (rc) return rc;
            rc = from_device<T>(ctx, theta + off, dth + off, cnt); if (rc) return rc;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

template <typename T>
int full_trace_impl(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles,
                    const T* axes, int64_t axes_len, int ny, int nx, T* ex, T* ey, T* rho, T* theta,
                    int64_t* count, double* rms, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    rc = check_sys(ctx, sys); if (rc) return rc;
    if (nb <= 0 || !bundles || !axes || ny <= 0 || nx <= 0 || !count || !rms)
        return fail(ORT_EINVAL, "bad full_trace arguments");
    const bool stats_only = !ex && !ey && !rho && !theta;        // spot statistics without the error vectors
    if (!stats_only && (!ex || !ey || !rho || !theta)) return fail(ORT_EINVAL, "ex, ey, rho, theta: all or none");
    const int S = sys->rows - 1;
    for (int b = 0; b < nb; ++b)
        if (bundles[b].stop <= 0 || bundles[b].stop > S) return fail(ORT_EINVAL, "bundle %d: full_trace needs a stop index in 1..%d", b, S);
    const int64_t rpb = (int64_t)ny * nx;
    if (rpb > 0x7fffffffLL - kTile) return fail(ORT_EINVAL, "bundle of %lld rays is too large (max 2^31)", (long long)rpb);
    const bool devp = flags & ORT_DEVICE_PTRS;

    TraceParams<T> p;
    memset(&p, 0, sizeof p);
    p.recs = Sel<T>::rec(sys); p.polys = Sel<T>::poly(sys); p.arms = Sel<T>::arms(sys); p.S = S; p.apert2 = Sel<T>::ap2(sys);
    rc = upload_bundles<T>(ctx, sys, nb, bundles, ny, nx, axes_len, flags, &p.bundles); if (rc) return rc;
    p.ny = ny; p.nx = nx; p.rpb = rpb; p.tiles_per_bundle = (int)((rpb + kTile - 1) / kTile);
    p.raybasis = (flags & ORT_RAYBASIS) ? 1 : 0;
    if (devp) p.axes = axes;
    else { rc = to_device<T>(ctx, SL_AXES, axes, (size_t)axes_len, &p.axes); if (rc) return rc; }
    rc = make_rb_slopes<T>(ctx, p, nb); if (rc) return rc;
    return run_full_trace<T>(ctx, p, nb, ex, ey, rho, theta, count, rms, flags);
}

template <typename T>
int spot_batch_impl(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                    const double* K, const double* coef, int ncoef,
                    const double* a, const double* hprime, int nfields, const double* fields, int k_rays,
                    ort_first_order* fo_out, T* ex, T* ey, T* rho, T* theta, int64_t* count, double* rms, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (nsys <= 0 || rows < 2 || rows + 1 > ORT_MAX_ROWS || !R || !t || !n || !a || !hprime || nfields <= 0 || !fields ||
        k_rays < 2 || !count || !rms)
        return fail(ORT_EINVAL, "bad spot_batch arguments");
    if ((ex || ey || rho || theta) && (!ex || !ey || !rho || !theta)) return fail(ORT_EINVAL, "ex, ey, rho, theta: all or none");
    if (ncoef < 0 || ncoef > ORT_MAX_NCOEF || (ncoef > 0 && !coef)) return fail(ORT_EINVAL, "bad coefficient table (ncoef 0..%d)", ORT_MAX_NCOEF);
    if (!coef) ncoef = 0;
    const bool layout = K || ncoef > 0;                              // the input is a Layout{Aspheric}: Q16 dispatch
    const bool devp = flags & ORT_DEVICE_PTRS;
    if (!devp) for (int f = 0; f < nfields; ++f)
        if (!(std::fabs(fields[f]) <= 1.0)) return fail(ORT_EDOMAIN, "DomainError with %g: Domain: |H| <= 1.0", fields[f]);
    const int64_t na64 = (int64_t)nsys * nfields;
    if (na64 > 0x7fffffffLL / 4) return fail(ORT_EINVAL, "too many (system, field) pairs");
    const int na = (int)na64, k2 = k_rays / 2, S = rows;          // extended system: rows + 1 rows -> S = rows iterations
    const int64_t rpb = (int64_t)k_rays * k2;
    if (rpb > 0x7fffffffLL - kTile) return fail(ORT_EINVAL, "bundle of %lld rays is too large (max 2^31)", (long long)rpb);
    const size_t nr = (size_t)nsys * rows;
    if (!devp) for (int s = 0; s < nsys; ++s)
        if (t[(size_t)s * rows + rows - 1] != 0.0)
            return fail(ORT_EINVAL, "system %d: spot_batch expects a last thickness of 0 (image space)", s);
    const double *dR = R, *dt = t, *dn = n, *dK = K, *dcoef = coef, *da = a, *dh = hprime, *dfields = fields;
    const size_t n_a = (size_t)nsys * (rows - 1), n_c = (size_t)nsys * rows * (size_t)ncoef;
    const size_t in_cnt = 3 * nr + (K ? nr : 0) + n_c + n_a + (size_t)nsys + (size_t)nfields;   // doubles, packed: one H2D copy
    // One device block [inputs | flag | results]: the H2D copy brings the inputs and a zeroed convergence flag, ONE D2H copy
    // takes [flag | count | rms | first-order | error vectors (when asked for and small enough)] back through pinned memory
    const size_t cap = (size_t)2 * (size_t)rpb;
    const size_t vec_bytes = ex ? 4 * (size_t)na * cap * sizeof(T) : 0;
    const bool packed_vec = ex && !devp && vec_bytes <= kPackedVectorBytes;
    const size_t in_bytes = in_cnt * sizeof(double);
    const size_t o_flag = in_bytes, o_cnt = o_flag + 16, o_rms = o_cnt + (size_t)na * sizeof(int64_t),
                 o_fo = o_rms + (size_t)na * sizeof(double), o_vec = (o_fo + (size_t)nsys * sizeof(FirstOrderOut) + 255) & ~(size_t)255,
                 pack_bytes = o_vec + (packed_vec ? vec_bytes : 0);
    unsigned char* hpin = nullptr; unsigned char* dpack = nullptr;
    // A small call (the reference's own: one system, one field, 2,048 rays) does not copy at all: its kernels read the
    // inputs from, and write the results to, the pinned host block itself — two copy dispatches and their launch gaps
    // (~2 x 4 us of a ~40 us device timeline, more on the host side) for a few PCIe round trips inside the kernels
    const bool zero_copy = !devp && na <= kSmallPairs && !(flags & ORT_NO_SMALL_PATH) && pack_bytes <= kZeroCopyBytes;
    if (!devp) {
        rc = ctx->pinned(pack_bytes, &hpin); if (rc) return rc;      // no reuse inside the call, no mid-call sync
        if (zero_copy) { void* dp0 = nullptr; HIP_TRY(hipHostGetDevicePointer(&dp0, hpin, 0)); dpack = static_cast<unsigned char*>(dp0); }
        else { rc = dev_out<unsigned char>(ctx, SL_SB_PACK, pack_bytes, &dpack); if (rc) return rc; }
        double* hp = reinterpret_cast<double*>(hpin); double* dp = reinterpret_cast<double*>(dpack);
        size_t o = 0;
        auto put = [&](const double* src, size_t cnt, const double** dev) { memcpy(hp + o, src, cnt * sizeof(double)); *dev = dp + o; o += cnt; };
        put(R, nr, &dR); put(t, nr, &dt); put(n, nr, &dn);
        if (K) put(K, nr, &dK);
        if (ncoef > 0) put(coef, n_c, &dcoef);
        put(a, n_a, &da); put(hprime, (size_t)nsys, &dh); put(fields, (size_t)nfields, &dfields);
        memset(hpin + o_flag, 0, 16);                                // the flag travels with the inputs: no fill launch
        if (!zero_copy) HIP_TRY(hipMemcpyAsync(dpack, hpin, in_bytes + 16, hipMemcpyHostToDevice, ctx->stream));
    }
    FirstOrderOut* d_fo; SurfRec<T>* d_rec; DevBundle<T>* d_bd; T* d_axes; int* d_flag; T* d_poly = nullptr;
    if (dpack) d_fo = reinterpret_cast<FirstOrderOut*>(dpack + o_fo);
    else { rc = dev_out<FirstOrderOut>(ctx, SL_SB_FO, (size_t)nsys, &d_fo); if (rc) return rc; }
    rc = dev_out<SurfRec<T>>(ctx, SL_SB_REC, nr, &d_rec); if (rc) return rc;
    if (ncoef > 0) { rc = dev_out<T>(ctx, SL_SB_CEXT, nr * kPolyRec, &d_poly); if (rc) return rc; }
    rc = dev_out<DevBundle<T>>(ctx, SL_BUNDLES, (size_t)na, &d_bd); if (rc) return rc;
    ctx->bundle_cache.clear();                                  // SL_BUNDLES no longer mirrors a host array
    rc = dev_out<T>(ctx, SL_AXES, (size_t)na * (k_rays + k2), &d_axes); if (rc) return rc;
    hipStream_t st = ctx->stream;
    if (dpack) d_flag = reinterpret_cast<int*>(dpack + o_flag);
    else {
        rc = dev_out<int>(ctx, SL_SB_FLAG, 1, &d_flag); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), st));
    }
    TraceParams<T> p;
    memset(&p, 0, sizeof p);
    p.recs = d_rec; p.polys = d_poly; p.S = S; p.bundles = d_bd; p.axes = d_axes;
    // which kernel build (device-resident prescriptions cannot be inspected: coefficients -> every arm; conic constants, or
    // Float64, where a row may be too weak for the centre form -> general arms)
    if (!devp) {
        // host-visible prescriptions: classify every row exactly as the device-side table builder will (make_poly_rec /
        // make_rec, ort_device.hpp) and run the build its rows need — an even asphere gets the even-asphere build
        int needs = 0;
        T prec[kPolyRec];
        for (int s = 0; s < nsys; ++s)
            for (int i = 0; i + 1 < rows; ++i) {
                const size_t r0 = (size_t)s * rows;
                int nc = 0, pcls = 0;
                if (ncoef > 0) pcls = make_poly_rec<T>(prec, coef + (r0 + i + 1) * (size_t)ncoef, ncoef, &nc);
                SurfRec<T> rec;
                needs |= make_rec<T>(rec, (T)t[r0 + i], (T)R[r0 + i + 1], (T)n[r0 + i], (T)n[r0 + i + 1], K ? (T)K[r0 + i + 1] : T(0), nc, pcls);
            }
        p.arms = arms_of_needs(needs);                              // (the appended image row is a plane: basic arms)
    } else {
        p.arms = ncoef > 0 ? ARMS_POLY : ((K != nullptr || sizeof(T) == 8) ? ARMS_GENERAL : ARMS_BASIC);
    }
    p.ny = k_rays; p.nx = k2; p.rpb = rpb; p.tiles_per_bundle = (int)((rpb + kTile - 1) / kTile);
    auto nblk = [](int64_t n, int b) { return dim3((unsigned)((n + b - 1) / b)); };
    const double* coefp = ncoef > 0 ? dcoef : (const double*)nullptr;
    // solve (RayTracing.jl:302-323) -> tables -> aiming requests -> aiming (:223-296) -> bundles + axis end
    // points (PupilSampling.jl:94-122) -> axes -> trace + stop filter + tile moments -> per-bundle RMS
    if (na <= kSmallPairs && !(flags & ORT_NO_SMALL_PATH)) {
        // a few pairs: all of it in ONE launch, one wave per pair (k_small_prepare; the same device functions)
        hipLaunchKernelGGL((k_small_prepare<T>), dim3((unsigned)na), dim3(64), 0, st, nsys, nfields, rows, dR, dt, dn, dK, coefp, ncoef,
                           da, dh, dfields, k_rays, k2, layout ? 1 : 0, 587.5618e-6, d_fo, d_rec, d_poly, d_bd, d_axes, d_flag,
                           (flags & ORT_FAST_MATH) ? 1 : 0);
    } else {
        MerSurf *d_mf, *d_mr; double *d_tlf, *d_tlr, *d_ends, *d_crev = nullptr; AimIn* d_ain; AimOut* d_aout;
        if (ncoef > 0) { rc = dev_out<double>(ctx, SL_SB_CREV, n_c, &d_crev); if (rc) return rc; }
        rc = dev_out<MerSurf>(ctx, SL_SB_MF, (size_t)nsys * (rows - 1), &d_mf); if (rc) return rc;
        rc = dev_out<MerSurf>(ctx, SL_SB_MR, (size_t)nsys * (rows - 1), &d_mr); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_SB_TLF, (size_t)nsys, &d_tlf); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_SB_TLR, (size_t)nsys, &d_tlr); if (rc) return rc;
        rc = dev_out<AimIn>(ctx, SL_SB_AIN, (size_t)na, &d_ain); if (rc) return rc;
        rc = dev_out<AimOut>(ctx, SL_SB_AOUT, (size_t)na, &d_aout); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_SB_ENDS, (size_t)na * 4, &d_ends); if (rc) return rc;
        hipLaunchKernelGGL(k_first_order, nblk(nsys, 64), dim3(64), 0, st, nsys, rows, dR, dt, dn, da, (const double*)nullptr, dh,
                           587.5618e-6, d_fo, (double*)nullptr, (double*)nullptr);
        hipLaunchKernelGGL((k_build_tables<T>), nblk((int64_t)nr, kBlock), dim3(kBlock), 0, st, nsys, rows, dR, dt, dn, dK,
                           coefp, ncoef, d_fo, d_rec, d_poly, d_mf, d_mr, d_crev, d_tlf, d_tlr);
        hipLaunchKernelGGL(k_build_aim, nblk(na, kBlock), dim3(kBlock), 0, st, nsys, nfields, rows, d_fo, da, dfields, layout ? 1 : 0, d_ain);
        hipLaunchKernelGGL(k_aim, nblk((int64_t)na * 4, 64), dim3(64), 0, st, na, d_ain, d_mf, coefp, d_tlf,
                           d_mr, (const double*)d_crev, d_tlr, rows - 1, ncoef, d_aout, 1, (flags & ORT_FAST_MATH) ? 1 : 0);
        hipLaunchKernelGGL((k_build_bundles<T>), nblk(na, kBlock), dim3(kBlock), 0, st, na, k_rays, k2, d_ain, d_aout, d_bd, d_ends, d_flag);
        hipLaunchKernelGGL((k_make_axes<T>), nblk((int64_t)na * (k_rays + k2), kBlock), dim3(kBlock), 0, st, na, k_rays, k2, d_ends, d_axes);
    }
    HIP_TRY(hipGetLastError());
    if (devp) {
        rc = run_full_trace<T>(ctx, p, na, ex, ey, rho, theta, count, rms, flags); if (rc) return rc;
        if (fo_out) HIP_TRY(hipMemcpyAsync(fo_out, d_fo, (size_t)nsys * sizeof(FirstOrderOut), hipMemcpyDeviceToDevice, st));
        return ORT_OK;       // asynchronous; a failed aiming shows as NaN RMS of that bundle
    }
    int64_t* d_cnt = reinterpret_cast<int64_t*>(dpack + o_cnt); double* d_rms = reinterpret_cast<double*>(dpack + o_rms);
    if (!ex || packed_vec) {
        // (count, rms) — and the error vectors, when they fit the packed block — land behind the inputs on the device; ONE
        // copy brings them, the first-order structs and the convergence flag back through the pinned buffer
        T* dv = reinterpret_cast<T*>(dpack + o_vec);
        const size_t slab = (size_t)na * cap;
        rc = run_full_trace<T>(ctx, p, na, ex ? dv : (T*)nullptr, ex ? dv + slab : (T*)nullptr, ex ? dv + 2 * slab : (T*)nullptr,
                               ex ? dv + 3 * slab : (T*)nullptr, d_cnt, d_rms, flags | ORT_DEVICE_PTRS);
        if (rc) return rc;
        if (!zero_copy) HIP_TRY(hipMemcpyAsync(hpin + o_flag, dpack + o_flag, pack_bytes - o_flag, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        memcpy(count, hpin + o_cnt, (size_t)na * sizeof(int64_t));
        memcpy(rms, hpin + o_rms, (size_t)na * sizeof(double));
        // the stages ran with device pointers, where a faulted look-back shows as count = -1 (k_ft_finalize): a host caller
        // is promised an error code instead (ort.h)
        if (ex && (flags & ORT_FT_LOOKBACK))
            for (int b = 0; b < na; ++b)
                if (count[b] < 0)
                    return fail(ORT_EHIP, "full_trace in-launch hand-off fault: the survivors' offsets are not trustworthy, nothing was returned");
        if (ex) {
            const T* hv = reinterpret_cast<const T*>(hpin + o_vec);
            for (int b = 0; b < na; ++b) {                           // the valid entries of every slab
                const size_t off = (size_t)b * cap, cnt_b = (size_t)std::max<int64_t>(0, count[b]);
                memcpy(ex + off, hv + off, cnt_b * sizeof(T));            memcpy(ey + off, hv + slab + off, cnt_b * sizeof(T));
                memcpy(rho + off, hv + 2 * slab + off, cnt_b * sizeof(T)); memcpy(theta + off, hv + 3 * slab + off, cnt_b * sizeof(T));
            }
        }
    } else {
        rc = run_full_trace<T>(ctx, p, na, ex, ey, rho, theta, count, rms, flags); if (rc) return rc;
        if (!zero_copy) HIP_TRY(hipMemcpyAsync(hpin + o_flag, dpack + o_flag, o_vec - o_flag, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (fo_out) memcpy(fo_out, hpin + o_fo, (size_t)nsys * sizeof(FirstOrderOut));
    int flag = 0;
    memcpy(&flag, hpin + o_flag, sizeof(int));
    if (flag) return fail(ORT_EHIP, "ray aiming did not converge for at least one (system, field) pair");
    return ORT_OK;
}

}  // namespace

// ======================================================================================
extern "C" {

int ort_version(void) { return ORT_VERSION; }
const char* ort_last_error(void) { return g_err.c_str(); }

int ort_ctx_create(int device, void* stream, ort_ctx** out)
{
    if (!out) return fail(ORT_EINVAL, "null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(ORT_EHIP, "no HIP device available (%s); this engine has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(ORT_EINVAL, "device %d out of range (0..%d)", device, count - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ORT_EHIP, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    ort_ctx* c = new (std::nothrow) ort_ctx();
    if (!c) return fail(ORT_ENOMEM, "out of host memory");
    c->device = device;
    if (stream) { c->stream = static_cast<hipStream_t>(stream); c->own_stream = false; }
    else {
        hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (es != hipSuccess) { delete c; return fail(ORT_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(es)); }
        c->own_stream = true;
    }
    hipError_t e0 = hipEventCreate(&c->ev0), e1 = hipEventCreate(&c->ev1);
    if (e0 != hipSuccess || e1 != hipSuccess) { delete c; return fail(ORT_EHIP, "hipEventCreate failed"); }
    *out = c;
    return ORT_OK;
}

int ort_ctx_destroy(ort_ctx* ctx)
{
    if (!ctx) return ORT_OK;
    hipError_t e = hipSetDevice(ctx->device); (void)e;
    e = hipStreamSynchronize(ctx->stream); (void)e;
    // communicators that outlive their context (a garbage collector's order): their streams may still hold this context's
    // scratch in flight — drain them — and they must not reach back into freed memory: detach them (their entry points then
    // fail with "null context"; ort_comm_destroy still releases them)
    (void)sync_comm_streams(ctx);
    for (ort_comm* c : ctx->comms) detach_comm(c);
    for (auto& s : ctx->slot) s.release();
    if (ctx->pin) { e = hipHostFree(ctx->pin); (void)e; }
    if (ctx->ev0) { e = hipEventDestroy(ctx->ev0); (void)e; }
    if (ctx->ev1) { e = hipEventDestroy(ctx->ev1); (void)e; }
    if (ctx->own_stream && ctx->stream) { e = hipStreamDestroy(ctx->stream); (void)e; }
    delete ctx;
    return ORT_OK;
}

int ort_ctx_set_stream(ort_ctx* ctx, void* stream)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!stream) return fail(ORT_EINVAL, "null stream");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { HIP_TRY(hipStreamDestroy(ctx->stream)); ctx->own_stream = false; }
    ctx->stream = static_cast<hipStream_t>(stream);
    return ORT_OK;
}

int ort_ctx_synchronize(ort_ctx* ctx)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ORT_OK;
}

int ort_ctx_timer_start(ort_ctx* ctx)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return ORT_OK;
}

int ort_ctx_timer_stop(ort_ctx* ctx, float* ms)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!ms) return fail(ORT_EINVAL, "null ms");
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return ORT_OK;
}

int ort_ctx_device_info(ort_ctx* ctx, char* name, int name_len, int* cus, int* clock_mhz, int64_t* mem_bytes)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) { snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName); }
    if (cus) *cus = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (mem_bytes) *mem_bytes = (int64_t)prop.totalGlobalMem;
    return ORT_OK;
}

// --------------------------------------------------------------------------------------
// Device buffers for hosts without a GPU array library of their own (the Julia shim): with these and
// ORT_DEVICE_PTRS a caller keeps ray-sized results on the device and downloads only what it reads.
int ort_device_malloc(ort_ctx* ctx, size_t bytes, void** out)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!out) return fail(ORT_EINVAL, "null out");
    *out = nullptr;
    if (bytes == 0) return ORT_OK;
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) { *out = nullptr; return fail(ORT_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return ORT_OK;
}

int ort_device_free(ort_ctx* ctx, void* p)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!p) return ORT_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));                 // no launch of this context may still use it,
    rc = sync_comm_streams(ctx); if (rc) return rc;             // nor a collective still in flight on a communicator's own stream
    HIP_TRY(hipFree(p));
    return ORT_OK;
}

#ifdef ORT_PHASE_CLOCKS
// measurement build only: the stamps of the last launches' block 0 (scripts/phase_clocks.py)
extern "C" int ort_debug_phase_clocks(ort_ctx* ctx, unsigned long long* out32)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(ort::g_phase), 32 * sizeof(unsigned long long)));
    return ORT_OK;
}
#endif

#ifdef ORT_COUNT_RETRACE
// measurement build only: {tile-waves traced by MATH_FAST kernels, tile-waves that traced again with the reference sequence} since
// the last reset (scripts/run_workload.py --retrace)
extern "C" int ort_debug_retrace_counts(ort_ctx* ctx, unsigned long long* out2, int reset)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpyFromSymbol(out2, HIP_SYMBOL(ort::g_retrace), 2 * sizeof(unsigned long long)));
    if (reset) { const unsigned long long z[2] = {0, 0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(ort::g_retrace), z, sizeof z)); }
    return ORT_OK;
}
#endif

// Testing aid (tests/test_gpu_parity.py): shift the context's look-back ticket base against the device counter — the
// bookkeeping fault the look-back route must REPORT (ORT_EHIP) rather than misplace survivors.
int ort_ctx_test_skew_tickets(ort_ctx* ctx, int64_t delta)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    ctx->ft_ticket_base += (unsigned long long)delta;
    return ORT_OK;
}

int ort_ctx_test_fused_no_scan(ort_ctx* ctx, int on)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    ctx->test_fused_no_scan = on != 0;
    return ORT_OK;
}

// wavegrad(eps, lambda) (src/PupilSampling.jl:165-167) of full_trace results, device or host buffers.
int ort_wavegrad_f64(ort_ctx* ctx, int nb, int64_t cap, const int64_t* count, const double* nu, double lambda,
                     const double* ex, const double* ey, double* gx, double* gy, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (nb <= 0 || cap <= 0 || !count || !nu || !ex || !ey || !gx || !gy || !(lambda > 0.0)) return fail(ORT_EINVAL, "bad wavegrad arguments");
    const bool devp = flags & ORT_DEVICE_PTRS;
    const int64_t chunks = (cap + kTile - 1) / kTile;
    if (chunks * nb > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large");
    const int64_t* dcount = count; const double *dnu = nu, *dex = ex, *dey = ey; double *dgx = gx, *dgy = gy;
    const size_t n = (size_t)nb * (size_t)cap;
    if (!devp) {
        rc = to_device<int64_t>(ctx, SL_IN0, count, (size_t)nb, &dcount); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN1, nu, (size_t)nb, &dnu); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN2, ex, n, &dex); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN3, ey, n, &dey); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT0, n, &dgx); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT1, n, &dgy); if (rc) return rc;
    }
    hipLaunchKernelGGL((k_wavegrad<double>), dim3((unsigned)(chunks * nb)), dim3(kBlock), 0, ctx->stream, cap, (int)chunks, dcount, dnu,
                       lambda, dex, dey, dgx, dgy);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<double>(ctx, gx, dgx, n); if (rc) return rc;
        rc = from_device<double>(ctx, gy, dgy, n); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

int ort_device_upload(ort_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (bytes == 0) return ORT_OK;
    if (!dst || !src) return fail(ORT_EINVAL, "null pointer");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ORT_OK;
}

int ort_device_download(ort_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (bytes == 0) return ORT_OK;
    if (!dst || !src) return fail(ORT_EINVAL, "null pointer");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ORT_OK;
}

// --------------------------------------------------------------------------------------
int ort_system_create(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                      const double* K, const double* coef, int ncoef, ort_system** out)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!out) return fail(ORT_EINVAL, "null out");
    *out = nullptr;
    if (nsys <= 0 || !R || !t || !n) return fail(ORT_EINVAL, "bad system arguments");
    if (rows < 2 || rows > ORT_MAX_ROWS) return fail(ORT_EINVAL, "rows %d outside 2..%d", rows, ORT_MAX_ROWS);
    if (ncoef < 0 || ncoef > ORT_MAX_NCOEF) return fail(ORT_EINVAL, "ncoef %d outside 0..%d", ncoef, ORT_MAX_NCOEF);
    if (!coef) ncoef = 0;
    if (ncoef == 0) coef = nullptr;
    const int S = rows - 1;
    std::vector<SurfRec<double>> r64; std::vector<double> p64;
    std::vector<SurfRec<float>> r32; std::vector<float> p32;
    const int arms64 = build_records<double>(nsys, rows, ncoef, R, t, n, K, coef, r64, p64);
    const int arms32 = build_records<float>(nsys, rows, ncoef, R, t, n, K, coef, r32, p32);
    const size_t n_c64 = ncoef > 0 ? (size_t)nsys * rows * ncoef : 0;             // raw rows for the meridional kernels
    std::vector<MerSurf> mer((size_t)nsys * S);
    ort_system* sys = new (std::nothrow) ort_system();
    if (!sys) return fail(ORT_ENOMEM, "out of host memory");
    sys->t_last.resize((size_t)nsys);
    for (int s = 0; s < nsys; ++s) {
        for (int i = 0; i < S; ++i) {
            const SurfRec<double>& r = r64[(size_t)s * S + i];
            MerSurf m;
            memset(&m, 0, sizeof m);
            m.t = r.t; m.R = r.R; m.sgn = r.sgn; m.K = K ? K[(size_t)s * rows + i + 1] : 0.0;
            m.n1 = n[(size_t)s * rows + i]; m.n2 = n[(size_t)s * rows + i + 1];
            m.eta = m.n1 / m.n2; m.invR = r.finite ? 1.0 / m.R : 0.0;
            m.finite = r.finite; m.ncoef = r.ncoef;
            mer[(size_t)s * S + i] = m;
        }
        sys->t_last[(size_t)s] = t[(size_t)s * rows + rows - 1];
    }
    sys->ctx = ctx; sys->nsys = nsys; sys->rows = rows; sys->ncoef = ncoef; sys->arms64 = arms64; sys->arms32 = arms32;
    // one device slab for every table of the batch (one hipMalloc, one staged copy): allocation
    // calls dominate the upload of 10^4-instance batches otherwise
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_r64 = pad(r64.size() * sizeof(SurfRec<double>)), b_r32 = pad(r32.size() * sizeof(SurfRec<float>));
    const size_t b_mer = pad(mer.size() * sizeof(MerSurf)), b_tl = pad(sys->t_last.size() * sizeof(double));
    const size_t b_c64 = pad(n_c64 * sizeof(double)), b_p64 = pad(p64.size() * sizeof(double)), b_p32 = pad(p32.size() * sizeof(float));
    const size_t total = b_r64 + b_r32 + b_mer + b_tl + b_c64 + b_p64 + b_p32;
    std::vector<unsigned char> stage(total, 0);
    size_t o = 0;
    const size_t o_r64 = o; memcpy(stage.data() + o, r64.data(), r64.size() * sizeof(SurfRec<double>)); o += b_r64;
    const size_t o_r32 = o; memcpy(stage.data() + o, r32.data(), r32.size() * sizeof(SurfRec<float>)); o += b_r32;
    const size_t o_mer = o; memcpy(stage.data() + o, mer.data(), mer.size() * sizeof(MerSurf)); o += b_mer;
    const size_t o_tl = o; memcpy(stage.data() + o, sys->t_last.data(), sys->t_last.size() * sizeof(double)); o += b_tl;
    const size_t o_c64 = o; if (ncoef > 0) memcpy(stage.data() + o, coef, n_c64 * sizeof(double)); o += b_c64;
    const size_t o_p64 = o; if (ncoef > 0) memcpy(stage.data() + o, p64.data(), p64.size() * sizeof(double)); o += b_p64;
    const size_t o_p32 = o; if (ncoef > 0) memcpy(stage.data() + o, p32.data(), p32.size() * sizeof(float)); o += b_p32;
    {
        hipError_t e = hipMalloc(&sys->slab, total);
        if (e == hipSuccess) e = hipMemcpy(sys->slab, stage.data(), total, hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = fail(ORT_EHIP, "system upload failed: %s", hipGetErrorString(e)); }
    }
    if (!rc) {
        unsigned char* base = static_cast<unsigned char*>(sys->slab);
        sys->rec64 = reinterpret_cast<SurfRec<double>*>(base + o_r64);
        sys->rec32 = reinterpret_cast<SurfRec<float>*>(base + o_r32);
        sys->mer = reinterpret_cast<MerSurf*>(base + o_mer);
        sys->d_tlast = reinterpret_cast<double*>(base + o_tl);
        if (ncoef > 0) {
            sys->coef64 = reinterpret_cast<double*>(base + o_c64);
            sys->poly64 = reinterpret_cast<double*>(base + o_p64); sys->poly32 = reinterpret_cast<float*>(base + o_p32);
        }
    }
    if (rc) { ort_system_destroy(sys); return rc; }
    *out = sys;
    return ORT_OK;
}

int ort_system_destroy(ort_system* sys)
{
    if (!sys) return ORT_OK;
    hipError_t e;
    if (sys->ctx) { e = hipSetDevice(sys->ctx->device); (void)e; e = hipStreamSynchronize(sys->ctx->stream); (void)e; }
    if (sys->slab) { e = hipFree(sys->slab); (void)e; }
    if (sys->ap64) { e = hipFree(sys->ap64); (void)e; }
    delete sys;
    return ORT_OK;
}

int ort_system_set_apertures(ort_system* sys, const double* a)
{
    if (!sys || !sys->ctx) return fail(ORT_EINVAL, "null system");
    int rc = check_ctx(sys->ctx); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(sys->ctx->stream));            // no launch may still be reading the old table
    if (sys->ap64) { HIP_TRY(hipFree(sys->ap64)); sys->ap64 = nullptr; sys->ap32 = nullptr; }
    if (!a) return ORT_OK;
    const int S = sys->rows - 1;
    const size_t cnt = (size_t)sys->nsys * S;
    std::vector<double> h64(cnt); std::vector<float> h32(cnt);
    for (size_t i = 0; i < cnt; ++i) {
        if (!(a[i] >= 0.0)) return fail(ORT_EINVAL, "aperture %zu is negative or NaN", i);
        h64[i] = a[i] * a[i];
        const float af = (float)a[i];
        h32[i] = af * af;
    }
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, cnt * (sizeof(double) + sizeof(float))));
    sys->ap64 = static_cast<double*>(d);
    sys->ap32 = reinterpret_cast<float*>(sys->ap64 + cnt);
    HIP_TRY(hipMemcpy(sys->ap64, h64.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sys->ap32, h32.data(), cnt * sizeof(float), hipMemcpyHostToDevice));
    return ORT_OK;
}

int ort_system_rows(const ort_system* sys) { return sys ? sys->rows : fail(ORT_EINVAL, "null system"); }
int ort_system_count(const ort_system* sys) { return sys ? sys->nsys : fail(ORT_EINVAL, "null system"); }

// --------------------------------------------------------------------------------------
int ort_trace_skew_f64(ort_ctx* ctx, const ort_system* sys, int isys, int64_t nrays,
                       const double* y, const double* x, const double* U, const double* V,
                       double* xv, double* yv, int64_t ld, int32_t* status, unsigned flags)
{
    return trace_list_impl<double>(ctx, sys, isys, nrays, y, x, U, V, xv, yv, ld, status, flags);
}

int ort_trace_skew_f32(ort_ctx* ctx, const ort_system* sys, int isys, int64_t nrays,
                       const float* y, const float* x, const float* U, const float* V,
                       float* xv, float* yv, int64_t ld, int32_t* status, unsigned flags)
{
    return trace_list_impl<float>(ctx, sys, isys, nrays, y, x, U, V, xv, yv, ld, status, flags);
}

int ort_trace_grid_f64(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles,
                       const double* axes, int64_t axes_len, int ny, int nx,
                       const ort_grid_out_f64* out, unsigned flags)
{
    return trace_grid_impl<double, ort_grid_out_f64>(ctx, sys, nb, bundles, axes, axes_len, ny, nx, out, flags);
}

int ort_trace_grid_f32(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles,
                       const float* axes, int64_t axes_len, int ny, int nx,
                       const ort_grid_out_f32* out, unsigned flags)
{
    return trace_grid_impl<float, ort_grid_out_f32>(ctx, sys, nb, bundles, axes, axes_len, ny, nx, out, flags);
}

// --------------------------------------------------------------------------------------
int ort_make_axes_f64(ort_ctx* ctx, int nb, int ny, int nx, const double* ends, double* axes, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (nb <= 0 || ny <= 0 || nx <= 0 || !ends || !axes) return fail(ORT_EINVAL, "bad make_axes arguments");
    const bool devp = flags & ORT_DEVICE_PTRS;
    const int64_t total = (int64_t)nb * (ny + nx);
    const double* dends = ends; double* daxes = axes;
    if (!devp) {
        rc = to_device<double>(ctx, SL_IN0, ends, (size_t)nb * 4, &dends); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT0, (size_t)total, &daxes); if (rc) return rc;
    }
    const int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large");
    hipLaunchKernelGGL((k_make_axes<double>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, nb, ny, nx, dends, daxes);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<double>(ctx, axes, daxes, (size_t)total); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

// --------------------------------------------------------------------------------------
int ort_full_trace_f64(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles,
                       const double* axes, int64_t axes_len, int ny, int nx,
                       double* ex, double* ey, double* rho, double* theta,
                       int64_t* count, double* rms, unsigned flags)
{
    return full_trace_impl<double>(ctx, sys, nb, bundles, axes, axes_len, ny, nx, ex, ey, rho, theta, count, rms, flags);
}

int ort_full_trace_f32(ort_ctx* ctx, const ort_system* sys, int nb, const ort_bundle* bundles,
                       const float* axes, int64_t axes_len, int ny, int nx,
                       float* ex, float* ey, float* rho, float* theta,
                       int64_t* count, double* rms, unsigned flags)
{
    return full_trace_impl<float>(ctx, sys, nb, bundles, axes, axes_len, ny, nx, ex, ey, rho, theta, count, rms, flags);
}

// --------------------------------------------------------------------------------------
int ort_spot_batch_f64(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                       const double* a, const double* hprime, int nfields, const double* fields, int k_rays,
                       ort_first_order* fo_out, int64_t* count, double* rms, unsigned flags)
{
    return spot_batch_impl<double>(ctx, nsys, rows, R, t, n, nullptr, nullptr, 0, a, hprime, nfields, fields, k_rays, fo_out,
                                   nullptr, nullptr, nullptr, nullptr, count, rms, flags);
}

int ort_full_trace_batch_f64(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                             const double* a, const double* hprime, int nfields, const double* fields, int k_rays,
                             ort_first_order* fo_out, double* ex, double* ey, double* rho, double* theta,
                             int64_t* count, double* rms, unsigned flags)
{
    if (!ex || !ey || !rho || !theta) return fail(ORT_EINVAL, "full_trace_batch needs ex, ey, rho, theta (ort_spot_batch_f64 is the statistics-only call)");
    return spot_batch_impl<double>(ctx, nsys, rows, R, t, n, nullptr, nullptr, 0, a, hprime, nfields, fields, k_rays, fo_out,
                                   ex, ey, rho, theta, count, rms, flags);
}

int ort_full_trace_layout_batch_f64(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                                    const double* K, const double* coef, int ncoef,
                                    const double* a, const double* hprime, int nfields, const double* fields, int k_rays,
                                    ort_first_order* fo_out, double* ex, double* ey, double* rho, double* theta,
                                    int64_t* count, double* rms, unsigned flags)
{
    return spot_batch_impl<double>(ctx, nsys, rows, R, t, n, K, coef, ncoef, a, hprime, nfields, fields, k_rays, fo_out,
                                   ex, ey, rho, theta, count, rms, flags);
}

int ort_spot_batch_f32(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                       const double* a, const double* hprime, int nfields, const double* fields, int k_rays,
                       ort_first_order* fo_out, int64_t* count, double* rms, unsigned flags)
{
    return spot_batch_impl<float>(ctx, nsys, rows, R, t, n, nullptr, nullptr, 0, a, hprime, nfields, fields, k_rays, fo_out,
                                  nullptr, nullptr, nullptr, nullptr, count, rms, flags);
}

// --------------------------------------------------------------------------------------
int ort_trace_meridional_f64(ort_ctx* ctx, const ort_system* sys, int isys, int64_t nrays,
                             const double* y, const double* U,
                             double* y_out, double* U_out, double* ts_out, int64_t ld, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    rc = check_sys(ctx, sys); if (rc) return rc;
    if (isys < 0 || isys >= sys->nsys) return fail(ORT_EINVAL, "system index %d out of range", isys);
    if (nrays < 0 || !y || !U || !y_out || !U_out) return fail(ORT_EINVAL, "bad meridional arguments");
    if (nrays == 0) return ORT_OK;
    if (ld < nrays) return fail(ORT_EINVAL, "ld %lld < rays %lld", (long long)ld, (long long)nrays);
    const int rows = sys->rows, S = rows - 1;
    const MerSurf* surf = sys->mer + (size_t)isys * S;
    const double* coefs = sys->coef64 ? sys->coef64 + (size_t)isys * rows * sys->ncoef : nullptr;
    const int layout = (flags & ORT_LAYOUT_INPUT) ? 1 : 0;
    const int64_t blocks = (nrays + kBlock - 1) / kBlock;
    const double t_last = sys->t_last[(size_t)isys];
    // DomainError record of this launch: [0] = min over (ray << 8 | surface), [1] = count
    unsigned long long* ddom;
    rc = dev_out<unsigned long long>(ctx, SL_DOMAIN, 2, &ddom); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ddom, 0xff, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipMemsetAsync(ddom + 1, 0, sizeof(unsigned long long), ctx->stream));
    if (flags & ORT_DEVICE_PTRS) {
        hipLaunchKernelGGL(k_trace_meridional, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                           surf, coefs, S, sys->ncoef, layout, t_last, nrays, y, U, y_out, U_out, ts_out, ld, ddom);
        HIP_TRY(hipGetLastError());
        return ORT_OK;
    }
    const double *dy, *dU; double *oy, *oU, *ots = nullptr;
    rc = to_device<double>(ctx, SL_IN0, y, (size_t)nrays, &dy); if (rc) return rc;
    rc = to_device<double>(ctx, SL_IN1, U, (size_t)nrays, &dU); if (rc) return rc;
    rc = dev_out<double>(ctx, SL_OUT0, (size_t)rows * nrays, &oy); if (rc) return rc;
    rc = dev_out<double>(ctx, SL_OUT1, (size_t)rows * nrays, &oU); if (rc) return rc;
    if (ts_out) { rc = dev_out<double>(ctx, SL_OUT2, (size_t)rows * nrays, &ots); if (rc) return rc; }
    hipLaunchKernelGGL(k_trace_meridional, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                       surf, coefs, S, sys->ncoef, layout, t_last, nrays, dy, dU, oy, oU, ots, nrays, ddom);
    HIP_TRY(hipGetLastError());
    const size_t w = (size_t)nrays * sizeof(double);
    HIP_TRY(hipMemcpy2DAsync(y_out, ld * sizeof(double), oy, w, w, rows, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpy2DAsync(U_out, ld * sizeof(double), oU, w, w, rows, hipMemcpyDeviceToHost, ctx->stream));
    if (ts_out) HIP_TRY(hipMemcpy2DAsync(ts_out, ld * sizeof(double), ots, w, w, rows, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long hdom[2] = {~0ull, 0ull};
    HIP_TRY(hipMemcpyAsync(hdom, ddom, sizeof hdom, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hdom[0] != ~0ull)
        return fail(ORT_EDOMAIN, "DomainError: asin(x) is not defined for |x| > 1 (y / R at a spherical surface, src/RayTracing.jl:162): "
                                 "ray %llu at surface row %u, %llu ray-surface(s) in all; those rays are NaN from there on, every other output is valid",
                    hdom[0] >> 8, (unsigned)(hdom[0] & 0xff), hdom[1]);
    return ORT_OK;
}

int ort_ctx_domain_error(ort_ctx* ctx, int64_t* ray, int* surface, int64_t* count)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (ray) *ray = -1;
    if (surface) *surface = 0;
    if (count) *count = 0;
    if (!ctx->slot[SL_DOMAIN].p) return ORT_OK;
    unsigned long long hdom[2];
    HIP_TRY(hipMemcpyAsync(hdom, ctx->slot[SL_DOMAIN].p, sizeof hdom, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hdom[0] == ~0ull) return ORT_OK;
    if (ray) *ray = (int64_t)(hdom[0] >> 8);
    if (surface) *surface = (int)(hdom[0] & 0xff);
    if (count) *count = (int64_t)hdom[1];
    return ORT_OK;
}

// --------------------------------------------------------------------------------------
int ort_aim_f64(ort_ctx* ctx, const ort_system* fwd, const ort_system* rev, int n,
                const ort_aim_in* in, ort_aim_out* out, unsigned flags)
{
    static_assert(sizeof(ort_aim_in) == sizeof(AimIn) && sizeof(ort_aim_out) == sizeof(AimOut), "ABI struct mismatch");
    int rc = check_ctx(ctx); if (rc) return rc;
    rc = check_sys(ctx, fwd); if (rc) return rc;
    rc = check_sys(ctx, rev); if (rc) return rc;
    if (n < 0 || !in || !out) return fail(ORT_EINVAL, "bad aim arguments");
    if (n == 0) return ORT_OK;
    if (n > (1 << 29)) return fail(ORT_EINVAL, "too many aiming requests (4 lanes each)");
    if (fwd->rows != rev->rows || fwd->nsys != rev->nsys || fwd->ncoef != rev->ncoef)
        return fail(ORT_EINVAL, "forward and reversed system batches differ in shape");
    const bool devp = flags & ORT_DEVICE_PTRS;
    const int S = fwd->rows - 1;
    if (!devp) {
        for (int i = 0; i < n; ++i) {
            if (in[i].system < 0 || in[i].system >= fwd->nsys) return fail(ORT_EINVAL, "aim %d: system %d out of range", i, in[i].system);
            if (in[i].stop < 1 || in[i].stop > S) return fail(ORT_EINVAL, "aim %d: stop %d out of range 1..%d", i, in[i].stop, S);
            if (!(std::fabs(in[i].H) <= 1.0)) return fail(ORT_EDOMAIN, "DomainError with %g: Domain: |H| <= 1.0", in[i].H);
        }
    }
    const AimIn* din = reinterpret_cast<const AimIn*>(in);
    AimOut* dout = reinterpret_cast<AimOut*>(out);
    if (!devp) {
        rc = to_device<AimIn>(ctx, SL_IN0, reinterpret_cast<const AimIn*>(in), (size_t)n, &din); if (rc) return rc;
        rc = dev_out<AimOut>(ctx, SL_OUT0, (size_t)n, &dout); if (rc) return rc;
    }
    hipLaunchKernelGGL(k_aim, dim3((unsigned)(((int64_t)n * 4 + 63) / 64)), dim3(64), 0, ctx->stream, n, din,
                       fwd->mer, fwd->coef64, fwd->d_tlast, rev->mer, rev->coef64, rev->d_tlast, S, fwd->ncoef, dout,
                       (flags & ORT_AIM_EDGE_AS_FOUND) ? 0 : 1, (flags & ORT_FAST_MATH) ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<AimOut>(ctx, reinterpret_cast<AimOut*>(out), dout, (size_t)n); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

int ort_fan_f64(ort_ctx* ctx, const ort_system* sys, int n, const ort_fan_in* in, int k_rays, int descending,
                double* y_XP, double* eps, unsigned flags)
{
    static_assert(sizeof(ort_fan_in) == sizeof(FanIn), "ABI struct mismatch");
    int rc = check_ctx(ctx); if (rc) return rc;
    rc = check_sys(ctx, sys); if (rc) return rc;
    if (n < 0 || !in || k_rays < 1 || !y_XP || !eps) return fail(ORT_EINVAL, "bad fan arguments");
    if (n == 0) return ORT_OK;
    const bool devp = flags & ORT_DEVICE_PTRS;
    const int64_t total = (int64_t)n * k_rays;
    if ((total + kBlock - 1) / kBlock > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large");
    if (!devp)
        for (int i = 0; i < n; ++i)
            if (in[i].system < 0 || in[i].system >= sys->nsys) return fail(ORT_EINVAL, "fan %d: system %d out of range", i, in[i].system);
    const FanIn* din = reinterpret_cast<const FanIn*>(in);
    double *dy = y_XP, *de = eps;
    if (!devp) {
        rc = to_device<FanIn>(ctx, SL_IN0, reinterpret_cast<const FanIn*>(in), (size_t)n, &din); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT0, (size_t)total, &dy); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT1, (size_t)total, &de); if (rc) return rc;
    }
    hipLaunchKernelGGL(k_fan, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, n, k_rays, descending ? 1 : 0,
                       din, sys->mer, sys->coef64, sys->d_tlast, sys->rows - 1, sys->ncoef, dy, de);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<double>(ctx, y_XP, dy, (size_t)total); if (rc) return rc;
        rc = from_device<double>(ctx, eps, de, (size_t)total); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

static int first_order_impl(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                            const double* a, const double* dn, const double* hprime, double lambda,
                            ort_first_order* out, double* surf, double* inc, unsigned flags)
{
    static_assert(sizeof(ort_first_order) == sizeof(FirstOrderOut), "ABI struct mismatch");
    static_assert((int)ORT_SURF_COUNT == (int)SURF_COUNT && (int)ORT_SURF_TANGENTIAL == (int)SURF_TANGENTIAL, "ABI enum mismatch");
    int rc = check_ctx(ctx); if (rc) return rc;
    if (nsys <= 0 || rows < 2 || rows > ORT_MAX_ROWS || !R || !t || !n || !a || !hprime || !out || !(lambda > 0.0))
        return fail(ORT_EINVAL, "bad first_order arguments");
    const bool devp = flags & ORT_DEVICE_PTRS;
    const double *dR = R, *dt = t, *dnn = n, *da = a, *ddn = dn, *dh = hprime;
    FirstOrderOut* dout = reinterpret_cast<FirstOrderOut*>(out);
    double *dsurf = surf, *dinc = inc;
    const size_t nr = (size_t)nsys * rows, ns = (size_t)nsys * (rows - 1);
    if (!devp) {
        // Lens() keeps the last row (k == rows, src/RayTracing.jl:47-50) when the last thickness is finite and
        // non-zero; the reference then needs `rows` semi-diameters (a ./ y[2:end], :215) and throws a
        // DimensionMismatch for the rows-1 this ABI carries
        for (int s = 0; s < nsys; ++s) {
            const double tl = t[(size_t)s * rows + rows - 1];
            if (tl != 0.0 && std::isfinite(tl))
                return fail(ORT_EINVAL, "system %d: last thickness %g is finite and non-zero: the reference keeps the last "
                                        "lens row and needs %d semi-diameters (DimensionMismatch); end the prescription in image space (t = 0)",
                            s, tl, rows);
        }
        rc = to_device<double>(ctx, SL_IN0, R, nr, &dR); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN1, t, nr, &dt); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN2, n, nr, &dnn); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN3, a, ns, &da); if (rc) return rc;
        if (dn) { rc = to_device<double>(ctx, SL_TAB0, dn, nr, &ddn); if (rc) return rc; }
        rc = to_device<double>(ctx, SL_TAB1, hprime, (size_t)nsys, &dh); if (rc) return rc;
        rc = dev_out<FirstOrderOut>(ctx, SL_OUT0, (size_t)nsys, &dout); if (rc) return rc;
        if (surf) { rc = dev_out<double>(ctx, SL_OUT1, (size_t)SURF_COUNT * ns, &dsurf); if (rc) return rc; }
        if (inc) { rc = dev_out<double>(ctx, SL_OUT2, (size_t)4 * ns, &dinc); if (rc) return rc; }
    }
    hipLaunchKernelGGL(k_first_order, dim3((unsigned)((nsys + 63) / 64)), dim3(64), 0, ctx->stream,
                       nsys, rows, dR, dt, dnn, da, ddn, dh, lambda, dout, dsurf, dinc);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<FirstOrderOut>(ctx, reinterpret_cast<FirstOrderOut*>(out), dout, (size_t)nsys); if (rc) return rc;
        if (surf) { rc = from_device<double>(ctx, surf, dsurf, (size_t)SURF_COUNT * ns); if (rc) return rc; }
        if (inc) { rc = from_device<double>(ctx, inc, dinc, (size_t)4 * ns); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

int ort_first_order_f64(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                        const double* a, const double* dn, const double* hprime, double lambda,
                        ort_first_order* out, unsigned flags)
{
    return first_order_impl(ctx, nsys, rows, R, t, n, a, dn, hprime, lambda, out, nullptr, nullptr, flags);
}

int ort_aberrations_f64(ort_ctx* ctx, int nsys, int rows, const double* R, const double* t, const double* n,
                        const double* a, const double* dn, const double* hprime, double lambda,
                        ort_first_order* out, double* surf, double* inc, unsigned flags)
{
    return first_order_impl(ctx, nsys, rows, R, t, n, a, dn, hprime, lambda, out, surf, inc, flags);
}

// --------------------------------------------------------------------------------------
int ort_trace_paraxial_f64(ort_ctx* ctx, int nlens, int k, const double* tau, const double* phi, const double* a,
                           int64_t rays_per_lens, const double* y, const double* w,
                           double* rt_y, double* rt_w, int64_t ld, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (nlens <= 0 || k <= 0 || k > ORT_MAX_ROWS || !tau || !phi || rays_per_lens < 0 || !y || !w || !rt_y || !rt_w)
        return fail(ORT_EINVAL, "bad paraxial arguments");
    const int64_t N = (int64_t)nlens * rays_per_lens;
    if (N == 0) return ORT_OK;
    if (ld < N) return fail(ORT_EINVAL, "ld %lld < rays %lld", (long long)ld, (long long)N);
    const int clip = (flags & ORT_CLIP) ? 1 : 0;
    const int bpl = (int)((rays_per_lens + kBlock - 1) / kBlock);
    const int64_t blocks = (int64_t)nlens * bpl;
    if (blocks > 0x7fffffffLL) return fail(ORT_EINVAL, "launch too large");
    const bool devp = flags & ORT_DEVICE_PTRS;
    // the lens table is always small host data unless ORT_DEVICE_PTRS says otherwise
    const double *dtau = tau, *dphi = phi, *da = a, *dy = y, *dw = w;
    double *oy = rt_y, *ow = rt_w; int64_t old = ld;
    if (!devp) {
        rc = to_device<double>(ctx, SL_TAB0, tau, (size_t)nlens * k, &dtau); if (rc) return rc;
        rc = to_device<double>(ctx, SL_TAB1, phi, (size_t)nlens * k, &dphi); if (rc) return rc;
        if (a) { rc = to_device<double>(ctx, SL_TAB2, a, (size_t)nlens * k, &da); if (rc) return rc; }
        rc = to_device<double>(ctx, SL_IN0, y, (size_t)N, &dy); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN1, w, (size_t)N, &dw); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT0, (size_t)(k + 1) * N, &oy); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT1, (size_t)(k + 1) * N, &ow); if (rc) return rc;
        old = N;
    }
    hipLaunchKernelGGL(k_trace_paraxial, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                       k, dtau, dphi, da, clip, rays_per_lens, bpl, dy, dw, oy, ow, old);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        const size_t wd = (size_t)N * sizeof(double);
        HIP_TRY(hipMemcpy2DAsync(rt_y, ld * sizeof(double), oy, wd, wd, k + 1, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpy2DAsync(rt_w, ld * sizeof(double), ow, wd, wd, k + 1, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

// --------------------------------------------------------------------------------------
int ort_abcd_f64(ort_ctx* ctx, int nlens, int k, const double* tau, const double* phi, double* M, unsigned flags)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (nlens <= 0 || k <= 0 || !tau || !phi || !M) return fail(ORT_EINVAL, "bad abcd arguments");
    const bool devp = flags & ORT_DEVICE_PTRS;
    const double *dtau = tau, *dphi = phi; double* dM = M;
    if (!devp) {
        rc = to_device<double>(ctx, SL_TAB0, tau, (size_t)nlens * k, &dtau); if (rc) return rc;
        rc = to_device<double>(ctx, SL_TAB1, phi, (size_t)nlens * k, &dphi); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT0, (size_t)nlens * 4, &dM); if (rc) return rc;
    }
    const int blocks = (nlens + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_abcd, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, nlens, k, dtau, dphi, dM);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<double>(ctx, M, dM, (size_t)nlens * 4); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

static int abcd_apply(ort_ctx* ctx, const double* M, int64_t nv, const double* v, const double* tau,
                      const double* tau_p, double* out, unsigned flags, int reverse)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!M || nv < 0 || !v || !tau || !tau_p || !out) return fail(ORT_EINVAL, "bad abcd transfer arguments");
    if (nv == 0) return ORT_OK;
    const bool devp = flags & ORT_DEVICE_PTRS;
    const double *dM = M, *dv = v, *dt = tau, *dtp = tau_p; double* dout = out;
    if (!devp) {
        rc = to_device<double>(ctx, SL_TAB0, M, 4, &dM); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN0, v, (size_t)nv * 2, &dv); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN1, tau, (size_t)nv, &dt); if (rc) return rc;
        rc = to_device<double>(ctx, SL_IN2, tau_p, (size_t)nv, &dtp); if (rc) return rc;
        rc = dev_out<double>(ctx, SL_OUT0, (size_t)nv * 2, &dout); if (rc) return rc;
    }
    const int64_t blocks = (nv + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_abcd_transfer, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, dM, nv, dv, dt, dtp, dout, reverse);
    HIP_TRY(hipGetLastError());
    if (!devp) {
        rc = from_device<double>(ctx, out, dout, (size_t)nv * 2); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return ORT_OK;
}

int ort_abcd_transfer_f64(ort_ctx* ctx, const double* M, int64_t nv, const double* v, const double* tau,
                          const double* tau_p, double* out, unsigned flags)
{
    return abcd_apply(ctx, M, nv, v, tau, tau_p, out, flags, 0);
}

int ort_abcd_reverse_transfer_f64(ort_ctx* ctx, const double* M, int64_t nv, const double* v, const double* tau_p,
                                  const double* tau, double* out, unsigned flags)
{
    return abcd_apply(ctx, M, nv, v, tau, tau_p, out, flags, 1);
}

}  // extern "C"

// --------------------------------------------------------------------------------------
// RCCL, loaded lazily.  Only the entry points the reassembly needs.
struct Id128 { char b[ORT_UNIQUE_ID_BYTES]; };   // ncclUniqueId, passed by value
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
Rccl g_rccl;
constexpr int kNcclInt64 = 4, kNcclFloat32 = 7, kNcclFloat64 = 8;

int rccl_load()
{
    if (g_rccl.h) return ORT_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(ORT_EHIP, "cannot load librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.Broadcast || !g_rccl.CommDestroy ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return fail(ORT_EHIP, "librccl.so lacks a required symbol");
    g_rccl.h = h;
    return ORT_OK;
}
#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        int r_ = (expr);                                                                           \
        if (r_ != 0) return fail(ORT_EHIP, "%s failed: %s", #expr,                                 \
                                 g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
    } while (0)
}  // namespace

// The collectives run on the communicator's OWN stream, ordered after the work already queued on the context's
// stream (an event), so that the reassembly of one shard overlaps the trace of the next; ort_comm_wait orders the
// context's stream after them again, ort_comm_synchronize blocks the host.
struct ort_comm {
    ort_ctx* ctx = nullptr;          // null once the context has been destroyed ahead of the communicator (ort_ctx_destroy)
    int device = 0;
    void* comm = nullptr;
    int nranks = 0, rank = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr;
    static constexpr int kRing = 8;
    hipEvent_t ev_ring[kRing] = {};  // completion of the last kRing collectives, oldest overwritten
    unsigned long long issued = 0;   // collectives issued so far
    int64_t* d_counts = nullptr;     // [nranks + 1] device scratch of the ragged gather (own count at [nranks])
    int64_t* h_counts = nullptr;     // page-locked mirror
};

namespace {
int sync_comm_streams(ort_ctx* ctx)
{
    for (ort_comm* c : ctx->comms)
        if (c && c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    return ORT_OK;
}
void detach_comm(ort_comm* c) { if (c) c->ctx = nullptr; }
int comm_begin(ort_comm* c)          // comm stream waits for everything queued on the context's stream so far
{
    HIP_TRY(hipEventRecord(c->ev_in, c->ctx->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    return ORT_OK;
}
int comm_end(ort_comm* c)
{
    HIP_TRY(hipEventRecord(c->ev_ring[c->issued % ort_comm::kRing], c->stream));
    ++c->issued;
    return ORT_OK;
}
}  // namespace

extern "C" {

int ort_comm_unique_id(void* id128)
{
    if (!id128) return fail(ORT_EINVAL, "null id");
    int rc = rccl_load(); if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id128));
    return ORT_OK;
}

int ort_comm_create(ort_ctx* ctx, int nranks, int rank, const void* id128, ort_comm** out)
{
    int rc = check_ctx(ctx); if (rc) return rc;
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ORT_EINVAL, "bad communicator arguments");
    *out = nullptr;
    rc = rccl_load(); if (rc) return rc;
    Id128 id;
    memcpy(id.b, id128, ORT_UNIQUE_ID_BYTES);
    ort_comm* c = new (std::nothrow) ort_comm();
    if (!c) return fail(ORT_ENOMEM, "out of host memory");
    c->ctx = ctx; c->device = ctx->device; c->nranks = nranks; c->rank = rank;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    for (int i = 0; i < ort_comm::kRing && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_ring[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_counts, (size_t)(nranks + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_counts, (size_t)(nranks + 1) * sizeof(int64_t), hipHostMallocDefault);
    if (e != hipSuccess) { ort_comm_destroy(c); return fail(ORT_EHIP, "communicator resources: %s", hipGetErrorString(e)); }
    int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) { c->comm = nullptr; ort_comm_destroy(c); return fail(ORT_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); }
    ctx->comms.push_back(c);
    *out = c;
    return ORT_OK;
}

int ort_comm_destroy(ort_comm* comm)
{
    if (!comm) return ORT_OK;
    hipError_t e = hipSetDevice(comm->device); (void)e;
    if (comm->stream) { e = hipStreamSynchronize(comm->stream); (void)e; }
    if (comm->ctx) {                                             // (a context destroyed first has detached itself, ort_ctx_destroy)
        for (auto& q : comm->ctx->comms) if (q == comm) q = nullptr;
        e = hipStreamSynchronize(comm->ctx->stream); (void)e;
    }
    if (comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm->comm);
    if (comm->d_counts) { e = hipFree(comm->d_counts); (void)e; }
    if (comm->h_counts) { e = hipHostFree(comm->h_counts); (void)e; }
    if (comm->ev_in) { e = hipEventDestroy(comm->ev_in); (void)e; }
    for (hipEvent_t ev : comm->ev_ring) if (ev) { e = hipEventDestroy(ev); (void)e; }
    if (comm->stream) { e = hipStreamDestroy(comm->stream); (void)e; }
    delete comm;
    return ORT_OK;
}

int ort_comm_size(const ort_comm* comm) { return comm ? comm->nranks : fail(ORT_EINVAL, "null communicator"); }
int ort_comm_rank(const ort_comm* comm) { return comm ? comm->rank : fail(ORT_EINVAL, "null communicator"); }

int ort_comm_wait_lag(ort_comm* comm, int lag)
{
    if (!comm) return fail(ORT_EINVAL, "null communicator");
    if (lag < 0 || lag >= ort_comm::kRing) return fail(ORT_EINVAL, "lag %d outside 0..%d", lag, ort_comm::kRing - 1);
    int rc = check_ctx(comm->ctx); if (rc) return rc;
    if (comm->issued < (unsigned long long)lag + 1) return ORT_OK;          // nothing that old was issued
    HIP_TRY(hipStreamWaitEvent(comm->ctx->stream, comm->ev_ring[(comm->issued - 1 - lag) % ort_comm::kRing], 0));
    return ORT_OK;
}

int ort_comm_wait(ort_comm* comm) { return ort_comm_wait_lag(comm, 0); }

int ort_comm_synchronize(ort_comm* comm)
{
    if (!comm) return fail(ORT_EINVAL, "null communicator");
    int rc = check_ctx(comm->ctx); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(comm->stream));
    return ORT_OK;
}

// ONE ncclAllGather of the packed slab [2][count] (x at +0, y at +count) -> [nranks][2][count].
int ort_allgather_hits_packed_f64(ort_comm* comm, const double* hits, int64_t count, double* gathered)
{
    if (!comm || !hits || !gathered || count < 0) return fail(ORT_EINVAL, "bad all-gather arguments");
    int rc = check_ctx(comm->ctx); if (rc) return rc;
    if (count == 0) return ORT_OK;
    rc = comm_begin(comm); if (rc) return rc;
    RCCL_TRY(g_rccl.AllGather(hits, gathered, (size_t)(2 * count), kNcclFloat64, comm->comm, comm->stream));
    return comm_end(comm);
}

int ort_allgather_hits_packed_f32(ort_comm* comm, const float* hits, int64_t count, float* gathered)
{
    if (!comm || !hits || !gathered || count < 0) return fail(ORT_EINVAL, "bad all-gather arguments");
    int rc = check_ctx(comm->ctx); if (rc) return rc;
    if (count == 0) return ORT_OK;
    rc = comm_begin(comm); if (rc) return rc;
    RCCL_TRY(g_rccl.AllGather(hits, gathered, (size_t)(2 * count), kNcclFloat32, comm->comm, comm->stream));
    return comm_end(comm);
}

// Separate x / y arrays.  When they are the two halves of one packed slab (yf == xf + count, gy == gx + nranks*count
// is NOT that layout) the call is the packed one; otherwise the two slabs go out as ONE fused RCCL launch.
int ort_allgather_hits_f64(ort_comm* comm, const double* xf, const double* yf, int64_t count, double* gx, double* gy)
{
    if (!comm || !xf || !yf || !gx || !gy || count < 0) return fail(ORT_EINVAL, "bad all-gather arguments");
    int rc = check_ctx(comm->ctx); if (rc) return rc;
    if (count == 0) return ORT_OK;
    rc = comm_begin(comm); if (rc) return rc;
    RCCL_TRY(g_rccl.GroupStart());
    int r1 = g_rccl.AllGather(xf, gx, (size_t)count, kNcclFloat64, comm->comm, comm->stream);
    int r2 = r1 ? 0 : g_rccl.AllGather(yf, gy, (size_t)count, kNcclFloat64, comm->comm, comm->stream);
    int r3 = g_rccl.GroupEnd();              // always closed: an open group would swallow every later collective
    RCCL_TRY(r1); RCCL_TRY(r2); RCCL_TRY(r3);
    return comm_end(comm);
}

// Ragged reassembly (compacted survivors): counts first (one 8-byte all-gather, read by the host), then every
// rank's slab lands at its exclusive offset — a group of ncclBroadcast, one per rank, fused into one launch.
int ort_allgather_ragged_f64(ort_comm* comm, const double* values, int64_t count, double* gathered, int64_t capacity,
                             int64_t* counts)
{
    if (!comm || count < 0 || (count > 0 && !values) || !gathered || capacity < 0) return fail(ORT_EINVAL, "bad ragged all-gather arguments");
    int rc = check_ctx(comm->ctx); if (rc) return rc;
    const int nr = comm->nranks;
    rc = comm_begin(comm); if (rc) return rc;
    comm->h_counts[nr] = count;
    HIP_TRY(hipMemcpyAsync(comm->d_counts + nr, comm->h_counts + nr, sizeof(int64_t), hipMemcpyHostToDevice, comm->stream));
    RCCL_TRY(g_rccl.AllGather(comm->d_counts + nr, comm->d_counts, 1, kNcclInt64, comm->comm, comm->stream));
    HIP_TRY(hipMemcpyAsync(comm->h_counts, comm->d_counts, (size_t)nr * sizeof(int64_t), hipMemcpyDeviceToHost, comm->stream));
    HIP_TRY(hipStreamSynchronize(comm->stream));
    int64_t total = 0;
    for (int r = 0; r < nr; ++r) { if (counts) counts[r] = comm->h_counts[r]; total += comm->h_counts[r]; }
    if (total > capacity) return fail(ORT_EINVAL, "ragged all-gather: %lld entries exceed the capacity %lld", (long long)total, (long long)capacity);
    RCCL_TRY(g_rccl.GroupStart());
    int bad = 0; int64_t off = 0;
    for (int r = 0; r < nr && !bad; ++r) {
        const int64_t c = comm->h_counts[r];
        if (c > 0) bad = g_rccl.Broadcast(values, gathered + off, (size_t)c, kNcclFloat64, r, comm->comm, comm->stream);
        off += c;
    }
    int r3 = g_rccl.GroupEnd();
    RCCL_TRY(bad); RCCL_TRY(r3);
    return comm_end(comm);
}

}  // extern "C"

// ort_kernels.hpp — gfx950 kernels of the batched ray-trace engine.
//
// Layout in HBM
//   surface table   SurfRec<T>[nsys][S]  (+ polynomial records T[nsys][S][kPolyRec]); a workgroup copies
//                   the S records of ITS system into LDS once (<= 8 KiB + 12 KiB of coefficients).
//   rays            never stored: a bundle's rays are generated from two 1-D axes (L2
//                   resident) — or read once, coalesced, from SoA lists.
//   history         T[S][ld] surface-major: lane l of a wave writes 16 B (RPT = 2 adjacent
//                   rays) at consecutive addresses -> 1 KiB per store instruction.
//   summary         SoA T[N] x {xf, yf, xs, ys} + int32 status[N].
//   full_trace      workspace T[tiles][512] x {ex, ey, r, theta}: a tile fills only its compacted survivors of its own
//                   slot; outputs [nb][2*rpb] (first half = survivors in ray order, second half = mirror).
// One thread owns RPT = 2 adjacent rays: 16-byte stores and two independent FP64 div/sqrt
// dependency chains per lane.  Workgroup = 256 threads = 512 rays; a launch is
// nb * ceil(ny*nx/512) workgroups (>> 256 CUs at every BASELINE config but #1).  Output is
// streamed once and never re-read, so no XCD-aware remap is needed: the only shared data is
// the <= 5 KiB table, resident in every XCD's L2.
#pragma once

#include "ort_device.hpp"

#include <type_traits>

namespace ort {

#ifdef ORT_PHASE_CLOCKS              // measurement build only (scripts/phase_clocks.py): shader-clock stamps of block 0
__device__ unsigned long long g_phase[32];
#define ORT_PHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_phase[i] = __builtin_readcyclecounter(); g_phase[16 + (i)] = wall_clock64(); } } while (0)
#else
#define ORT_PHASE(i) do { } while (0)
#endif
#ifdef ORT_COUNT_RETRACE             // measurement build only (scripts/run_workload.py --retrace): [0] = tile-waves traced by a
__device__ unsigned long long g_retrace[2];   // MATH_FAST kernel, [1] = those that traced again with the reference sequence
#endif


// Tunables (compile-time; the defaults are the measured best).
#ifndef ORT_POLY_WAVES
#define ORT_POLY_WAVES (ORT_MIN_WAVES - 1)   // waves per SIMD the polynomial-arm builds are compiled for
#endif
#ifndef ORT_RPT
#define ORT_RPT 2            // rays per lane
#endif
#ifndef ORT_NT_STORES
#define ORT_NT_STORES 1      // non-temporal history stores
#endif
#ifndef ORT_APERTURES
#define ORT_APERTURES 1   // 0 compiles the clear-aperture extension out (A/B builds)
#endif
#ifndef ORT_FT_DEBUG
#define ORT_FT_DEBUG 0
#endif
#ifndef ORT_MIN_WAVES
#define ORT_MIN_WAVES 5      // __launch_bounds__ 2nd argument (waves per SIMD): 96 VGPRs, no spill in the fast history kernel
#endif
#ifndef ORT_WAVES_NOHIST
#define ORT_WAVES_NOHIST ORT_MIN_WAVES   // the same for the kernels without history stores (A/B: 6 = 80 VGPRs spills)
#endif
#ifndef ORT_WAVES_F32
#define ORT_WAVES_F32 ORT_WAVES_NOHIST   // the same for the Float32 kernels without history stores
#endif
#ifndef ORT_BLOCK
#define ORT_BLOCK 256          // threads per workgroup (A/B builds: 128 halves the tile and the end-of-launch tail)
#endif
constexpr int kBlock = ORT_BLOCK;
constexpr int kRPT = ORT_RPT;
constexpr int kTile = kBlock * kRPT;
constexpr int kMaxRows = 64;
constexpr int kMaxCoef = 12;

template <typename T>
struct DevBundle {
    int32_t system;
    int32_t stop;       // 0-based loop index captured as the stop hit, -1 = none
    T u, v;             // slopes tan U, tan V (host libm)
    T k0, k1, k2;       // normalize([v, u, 1]) — the same for every ray of the bundle (:40-41), done once on the host
    T a_stop, hprime;
    T ybar, z0;
    int64_t yoff, xoff;
};

typedef double dvec2_t __attribute__((ext_vector_type(2)));
typedef float fvec2_t __attribute__((ext_vector_type(2)));
template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = dvec2_t; };
template <> struct Vec2<float> { using type = fvec2_t; };

template <typename T>
struct TraceParams {
    const SurfRec<T>* recs;     // [nsys][S]
    const T* polys;             // [nsys][S][kPolyRec] polynomial records (ort_device.hpp) or null
    int arms;                   // highest ARMS level among the batch's rows (host: which kernel build to launch)
    int S;                      // loop iterations = rows - 1
    // grid source
    const DevBundle<T>* bundles;
    const T* axes;
    int ny, nx;
    int64_t rpb;                // rays per bundle = ny*nx
    int tiles_per_bundle;
    // list source
    const T* ly; const T* lx; const T* lU; const T* lV;
    int64_t nrays; int isys; int slopes_given;
    int raybasis;
    const T* rb_slopes;         // raybasis: [nb][ny + nx] launch slopes per pupil row | column (k_make_slope_axes), never null then
    const T* apert2;            // [nsys][S] squared clear semi-diameters (extension, off when null)
    // outputs
    T* xv; T* yv; int64_t ld;
    T* xf; T* yf; T* xs; T* ys;
    int32_t* status;
    // full_trace: where the trace kernel puts a tile's compacted survivors (ex, ey, UN-normalised stop radius, theta).
    // FT_FULL: out_* = workspace [tiles][kTile], a slot per tile.  FT_LOOKBACK: out_* = the caller's
    // ex / ey / rho / theta [nb][2*rpb], at the offset found by a decoupled look-back over the bundle's tiles
    T* out_ex; T* out_ey; T* out_r; T* out_th;
    unsigned long long* ft_state;       // [tiles] look-back words: state (2 bits) | epoch (30 bits) | prefix (32 bits)
    unsigned long long* ft_ticket;      // tiles are taken in ticket order: every predecessor of a tile has started
    unsigned long long ft_ticket_base;  // tickets handed out by earlier launches
    unsigned ft_epoch;                  // launch number: words of earlier launches read as "not yet written"
    int* ft_err;                        // look-back fault word: bit 0 = a wait hit its poll cap, bit 1 = a ticket outside the grid
    int32_t* tile_cnt; double* tile_sx; double* tile_sy; double* tile_rmax;
    double* tile_m2x; double* tile_m2y;     // FT_WALK: a partial's sums of squared deviations about its means
    int walk_spans;                         // FT_WALK / FT_WALK1: spans (of kWalkTiles tiles / of one tile) per bundle
    int walk_group;                         //          consecutive spans of one bundle a workgroup walks (launch shape only: no result depends on it)
    // FT_FUSED: the second pass inside the trace launch (workgroup i traces tile i and places tile i - fuse_lag)
    int64_t* tile_off; struct FtBundleAgg* agg; double* tile_sq;
    T* fin_ex; T* fin_ey; T* fin_rho; T* fin_th;       // the caller's ex / ey / rho / theta [nb][2*rpb]
    int* ft_done;                           // [nb] tiles of the bundle traced so far (back to 0 once its scan has run)
    unsigned* ft_ready;                     // [nb] = ft_epoch once the bundle's offsets and aggregates are published
    int fuse_ntiles, fuse_lag;              // nb * tiles_per_bundle; tiles_per_bundle + margin
    int fuse_scan_lag;                      // the workgroup fuse_scan_lag - 1 past a bundle's last tile runs that bundle's scan (< margin)
    int fuse_spin_cap;                      // polls before a waiting workgroup gives up and raises the fault word
};

// Two adjacent rays of one lane: one 2*sizeof(T) streaming store when the address allows.
template <typename T>
__device__ __forceinline__ void store_pair(T* base, int64_t g, bool two, T a, T b)
{
    const bool aligned = (reinterpret_cast<uintptr_t>(base + g) & (2 * sizeof(T) - 1)) == 0;
    if (two && aligned) {
        typename Vec2<T>::type v2; v2.x = a; v2.y = b;
#if ORT_NT_STORES
        __builtin_nontemporal_store(v2, reinterpret_cast<typename Vec2<T>::type*>(base + g));
#else
        *reinterpret_cast<typename Vec2<T>::type*>(base + g) = v2;
#endif
    } else {
#if ORT_NT_STORES
        __builtin_nontemporal_store(a, base + g);
        if (two) __builtin_nontemporal_store(b, base + g + 1);
#else
        base[g] = a;
        if (two) base[g + 1] = b;
#endif
    }
}

// Unconditional 2-ray vector store (the caller has established alignment and liveness).
template <typename T>
__device__ __forceinline__ void store_vec2(T* p, T a, T b)
{
    typename Vec2<T>::type v2; v2.x = a; v2.y = b;
#if ORT_NT_STORES
    __builtin_nontemporal_store(v2, reinterpret_cast<typename Vec2<T>::type*>(p));
#else
    *reinterpret_cast<typename Vec2<T>::type*>(p) = v2;
#endif
}

// ------------------------------------------------------------------------------------
// 16-byte streaming of a compacted tile held in LDS to an arbitrarily aligned place of an output array: `head`
// scalar elements up to the first 128-BYTE LINE boundary of the destination, then one aligned vector per thread and trip
// (2 doubles / 4 floats, non-temporal: every wave-wide store covers 8 whole lines), then a scalar tail.  Cutting at
// 16-byte boundaries only left a partial line at each end of every wave store, to be completed by its neighbour's
// partial line: ~35 % slower per byte (k_ft_place, profiles/r03_place_ab.log).  F maps the staged value to the stored one.
template <typename T, typename F>
__device__ __forceinline__ void stream_out(T* __restrict__ dst, const T* __restrict__ lds, int c, int tid, F f)
{
    constexpr int V = 16 / (int)sizeof(T), L = 128 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int mis = (int)((reinterpret_cast<uintptr_t>(dst) & 127) / sizeof(T));    // elements past a 128-byte line
    const int head = min(c, mis ? L - mis : 0);
    const int nvec = (c - head) / V;
    if (tid < head) __builtin_nontemporal_store(f(lds[tid]), dst + tid);
    for (int g = tid; g < nvec; g += kBlock) {
        const int j = head + g * V;
        vec_t v;
#pragma unroll
        for (int q = 0; q < V; ++q) v[q] = f(lds[j + q]);
        __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(dst + j));
    }
    const int j = head + nvec * V + tid;                                            // < V - 1 elements left
    if (j < c) __builtin_nontemporal_store(f(lds[j]), dst + j);
}

// lane `lane` (wave-uniform) of a double, as a SCALAR value
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------------------------
// Data handed from one workgroup to another INSIDE a launch (the fused full_trace route, FT_FUSED).  The XCDs' L2s are not
// coherent with each other and a CU's vector L1 is never refreshed by another CU's stores: every such byte is stored
// write-through (`sc1`) and drained by its wave (`s_waitcnt vmcnt(0)`) ahead of the workgroup's barrier and the ONE lane that
// signals (an agent-scope atomic), and every load of it is an `sc1` load (past the L1) behind the poll / the returned add —
// no fence anywhere (`__threadfence()` per workgroup ran this route 10 x slower: it writes back and invalidates whole caches).
// 16-byte accesses go through raw buffer instructions (aux 16 = sc1) off a wave-uniform base, words through relaxed
// agent-scope atomics (the same instructions with sc1).
typedef unsigned uvec4_t __attribute__((ext_vector_type(4)));
typedef unsigned uvec2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pub_rsrc(const void* base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <typename V> __device__ __forceinline__ void pub_store16(__amdgpu_buffer_rsrc_t r, int byte_off, V v)
{
    static_assert(sizeof(V) == 16, "16-byte vectors");
    uvec4_t u; __builtin_memcpy(&u, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);
}
template <typename V> __device__ __forceinline__ V pub_load16(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    static_assert(sizeof(V) == 16, "16-byte vectors");
    const uvec4_t u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    V v; __builtin_memcpy(&v, &u, 16); return v;
}
template <typename T> __device__ __forceinline__ T pub_load1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    T v;
    if constexpr (sizeof(T) == 8) { const uvec2_t u = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16); __builtin_memcpy(&v, &u, 8); }
    else { const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16); __builtin_memcpy(&v, &u, 4); }
    return v;
}
__device__ __forceinline__ void pub_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int32_t pub_get(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int64_t pub_get(const int64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double pub_get(const double* p)
{
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void pub_put(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pub_put(int64_t* p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pub_put(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double dev_tan(double a) { return ::tan(a); }
__device__ __forceinline__ float dev_tan(float a) { return ::tanf(a); }
__device__ __forceinline__ double dev_hypot(double a, double b) { return ::hypot(a, b); }
__device__ __forceinline__ float dev_hypot(float a, float b) { return ::hypotf(a, b); }
__device__ __forceinline__ double dev_atan2(double a, double b) { return ::atan2(a, b); }
__device__ __forceinline__ float dev_atan2(float a, float b) { return ::atan2f(a, b); }

// ------------------------------------------------------------------------------------
// full_trace, stage B: per bundle, exclusive scan of the tile survivor counts (tile_off, FT_FULL route; may be
// null) and the bundle aggregates (count, centroid, max radius).  One workgroup per bundle; fixed shapes:
// bitwise reproducible.
// ------------------------------------------------------------------------------------
struct FtBundleAgg {
    int64_t m;        // survivors (first half)
    double mux, muy;  // centroid of the mirrored set  (PupilSampling.jl:171)
    double rmax;      // maximum(r)                    (:142)
    double sq;        // sum of squared deviations (filled by k_ft_finalize)
};

struct FtScanShared { int64_t w[kBlock / 64]; double rx[kBlock], ry[kBlock], rm[kBlock]; };

// bundle b, by kBlock threads tid = 0 .. kBlock-1 of a workgroup (also the first stage of k_ft_small_finish, where the
// workgroup holds more threads: those pass active = false and only keep the barriers company)
// PUB: inside the fused launch — the tile words were published by other workgroups of this launch, and the offsets and the
// aggregates written here are read by others: every access a pub_get / pub_put (the caller drains and signals)
template <bool PUB = false>
__device__ __forceinline__ void ft_scan_body(int b, const int32_t* tile_cnt, const double* tile_sx, const double* tile_sy,
                                             const double* tile_rmax, int tiles_per_bundle,
                                             int64_t* tile_off, FtBundleAgg* agg, FtScanShared& sh, int tid, bool active)
{
    auto get = [](auto* q) { if constexpr (PUB) return pub_get(q); else return *q; };
    // thread t owns the contiguous chunk of `per` tiles [t per, (t+1) per): a serial pass for its total, ONE block
    // scan of the 256 totals, a serial pass writing the offsets — two passes and one scan whatever the tile count
    // (8192 tiles per bundle at 2048^2: 32 per thread), where a block-wide scan per 256 tiles took 32 rounds
    int64_t* s_w = sh.w; double* s_rx = sh.rx; double* s_ry = sh.ry; double* s_rm = sh.rm;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)b * tiles_per_bundle;
    const int per = (tiles_per_bundle + kBlock - 1) / kBlock;
    const int t0 = tid * per, t1 = active ? min(tiles_per_bundle, t0 + per) : t0;
    int64_t mine = 0;
    double ax = 0.0, ay = 0.0, mx = -1.0;
    for (int t = t0; t < t1; ++t) {
        mine += get(tile_cnt + base + t);
        ax += get(tile_sx + base + t); ay += get(tile_sy + base + t); mx = fmax(mx, get(tile_rmax + base + t));
    }
    int64_t v = mine;                                            // inclusive wave scan of the per-thread totals
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t nbv = __shfl_up(v, off);
        if (lane >= off) v += nbv;
    }
    if (active) {
        if (lane == 63) s_w[wave] = v;
        s_rx[tid] = ax; s_ry[tid] = ay; s_rm[tid] = mx;
    }
    __syncthreads();
    int64_t woff = 0, total = 0;
    for (int w = 0; w < kBlock / 64; ++w) { woff += (w < wave) ? s_w[w] : 0; total += s_w[w]; }
    if (tile_off) {
        int64_t o = woff + v - mine;                             // exclusive offset of this thread's first tile
        for (int t = t0; t < t1; ++t) {
            if constexpr (PUB) pub_put(tile_off + base + t, o); else tile_off[base + t] = o;
            o += get(tile_cnt + base + t);
        }
    }
    // deterministic tree over the 256 per-thread partials
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (active && tid < off) { s_rx[tid] += s_rx[tid + off]; s_ry[tid] += s_ry[tid + off]; s_rm[tid] = fmax(s_rm[tid], s_rm[tid + off]); }
        __syncthreads();
    }
    if (active && tid == 0) {
        const int64_t m = total;
        FtBundleAgg a;
        a.m = m;
        // mean of [ex; -ex] and of [ey; ey] over n = 2m entries
        a.mux = m ? (s_rx[0] + (-s_rx[0])) / (double)(2 * m) : 0.0;
        a.muy = m ? (s_ry[0] + s_ry[0]) / (double)(2 * m) : 0.0;
        a.rmax = s_rm[0];
        a.sq = 0.0;
        if constexpr (PUB) {
            pub_put(&agg[b].m, a.m); pub_put(&agg[b].mux, a.mux); pub_put(&agg[b].muy, a.muy); pub_put(&agg[b].rmax, a.rmax);
            pub_put(&agg[b].sq, a.sq);
        } else agg[b] = a;
    }
}

__global__ __launch_bounds__(kBlock) void k_ft_scan(const int32_t* __restrict__ tile_cnt, const double* __restrict__ tile_sx,
                                                    const double* __restrict__ tile_sy, const double* __restrict__ tile_rmax,
                                                    int tiles_per_bundle, int64_t* __restrict__ tile_off, FtBundleAgg* __restrict__ agg)
{
    __shared__ FtScanShared sh;
    ft_scan_body(blockIdx.x, tile_cnt, tile_sx, tile_sy, tile_rmax, tiles_per_bundle, tile_off, agg, sh, threadIdx.x, true);
}

// ------------------------------------------------------------------------------------
// full_trace, stage C of the FT_FULL route: one workgroup per tile moves the tile's compacted survivors from its
// workspace slot to both halves of the bundle's output slab — first half at the tile's exclusive offset (ray
// order, :134-137), mirror [-ex; ey; rho; pi - theta] at offset m (:139-144), rho = r / maximum(r) (:142) — and
// sums the squared deviations about the centroid (two-pass sigma, :169-173).  32 B per survivor read (16-byte loads
// from the 16-byte-aligned slot), staged in LDS, 64 B written as aligned 16-byte streaming stores whatever the parity of
// the tile's offset and of m (stream_out): the kernel is bound by HBM, and 8-byte accesses ran it at half the store rate.
// ------------------------------------------------------------------------------------
template <typename T> struct FtPlaceShared { double wsq[kBlock / 64]; };

// x / d with r = 1 / d (correctly rounded) given: the quotient, its exact remainder, one correction — the correctly
// rounded quotient (Markstein), three operations per element instead of a division sequence each.
__device__ __forceinline__ double place_div(double x, double d, double r) { return ieee_div_nofix(x, d, r); }
__device__ __forceinline__ float place_div(float x, float d, float) { return x / d; }

// One half of a tile's placement: c compacted entries of the four workspace arrays at element `src` go to element `dst` of
// the four outputs.  The DESTINATION decides the chunking, and at the granularity the memory system writes in: `head`
// single elements up to the destination's first 128-byte line boundary, then 16-byte non-temporal stores of which every
// wave-wide instruction covers 8 WHOLE lines, then a tail.  Cut at 16-byte boundaries only, every wave store left a
// partial line at each end — completed later by its neighbour's partial line — and the pass ran 35 % slower per byte
// than with every slot full and aligned (ORT_PLACE_DEBUG A/Bs, profiles/r03_place_ab.log: 700 -> 477 us once each tile
// started on a line).  The loads take whatever alignment that leaves them (the slot is L2-resident: a misaligned 16-byte
// load costs one extra line per wave).  No staging buffer, no barrier.  FIRST half: also the squared deviations of the
// mirrored pair about the centroid (:169-173); MIRROR: [-ex; ey; rho; pi - theta] (:139-144).
template <typename T, bool MIRROR, bool PUB = false>
__device__ __forceinline__ double place_half(const T* __restrict__ w_ex, const T* __restrict__ w_ey, const T* __restrict__ w_r,
                                             const T* __restrict__ w_th, int64_t src, T* __restrict__ ex, T* __restrict__ ey,
                                             T* __restrict__ rho, T* __restrict__ theta, int64_t dst, int c, int tid,
                                             const FtBundleAgg& a, T rmax, T rinv)
{
    constexpr int V = 16 / (int)sizeof(T), L = 128 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    double sq = 0.0;
    // PUB (the fused launch): the slot was written by another workgroup of this launch — sc1 loads off the slot's base
    const __amdgpu_buffer_rsrc_t rx = pub_rsrc(PUB ? w_ex + src : nullptr), ry = pub_rsrc(PUB ? w_ey + src : nullptr),
                                 rr = pub_rsrc(PUB ? w_r + src : nullptr), rt = pub_rsrc(PUB ? w_th + src : nullptr);
    auto one = [&](int j) {
        T vx, vy, vr, vt;
        if constexpr (PUB) {
            vx = pub_load1<T>(rx, j * (int)sizeof(T)); vy = pub_load1<T>(ry, j * (int)sizeof(T));
            vr = pub_load1<T>(rr, j * (int)sizeof(T)); vt = pub_load1<T>(rt, j * (int)sizeof(T));
        } else { vx = w_ex[src + j]; vy = w_ey[src + j]; vr = w_r[src + j]; vt = w_th[src + j]; }
        if (!MIRROR) {
            const double dx1 = (double)vx - a.mux, dx2 = -(double)vx - a.mux, dy = (double)vy - a.muy;
            sq += (dx1 * dx1 + dx2 * dx2) + (dy * dy + dy * dy);
        }
        __builtin_nontemporal_store(MIRROR ? -vx : vx, ex + dst + j);                                   // :141
        __builtin_nontemporal_store(vy, ey + dst + j);                                                  // :140
        __builtin_nontemporal_store(place_div(vr, rmax, rinv), rho + dst + j);                          // :142,143
        __builtin_nontemporal_store(MIRROR ? (T)3.141592653589793 - vt : vt, theta + dst + j);          // :144
    };
    // vector stores need the four outputs equally placed inside a 16-byte word (they sit at the same element offset of
    // equally aligned arrays unless the caller passed odd pointers); the line cut is taken from ex
    const unsigned mis16 = (unsigned)(reinterpret_cast<uintptr_t>(ex + dst) & 15);
    const bool same = (reinterpret_cast<uintptr_t>(ey + dst) & 15) == mis16 && (reinterpret_cast<uintptr_t>(rho + dst) & 15) == mis16 &&
                      (reinterpret_cast<uintptr_t>(theta + dst) & 15) == mis16 && mis16 % sizeof(T) == 0;
    if (!same) {
        for (int j = tid; j < c; j += kBlock) one(j);
        return sq;
    }
    const int mis = (int)((reinterpret_cast<uintptr_t>(ex + dst) & 127) / sizeof(T));     // elements past a 128-byte line
    const int head = min(c, mis ? L - mis : 0);
    const int nvec = (c - head) / V;
    if (tid < head) one(tid);
    for (int g = tid; g < nvec; g += kBlock) {                                            // a wave: 1 KiB from a line boundary
        const int j = head + g * V;
        vec_t vx, vy, vr, vt;
        if constexpr (PUB) {
            vx = pub_load16<vec_t>(rx, j * (int)sizeof(T)); vy = pub_load16<vec_t>(ry, j * (int)sizeof(T));
            vr = pub_load16<vec_t>(rr, j * (int)sizeof(T)); vt = pub_load16<vec_t>(rt, j * (int)sizeof(T));
        } else {
            __builtin_memcpy(&vx, w_ex + src + j, sizeof(vec_t)); __builtin_memcpy(&vy, w_ey + src + j, sizeof(vec_t));
            __builtin_memcpy(&vr, w_r + src + j, sizeof(vec_t));  __builtin_memcpy(&vt, w_th + src + j, sizeof(vec_t));
        }
        vec_t ox, orr, ot;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            if (!MIRROR) {
                const double dx1 = (double)vx[q] - a.mux, dx2 = -(double)vx[q] - a.mux, dy = (double)vy[q] - a.muy;
                sq += (dx1 * dx1 + dx2 * dx2) + (dy * dy + dy * dy);
            }
            ox[q] = MIRROR ? -vx[q] : vx[q];
            orr[q] = place_div(vr[q], rmax, rinv);
            ot[q] = MIRROR ? (T)3.141592653589793 - vt[q] : vt[q];
        }
        __builtin_nontemporal_store(ox, reinterpret_cast<vec_t*>(ex + dst + j));
        __builtin_nontemporal_store(vy, reinterpret_cast<vec_t*>(ey + dst + j));
        __builtin_nontemporal_store(orr, reinterpret_cast<vec_t*>(rho + dst + j));
        __builtin_nontemporal_store(ot, reinterpret_cast<vec_t*>(theta + dst + j));
    }
    const int jt = head + nvec * V + tid;                                                 // < V - 1 elements left
    if (jt < c) one(jt);
    return sq;
}

// One tile's placement by kBlock threads: `c` survivors of slot `bid` to element `dst` of the bundle's first half and to
// dst + m of its mirror half.  Returns this thread's share of the tile's squared deviations.
template <typename T, bool PUB = false>
__device__ __forceinline__ double ft_place_tile(int bid, int c, int64_t dst, const FtBundleAgg& a,
                                                const T* __restrict__ w_ex, const T* __restrict__ w_ey,
                                                const T* __restrict__ w_r, const T* __restrict__ w_th,
                                                T* __restrict__ ex, T* __restrict__ ey, T* __restrict__ rho, T* __restrict__ theta,
                                                int tid)
{
    const int64_t src = (int64_t)bid * kTile;                                     // the tile's slot: 16-byte aligned, kTile entries
    const T rmax = (T)a.rmax, rinv = T(1) / rmax;                                 // r ./ maximum(r)  (:142)
    const double sq = place_half<T, false, PUB>(w_ex, w_ey, w_r, w_th, src, ex, ey, rho, theta, dst, c, tid, a, rmax, rinv);
    place_half<T, true, PUB>(w_ex, w_ey, w_r, w_th, src, ex, ey, rho, theta, dst + a.m, c, tid, a, rmax, rinv);
    return sq;
}

// One tile's placement laid out for LATENCY (the fused launch, FT_FUSED, PUB = true: the workgroup that runs it holds a trace
// kernel's registers while it waits; k_ft_place, PUB = false): what ft_place_tile<T, PUB> does, element for element and sum
// for sum.  EVERY load of
// both halves goes out first (a thread owns at most one head element, one 16-byte vector and one tail element per half:
// kTile / V <= kBlock), then `between()` runs (the caller's publish of its own tile: its wait covers these loads too), then
// the arithmetic and all the stores; nothing waits for a store.
template <typename T, bool PUB = true, typename F>
__device__ __forceinline__ double ft_place_tile_pub(int bid, int c, int64_t dst, const FtBundleAgg& a,
                                                    const T* w_ex, const T* w_ey, const T* w_r, const T* w_th,
                                                    T* __restrict__ ex, T* __restrict__ ey, T* __restrict__ rho, T* __restrict__ theta,
                                                    int tid, F between)
{
    constexpr int V = 16 / (int)sizeof(T), L = 128 / (int)sizeof(T), B = (int)sizeof(T);
    static_assert(kTile / V <= kBlock, "one vector per thread and half");
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int64_t src = (int64_t)bid * kTile;
    const T rmax = (T)a.rmax, rinv = T(1) / rmax;                                 // r ./ maximum(r)  (:142)
    bool same = true;
    int head[2], nvec[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t d = dst + (q ? a.m : 0);
        const unsigned mis16 = (unsigned)(reinterpret_cast<uintptr_t>(ex + d) & 15);
        same = same && (reinterpret_cast<uintptr_t>(ey + d) & 15) == mis16 && (reinterpret_cast<uintptr_t>(rho + d) & 15) == mis16 &&
               (reinterpret_cast<uintptr_t>(theta + d) & 15) == mis16 && mis16 % sizeof(T) == 0;
        const int mis = (int)((reinterpret_cast<uintptr_t>(ex + d) & 127) / sizeof(T));
        head[q] = min(c, mis ? L - mis : 0);
        nvec[q] = (c - head[q]) / V;
    }
    if (!same) {                                                 // odd output pointers: the element-wise path
        between();
        return ft_place_tile<T, PUB>(bid, c, dst, a, w_ex, w_ey, w_r, w_th, ex, ey, rho, theta, tid);
    }
    // PUB: the slot was written by another workgroup of this launch (sc1 loads); else by an earlier launch (plain loads)
    const __amdgpu_buffer_rsrc_t rx = pub_rsrc(PUB ? w_ex + src : nullptr), ry = pub_rsrc(PUB ? w_ey + src : nullptr),
                                 rr = pub_rsrc(PUB ? w_r + src : nullptr), rt = pub_rsrc(PUB ? w_th + src : nullptr);
    auto ld1 = [&](const T* w, __amdgpu_buffer_rsrc_t r, int j) -> T {
        if constexpr (PUB) return pub_load1<T>(r, j * B); else return w[src + j];
    };
    auto ld16 = [&](const T* w, __amdgpu_buffer_rsrc_t r, int j) -> vec_t {
        if constexpr (PUB) return pub_load16<vec_t>(r, j * B);
        else { vec_t v; __builtin_memcpy(&v, w + src + j, sizeof(vec_t)); return v; }
    };
    T hx[2], hy[2], hr[2], ht[2], tx[2], ty[2], tr[2], tt[2];
    vec_t vx[2], vy[2], vr[2], vt[2];
    int jv[2], jt[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        hx[q] = hy[q] = hr[q] = ht[q] = tx[q] = ty[q] = tr[q] = tt[q] = T(0);
        vx[q] = vy[q] = vr[q] = vt[q] = vec_t(T(0));
        jv[q] = head[q] + tid * V; jt[q] = head[q] + nvec[q] * V + tid;
        if (tid < head[q]) { hx[q] = ld1(w_ex, rx, tid); hy[q] = ld1(w_ey, ry, tid); hr[q] = ld1(w_r, rr, tid); ht[q] = ld1(w_th, rt, tid); }
        if (tid < nvec[q]) {
            vx[q] = ld16(w_ex, rx, jv[q]); vy[q] = ld16(w_ey, ry, jv[q]);
            vr[q] = ld16(w_r, rr, jv[q]); vt[q] = ld16(w_th, rt, jv[q]);
        }
        if (jt[q] < c) { tx[q] = ld1(w_ex, rx, jt[q]); ty[q] = ld1(w_ey, ry, jt[q]); tr[q] = ld1(w_r, rr, jt[q]); tt[q] = ld1(w_th, rt, jt[q]); }
    }
    between();
    double sq = 0.0;
    auto dev2 = [&](T x, T y) {                                  // the mirrored pair about the centroid (:169-173), as place_half takes it
        const double dx1 = (double)x - a.mux, dx2 = -(double)x - a.mux, dy = (double)y - a.muy;
        sq += (dx1 * dx1 + dx2 * dx2) + (dy * dy + dy * dy);
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t d = dst + (q ? a.m : 0);
        const bool mir = q == 1;
        auto one = [&](int j, T x, T y, T r, T t) {
            if (!mir) dev2(x, y);
            __builtin_nontemporal_store(mir ? -x : x, ex + d + j);                                      // :141
            __builtin_nontemporal_store(y, ey + d + j);                                                 // :140
            __builtin_nontemporal_store(place_div(r, rmax, rinv), rho + d + j);                         // :142,143
            __builtin_nontemporal_store(mir ? (T)3.141592653589793 - t : t, theta + d + j);             // :144
        };
        if (tid < head[q]) one(tid, hx[q], hy[q], hr[q], ht[q]);
        if (tid < nvec[q]) {
            vec_t ox, orr, ot;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if (!mir) dev2(vx[q][e], vy[q][e]);
                ox[e] = mir ? -vx[q][e] : vx[q][e];
                orr[e] = place_div(vr[q][e], rmax, rinv);
                ot[e] = mir ? (T)3.141592653589793 - vt[q][e] : vt[q][e];
            }
            __builtin_nontemporal_store(ox, reinterpret_cast<vec_t*>(ex + d + jv[q]));
            __builtin_nontemporal_store(vy[q], reinterpret_cast<vec_t*>(ey + d + jv[q]));
            __builtin_nontemporal_store(orr, reinterpret_cast<vec_t*>(rho + d + jv[q]));
            __builtin_nontemporal_store(ot, reinterpret_cast<vec_t*>(theta + d + jv[q]));
        }
        if (jt[q] < c) one(jt[q], tx[q], ty[q], tr[q], tt[q]);
    }
    return sq;
}

// tile `bid` (= bundle * tiles_per_bundle + tile), by kBlock threads tid = 0 .. kBlock-1 of a workgroup: the second stage of
// k_ft_small_finish, whose workgroup places four tiles at a time; active = false: no tile for this group
template <typename T>
__device__ __forceinline__ void ft_place_body(int bid, const T* __restrict__ w_ex, const T* __restrict__ w_ey,
                                              const T* __restrict__ w_r, const T* __restrict__ w_th,
                                              int64_t rpb, int tiles_per_bundle,
                                              const int32_t* __restrict__ tile_cnt, const int64_t* __restrict__ tile_off,
                                              const FtBundleAgg* __restrict__ agg,
                                              T* __restrict__ ex, T* __restrict__ ey,
                                              T* __restrict__ rho, T* __restrict__ theta,
                                              double* __restrict__ tile_sq, FtPlaceShared<T>& sh, int tid, bool active)
{
    double* s_wsq = sh.wsq;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = active ? bid / tiles_per_bundle : 0;
    const int c = active ? tile_cnt[bid] : 0;
    const FtBundleAgg a = agg[b];
    const int64_t dst = (int64_t)b * 2 * rpb + (active ? tile_off[bid] : 0);
    double sq = ft_place_tile<T>(bid, c, dst, a, w_ex, w_ey, w_r, w_th, ex, ey, rho, theta, tid);
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off);
    if (lane == 0) s_wsq[wave] = sq;
    __syncthreads();
    if (active && tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += s_wsq[w];
        tile_sq[bid] = t;
    }
}

// ------------------------------------------------------------------------------------
// The hot kernel.  GRID: rays generated from bundle axes; otherwise read from lists.
// HIST: write per-surface history.  SUMM: write image/stop hits + status.
// FT: full_trace epilogue.  The stop filter and an order-preserving compaction of the tile's survivors (wave
// ballot + popcount prefix, wave offsets through LDS) run in this kernel; where the compacted tile goes is the mode:
//   FT_FULL      to the tile's own slot of a workspace [N]; k_ft_scan turns the tile counts into offsets and
//                k_ft_place moves the survivors (25 B/ray read) to both halves of the output — no workgroup waits
//                for another (the default: fastest measured);
//   FT_LOOKBACK  straight to its final position in the first half, found by a decoupled look-back over the
//                bundle's earlier tiles (tiles taken in ticket order); k_ft_mirror adds the second half — least
//                HBM traffic (84 vs 103 B/ray measured), but every tile waits for its predecessors' counts and the kernel holds
//                4 waves per SIMD instead of 5 (ORT_FT_LOOKBACK; 5 % slower);
//   FT_FUSED     (ORT_FT_FUSED, two or more bundles) FT_FULL with the second pass INSIDE the launch: the workgroup that has traced
//                tile i into its slot then places tile i - lag (lag = tiles per bundle + a margin: a tile of a bundle complete
//                long ago) through k_ft_place's body, designated workgroups run the bundles' scans, the launch ends with `lag`
//                workgroups that only place — the HBM-bound pass overlaps the issue-bound one (config 3: -13 % at sustained
//                clocks, bit-identical).  Hand-offs: sc1 stores / sc1 loads, no fence (pub_* above);
//   FT_WALK, FT_WALK1   the statistics-only route: nothing ray-sized is written.  A workgroup WALKS consecutive tiles of its bundle
//                with the table staged once and no barrier after that; every lane carries (n, sum d, sum d^2) of its
//                survivors about the first survivor its wave met, and every SPAN of kSpan tiles each wave folds its lanes
//                (one shuffle tree) and writes ONE partial (n, mean, M2) — four per span, merged by k_ft_stats_reduce.  The span
//                length is a function of the bundle's shape alone (one tile for bundles of <= 32 tiles — the reference's own
//                call: any number of tiles per workgroup, down to one for a call that small —, kWalkTiles beyond: the fold
//                costs 7-11 % when taken per tile on BASELINE configs 3 and 5), so what a partial holds, and with it a bundle's
//                statistics, depends neither on how many spans a workgroup walks (walk_group: chosen per launch to keep the
//                chip full) nor on what else is in the launch.  Per workgroup — not per tile — is what a one-tile-per-
//                workgroup route pays: staging + three dependent round trips before the first ray moves, 1.28 M times on
//                BASELINE config 5 (profiles/r04_ab_walk_tiles_per_workgroup.log).
enum { FT_NONE = 0, FT_FULL = 1, FT_WALK1 = 2, FT_LOOKBACK = 3, FT_WALK = 4, FT_FUSED = 5 };   // FT_WALK1: FT_WALK with spans of ONE tile (compile-time: the
                                                                               // span test of the long-span kernel stays a constant)
#ifndef ORT_STATUS_ON_DEMAND
#define ORT_STATUS_ON_DEMAND 1   // summary kernels count the per-surface status only when the caller passed a status array (A/B: 0)
#endif
#ifndef ORT_STOP_EXIT
#define ORT_STOP_EXIT 1       // full_trace kernels: a wave whose rays are all outside the stop ends its surface loop there (A/B: 0 = none, 3 = every kernel)
#endif
#ifndef ORT_WALK_TILES
#define ORT_WALK_TILES 8     // tiles per span of the FT_WALK route for bundles of more than kSmallTiles tiles (a power of two)
#endif
constexpr int kWalkTiles = ORT_WALK_TILES;
#ifndef ORT_SUMM_WALK_F64
#define ORT_SUMM_WALK_F64 0   // A/B build: the Float64 summary kernels walk tiles as the Float32 ones do
#endif
template <typename T> constexpr bool kSummaryWalks = sizeof(T) == 4 || ORT_SUMM_WALK_F64;
constexpr int kStatusVignetted = 1 << 17, kStatusVigShift = 20;
// ------------------------------------------------------------------------------------
// ARMS: the row classes this build carries (surface_step_n): the batch's highest row decides (ort_system::arms).
// RPT = rays per lane.  2 (kRPT) everywhere but the small-problem route (a bundle of a few tiles: the reference's own
// call is 2,048 rays), whose launches are a handful of waves, each alone on its SIMD: there the time is the length of
// one wave's instruction stream, and RPT = 1 — the same 512-ray tile on 512 threads — halves it.  Same results bit for
// bit (the tile sums are taken in the RPT = 2 order, tile_sum2 below).
template <typename T, int MATH, int ARMS, bool GRID, bool HIST, bool SUMM, int FT, int RPT = kRPT>
__global__ __launch_bounds__(kTile / RPT, ARMS >= ARMS_EVEN ? ORT_POLY_WAVES : ((HIST && SUMM) || FT == 3 /* FT_LOOKBACK */ || ((FT == 4 || FT == 2) /* FT_WALK, FT_WALK1 */ && sizeof(T) == 8)) ? ORT_MIN_WAVES - 1 : HIST ? ORT_MIN_WAVES : sizeof(T) == 4 ? ORT_WAVES_F32 : ORT_WAVES_NOHIST)
void k_trace(TraceParams<T> p)   // both outputs, look-back epilogue, Float64 running sums, polynomial arms: 128 VGPRs (at 96 they park tens of values per row in scratch)
{
    constexpr int NT = kTile / RPT;                              // threads per workgroup: one tile of kTile rays
    constexpr int kSumWaves = kBlock / 64;                       // waves of the RPT = 2 shape: the order the tile sums are taken in
    static_assert(RPT == 1 || RPT == 2, "one or two rays per lane");
    constexpr bool POLY = ARMS >= ARMS_EVEN;
    constexpr bool WALK = FT == FT_WALK || FT == FT_WALK1;
    constexpr int kSpan = FT == FT_WALK1 ? 1 : kWalkTiles;       // tiles per span of the statistics-only route
    // Float32 summary-mode grid launches walk walk_group consecutive tiles of a bundle per workgroup too (a tile's output does
    // not depend on it): BASELINE config 5's hit payload 12.5 -> 10.5 ms.  Float64: measured, no gain (the per-tile trace is
    // twice as long, the workgroup's start-up hides behind it; profiles/r04_ab_summary_walk.log)
    constexpr bool SWALK = kSummaryWalks<T> && GRID && SUMM && !HIST && FT == FT_NONE && RPT == kRPT;
    static_assert(!WALK || (GRID && !HIST && !SUMM && RPT == kRPT), "FT_WALK: grid source, no other output");
    __shared__ SurfRec<T> s_rec[kMaxRows];
    __shared__ __attribute__((aligned(16))) T s_poly[POLY ? kMaxRows * kPolyLds : 1];
    __shared__ int s_wcnt[NT / 64];
    __shared__ double s_wsx[kSumWaves], s_wsy[kSumWaves], s_wmax[NT / 64];
    __shared__ double s_px[RPT == 1 ? NT / 64 : 1][32], s_py[RPT == 1 ? NT / 64 : 1][32];   // RPT = 1: pair sums (tile_sum2)
    __shared__ __attribute__((aligned(16))) T s_c[4][(FT == FT_FULL || FT == FT_LOOKBACK || FT == FT_FUSED) ? kTile : 1];   // a tile's compacted survivors

    const int tid = threadIdx.x;
    const int S = p.S;
    ORT_PHASE(8);
    // FT_LOOKBACK: the tile index is a TICKET, not blockIdx — the look-back below waits on tiles with lower indices,
    // and a ticket order guarantees they are running or done whatever order the hardware dispatches blocks in
    constexpr bool FUSED = FT == FT_FUSED;
    constexpr bool kCompact = FT == FT_FULL || FT == FT_LOOKBACK || FUSED;
    __shared__ unsigned s_bid;
    __shared__ int s_arrive;                                     // FT_FUSED: waves of this workgroup whose stores have left
    if (FUSED && tid == 0) s_arrive = 0;                         // (read after the table's barrier at the earliest)
    if (FT == FT_LOOKBACK) {
        if (tid == 0) {
#if ORT_FT_DEBUG & 2            /* A/B only: blockIdx order instead of tickets (no progress guarantee) */
            unsigned long long tk = blockIdx.x;
#else
            unsigned long long tk = atomicAdd(p.ft_ticket, 1ull) - p.ft_ticket_base;
#endif
            if (tk >= gridDim.x) {                                   // host bookkeeping fault: never index out of the grid, and say so —
                tk = gridDim.x - 1;                                  // the call returns an error instead of misplaced survivors
                atomicOr(p.ft_err, 2);
            }
            // tickets walk the bundles round-robin (ticket = tile * nb + bundle): the tiles in flight at any time
            // are spread over all the bundles' chains, so each look-back chain below is nb times shorter; a tile's
            // predecessors in its bundle still hold lower tickets
            const unsigned nbn = gridDim.x / (unsigned)p.tiles_per_bundle;
            s_bid = (unsigned)(tk % nbn) * (unsigned)p.tiles_per_bundle + (unsigned)(tk / nbn);
        }
        __syncthreads();
    }
    const unsigned bid = (FT == FT_LOOKBACK) ? s_bid : blockIdx.x;
    // FT_FUSED: the launch holds fuse_lag workgroups past the last tile; they trace nothing and place the last tiles
    const bool tracing = !FUSED || bid < (unsigned)p.fuse_ntiles;
    // FT_FUSED: the tile this workgroup places once it has traced its own (pj), and — read beside the bundle record and the
    // table, in their round trips — whether that tile's bundle is published yet and, if so, the tile's count and offset and
    // the bundle's aggregates (wave-uniform: kept in scalar registers across the trace)
    const int64_t pj = FUSED ? (int64_t)bid - p.fuse_lag : -1;
    const bool placing = FUSED && pj >= 0 && pj < p.fuse_ntiles;
    const int pb = placing ? (int)(pj / p.tiles_per_bundle) : 0;
    unsigned pf_rdy = 0;
    if constexpr (FUSED) if (placing) pf_rdy = __hip_atomic_load(p.ft_ready + pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int sysid, b = 0;
    int64_t j0;          // first ray of this thread inside its bundle / list
    int64_t gbase;       // global index of ray j0
    int64_t limit;       // rays in this bundle / list
    unsigned tile_base = 0;   // first ray of this workgroup's tile inside its bundle (wave-uniform)
    int walk_n = 1;           // tiles this workgroup traces one after the other (FT_WALK: walk_group spans of its bundle)
    int walk_tile = 0;        // FT_WALK: the tile being traced (index inside the bundle)
    if (GRID) {
        int tile;
        if (WALK) {
            const int groups = (p.walk_spans + p.walk_group - 1) / p.walk_group;      // workgroups per bundle
            b = bid / groups;
            tile = (bid - b * groups) * p.walk_group * kSpan;
            walk_n = min(p.walk_group * kSpan, p.tiles_per_bundle - tile);
            walk_tile = tile;
        } else if (SWALK) {
            const int groups = (p.tiles_per_bundle + p.walk_group - 1) / p.walk_group;      // workgroups per bundle
            b = bid / groups;
            tile = (bid - b * groups) * p.walk_group;
            walk_n = min(p.walk_group, p.tiles_per_bundle - tile);
        } else if (tracing) {
            b = bid / p.tiles_per_bundle;
            tile = bid - b * p.tiles_per_bundle;
        } else {
            b = 0; tile = 0; walk_n = 0;
        }
        tile_base = (unsigned)tile * (unsigned)kTile;
        sysid = p.bundles[b].system;
        j0 = (int64_t)tile * kTile + (int64_t)tid * RPT;
        limit = p.rpb;
        gbase = (int64_t)b * p.rpb + j0;
    } else {
        sysid = p.isys;
        j0 = (int64_t)bid * kTile + (int64_t)tid * RPT;
        limit = p.nrays;
        gbase = j0;
    }
    const SurfRec<T>* __restrict__ grec = p.recs + (int64_t)sysid * S;
    const T* __restrict__ gpoly = (POLY && p.polys) ? p.polys + (int64_t)sysid * S * kPolyRec : nullptr;
    // clear-aperture extension (no reference counterpart): wave-uniform row pointer, null = off
    typedef const __attribute__((address_space(4))) T* CApPtr;  // wave-uniform address: scalar loads
    const CApPtr gap2 = (ORT_APERTURES && (SUMM || FT) && p.apert2) ? (CApPtr)(uintptr_t)(p.apert2 + (int64_t)sysid * S) : (CApPtr)0;

    bool pf_ok = false;
    int pf_c = 0; int64_t pf_off = 0, pf_m = 0; double pf_mux = 0.0, pf_muy = 0.0, pf_rmax = 0.0;
    auto fetch_placement = [&]() {                               // behind a ready word that matched: sc1 loads
        pf_c = pub_get(p.tile_cnt + pj); pf_off = pub_get(p.tile_off + pj);
        pf_m = pub_get(&p.agg[pb].m); pf_mux = pub_get(&p.agg[pb].mux); pf_muy = pub_get(&p.agg[pb].muy); pf_rmax = pub_get(&p.agg[pb].rmax);
    };
    auto uniform_placement = [&]() {                             // the same value in every lane -> scalar registers
        auto u64 = [](unsigned long long v) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
            return ((unsigned long long)hi << 32) | lo;
        };
        pf_c = __builtin_amdgcn_readfirstlane(pf_c);
        pf_off = (int64_t)u64((unsigned long long)pf_off); pf_m = (int64_t)u64((unsigned long long)pf_m);
        pf_mux = __longlong_as_double((long long)u64((unsigned long long)__double_as_longlong(pf_mux)));
        pf_muy = __longlong_as_double((long long)u64((unsigned long long)__double_as_longlong(pf_muy)));
        pf_rmax = __longlong_as_double((long long)u64((unsigned long long)__double_as_longlong(pf_rmax)));
    };
    if constexpr (FUSED) {
        pf_ok = placing && pf_rdy == p.ft_epoch;
        if (pf_ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: the loads below stay below the ready word's)
            fetch_placement();
        }
    }
    if (tracing) {                                               // (workgroup-uniform)
        // stage this system's table: S records of sizeof(SurfRec<T>) bytes, as 16-B words
        constexpr int kW = sizeof(SurfRec<T>) / 16;
        const uint4* src = reinterpret_cast<const uint4*>(grec);
        uint4* dst = reinterpret_cast<uint4*>(s_rec);
        for (int w = tid; w < S * kW; w += NT) dst[w] = src[w];
        if (POLY && gpoly) {
            // kPolyLds values per row: the even form ev | qd for MATH_FAST on rows that have one, else pc | dc
            for (int w = tid; w < S * kPolyLds; w += NT) {
                const int row = w / kPolyLds, e = w - row * kPolyLds;
                const bool even = MATH == MATH_FAST && (grec[row].cls & (CLS_PEVEN | CLS_FINITE)) == (CLS_PEVEN | CLS_FINITE);
                s_poly[w] = gpoly[row * kPolyRec + ((even && e < 12) ? 24 + e : e)];
            }
        }
        __syncthreads();
    }
    if constexpr (FUSED) if (pf_ok) uniform_placement();

    // FT_WALK: this lane's survivors so far — their number and the sums of (ex - Kx), (ey - Ky) and of their squares, (Kx, Ky) =
    // the FIRST survivor its wave meets (wave-uniform: scalar registers): every term is of the size of the spot, so nothing
    // cancels whatever the centroid's offset
    int wk_n = 0;
    bool wk_have = false;
    double wk_kx = 0.0, wk_ky = 0.0, wk_sx = 0.0, wk_sy = 0.0, wk_qx = 0.0, wk_qy = 0.0;
    for (int wt = 0; wt < walk_n; ++wt) {
        ORT_PHASE(14);
        Ray<T> ray[RPT];
        bool live[RPT];
        int32_t st[RPT];
        T xs_[RPT], ys_[RPT];
        int stopi = -1;
        T hprime = T(0), a_stop = T(0);
        // (a lambda: a wave whose rays leave the domain of the fast forms launches them a second time, see below)
        auto launch_rays = [&](auto math) {
            constexpr int M = decltype(math)::value;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int64_t j = j0 + r;
                live[r] = j < limit;
                const int64_t jj = live[r] ? j : (limit - 1);   // clamp: dead lanes retrace a valid ray
                T y, x, u, v;
                if (GRID) {
                    const DevBundle<T>& bd = p.bundles[b];
                    // (iy, ix) of ray jj without a per-lane 64-bit division: the tile's first ray is
                    // divided once in scalar registers, lanes walk forward from there.
                    unsigned iy, ix;
                    if (live[r] && p.nx >= 64) {
                        const unsigned iy0 = tile_base / (unsigned)p.nx;         // wave-uniform: scalar division
                        ix = (tile_base - iy0 * (unsigned)p.nx) + (unsigned)(tid * RPT + r);
                        iy = iy0;
                        while (ix >= (unsigned)p.nx) { ix -= (unsigned)p.nx; ++iy; }
                    } else {
                        iy = (unsigned)((uint32_t)jj / (uint32_t)p.nx);
                        ix = (unsigned)jj - iy * (unsigned)p.nx;
                    }
                    y = p.axes[bd.yoff + iy];
                    x = p.axes[bd.xoff + ix];
                    stopi = bd.stop; hprime = bd.hprime; a_stop = bd.a_stop;
                    if (p.raybasis) {                        // PupilSampling.jl:124-127 (Q8): U = (ybar - y) / z0, V = -x / z0, then
                        // tan(U), tan(V) (:38-39) — functions of the pupil row and of the pupil column alone: taken once per
                        // row / column by k_make_slope_axes (ny + nx tangents per bundle instead of 2 ny nx), read here like y, x
                        const T* sl = p.rb_slopes + (int64_t)b * (p.ny + p.nx);
                        u = sl[iy];
                        v = sl[p.ny + ix];
                    } else {                                 // shared field angles: direction cosines are bundle-uniform
                        ray[r].y = y; ray[r].x = x; ray[r].u = bd.u; ray[r].v = bd.v; ray[r].sprev = T(0);
                        ray[r].k0 = bd.k0; ray[r].k1 = bd.k1; ray[r].k2 = bd.k2;
                        st[r] = 1; xs_[r] = T(0); ys_[r] = T(0);
                        continue;
                    }
                } else {
                    y = p.ly[jj]; x = p.lx[jj];
                    u = p.lU[jj]; v = p.lV[jj];
                    if (!p.slopes_given) { u = dev_tan(u); v = dev_tan(v); }   // :38-39
                }
                ray_init<T, M>(ray[r], y, x, u, v);
                st[r] = 1;
                xs_[r] = T(0); ys_[r] = T(0);
            }
        };
        launch_rays(std::integral_constant<int, MATH>{});
        const bool two = (RPT > 1) && live[RPT - 1];
        // History stores: one wave-uniform decision, taken once — every lane of the wave owns two live
        // rays and both row bases keep 16-byte alignment on every surface (ld even) -> plain
        // 16-byte stores off a scalar row base; otherwise the guarded per-lane path.
        const int lane_off = tid * RPT;
        const int64_t blockbase = gbase - lane_off;                  // wave-uniform
        bool vec_all = false;
        if (HIST && RPT == 2) {
            const bool al = ((reinterpret_cast<uintptr_t>(p.xv + gbase) | reinterpret_cast<uintptr_t>(p.yv + gbase)) &
                             (2 * sizeof(T) - 1)) == 0 && (p.ld & 1) == 0;
            vec_all = __all(two && al);
        }

        const int stop_u = __builtin_amdgcn_readfirstlane(stopi);    // bundle-uniform: the stop capture is a scalar branch
        const bool want_status = p.status != nullptr;                // (kernel argument: scalar)
        // The surface loop in arithmetic policy M.  MATH_FAST returns whether a ray of this lane left the domain of
        // the fast forms (`odd`, ort_device.hpp).
        auto trace_surfaces = [&](auto math) -> bool {
            constexpr int M = decltype(math)::value;
            bool odd = false;
            if (M == MATH_FAST) {                                    // Inf / NaN launch data: the reference just computes with
#pragma unroll                                                   // them, and so does its own operation sequence (retrace)
                for (int r = 0; r < RPT; ++r)
                    odd = odd || t_class(ray[r].x, kClassNonFinite) || t_class(ray[r].y, kClassNonFinite) ||
                          t_class(ray[r].k0 + ray[r].k1, kClassNonFinite);
            }
            int s_lim = S;                                           // (drops to the stop row for a wave that lies outside the stop)
            for (int i = 0; i < s_lim; ++i) {
                const SurfRec<T>& rec = s_rec[i];
                const int cls = __builtin_amdgcn_readfirstlane(rec.cls);     // wave-uniform -> scalar branch
                // the staged block is laid out for the kernel's own policy; the (cold) MATH_IEEE retrace of a MATH_FAST kernel
                // reads its pc | dc block from the table itself
                const T* cf = (M == MATH) ? (s_poly + i * kPolyLds) : (gpoly ? gpoly + i * kPolyRec : nullptr);
                surface_step_n<T, M, RPT, ARMS>(ray, rec, cf, cls, i == S - 1, odd);
                if (SUMM || FT) {
                    if (SUMM && (!ORT_STATUS_ON_DEMAND || want_status)) {  // the full_trace epilogue reads the final NaN-ness only: no count;
                                                                         // nor does a summary call that asks for hits without status (scalar branch)
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            // NaN is sticky (every later transfer propagates it), so the 1-based index of the first
                            // NaN surface is 1 + the number of surfaces with ordered (x, y): one v_cmp_o + one add.
                            st[r] += __builtin_isunordered(ray[r].x, ray[r].y) ? 0 : 1;
                        }
                    }
                    if (i == stop_u) {
                        const T a2 = a_stop * a_stop, alim = (T)Near<T>::thr * a2;
                        // full_trace keeps only the rays inside the stop (:131-132): a wave whose rays are ALL outside it — by more
                        // than the margin inside which the reference's hypot decides, in either policy — has nothing left to
                        // contribute: the surface loop's bound drops to this row (a square pupil around a round stop: ~15 % of the
                        // waves of a 2048^2 bundle, at about half of their rows: BASELINE config 3 statistics-only -7 %, fused -1.3 %).
                        // As a `break` the exit changes how the compiler lays the loop out (Float32: 2,056 -> 1,541 instructions, yet
                        // ort_spot_batch_f32 8.15 -> 9.18 ms); as a bound it still costs the Float32 kernels 4 % (config 5's waves are
                        // whole pupil rows, never all outside) and the plain-sphere Float64 compaction kernel 3 %, so it is taken by
                        // the Float64 statistics kernels and the polynomial builds only (profiles/r04_ab_stop_exit.log;
                        // -DORT_STOP_EXIT=3: every full_trace kernel, =0: none)
                        constexpr bool kStopExit = ORT_STOP_EXIT && FT != FT_NONE && !SUMM && !HIST &&
                                                   (ORT_STOP_EXIT == 3 || (sizeof(T) == 8 && (WALK || POLY)));      // (which kernels: see below)
                        bool out_all = kStopExit && a_stop >= T(0);
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            xs_[r] = ray[r].x; ys_[r] = ray[r].y;
                            const T e = t_fma<T>(xs_[r], xs_[r], t_fma<T>(ys_[r], ys_[r], -a2));     // r^2 - a_stop^2
                            // within kNear of the stop's edge the filter r > a_stop (:132) is decided by the reference sequence
                            if (M == MATH_FAST) odd = odd || near_zero<T>(e, alim);
                            if (kStopExit) out_all = out_all && (!live[r] || e > alim);              // (NaN: not outside)
                        }
                        if (kStopExit && __all(out_all)) s_lim = i + 1;
                    }
                    if (gap2) {                                          // scalar branch: one s_cbranch when off
                        const T a2 = gap2[i];
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            // bit 17: outside the clear aperture of some surface; bits 20..27 count the surfaces
                            // passed before that -> 1-based index of the first vignetting surface = count + 1
                            const T r2 = ray[r].x * ray[r].x + ray[r].y * ray[r].y;
                            if (M == MATH_FAST) odd = odd || near_zero<T>(r2 - a2, (T)Near<T>::thr * a2);
                            st[r] |= (r2 > a2) ? kStatusVignetted : 0;
                            if (SUMM) st[r] += (st[r] & kStatusVignetted) ? 0 : (1 << kStatusVigShift);
                        }
                    }
                }
                if (HIST) {
                    if (vec_all) {
                        T* rx = p.xv + ((int64_t)i * p.ld + blockbase);  // scalar row base
                        T* ry = p.yv + ((int64_t)i * p.ld + blockbase);
                        store_vec2<T>(rx + lane_off, ray[0].x, ray[RPT - 1].x);
                        store_vec2<T>(ry + lane_off, ray[0].y, ray[RPT - 1].y);
                    } else if (live[0]) {
                        store_pair<T>(p.xv + (int64_t)i * p.ld, gbase, two, ray[0].x, ray[RPT - 1].x);
                        store_pair<T>(p.yv + (int64_t)i * p.ld, gbase, two, ray[0].y, ray[RPT - 1].y);
                    }
                }
            }
            return odd;
        };
        const bool odd_seen = trace_surfaces(std::integral_constant<int, MATH>{});
#ifdef ORT_COUNT_RETRACE
        if (MATH == MATH_FAST && (tid & 63) == 0) { atomicAdd(&g_retrace[0], 1ull); if (__any(odd_seen)) atomicAdd(&g_retrace[1], 1ull); }
#endif
        if (MATH == MATH_FAST && __builtin_expect(__any(odd_seen), 0)) {
            // A ray of this wave went where the reference's formulas are no longer the geometry the fast forms compute
            // (a far-cap hit, a direction refracted backward, a polynomial row outside its conic: possible only far
            // outside any clear aperture).  What the reference does there is defined by its operation sequence, so the
            // wave traces its rays again with exactly that — MATH_IEEE, bit-identical to the CPU reference; its history
            // stores land on the same addresses, after the first pass's have completed.
            __builtin_amdgcn_s_waitcnt(0);
            launch_rays(std::integral_constant<int, MATH_IEEE>{});
            trace_surfaces(std::integral_constant<int, MATH_IEEE>{});
        }

        ORT_PHASE(15);
        // The stop filter r > a_stop (PupilSampling.jl:131-132).  The reference sequence takes hypot.  MATH_FAST decides by
        // r^2 against a_stop^2 — the same outcome for every ray farther than kNear from the edge — and only the rays within
        // kNear of it (their wave has retraced with the reference sequence, see the stop capture above) take the hypot.
        auto outside_stop = [&](T xs, T ys, T& r2) -> bool {
            if (MATH == MATH_IEEE) { r2 = T(0); return dev_hypot(xs, ys) > a_stop; }
            const T a2 = a_stop < T(0) ? T(-1) : a_stop * a_stop;        // a negative radius passes nothing
            r2 = t_fma<T>(xs, xs, ys * ys);
            bool out = r2 > a2;
            if (near_zero<T>(r2 - a2, (T)Near<T>::thr * a2)) out = dev_hypot(xs, ys) > a_stop;
            return out;
        };

        if (SUMM) {
            if (live[0]) {
                if (p.xf) store_pair<T>(p.xf, gbase, two, ray[0].x, ray[RPT - 1].x);
                if (p.yf) store_pair<T>(p.yf, gbase, two, ray[0].y, ray[RPT - 1].y);
                if (p.xs) store_pair<T>(p.xs, gbase, two, xs_[0], xs_[RPT - 1]);
                if (p.ys) store_pair<T>(p.ys, gbase, two, ys_[0], ys_[RPT - 1]);
                if (p.status) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        if (!live[r]) continue;
                        int32_t s = st[r];
                        if (stopi >= 0) {
                            T r2;
                            if (outside_stop(xs_[r], ys_[r], r2)) s |= (1 << 16);
                        }
                        p.status[gbase + r] = s;
                    }
                }
            }
            if (SWALK) { tile_base += (unsigned)kTile; j0 += kTile; gbase += kTile; }      // the next tile
        }

        if (WALK) {
            // stop filter (PupilSampling.jl:129-137), then straight into the lane's running sums (:169-173): no barrier, no
            // shuffle, nothing per tile but this
            bool keep[RPT];
            double ex[RPT], ey[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const T xf = ray[r].x, yf = ray[r].y;
                T r2;
                const bool outside = outside_stop(xs_[r], ys_[r], r2);                        // :131-132
                keep[r] = !(outside || t_isnan(xf) || t_isnan(yf) || !live[r] || (st[r] & kStatusVignetted));
                ex[r] = (double)xf; ey[r] = (double)(yf - hprime);                             // :135, :134
            }
            if (!wk_have) {                                                                  // scalar branch: taken until the wave holds a survivor
#pragma unroll
                for (int r = RPT - 1; r >= 0; --r) {
                    const unsigned long long m = __ballot(keep[r]);
                    if (m) {
                        const int src = __builtin_ctzll(m);
                        wk_kx = readlane_f64(ex[r], src); wk_ky = readlane_f64(ey[r], src); wk_have = true;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const double dx = keep[r] ? ex[r] - wk_kx : 0.0, dy = keep[r] ? ey[r] - wk_ky : 0.0;
                wk_sx += dx; wk_sy += dy;
                wk_qx = __builtin_fma(dx, dx, wk_qx); wk_qy = __builtin_fma(dy, dy, wk_qy);
                wk_n += keep[r] ? 1 : 0;
            }
            tile_base += (unsigned)kTile; j0 += kTile; gbase += kTile; ++walk_tile;            // the next tile
            if (walk_tile % kSpan == 0 || wt == walk_n - 1) {
                // end of a span (wave-uniform): fold the wave's lanes — one K per wave, so the sums just add; a 64-lane shuffle
                // tree, fixed shape: bitwise reproducible — and write the wave's partial (n, mean, M2) of this span
                const int lane = tid & 63, wave = tid >> 6;
                double n = (double)wk_n;
                for (int off = 32; off > 0; off >>= 1) {
                    n += __shfl_down(n, off);
                    wk_sx += __shfl_down(wk_sx, off); wk_sy += __shfl_down(wk_sy, off);
                    wk_qx += __shfl_down(wk_qx, off); wk_qy += __shfl_down(wk_qy, off);
                }
                if (lane == 0) {
                    const int64_t o = (((int64_t)b * p.walk_spans + (walk_tile - 1) / kSpan) * (kBlock / 64)) + wave;
                    const double inv = n > 0.0 ? 1.0 / n : 0.0;
                    p.tile_cnt[o] = (int32_t)n;
                    p.tile_sx[o] = __builtin_fma(wk_sx, inv, wk_kx); p.tile_sy[o] = __builtin_fma(wk_sy, inv, wk_ky);
                    p.tile_m2x[o] = fmax(0.0, __builtin_fma(-wk_sx * inv, wk_sx, wk_qx));   // sum d^2 - (sum d)^2 / n about K ~ the mean
                    p.tile_m2y[o] = fmax(0.0, __builtin_fma(-wk_sy * inv, wk_sy, wk_qy));
                    p.tile_rmax[o] = -1.0;                                                   // (statistics only: no r)
                }
                wk_n = 0; wk_have = false;
                wk_kx = wk_ky = wk_sx = wk_sy = wk_qx = wk_qy = 0.0;
            }
        }
        if (kCompact) {
            // stop filter (PupilSampling.jl:129-137); the tile's survivors are compacted IN RAY ORDER (two 64-bit ballots +
            // popcount prefix per wave, wave offsets through LDS) and streamed to the tile's workspace slot (FT_FULL) or to
            // their final place in the first half of the bundle's output slab (FT_LOOKBACK).
            int cnt = 0; double sx = 0.0, sy = 0.0, rmax = -1.0;
            T exv[RPT], eyv[RPT], rv[RPT], thv[RPT];
            bool keep[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const T xf = ray[r].x, yf = ray[r].y;
                T r2;
                const bool outside = outside_stop(xs_[r], ys_[r], r2);                        // :131-132
                // r itself (rho, :136,142): the reference sequence keeps its hypot; MATH_FAST takes the root of r^2 (seeded from
                // r^2 + tiny: an on-axis ray gives 0, not 0 * inf), and the statistics-only route needs no r at all
                T ri = T(0);
                if (MATH == MATH_IEEE) ri = dev_hypot(xs_[r], ys_[r]);
                else if (kCompact) ri = r2 * fast_rsqrt(r2 + (sizeof(T) == 8 ? (T)1e-300 : (T)1e-36));
                const bool drop = outside || t_isnan(xf) || t_isnan(yf) || !live[r] || (st[r] & kStatusVignetted);
                keep[r] = !drop;
                if (kCompact) thv[r] = MATH == MATH_FAST ? fast_atan2(ys_[r], xs_[r]) : dev_atan2(ys_[r], xs_[r]);   // :133
                eyv[r] = yf - hprime;                                    // :134
                exv[r] = xf;                                             // :135
                rv[r] = drop ? T(-1) : ri;                               // :136, -1 marks a dropped ray
                if (!drop) { ++cnt; sx += (double)exv[r]; sy += (double)eyv[r]; rmax = fmax(rmax, (double)ri); }
            }
            const int lane = tid & 63, wave = tid >> 6;
            int rank0 = 0;
            if (kCompact) {
                const unsigned long long m0 = __ballot(keep[0]);
                const unsigned long long m1 = (RPT > 1) ? __ballot(keep[RPT - 1]) : 0ull;
                const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                rank0 = __popcll(m0 & lt) + __popcll(m1 & lt);
            }
            // tile aggregates: fixed-shape tree -> bitwise reproducible.  The floating-point sums are taken in ONE order whatever
            // RPT is — that of the two-rays-per-lane shape: a lane's two rays, a 64-lane shuffle tree over each group of 128
            // rays (-> s_wsx / s_wsy[group]), the groups in sequence by whoever reads them.  With one ray per lane a group
            // spans two waves: neighbouring lanes pair up, the pair sums meet in LDS, and the tree continues from its second level.
            auto tile_sum2 = [&](double x, double y) {
                if constexpr (RPT == 2) {
                    for (int off = 32; off > 0; off >>= 1) { x += __shfl_down(x, off); y += __shfl_down(y, off); }
                    if (lane == 0) { s_wsx[wave] = x; s_wsy[wave] = y; }
                } else {
                    x += __shfl_xor(x, 1); y += __shfl_xor(y, 1);            // a + b on both lanes of the pair: the same bits
                    if (!(lane & 1)) { s_px[wave][lane >> 1] = x; s_py[wave][lane >> 1] = y; }
                    __syncthreads();
                    if (wave < kSumWaves) {
                        double qx = 0.0, qy = 0.0;
                        if (lane < 32) {                                     // the tree's first level: lanes l and l + 32 of the group
                            qx = s_px[2 * wave][lane] + s_px[2 * wave + 1][lane];
                            qy = s_py[2 * wave][lane] + s_py[2 * wave + 1][lane];
                        }
                        for (int off = 16; off > 0; off >>= 1) { qx += __shfl_down(qx, off); qy += __shfl_down(qy, off); }
                        if (lane == 0) { s_wsx[wave] = qx; s_wsy[wave] = qy; }
                    }
                }
            };
            for (int off = 32; off > 0; off >>= 1) {
                cnt += __shfl_down(cnt, off);
                rmax = fmax(rmax, __shfl_down(rmax, off));
            }
            if (lane == 0) { s_wcnt[wave] = cnt; s_wmax[wave] = rmax; }
            tile_sum2(sx, sy);
            __syncthreads();
            if (kCompact) {
                T* const s_cx = s_c[0]; T* const s_cy = s_c[1]; T* const s_cr = s_c[2]; T* const s_ct = s_c[3];
                __shared__ long long s_base;
                int woff = 0, c = 0;
                for (int w = 0; w < NT / 64; ++w) { woff += (w < wave) ? s_wcnt[w] : 0; c += s_wcnt[w]; }
                int k = woff + rank0;
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (keep[r]) { s_cx[k] = exv[r]; s_cy[k] = eyv[r]; s_cr[k] = rv[r]; s_ct[k] = thv[r]; ++k; }
                const int tile = (int)(bid - (unsigned)b * (unsigned)p.tiles_per_bundle);
                if (FT == FT_FULL || FUSED) {
                    if (tid == 0) {
                        double ax = 0.0, ay = 0.0, mx = -1.0;
                        for (int w = 0; w < kSumWaves; ++w) { ax += s_wsx[w]; ay += s_wsy[w]; }
                        for (int w = 0; w < NT / 64; ++w) mx = fmax(mx, s_wmax[w]);
                        if constexpr (FUSED) {                       // read by another workgroup of this launch (the bundle's scan)
                            pub_put(p.tile_cnt + bid, (int32_t)c); pub_put(p.tile_sx + bid, ax); pub_put(p.tile_sy + bid, ay);
                            pub_put(p.tile_rmax + bid, mx);
                        } else { p.tile_cnt[bid] = c; p.tile_sx[bid] = ax; p.tile_sy[bid] = ay; p.tile_rmax[bid] = mx; }
                    }
                } else if (wave == 0) {
                    // Exclusive offset of this tile among its bundle's survivors: decoupled look-back (Merrill & Garland) over
                    // the bundle's earlier tiles, 64 at a time.  One 8-byte word per tile carries everything, so relaxed
                    // agent-scope atomics suffice: state 1 = the tile's own count, state 2 = inclusive prefix.
                    double ax = 0.0, ay = 0.0, mx = -1.0;
                    for (int w = 0; w < kSumWaves; ++w) { ax += s_wsx[w]; ay += s_wsy[w]; }
                    for (int w = 0; w < NT / 64; ++w) mx = fmax(mx, s_wmax[w]);
                    const unsigned long long ep = (unsigned long long)(p.ft_epoch & 0x3fffffffu) << 32;
                    unsigned long long* stw = p.ft_state + (size_t)b * p.tiles_per_bundle;
                    if (lane == 0) {
                        p.tile_cnt[bid] = c; p.tile_sx[bid] = ax; p.tile_sy[bid] = ay; p.tile_rmax[bid] = mx;
                        __hip_atomic_store(stw + tile, ((tile == 0 ? 2ull : 1ull) << 62) | ep | (unsigned)c, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                    long long excl = 0;
#if ORT_FT_DEBUG & 1            /* A/B only: no look-back, every tile writes at its dense position (wrong offsets, same traffic) */
                    int hi = -1; excl = (long long)tile * kTile * 3 / 4;
#else
                    int hi = tile - 1;                               // next predecessor to look at
#endif
                    constexpr int kLook = 4;                         // predecessors per lane and round: windows of 256 tiles
                    while (hi >= 0) {
                        // lane l looks at tiles hi - kLook l - j, j = 0 .. kLook-1 (nearest first); every tile with a lower
                        // ticket is running or done, so the waits end; the cap is a guard against a host-side bookkeeping
                        // error only: it raises the fault word and the whole call returns an error
                        long long part = 0;                          // sum of this lane's words up to its first inclusive one
                        bool found = false;
#pragma unroll
                        for (int j = 0; j < kLook; ++j) {
                            const int t = hi - kLook * lane - j;
                            unsigned long long wd = 2ull << 62 | ep; // beyond the first tile: an inclusive prefix of 0
                            if (t >= 0) {
                                bool got = false;
                                for (int spin = 0; spin < (1 << 22) && !got; ++spin) {
                                    wd = __hip_atomic_load(stw + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    got = (wd >> 62) != 0 && (wd & (0x3fffffffull << 32)) == ep;
                                    if (!got) {
                                        // a fault already recorded (tickets outside the grid: predecessors that never run): stop waiting
                                        if ((spin & 255) == 255 && __hip_atomic_load(p.ft_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                                        __builtin_amdgcn_s_sleep(1);
                                    }
                                }
                                if (!got) { atomicOr(p.ft_err, 1); wd = 2ull << 62 | ep; }   // cap hit: the call fails (run_full_trace)
                            }
                            if (!found) part += (long long)(wd & 0xffffffffull);
                            found = found || ((wd >> 62) == 2);
                        }
                        const unsigned long long incl = __ballot(found);
                        const int first = incl ? __builtin_ctzll(incl) : 64;       // nearest lane holding an inclusive prefix
                        long long v = (lane <= first) ? part : 0;
                        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
                        excl += __shfl(v, 0);
                        if (incl) break;
                        hi -= 64 * kLook;
                    }
                    if (lane == 0) {
                        if (tile != 0)
                            __hip_atomic_store(stw + tile, (2ull << 62) | ep | (unsigned long long)(unsigned)(excl + c), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        s_base = excl;
                    }
                }
                __syncthreads();
                if (FT == FT_FULL || FUSED) {
                    // the tile's own slot of the workspace [tiles][kTile]: 16-byte aligned, so whole vectors go out (the
                    // entries past c inside the slot are never read)
                    constexpr int V = 16 / (int)sizeof(T);
                    typedef T vec_t __attribute__((ext_vector_type(V)));
                    const int64_t o0 = (int64_t)bid * kTile;
                    if constexpr (FUSED) {                           // write-through: another workgroup of this launch places the slot
                        const __amdgpu_buffer_rsrc_t rx = pub_rsrc(p.out_ex + o0), ry = pub_rsrc(p.out_ey + o0),
                                                     rr = pub_rsrc(p.out_r + o0), rt = pub_rsrc(p.out_th + o0);
                        for (int j = tid * V; j < c; j += NT * V) {
                            pub_store16(rx, j * (int)sizeof(T), *reinterpret_cast<const vec_t*>(s_cx + j));
                            pub_store16(ry, j * (int)sizeof(T), *reinterpret_cast<const vec_t*>(s_cy + j));
                            pub_store16(rr, j * (int)sizeof(T), *reinterpret_cast<const vec_t*>(s_cr + j));
                            pub_store16(rt, j * (int)sizeof(T), *reinterpret_cast<const vec_t*>(s_ct + j));
                        }
                    } else
                    for (int j = tid * V; j < c; j += NT * V) {
                        *reinterpret_cast<vec_t*>(p.out_ex + o0 + j) = *reinterpret_cast<const vec_t*>(s_cx + j);
                        *reinterpret_cast<vec_t*>(p.out_ey + o0 + j) = *reinterpret_cast<const vec_t*>(s_cy + j);
                        *reinterpret_cast<vec_t*>(p.out_r + o0 + j) = *reinterpret_cast<const vec_t*>(s_cr + j);
                        *reinterpret_cast<vec_t*>(p.out_th + o0 + j) = *reinterpret_cast<const vec_t*>(s_ct + j);
                    }
                } else {
                    // FT_LOOKBACK: its place in the bundle's output slab [2 rpb], any alignment
                    const int64_t o0 = (int64_t)b * 2 * p.rpb + s_base;
                    auto same = [](T v) { return v; };
                    stream_out<T>(p.out_ex + o0, s_cx, c, tid, same); stream_out<T>(p.out_ey + o0, s_cy, c, tid, same);
                    stream_out<T>(p.out_r + o0, s_cr, c, tid, same);  stream_out<T>(p.out_th + o0, s_ct, c, tid, same);
                }
            }
        }
    }
    if constexpr (FUSED) {
        // FT_FUSED: the second pass (HBM-bound) inside the trace launch (FP64-issue-bound).  Workgroup i has traced tile i into
        // its workspace slot (sc1 stores); below it (1) publishes the tile, (2) places tile pj = i - fuse_lag — a tile of a bundle that
        // was complete `margin` workgroups ago, so its wait is a safety net, not a queue — and (3), if it is the one named for
        // it, runs a bundle's scan and raises that bundle's ready word.  A workgroup only ever waits for workgroups with LOWER
        // indices (dispatched before it): no cycle; a poll cap raises the fault word and the call returns an error instead of
        // hanging.  Same bodies as k_ft_scan / k_ft_place: same bits.
        static_assert(!FUSED || RPT == kRPT, "FT_FUSED: kBlock threads per workgroup");
        __shared__ int s_go;
        const int tpb = p.tiles_per_bundle;
        const int lane = tid & 63, wave = tid >> 6;
        // (1) publish this workgroup's own tile, wave by wave: a wave whose sc1 stores have left (its wait covers the placement's
        // loads below too) counts itself in, and the wave that comes last adds the tile to its bundle's count — nobody waits
        // for an answer, and no barrier holds a wave that is done
        auto publish = [&]() {
            if (!tracing) return;
            pub_drain();
            if (lane == 0 && atomicAdd(&s_arrive, 1) == kBlock / 64 - 1)
                __hip_atomic_fetch_add(p.ft_done + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // (2) place tile pj
        bool published = false;
        if (placing) {                                           // (workgroup-uniform)
            // each wave read the ready word for itself when the workgroup started: only the workgroup's barrier makes one answer
            __shared__ int s_ok[kBlock / 64];
            if (lane == 0) s_ok[wave] = pf_ok;
            __syncthreads();
            bool all_ok = true;
            for (int w = 0; w < kBlock / 64; ++w) all_ok = all_ok && s_ok[w] != 0;
            if (!all_ok) {                                       // (workgroup-uniform) not published when some wave looked
                publish(); published = true;                     // (its own tile first: the wait below may be long)
                if (tid == 0) {
                    bool got = false;
                    for (int spin = 0; spin < p.fuse_spin_cap && !got; ++spin) {
                        got = __hip_atomic_load(p.ft_ready + pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.ft_epoch;
                        if (!got) {
                            if ((spin & 255) == 255 && __hip_atomic_load(p.ft_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                            __builtin_amdgcn_s_sleep(4);
                        }
                    }
                    if (!got) atomicOr(p.ft_err, 1);
                    s_go = got;
                }
                __syncthreads();
                if (s_go != 0 && !pf_ok) {                       // (per wave) the waves that have not read the tile's words yet
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: the loads below stay below the poll)
                    fetch_placement(); uniform_placement();
                }
                pf_ok = s_go != 0;
            }
            if (pf_ok) {
                FtBundleAgg a;
                a.m = pf_m; a.mux = pf_mux; a.muy = pf_muy; a.rmax = pf_rmax; a.sq = 0.0;
                const int64_t dst = (int64_t)pb * 2 * p.rpb + pf_off;
                double sq = ft_place_tile_pub<T>((int)pj, pf_c, dst, a, p.out_ex, p.out_ey, p.out_r, p.out_th,
                                                 p.fin_ex, p.fin_ey, p.fin_rho, p.fin_th, tid,
                                                 [&]() { if (!published) { publish(); published = true; } });
                for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o);
                if (lane == 0) p.tile_sq[pj * (kBlock / 64) + wave] = sq;       // k_ft_finalize<kBlock / 64> folds the waves in order
            }
        }
        if (!published) publish();
        // (3) the workgroup fuse_scan_lag - 1 past the last tile of a bundle runs that bundle's scan (k_ft_scan's body) once the
        // bundle's count is complete, and raises its ready word
        const int64_t sq0 = (int64_t)bid - (p.fuse_scan_lag - 1);                // = (bundle + 1) * tpb for such a workgroup
        if (sq0 > 0 && sq0 % tpb == 0 && sq0 <= p.fuse_ntiles) {                 // (workgroup-uniform)
            const int sb = (int)(sq0 / tpb) - 1;
            if (tid == 0) {
                bool got = false;
                for (int spin = 0; spin < p.fuse_spin_cap && !got; ++spin) {
                    got = pub_get(p.ft_done + sb) == tpb;
                    if (!got) {
                        if ((spin & 255) == 255 && __hip_atomic_load(p.ft_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                if (!got) atomicOr(p.ft_err, 1);
                s_go = got;
            }
            __syncthreads();
            if (s_go) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                ft_scan_body<true>(sb, p.tile_cnt, p.tile_sx, p.tile_sy, p.tile_rmax, tpb, p.tile_off, p.agg,
                                   *reinterpret_cast<FtScanShared*>(&s_c[0][0]), tid, true);
                pub_drain();
                __syncthreads();
                if (tid == 0) __hip_atomic_store(p.ft_ready + sb, p.ft_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// full_trace statistics-only, stage B: merge the per-tile (n, mean, M2) triples of a bundle with
// Chan's pairwise update (fixed order: bitwise reproducible) and emit count and RMS of the
// MIRRORED set [ex; -ex], [ey; ey] (src/PupilSampling.jl:140-141,169-173):
//   mean_x = 0, M2x_total = 2 (M2x + n mx^2);  mean_y = my, M2y_total = 2 M2y;  RMS^2 = (.)/(2n).
// ------------------------------------------------------------------------------------
struct Moments { double n, mx, my, qx, qy, rmax; };

__device__ __forceinline__ Moments chan_merge(const Moments& a, const Moments& b)
{
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    Moments o;
    o.n = a.n + b.n;
    const double dx = b.mx - a.mx, dy = b.my - a.my, w = a.n * b.n / o.n;
    o.mx = a.mx + dx * (b.n / o.n);
    o.my = a.my + dy * (b.n / o.n);
    o.qx = (a.qx + b.qx) + dx * dx * w;
    o.qy = (a.qy + b.qy) + dy * dy * w;
    o.rmax = fmax(a.rmax, b.rmax);
    return o;
}

__global__ __launch_bounds__(kBlock) void k_ft_stats_reduce(const int32_t* __restrict__ tile_cnt, const double* __restrict__ tile_mx,
                                                            const double* __restrict__ tile_my, const double* __restrict__ tile_m2x,
                                                            const double* __restrict__ tile_m2y, const double* __restrict__ tile_rmax,
                                                            int tiles_per_bundle, int64_t* __restrict__ count, double* __restrict__ rms)
{
    __shared__ Moments s_m[kBlock];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)b * tiles_per_bundle;
    // contiguous chunk of tiles per thread, merged in tile order, then a fixed tree over the threads
    const int per = (tiles_per_bundle + kBlock - 1) / kBlock;
    Moments acc = {0.0, 0.0, 0.0, 0.0, 0.0, -1.0};
    for (int q = 0; q < per; ++q) {
        const int t = tid * per + q;
        if (t < tiles_per_bundle) {
            const Moments m = {(double)tile_cnt[base + t], tile_mx[base + t], tile_my[base + t], tile_m2x[base + t],
                               tile_m2y[base + t], tile_rmax[base + t]};
            acc = chan_merge(acc, m);
        }
    }
    s_m[tid] = acc;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        if ((tid & (2 * off - 1)) == 0 && tid + off < kBlock) s_m[tid] = chan_merge(s_m[tid], s_m[tid + off]);
        __syncthreads();
    }
    if (tid == 0) {
        const Moments m = s_m[0];
        count[b] = 2 * (int64_t)m.n;
        rms[b] = m.n > 0.0 ? sqrt(((m.qx + m.n * m.mx * m.mx) + m.qy) / m.n) : __builtin_nan("");
    }
}

// The same merge for bundles of at most 64 partials (BASELINE config 5: 2 x 10^4 bundles of 32): ONE WAVE per bundle, the
// tree of k_ft_stats_reduce — partial l pairs with l + 1, then l + 2, l + 4, ... in index order; an empty slot is the identity of
// chan_merge — walked with shuffles instead of LDS and barriers: the same merges in the same order, the same bits, a fifth of
// the time (78 -> 16 us there).
__global__ __launch_bounds__(kBlock) void k_ft_stats_reduce_wave(const int32_t* __restrict__ tile_cnt, const double* __restrict__ tile_mx,
                                                                 const double* __restrict__ tile_my, const double* __restrict__ tile_m2x,
                                                                 const double* __restrict__ tile_m2y, const double* __restrict__ tile_rmax,
                                                                 int parts, int nb, int64_t* __restrict__ count, double* __restrict__ rms)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (b >= nb) return;                                           // wave-uniform
    const int64_t base = (int64_t)b * parts;
    Moments m = {0.0, 0.0, 0.0, 0.0, 0.0, -1.0};
    if (lane < parts)
        m = Moments{(double)tile_cnt[base + lane], tile_mx[base + lane], tile_my[base + lane], tile_m2x[base + lane], tile_m2y[base + lane],
                    tile_rmax[base + lane]};
    for (int off = 1; off < 64; off <<= 1) {
        const Moments o = {__shfl_down(m.n, off), __shfl_down(m.mx, off), __shfl_down(m.my, off), __shfl_down(m.qx, off),
                           __shfl_down(m.qy, off), __shfl_down(m.rmax, off)};
        if ((lane & (2 * off - 1)) == 0) m = chan_merge(m, o);    // (lanes past 64 - off read their own value: never merged)
    }
    if (lane == 0) {
        count[b] = 2 * (int64_t)m.n;
        rms[b] = m.n > 0.0 ? sqrt(((m.qx + m.n * m.mx * m.mx) + m.qy) / m.n) : __builtin_nan("");
    }
}

// kPlaceTiles consecutive tiles of one bundle per workgroup, their headers fetched together (one round trip for the group).
#ifndef ORT_PLACE_TILES
#define ORT_PLACE_TILES 2
#endif
constexpr int kPlaceTiles = ORT_PLACE_TILES;

template <typename T>
__global__ __launch_bounds__(kBlock) void k_ft_place(const T* __restrict__ w_ex, const T* __restrict__ w_ey,
                                                     const T* __restrict__ w_r, const T* __restrict__ w_th,
                                                     int64_t rpb, int tiles_per_bundle,
                                                     const int32_t* __restrict__ tile_cnt, const int64_t* __restrict__ tile_off,
                                                     const FtBundleAgg* __restrict__ agg,
                                                     T* __restrict__ ex, T* __restrict__ ey,
                                                     T* __restrict__ rho, T* __restrict__ theta,
                                                     double* __restrict__ tile_sq)
{
    __shared__ double s_wsq[kPlaceTiles][kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int groups = (tiles_per_bundle + kPlaceTiles - 1) / kPlaceTiles;
    const int b = blockIdx.x / groups, t0 = (blockIdx.x - b * groups) * kPlaceTiles;
    const FtBundleAgg a = agg[b];
    int c[kPlaceTiles]; int64_t off[kPlaceTiles];
#pragma unroll
    for (int q = 0; q < kPlaceTiles; ++q) {
        const bool in = t0 + q < tiles_per_bundle;
        const int bid = b * tiles_per_bundle + (in ? t0 + q : t0);
        c[q] = in ? tile_cnt[bid] : 0; off[q] = tile_off[bid];
    }
#pragma unroll
    for (int q = 0; q < kPlaceTiles; ++q) {
        // (every load of both halves ahead of the stores: the same sums in the same order as ft_place_tile, one dependent round
        // trip per tile less — 583 -> 576 us on BASELINE config 3, profiles/r04_ab_place_loads_first.log)
        double sq = ft_place_tile_pub<T, false>(b * tiles_per_bundle + t0 + q, c[q], (int64_t)b * 2 * rpb + off[q], a, w_ex, w_ey, w_r, w_th,
                                                ex, ey, rho, theta, tid, []() {});
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o);
        if (lane == 0) s_wsq[q][wave] = sq;
    }
    __syncthreads();
    if (tid < kPlaceTiles && t0 + tid < tiles_per_bundle) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += s_wsq[tid][w];
        tile_sq[b * tiles_per_bundle + t0 + tid] = t;
    }
}

// ------------------------------------------------------------------------------------
// full_trace, stage C of the FT_LOOKBACK route: survivors only.  The first half (written by the trace kernel) is read once
// (16-byte loads: chunks start on multiples of kTile inside the slab); rho is normalised in place (r ./ maximum(r), :142),
// the mirrored half [-ex; ey; rho; pi - theta] goes to offset m (:139-144) through LDS as aligned 16-byte stores
// (stream_out), and the squared deviations about the centroid are summed per chunk (two-pass sigma, :169-173).
// Grid: nb * chunks_per_bundle blocks of kTile entries; chunks beyond the bundle's m survivors leave at once.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ft_mirror(int64_t rpb, int chunks_per_bundle,
                                                      const FtBundleAgg* __restrict__ agg,
                                                      T* __restrict__ ex, T* __restrict__ ey,
                                                      T* __restrict__ rho, T* __restrict__ theta,
                                                      double* __restrict__ chunk_sq, const int* __restrict__ ft_err)
{
    constexpr int V = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    __shared__ __attribute__((aligned(16))) T s_v[4][kTile];
    __shared__ double s_wsq[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / chunks_per_bundle;
    const int chunk = blockIdx.x - b * chunks_per_bundle;
    const FtBundleAgg a = agg[b];
    const int64_t j0 = (int64_t)chunk * kTile;
    // a faulted look-back left counts that mean nothing: touch no output (k_ft_finalize reports it)
    if (j0 >= a.m || a.m > rpb || *ft_err) { if (tid == 0) chunk_sq[blockIdx.x] = 0.0; return; }
    const int c = (int)min((int64_t)kTile, a.m - j0);
    const int64_t o0 = (int64_t)b * 2 * rpb + j0;                                 // 2 rpb and j0 are even, multiples of 4 for kTile
    const bool al = ((reinterpret_cast<uintptr_t>(ex + o0) | reinterpret_cast<uintptr_t>(ey + o0) |
                      reinterpret_cast<uintptr_t>(rho + o0) | reinterpret_cast<uintptr_t>(theta + o0)) & 15) == 0;
    double sq = 0.0;
    for (int j = tid * V; j < c; j += kBlock * V) {
        vec_t vx, vy, vr, vt;
        if (al && j0 + j + V <= 2 * rpb) {                                       // whole vectors stay inside the slab
            vx = *reinterpret_cast<const vec_t*>(ex + o0 + j); vy = *reinterpret_cast<const vec_t*>(ey + o0 + j);
            vr = *reinterpret_cast<const vec_t*>(rho + o0 + j); vt = *reinterpret_cast<const vec_t*>(theta + o0 + j);
        } else {
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const bool in = j + q < c;
                vx[q] = in ? ex[o0 + j + q] : T(0); vy[q] = in ? ey[o0 + j + q] : T(0);
                vr[q] = in ? rho[o0 + j + q] : T(0); vt[q] = in ? theta[o0 + j + q] : T(0);
            }
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            vr[q] = vr[q] / (T)a.rmax;                                            // :142
            if (j + q < c) {
                const double dx1 = (double)vx[q] - a.mux, dx2 = -(double)vx[q] - a.mux;
                const double dy = (double)vy[q] - a.muy;
                sq += (dx1 * dx1 + dx2 * dx2) + (dy * dy + dy * dy);
            }
        }
        *reinterpret_cast<vec_t*>(&s_v[0][j]) = vx; *reinterpret_cast<vec_t*>(&s_v[1][j]) = vy;
        *reinterpret_cast<vec_t*>(&s_v[2][j]) = vr; *reinterpret_cast<vec_t*>(&s_v[3][j]) = vt;
    }
    __syncthreads();
    auto same = [](T v) { return v; };
    stream_out<T>(rho + o0, s_v[2], c, tid, same);                                                            // rho in place
    stream_out<T>(ex + o0 + a.m, s_v[0], c, tid, [](T v) { return -v; });                                    // :141
    stream_out<T>(ey + o0 + a.m, s_v[1], c, tid, same);                                                       // :140
    stream_out<T>(rho + o0 + a.m, s_v[2], c, tid, same);                                                      // :143
    stream_out<T>(theta + o0 + a.m, s_v[3], c, tid, [](T v) { return (T)3.141592653589793 - v; });           // :144
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off);
    if (lane == 0) s_wsq[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += s_wsq[w];
        chunk_sq[blockIdx.x] = t;
    }
}

// full_trace, stage D: sigma per bundle (PupilSampling.jl:169-173).
// ft_err (look-back route, else null): a faulted look-back leaves count = -1, rms = NaN — what a device-pointer caller sees.
// bundle b, by kBlock threads tid = 0 .. kBlock-1 of a workgroup (also the last stage of k_ft_small_finish); s_r: kBlock doubles of LDS.
template <int PARTS = 1>   // PARTS = kBlock / 64: the fused launch left a tile's sum as its waves' partials — folded here in k_ft_place's order
__device__ __forceinline__ void ft_finalize_body(int b, const double* __restrict__ tile_sq, int tiles_per_bundle,
                                                 const FtBundleAgg* __restrict__ agg,
                                                 int64_t* __restrict__ count, double* __restrict__ rms,
                                                 const int* __restrict__ ft_err, double* __restrict__ s_r, int tid, bool active)
{
    double acc = 0.0;
    if (active) {
        for (int t = tid; t < tiles_per_bundle; t += kBlock) {
            const double* q = tile_sq + ((int64_t)b * tiles_per_bundle + t) * PARTS;
            double ts = 0.0;
            for (int w = 0; w < PARTS; ++w) ts += q[w];
            acc += ts;
        }
        s_r[tid] = acc;
    }
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (active && tid < off) s_r[tid] += s_r[tid + off];
        __syncthreads();
    }
    if (active && tid == 0) {
        const int64_t m = agg[b].m;
        const bool fault = ft_err && *ft_err;
        count[b] = fault ? -1 : 2 * m;
        rms[b] = (m && !fault) ? sqrt(s_r[0] / (double)(2 * m)) : __builtin_nan("");
    }
}

template <int PARTS = 1>
__global__ __launch_bounds__(kBlock) void k_ft_finalize(const double* __restrict__ tile_sq, int tiles_per_bundle,
                                                        const FtBundleAgg* __restrict__ agg,
                                                        int64_t* __restrict__ count, double* __restrict__ rms,
                                                        const int* __restrict__ ft_err)
{
    __shared__ double s_r[kBlock];
    ft_finalize_body<PARTS>(blockIdx.x, tile_sq, tiles_per_bundle, agg, count, rms, ft_err, s_r, threadIdx.x, true);
}

// full_trace of SMALL bundles (a few tiles each: the reference's own call is 4): stages B, C and D — tile offsets and bundle
// aggregates, placement of both halves, sigma — by ONE workgroup per bundle in one launch, through the bodies of
// k_ft_scan, k_ft_place and k_ft_finalize (same operations in the same order: same numbers).  Three dependent launches
// of a few microseconds of work each cost more in launch boundaries than in work.  The workgroup holds kFinishGroups
// groups of kBlock threads: the first runs the scan and sigma stages, each places one tile at a time (their global
// round trips overlap instead of queueing).
constexpr int kFinishGroups = 4;
template <typename T>
__global__ __launch_bounds__(kBlock * kFinishGroups) void k_ft_small_finish(const T* __restrict__ w_ex, const T* __restrict__ w_ey,
                                                            const T* __restrict__ w_r, const T* __restrict__ w_th,
                                                            int64_t rpb, int tiles_per_bundle,
                                                            const int32_t* __restrict__ tile_cnt, const double* __restrict__ tile_sx,
                                                            const double* __restrict__ tile_sy, const double* __restrict__ tile_rmax,
                                                            int64_t* __restrict__ tile_off, FtBundleAgg* __restrict__ agg,
                                                            T* __restrict__ ex, T* __restrict__ ey, T* __restrict__ rho, T* __restrict__ theta,
                                                            double* __restrict__ tile_sq, int64_t* __restrict__ count, double* __restrict__ rms)
{
    __shared__ FtScanShared ss;
    __shared__ FtPlaceShared<T> ps[kFinishGroups];
    __shared__ double s_r[kBlock];
    const int b = blockIdx.x, tid = threadIdx.x & (kBlock - 1), grp = threadIdx.x / kBlock;
    ORT_PHASE(10);
    ft_scan_body(b, tile_cnt, tile_sx, tile_sy, tile_rmax, tiles_per_bundle, tile_off, agg, ss, tid, grp == 0);
    __syncthreads();                                             // tile_off, agg[b]: written above, read below by this workgroup
    ORT_PHASE(11);
    for (int tile0 = 0; tile0 < tiles_per_bundle; tile0 += kFinishGroups) {
        const int tile = tile0 + grp;
        ft_place_body<T>(b * tiles_per_bundle + tile, w_ex, w_ey, w_r, w_th, rpb, tiles_per_bundle, tile_cnt, tile_off, agg,
                         ex, ey, rho, theta, tile_sq, ps[grp], tid, tile < tiles_per_bundle);
        __syncthreads();                                         // ps[].wsq is reused; tile_sq is read below
    }
    ORT_PHASE(12);
    ft_finalize_body(b, tile_sq, tiles_per_bundle, agg, count, rms, nullptr, s_r, tid, grp == 0);
    ORT_PHASE(13);
}

// ------------------------------------------------------------------------------------
// Meridional real-ray trace, src/RayTracing.jl:145-169 (sag :75-88, tilt :98).
// One thread per ray; history [rows][ld].  Trig through ocml (tan/asin/atan/sin/cos).
// ------------------------------------------------------------------------------------
struct MerSurf {   // row i+1 of the prescription as seen by loop iteration i
    double t, R, sgn, K, n1, n2;
    double eta, invR;      // n1 / n2 and 1 / R (0 on a plane), once per row: the aiming traces (mer_plain_trace_to) run hundreds of times
    int32_t finite, ncoef;
};

// One loop iteration of src/RayTracing.jl:151-167.  Returns ts[i] (after :160).
__device__ __forceinline__ double mer_step(const MerSurf& s, const double* __restrict__ c, int layout_mode,
                                           double& y, double& U, double& sprev, bool& domain)
{
    const double tcur = s.t - sprev;                              // ts[i] after :161
    const double tU = ::tan(U);
    y = y + tU * tcur;                                            // :152
    double sg;
    if (s.finite) {                                               // :76
        const double beta = s.R - y * tU;                         // :77
        const double y2 = y * y;                                  // :78
        const double sec = 1.0 / ::cos(U);
        const double D = beta * beta - y2 * (sec * sec + s.K);    // :79
        sg = y2 / (beta + s.sgn * __builtin_sqrt(D));             // :81
        sg = sg + (s.ncoef > 0 ? poly_eval_loop<double>(c, s.ncoef, y) : 0.0);
        sg = (D >= 0.0) ? sg : __builtin_nan("");                 // :80,83
    } else sg = 0.0;                                              // :86
    y = y + sg * tU;                                              // :158
    sprev = sg;
    double theta;
    if (s.K == 0.0 && !layout_mode && s.ncoef == 0) {
        const double q = y / s.R;                                 // tilt(y, R) = y / R (:101)
        domain = domain || (fabs(q) > 1.0);                       // Base.asin throws DomainError there (:162); NaN here + the flag
        theta = ::asin(q);                                        // :162
    } else {
        double tl = s.sgn * y / __builtin_sqrt(s.R * s.R - y * y * (1.0 + s.K));   // :98
        tl = tl + (s.ncoef > 0 ? poly_deriv_loop<double>(c, s.ncoef, y) : 0.0);
        theta = ::atan(tl);
    }
    const double sin_ip = s.n1 * ::sin(U + theta) / s.n2;         // :163
    U = (fabs(sin_ip) <= 1.0) ? ::asin(sin_ip) - theta : __builtin_nan("");   // :164
    return tcur + sg;                                             // ts[i] += s (:160)
}

__global__ __launch_bounds__(kBlock) void k_trace_meridional(const MerSurf* __restrict__ surf, const double* __restrict__ coefs,
                                                             int S, int ncoef, int layout_mode, double t_last,
                                                             int64_t nrays, const double* __restrict__ y_in,
                                                             const double* __restrict__ U_in,
                                                             double* __restrict__ y_out, double* __restrict__ U_out,
                                                             double* __restrict__ ts_out, int64_t ld,
                                                             unsigned long long* __restrict__ dom)
{
    __shared__ MerSurf s_s[kMaxRows];
    __shared__ double s_c[kMaxRows * kMaxCoef];
    for (int w = threadIdx.x; w < S; w += kBlock) s_s[w] = surf[w];
    if (coefs) for (int w = threadIdx.x; w < S * ncoef; w += kBlock) s_c[w] = coefs[ncoef + w];
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= nrays) return;
    double y = y_in[g], U = U_in[g];
    y_out[g] = y; U_out[g] = U;                                   // :150
    double sprev = 0.0;
    for (int i = 0; i < S; ++i) {
        bool domain = false;
        const double tsi = mer_step(s_s[i], s_c + i * ncoef, layout_mode, y, U, sprev, domain);
        // the reference's raytrace(...) of THIS ray would throw a DomainError here (asin of |y / R| > 1, :162): the
        // first such (ray, surface) of the launch is recorded for the caller, the ray continues as NaN
        if (domain && dom) { atomicMin(dom, ((unsigned long long)g << 8) | (unsigned)(i + 1)); atomicAdd(dom + 1, 1ull); }
        if (ts_out) ts_out[(int64_t)i * ld + g] = tsi;
        y_out[(int64_t)(i + 1) * ld + g] = y;                     // :165
        U_out[(int64_t)(i + 1) * ld + g] = U;                     // :166
    }
    if (ts_out) ts_out[(int64_t)S * ld + g] = t_last - sprev;     // ts[end] -= s (:161)
}

// ------------------------------------------------------------------------------------
// Batched ray aiming (SURVEY §8f "next #1"): four lanes per (system, field) run the Newton
// drivers of src/RayTracing.jl:223-240 (real marginal), :265-296 (real chief, on the reversed
// system) and the edge-ray search of src/PupilSampling.jl:67-83 (restated as the same
// FD-Newton, see api._trace_edge_rays), and emits the aiming scalars of
// src/PupilSampling.jl:94-103.  Each trace is mer_step over the system's table.
// ------------------------------------------------------------------------------------
struct AimIn {
    int32_t system;        // index into the forward and the reversed table batch
    int32_t stop;          // system.stop
    int32_t layout_fwd;    // Q16 flags of the forward / reversed prescriptions
    int32_t layout_rev;
    double H;              // |H| <= 1
    double y_marg;         // system.marginal.y[1]          :225
    double a_stop;         // system.a[stop]                :227
    double chief_y_end;    // system.chief.y[end]           :279
    double chief_u_end;    // system.chief.u[end]           :280
    double f;              // system.f  (h' = u f, PupilSampling.jl:103)
    double atol;           // sqrt(eps())
};
struct AimOut {
    double U, y1, y2, y_EP, hprime, EP_t, Ubar;
    double XP_t;           // real_chief.z[end] - real_chief.z[end-1]   (RayTracing.jl:294; TSA, SeidelAberrations.jl:120)
    int32_t iters;         // Newton iterations spent (all loops)
    int32_t ok;            // 1 = every loop converged
};

struct MerEnd { double y_stop, y_last, U_last, z_last, z_prev, y_first, s_last;
                double sU, cU; };   // plain systems (mer_plain_trace_to): sin / cos of the last angle; U_last is taken from them once, after the loop

__device__ inline MerEnd mer_trace_to(const MerSurf* __restrict__ surf, const double* __restrict__ coefs, int S, int ncoef,
                                      int layout_mode, double t_last, double y, double U, int stop_idx)
{
    MerEnd e;
    e.y_stop = __builtin_nan(""); e.y_first = y;
    double sprev = 0.0, z = 0.0, zp = 0.0;
    bool domain = false;                                         // (aiming traces: a NaN loss ends the Newton loop)
    for (int i = 0; i < S; ++i) {
        const double tsi = mer_step(surf[i], coefs ? coefs + (int64_t)(i + 1) * ncoef : nullptr, layout_mode, y, U, sprev, domain);
        zp = z; z = (i == 0) ? tsi : z + tsi;                    // cumsum(ts)  (Types.jl:61-63)
        if (i + 1 == stop_idx) e.y_stop = y;                     // ray.y[begin+stop]
        if (i == 0) e.y_first = y;                               // ray.y[2]
    }
    e.s_last = sprev;                                            // sag at the last surface
    zp = z; z = z + (t_last - sprev);                            // last ts entry
    e.y_last = y; e.U_last = U; e.z_last = z; e.z_prev = zp;
    return e;
}

// The same trace for a PLAIN prescription (every row a sphere or a plane, K = 0, p = zero) without a single
// trigonometric call: the ray carries (sin U, cos U); a sphere is met in centre form — with Q = P - C the path length is
// d = -b - sign(R) sqrt(b^2 - |Q0|^2 + R^2), b = Q0 . k —, the normal is -Q / R and Snell's law is applied to the
// direction vector, k' = eta k + (cos I' - eta cos I) n.  Algebraically the reference's loop (RayTracing.jl:151-167:
// y += tan U t; sag; theta = asin(y / R); U' = asin(n sin(U + theta) / n') - theta), evaluated with two square roots per
// surface where that one takes five libm calls.  The aiming loops are SERIAL chains of such traces (:223-296,
// PupilSampling.jl:67-83) and stop at |loss| <= sqrt(eps) (RayTracing.jl:1): their latency, not their last bits, is what
// a single full_trace call feels (config 1: 77 us of 140 us device time were these chains).  Rows with a conic constant
// or a polynomial keep mer_step.  A miss (disc < 0) or total internal reflection gives NaN from there on, as there (:83,164).
// sqrt to ~1 ulp from the hardware seed and two corrections (the residual form of ieee_sqrt without its last step);
// 0 -> 0, negative -> NaN.  The loops it feeds stop at 1.5e-8.
__device__ __forceinline__ double aim_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x + 1e-300);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}

// sin and cos of a launch angle of the aiming loops: |w| <= pi / 4 (field angles) by their Taylor series to x^17 / x^16
// (remainders below 2^-60 there), larger angles through libm.  The chief-ray loop takes one pair per iteration (:280-285).
__device__ __forceinline__ void aim_sincos(double w, double& sw, double& cw)
{
    if (__builtin_fabs(w) > 0.7853981633974483) { sw = ::sin(w); cw = ::cos(w); return; }
    const double x2 = w * w;
    double s = 2.8114572543455206e-15;                                // 1 / 17!
    s = __builtin_fma(s, x2, -7.6471637318198164e-13);                // -1 / 15!
    s = __builtin_fma(s, x2, 1.6059043836821613e-10);                 // 1 / 13!
    s = __builtin_fma(s, x2, -2.5052108385441720e-08);                // -1 / 11!
    s = __builtin_fma(s, x2, 2.7557319223985893e-06);                 // 1 / 9!
    s = __builtin_fma(s, x2, -1.9841269841269841e-04);                // -1 / 7!
    s = __builtin_fma(s, x2, 8.3333333333333332e-03);                 // 1 / 5!
    s = __builtin_fma(s, x2, -1.6666666666666666e-01);                // -1 / 3!
    sw = __builtin_fma(s * x2, w, w);
    double c = 4.7794773323873853e-14;                                // 1 / 16!
    c = __builtin_fma(c, x2, -1.1470745597729725e-11);                // -1 / 14!
    c = __builtin_fma(c, x2, 2.0876756987868100e-09);                 // 1 / 12!
    c = __builtin_fma(c, x2, -2.7557319223985888e-07);                // -1 / 10!
    c = __builtin_fma(c, x2, 2.4801587301587302e-05);                 // 1 / 8!
    c = __builtin_fma(c, x2, -1.3888888888888889e-03);                // -1 / 6!
    c = __builtin_fma(c, x2, 4.1666666666666664e-02);                 // 1 / 4!
    c = __builtin_fma(c, x2, -0.5);
    cw = __builtin_fma(c, x2, 1.0);
}

__device__ inline MerEnd mer_plain_trace_to(const MerSurf* __restrict__ surf, int S, double t_last, double y, double sU, double cU,
                                            int stop_idx)
{
    MerEnd e;
    e.y_stop = __builtin_nan(""); e.y_first = y; e.U_last = 0.0;
    double sprev = 0.0, z = 0.0, zp = 0.0;
    MerSurf nx = surf[0];
    for (int i = 0; i < S; ++i) {
        const MerSurf s = nx;
        nx = surf[i + 1 < S ? i + 1 : i];                             // the next row's record is fetched under this row's arithmetic
        const double tcur = s.t - sprev;                              // ts[i] after :161
        // row constants (independent of the ray: they fill the issue slots the ray's dependent chain leaves empty)
        const double eta = s.eta, eta2 = eta * eta, ome2 = 1.0 - eta2;
        double sg;
        if (s.finite) {
            const double R2 = s.R * s.R, e2c2 = eta2 * (s.invR * s.invR), ec = eta * __builtin_fabs(s.invR);
            const double Qz0 = sprev - (s.t + s.R);                   // the ray point relative to the centre of curvature
            const double b = __builtin_fma(Qz0, cU, y * sU);
            const double disc = __builtin_fma(b, b, R2 - __builtin_fma(Qz0, Qz0, y * y));
            // cos^2 I = disc / R^2: the second root's radicand does not wait for the first root
            const double D2 = __builtin_fma(e2c2, disc, ome2);
            const double sq = aim_sqrt(disc);                         // NaN: the ray misses (:83)
            const double cr = aim_sqrt(D2);                           // D2 < 0: total internal reflection -> NaN (:164)
            const double d = -__builtin_fma(s.sgn, sq, b);
            y = __builtin_fma(d, sU, y);
            const double Qz = __builtin_fma(d, cU, Qz0);
            sg = Qz + s.R;                                            // sag (:155)
            const double gc = __builtin_fma(-ec, sq, cr) * s.invR;    // (cos I' - eta cos I) / R,  cos I = |Q . k| / |R|
            sU = __builtin_fma(-gc, y, eta * sU);
            cU = __builtin_fma(-gc, Qz, eta * cU);
        } else {
            y = __builtin_fma(sU * fast_rcp(cU), tcur, y);            // :152
            sg = 0.0;                                                 // :86
            const double D2 = __builtin_fma(eta2 * cU, cU, ome2);
            sU = eta * sU;                                            // normal (0, 1): only the axial component refracts
            cU = aim_sqrt(D2);
        }
        sprev = sg;
        const double tsi = tcur + sg;                                 // ts[i] += s (:160)
        zp = z; z = (i == 0) ? tsi : z + tsi;                         // cumsum(ts)  (Types.jl:61-63)
        if (i + 1 == stop_idx) e.y_stop = y;                          // ray.y[begin+stop]
        if (i == 0) e.y_first = y;                                    // ray.y[2]
    }
    e.s_last = sprev;
    zp = z; z = z + (t_last - sprev);
    e.y_last = y; e.z_last = z; e.z_prev = zp; e.sU = sU; e.cU = cU;
    return e;
}

// FD-Newton on one scalar `v` shared by a PAIR of lanes: the even lane traces at v, the odd lane at v + eps
// (the reference's forward difference, RayTracing.jl:230,283), both apply the same update — the serial
// loop's arithmetic, two traces per round side by side.  `trace(v)` returns the MerEnd of a ray launched
// with the pair's free variable set to v; loss = y_stop - target.  The two pairs of a 4-lane group run
// different problems concurrently; the loop is group-uniform so the shuffles always see live lanes.
// inside_edge (the edge rays of the pupil grid) — a rule FITTED to one published figure, not a restatement of
// Optim.BFGS: a search that ends OUTSIDE its target |y_stop| = a_stop takes one more Newton step, to atol inside, so
// the grid's two edge rays pass the stop filter r > a_stop (PupilSampling.jl:132).  The reference's Tessar spot size
// (docs figure, 0.11975) is reproduced with them and is 0.64 % off without; on any other system which side the
// reference's BFGS ends on is unpinned, so the survivor count may differ from it by the x = 0 rays of the first and
// last pupil row (tests/test_gpu_parity.py::test_edge_rule_only_moves_the_two_edge_rays).
template <typename F>
__device__ __forceinline__ void pair_newton(F&& trace, double& v, double target, double atol, int cap, bool cap_fails,
                                            int pairbase, int role, MerEnd& e, double& loss, int& iters, int& ok,
                                            bool inside_edge = false)
{
    const double eps = 1.4901161193847656e-08;                   // const ϵ = sqrt(eps()), RayTracing.jl:1
    const bool pert = role & 1;
    bool conv = false;
    int it = 0;
    e.y_stop = e.y_last = e.U_last = e.z_last = e.z_prev = e.y_first = e.s_last = 0.0;
    while (true) {
        ORT_PHASE(4);
        if (!conv) e = trace(pert ? v + eps : v);
        ORT_PHASE(9);
        const double L = e.y_stop - target;
        const double L0 = __shfl(L, pairbase, 4), L1 = __shfl(L, pairbase + 1, 4);
        if (!conv) {
            loss = L0;
            if (!(fabs(L0) > atol)) {                            // NaN ends the loop like the reference (:229,282)
                conv = true;
                if (inside_edge && L0 * target > 0.0) {
                    const double dv = (L0 + copysign(atol, target)) * eps / (L1 - L0);
                    if (fabs(dv) <= 1e300) v -= dv;              // (a flat loss gives inf / NaN: keep the end point)
                }
            }
            else if (it >= cap) { conv = true; if (cap_fails) ok = 0; }
            else { v -= L0 * eps / (L1 - L0); ++it; }            // :231,284
        }
        const int other = __shfl((int)conv, role ^ 2, 4);
        if (conv && other) break;
    }
    iters += it;
}

// Four lanes per (system, field): lanes 0-1 = chief pair, lanes 2-3 = marginal pair, then lanes 0-1 / 2-3 =
// the two edge rays.  Same operations on the same values as the serial drivers, a quarter of the latency.
// `role` = lane index inside its group of four (the group's lanes must be consecutive lanes of one wave); F / Rv = the
// request's forward / reversed tables, cF / cR their coefficient rows (or null).  Returns the result on every lane.
__device__ inline AimOut aim_group(const AimIn& a, int role, const MerSurf* __restrict__ F, const double* __restrict__ cF, double tlF,
                                   const MerSurf* __restrict__ Rv, const double* __restrict__ cR, double tlR, int S, int ncoef,
                                   bool edge_inside = true, bool fast = false)
{
    const int pair = role >> 1, pairbase = pair * 2;
    const int rows = S + 1;
    // ORT_FAST_MATH and a plain prescription (spheres and planes only, both ways): the trig-free trace (mer_plain_trace_to).
    // The default policy keeps the reference's own sequence (mer_trace_to: tan / asin / atan, RayTracing.jl:151-167) — the
    // loops stop at |residual| <= sqrt(eps), so two forms of the same function may end on iterates a few 1e-9 apart
    bool plain = fast;
    for (int i = 0; i < S; ++i) plain = plain && F[i].K == 0.0 && F[i].ncoef == 0 && Rv[i].K == 0.0 && Rv[i].ncoef == 0;
    int iters = 0, ok = 1;
    // ---- phase 1: real chief ray on the reversed system (RayTracing.jl:278-286) | real marginal ray (:225-233)
    const int stop_rev = rows - a.stop;                          // :278
    const double ybp = a.chief_y_end;                            // :279
    double v = pair == 0 ? -a.chief_u_end : a.y_marg;            // :280 | :225
    MerEnd e; double loss = 0.0;
    double sv = 0.0, cv = 1.0;                                   // sin / cos of the chief pair's last launch angle (plain path)
    // ONE call for both pairs (lane-selected table and launch data): two calls would run one after the other
    const bool chief = pair == 0;
    const MerSurf* tab = chief ? Rv : F;
    const double* ctab = chief ? cR : cF;
    const double tl1 = chief ? tlR : tlF;
    const int stop1 = chief ? stop_rev : a.stop, lay1 = chief ? a.layout_rev : a.layout_fwd;
    pair_newton([&](double w) {
                    if (plain) {
                        double s1, c1;
                        aim_sincos(chief ? w : 0.0, s1, c1);              // the marginal ray is launched along the axis: (0, 1) exactly
                        if (chief) { sv = s1; cv = c1; }
                        return mer_plain_trace_to(tab, S, tl1, chief ? ybp : w, s1, c1, stop1);
                    }
                    return mer_trace_to(tab, ctab, S, ncoef, lay1, tl1, chief ? ybp : w, chief ? w : 0.0, stop1);
                }, v, pair == 0 ? 0.0 : a.a_stop, a.atol, 200, true, pairbase, role, e, loss, iters, ok);
    ORT_PHASE(3);
    // results live on the base lanes: chief on lane 0, marginal on lane 2
    double ub1, t_ub1, t_v0;                                     // ū[1], tan(ū[1]), tan(-launch angle of the reversed trace)
    if (plain) {                                                 // the tangents are quotients of what the trace carries; the angle
        const double sl = __shfl(e.sU, 0, 4), cl = __shfl(e.cU, 0, 4);   // itself is needed once (U = H ū[1] below)
        ub1 = -::atan2(sl, cl);                                  // ū[1] = -reverse(ray.u)[1]          :289
        t_ub1 = -sl / cl;
        t_v0 = -__shfl(sv, 0, 4) / __shfl(cv, 0, 4);
    } else {
        ub1 = -__shfl(e.U_last, 0, 4);
        t_ub1 = ::tan(ub1);
        t_v0 = ::tan(-__shfl(v, 0, 4));
    }
    const double yb2 = __shfl(e.y_last, 0, 4);                   // ȳ[2] = reverse(ray.y)[1]           :287
    const double z2 = __shfl(e.z_last, 0, 4) - __shfl(e.z_prev, 0, 4);   // z[2] = ray.z[end] - ray.z[end-1]   :292
    const double EP_t = -yb2 / t_ub1 + z2;                       // :293
    // z[end] - z[end-1] = -ȳ[end-1] / tan(ū[end-1])  (:294):  ȳ[end-1] = ray.y[2] (the last real surface),
    // ū[end-1] = -ray.u[1] (the converged launch angle of the reversed trace)
    const double XP_t = -__shfl(e.y_first, 0, 4) / t_v0;
    const double y_EP = fabs(__shfl(v, 2, 4));                   // PupilSampling.jl:98 (real_marginal.y[1])
    // ---- phase 2: field, edge rays (PupilSampling.jl:94-100)
    const double U = a.H * ub1;                                  // :96
    double sinU, cosU;
    aim_sincos(U, sinU, cosU);
    const double u = plain ? sinU / cosU : ::tan(U);             // :97
    const double astop = fabs(a.a_stop);                         // :91
    const double target = pair == 0 ? astop : -astop;
    const double y0 = (pair == 0 ? y_EP : -y_EP) - u * EP_t;     // :99
    double yy = y0;
    int ok2 = 1;
    ORT_PHASE(5);
    // only ray.y[begin+stop] feeds the loss (PupilSampling.jl:70,76): the traces of this search end at the stop
    pair_newton([&](double w) { return plain ? mer_plain_trace_to(F, a.stop, tlF, w, sinU, cosU, a.stop)
                                             : mer_trace_to(F, cF, a.stop, ncoef, a.layout_fwd, tlF, w, U, a.stop); },
                yy, target, a.atol, 100, false, pairbase, role, e, loss, iters, ok2, edge_inside);
    if (!(fabs(loss) <= 1e300)) yy = y0;                         // isnan(Δ) ? Inf : Δ keeps the start point (:72,78)
    AimOut o;
    o.U = U; o.y1 = __shfl(yy, 0, 4); o.y2 = __shfl(yy, 2, 4); o.y_EP = y_EP; o.hprime = u * a.f; o.EP_t = EP_t; o.Ubar = ub1; o.XP_t = XP_t;
    o.iters = __shfl(iters, 0, 4) + __shfl(iters, 2, 4);         // lanes 0 and 2 carry their pairs' counts
    o.ok = __shfl(ok, 0, 4) & __shfl(ok, 2, 4);
    return o;
}

__global__ __launch_bounds__(64) void k_aim(int n, const AimIn* __restrict__ in,
                                            const MerSurf* __restrict__ fwd, const double* __restrict__ cfwd, const double* __restrict__ tl_fwd,
                                            const MerSurf* __restrict__ rev, const double* __restrict__ crev, const double* __restrict__ tl_rev,
                                            int S, int ncoef, AimOut* __restrict__ out, int edge_inside, int fast)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    const int aim = g >> 2, role = g & 3;
    const bool valid = aim < n;
    const AimIn a = in[valid ? aim : n - 1];                     // tail lanes shadow the last request: uniform shuffles
    const int rows = S + 1;
    const AimOut o = aim_group(a, role, fwd + (int64_t)a.system * S, cfwd ? cfwd + (int64_t)a.system * rows * ncoef : nullptr,
                               tl_fwd[a.system], rev + (int64_t)a.system * S, crev ? crev + (int64_t)a.system * rows * ncoef : nullptr,
                               tl_rev[a.system], S, ncoef, edge_inside != 0, fast != 0);
    if (valid && role == 0) out[aim] = o;
}

// ------------------------------------------------------------------------------------
// Meridional fans (SURVEY §8f "next #4"): the ray set of `TSA` (src/SeidelAberrations.jl:116-137) — and, in
// descending order with another back focal distance, of the caustic plot (ext/MakieExtension.jl:353-398) — for
// many systems in ONE launch.  One thread per (request, ray): y = range(y_m / k, y_m, k)[i] (or its reverse), U = 0,
// traced with mer_step; then
//     y_XP = ray.y[end] + tan(ray.u[end]) * XP_t                                 (:131; transfer :107-115)
//     eps  = ray.y[end] + tan(ray.u[end]) * (BFD - sag(ray)),  sag(ray) = ray.z[end-1] - ray.z[end]   (:130,132; :91)
// The last ray (y = y_m) is the real marginal ray itself: the reference takes it from trace_marginal_ray (:127-128)
// instead of tracing it again — the same trace of the same launch data, so the same numbers — and measures ITS sag from
// the paraxial vertex (see below).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double dd_range_elem(double a, double b, int n, int i);   // below (pupil axes)

struct FanIn {
    int32_t system;        // index into the system batch
    int32_t layout_mode;   // 1 = the prescription is a Layout{Aspheric} (Q16)
    double y_marg;         // real_marginal.y[1]
    double XP_t;           // real_chief.z[end] - real_chief.z[end-1]
    double BFD;            // paraxial (or marginal, caustic) back focal distance from the last vertex
};

__global__ __launch_bounds__(kBlock) void k_fan(int n, int k_rays, int descending, const FanIn* __restrict__ in,
                                                const MerSurf* __restrict__ surf, const double* __restrict__ coefs,
                                                const double* __restrict__ t_last, int S, int ncoef,
                                                double* __restrict__ y_XP, double* __restrict__ eps)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= (int64_t)n * k_rays) return;
    const int q = (int)(g / k_rays), i = (int)(g - (int64_t)q * k_rays);
    const FanIn f = in[q];
    const double lo = f.y_marg / (double)k_rays;                                  // range(y_m / k, y_m, k)  (:121)
    const double y0 = descending ? dd_range_elem(f.y_marg, lo, k_rays, i) : dd_range_elem(lo, f.y_marg, k_rays, i);
    const MerSurf* F = surf + (int64_t)f.system * S;
    const double* cF = coefs ? coefs + (int64_t)f.system * (S + 1) * ncoef : nullptr;
    const double tl = t_last[f.system];
    const MerEnd e = mer_trace_to(F, cF, S, ncoef, f.layout_mode, tl, y0, 0.0, -1);
    const double tu = ::tan(e.U_last);
    // sag(ray) = ray.z[end-1] - ray.z[end] = -ts[end] = s_last - t[end] for a traced ray (:130,132; RayTracing.jl:91); the
    // marginal ray (y = y_m) takes sag(real, paraxial) = real.z[end-1] - paraxial.z[end-1] = s_last, no thickness (:125-127,
    // RayTracing.jl:93-95).  The two agree when the prescription ends in image space (t[end] = 0), as the reference's do.
    // The caustic set (descending) re-traces the marginal ray like every other (ext/MakieExtension.jl:369-371).
    const bool marginal = !descending && i == k_rays - 1;
    const double sag = marginal ? e.s_last : e.s_last - tl;
    y_XP[g] = e.y_last + tu * f.XP_t;
    eps[g] = e.y_last + tu * (f.BFD - sag);
}

// ------------------------------------------------------------------------------------
// Paraxial y-nu trace, src/RayTracing.jl:127-143 (+ :55-69).  One thread per ray; lens
// table of the ray's lens staged in LDS.  grid.x = lens * blocks_per_lens + chunk.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_trace_paraxial(int k, const double* __restrict__ tau, const double* __restrict__ phi,
                                                           const double* __restrict__ a, int clip,
                                                           int64_t rays_per_lens, int blocks_per_lens,
                                                           const double* __restrict__ y_in, const double* __restrict__ w_in,
                                                           double* __restrict__ rt_y, double* __restrict__ rt_w, int64_t ld)
{
    __shared__ double s_tau[kMaxRows], s_phi[kMaxRows], s_a[kMaxRows];
    const int lens = blockIdx.x / blocks_per_lens;
    const int chunk = blockIdx.x - lens * blocks_per_lens;
    for (int i = threadIdx.x; i < k; i += kBlock) {
        s_tau[i] = tau[(int64_t)lens * k + i];
        s_phi[i] = phi[(int64_t)lens * k + i];
        s_a[i] = a ? a[(int64_t)lens * k + i] : __builtin_inf();
    }
    __syncthreads();
    const int64_t r = (int64_t)chunk * kBlock + threadIdx.x;
    if (r >= rays_per_lens) return;
    const int64_t g = (int64_t)lens * rays_per_lens + r;
    double y = y_in[g], w = w_in[g];
    rt_y[g] = y; rt_w[g] = w;                                     // :132
    bool dead = false;
    for (int i = 0; i < k; ++i) {
        const double tq = s_tau[i];
        const double yp = __builtin_isfinite(tq) ? y + w * tq : y;    // :61-64
        const double wp = w - yp * s_phi[i];                          // :66-69
        y = yp; w = wp;
        if (clip && !dead && (fabs(y) - s_a[i] > 1e-13)) dead = true; // :135-137
        rt_y[(int64_t)(i + 1) * ld + g] = dead ? __builtin_nan("") : y;
        rt_w[(int64_t)(i + 1) * ld + g] = dead ? __builtin_nan("") : w;
    }
}

// ------------------------------------------------------------------------------------
// Pupil axes on the device: row b = [range(y_first, y_last, ny) | range(x_first, x_last, nx)]
// (src/PupilSampling.jl:121-122).  x_i = (a (m - i) + b i) / m, m = n - 1, in double-double
// arithmetic: exact numerator (integer weights, FMA products), quotient carried to ~106 bits and
// EXACT whenever the true value is a short binary fraction, so ties round half-to-even — the same
// algorithm, and the same bits, as the host mirror api.linrange_batch.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void dd_two_sum(double a, double b, double& s, double& e)
{
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}

__device__ __forceinline__ double dd_range_elem(double a, double b, int n, int i)
{
    if (n <= 1 || i <= 0) return a;
    if (i >= n - 1) return b;
    const double m = (double)(n - 1), w2 = (double)i, w1 = m - w2;
    const double p1 = a * w1, e1 = __builtin_fma(a, w1, -p1);
    const double p2 = b * w2, e2 = __builtin_fma(b, w2, -p2);
    double sh, se;
    dd_two_sum(p1, p2, sh, se);
    se = se + (e1 + e2);
    double nh, nl;
    dd_two_sum(sh, se, nh, nl);
    const double q1 = nh / m;
    const double ph = q1 * m, pl = __builtin_fma(q1, m, -ph);
    const double q2 = (((nh - ph) - pl) + nl) / m;
    return q1 + q2;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_make_axes(int nb, int ny, int nx, const double* __restrict__ ends,
                                                      T* __restrict__ axes)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int per = ny + nx;
    if (g >= (int64_t)nb * per) return;
    const int b = (int)(g / per), j = (int)(g - (int64_t)b * per);
    const double* e = ends + (int64_t)b * 4;
    axes[g] = (T)((j < ny) ? dd_range_elem(e[0], e[1], ny, j) : dd_range_elem(e[2], e[3], nx, j - ny));
}

// Launch slopes of the finite-conjugate rule (src/PupilSampling.jl:124-127 then :38-39): per bundle, tan((ybar - y) / z0) for its
// ny pupil rows and tan(-x / z0) for its nx columns — the same expressions, operand for operand, the trace kernel used to
// evaluate per ray.  out : [nb][ny + nx].
template <typename T>
__global__ __launch_bounds__(kBlock) void k_make_slope_axes(int nb, int ny, int nx, const DevBundle<T>* __restrict__ bundles,
                                                            const T* __restrict__ axes, T* __restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int per = ny + nx;
    if (g >= (int64_t)nb * per) return;
    const int b = (int)(g / per), j = (int)(g - (int64_t)b * per);
    const DevBundle<T>& bd = bundles[b];
    out[g] = (j < ny) ? dev_tan((bd.ybar - axes[bd.yoff + j]) / bd.z0) : dev_tan(-axes[bd.xoff + (j - ny)] / bd.z0);
}

// ------------------------------------------------------------------------------------
// Batched first-order solve + Seidel sums (SURVEY §8f "next #3"): one thread per system runs
// Lens(surfaces) (src/RayTracing.jl:38-53), the two paraxial traces and the marginal / chief
// construction of `_solve` (:208-221, :246-263, :302-323) and the third-order sums of
// `aberrations` (src/SeidelAberrations.jl:6-53).  O(rows) work per system: for Monte-Carlo
// tolerance runs (BASELINE config 5) thousands of instances go out as one launch.
// ------------------------------------------------------------------------------------
struct FirstOrderOut {
    double f, EBFD, EFFD, N, FOV, EP_D, EP_t, XP_D, XP_t, H;
    double y_marg, chief_y_end, chief_u_end, nu_end, BFD, PN;
    double W040, W131, W222, W220, W311, W020, W111, W220P;
    int32_t stop, k;
};

enum { SURF_SPHERICAL = 0, SURF_COMA, SURF_ASTIGMATISM, SURF_SAGITTAL, SURF_DISTORTION, SURF_AXIAL, SURF_LATERAL,
       SURF_PETZVAL, SURF_MEDIAL, SURF_TANGENTIAL, SURF_COUNT };   // = ORT_SURF_* of ort.h

// Work arrays of one first-order solve: private (scratch) arrays in the batched kernel, LDS in the one-wave small-problem
// kernel (k_small_prepare), where their latency is on the critical path.
struct FirstOrderWork { double *tau, *phi, *y1, *w1, *y2, *w2, *yc, *wc; };

// solve(surfaces, a, h') of ONE system: Lens, the two paraxial traces, stop, marginal and chief rays, first-order
// properties (everything of FirstOrderOut but the Seidel sums).  R, t, n, a: the system's own columns (global or LDS).
// WAVE = false: one thread does it all (k_first_order: one system per thread).  WAVE = true: the 64 lanes of a
// one-wave workgroup share ONE system whose work arrays live in LDS (k_small_prepare) — the element-wise stages
// (tau / phi, a ./ y, the scaling, the chief arrays: a division or two per element) run one lane per element, the
// recurrences and the scalar tail on lane 0; the same operations on the same values either way, `o` valid on lane 0.
template <bool WAVE>
__device__ __forceinline__ void first_order_paraxial(int lane, int rows, const double* __restrict__ R, const double* __restrict__ t,
                                                     const double* __restrict__ n, const double* __restrict__ a,
                                                     double h, const FirstOrderWork& W, FirstOrderOut& o)
{
    double* tau = W.tau; double* phi = W.phi; double* y1 = W.y1; double* w1 = W.w1; double* y2 = W.y2; double* w2 = W.w2;
    double* yc = W.yc; double* wc = W.wc;
    const int step = WAVE ? 64 : 1;
    const bool lead = !WAVE || lane == 0;
    auto sync = [&]() { if (WAVE) __syncthreads(); };
    // Lens(surfaces)  RayTracing.jl:38-53
    for (int i = lane; i < rows; i += step) {
        const double ti = (i == 0 && !__builtin_isfinite(t[0])) ? 0.0 : t[i];      // :42
        tau[i] = ti / n[i];                                                          // :43
        phi[i] = (i < rows - 1) ? (n[i + 1] - n[i]) / R[i + 1] : 0.0;               // :45,50
    }
    sync();
    const double tl = t[rows - 1];
    const int k = (tl == 0.0 || !__builtin_isfinite(tl)) ? rows - 1 : rows;          // :47-48
    // paraxial traces (1, 0) and (0, 1)  :127-141
    if (lead) {
        double ya = 1.0, wa = 0.0, yb = 0.0, wb = 1.0;
        y1[0] = ya; w1[0] = wa; y2[0] = yb; w2[0] = wb;
        for (int i = 0; i < k; ++i) {
            const bool fin = __builtin_isfinite(tau[i]);
            ya = fin ? ya + wa * tau[i] : ya;  wa = wa - ya * phi[i];
            yb = fin ? yb + wb * tau[i] : yb;  wb = wb - yb * phi[i];
            y1[i + 1] = ya; w1[i + 1] = wa; y2[i + 1] = yb; w2[i + 1] = wb;
        }
    }
    sync();
    o.k = k;
    o.f = -(1.0 / w1[k]);                                                            // :213
    o.EBFD = y1[k] * o.f;                                                            // :214
    // `a` holds rows-1 semi-diameters (ort.h); the reference needs length(a) == k (sv = a ./ y[2:end], :215) and
    // throws a DimensionMismatch otherwise: the host entry points reject k == rows, the bound keeps a device-pointer
    // caller inside its own system's slab
    const int ka = k < rows - 1 ? k : rows - 1;
    for (int i = lane; i < ka; i += step) yc[i] = a[i] / y1[i + 1];                  // sv, parked in the chief array (free until :258)
    sync();
    int stop = 0; double s = yc[0];
    for (int i = 1; i < ka; ++i) { const double sv = yc[i]; if (sv < s) { s = sv; stop = i; } }   // findmin :215-216
    o.stop = stop + 1;
    sync();
    for (int i = lane; i <= k; i += step) { y1[i] *= s; w1[i] *= s; }                // :217
    sync();
    if (lead) { w1[k + 1] = w1[k]; y1[k + 1] = (w1[k] == 0.0) ? y1[k] : 0.0; }       // extend :202-206
    sync();
    // chief  :246-263
    const double y_stop = y1[stop + 1], y2_stop = y2[stop + 1];
    const double nub = -w1[k + 1] * h / y1[1];                                       // :256
    for (int i = 1 + lane; i <= k; i += step) {                                      // :258
        yc[i] = nub * (y2[i] - y1[i] * y2_stop / y_stop);
        wc[i] = nub * (w2[i] - w1[i] * y2_stop / y_stop);
    }
    sync();
    if (lead) { yc[0] = 0.0; wc[0] = nub; yc[k + 1] = h; wc[k + 1] = wc[k]; }        // :259,260
    sync();
    // _solve  :302-323
    const double ybar = yc[1], nubp = wc[k + 1], ym = y1[0], ybpb = yc[k];
    const double d = (h - nubp * o.f - ybar) / nub;
    o.EFFD = d - o.f;
    o.PN = (n[rows - 1] - n[0]) * o.f;
    o.EP_D = fabs(ym) * 2.0; o.EP_t = -ybar / nub;
    o.H = nub * ym;
    o.XP_D = fabs(2.0 * o.H / nubp); o.XP_t = -ybpb / nubp;
    o.N = fabs(o.f / o.EP_D);
    auto next = [&](int i) { return n[i < rows ? i : rows - 1]; };                   // n = [n; n[end]]  Types.jl:39
    o.FOV = 2.0 * (::atan(fabs(wc[0] / next(0))) * 57.29577951308232);               // 2atand  :319
    o.y_marg = ym; o.chief_y_end = yc[k + 1]; o.chief_u_end = wc[k + 1] / next(k + 1); o.nu_end = w1[k + 1];
    o.BFD = -y1[k] / (w1[k] / next(k));                                              // Types.jl:44 (t[end] of the marginal)
}

// aberrations, SeidelAberrations.jl:6-53: the contributions of surface i (loop index, row i + 1) from the marginal and
// chief arrays first_order_paraxial left in W.  Independent per surface: the small-problem kernel runs them one lane
// each; the sums are taken in surface order either way (seidel_accumulate).
struct SeidelTerms { double sph, coma, ast, ptz, dist, axl, lat; };
__device__ __forceinline__ SeidelTerms seidel_terms(int i, int rows, const double* __restrict__ R, const double* __restrict__ n,
                                                    const double* __restrict__ dn, double H, double lam, const FirstOrderWork& W,
                                                    double& A_out)
{
    const double* y1 = W.y1; const double* w1 = W.w1; const double* yc = W.yc;
    auto next = [&](int j) { return n[j < rows ? j : rows - 1]; };
    const double ni = next(i), nj = next(i + 1), yi = y1[i + 1], ybi = yc[i + 1], Ri = R[i + 1];
    const double ui = w1[i] / ni, uj = w1[i + 1] / nj;
    const double A = w1[i] + ni * yi / Ri;
    const double Ab = (H + A * ybi) / yi;
    const double yD = yi * (uj / nj - ui / ni);
    const double yd = dn ? yi * (dn[i + 1] / nj - dn[i] / ni) : 0.0;
    const double inj = 1.0 / nj, ini = 1.0 / ni;
    const double Dn2 = inj * inj - ini * ini;
    const double P = (inj - ini) / Ri;
    SeidelTerms q;
    q.sph = -(A * A) * yD / (8.0 * lam);
    q.coma = -A * Ab * yD / (2.0 * lam);
    q.ast = -(Ab * Ab) * yD / (2.0 * lam);
    q.ptz = -(H * H) * P / (4.0 * lam);
    q.dist = -Ab * ((Ab * Ab) * yi * Dn2 - (H + Ab * yi) * ybi * P) / (2.0 * lam);
    q.axl = A * yd / (2.0 * lam); q.lat = Ab * yd / lam;
    A_out = A;
    return q;
}

// solve(surfaces, a, h') + aberrations(...) of ONE system g of the batch (see k_first_order below).
__device__ __forceinline__ void first_order_core(int g, int nsys, int rows, const double* __restrict__ R, const double* __restrict__ t,
                                                 const double* __restrict__ n, const double* __restrict__ a,
                                                 const double* __restrict__ dn, double h, double lam, const FirstOrderWork& W,
                                                 FirstOrderOut& o, double* __restrict__ surf, double* __restrict__ inc)
{
    first_order_paraxial<false>(0, rows, R, t, n, a, h, W, o);
    auto next = [&](int i) { return n[i < rows ? i : rows - 1]; };                   // n = [n; n[end]]  Types.jl:39
    double W040 = 0, W131 = 0, W222 = 0, W311 = 0, W220P = 0, W020 = 0, W111 = 0;
    for (int i = 0; i < rows - 1; ++i) {
        double A;
        const SeidelTerms q = seidel_terms(i, rows, R, n, dn, o.H, lam, W, A);
        const double sph = q.sph, coma = q.coma, ast = q.ast, ptz = q.ptz, dist = q.dist, axl = q.axl, lat = q.lat;
        W040 += sph; W131 += coma; W222 += ast; W311 += dist; W220P += ptz;
        W020 += axl; W111 += lat;
        if (surf) {      // per-surface contributions, SeidelAberrations.jl:25-34: [component][system][surface]
            const int64_t cs = (int64_t)nsys * (rows - 1), o = (int64_t)g * (rows - 1) + i;
            surf[SURF_SPHERICAL * cs + o] = sph; surf[SURF_COMA * cs + o] = coma; surf[SURF_ASTIGMATISM * cs + o] = ast;
            surf[SURF_SAGITTAL * cs + o] = ptz + ast / 2.0; surf[SURF_DISTORTION * cs + o] = dist;
            surf[SURF_AXIAL * cs + o] = axl; surf[SURF_LATERAL * cs + o] = lat; surf[SURF_PETZVAL * cs + o] = ptz;
            surf[SURF_MEDIAL * cs + o] = ptz + ast; surf[SURF_TANGENTIAL * cs + o] = ptz + 1.5 * ast;
        }
        if (inc) {       // incidences(surfaces, system) = [ni nī i ī], RayTracing.jl:338-353: [column][system][surface]
            const int64_t cs = (int64_t)nsys * (rows - 1), o = (int64_t)g * (rows - 1) + i;
            const double ni = next(i), ybi = W.yc[i + 1], Ri = R[i + 1];
            const double nib = W.wc[i] + ni * ybi / Ri;
            inc[0 * cs + o] = A; inc[1 * cs + o] = nib; inc[2 * cs + o] = A / ni; inc[3 * cs + o] = nib / ni;
        }
    }
    o.W040 = W040; o.W131 = W131; o.W222 = W222; o.W311 = W311; o.W220P = W220P;
    o.W220 = W220P + 0.5 * W222; o.W020 = W020; o.W111 = W111;
}

__global__ __launch_bounds__(64) void k_first_order(int nsys, int rows, const double* __restrict__ Rg, const double* __restrict__ tg,
                                                    const double* __restrict__ ng, const double* __restrict__ ag,
                                                    const double* __restrict__ dng, const double* __restrict__ hp,
                                                    double lam, FirstOrderOut* __restrict__ out,
                                                    double* __restrict__ surf, double* __restrict__ inc)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nsys) return;
    double tau[kMaxRows], phi[kMaxRows];
    double y1[kMaxRows + 2], w1[kMaxRows + 2], y2[kMaxRows + 1], w2[kMaxRows + 1], yc[kMaxRows + 2], wc[kMaxRows + 2];
    const FirstOrderWork W = {tau, phi, y1, w1, y2, w2, yc, wc};
    FirstOrderOut o;
    first_order_core(g, nsys, rows, Rg + (int64_t)g * rows, tg + (int64_t)g * rows, ng + (int64_t)g * rows, ag + (int64_t)g * (rows - 1),
                     dng ? dng + (int64_t)g * rows : nullptr, hp[g], lam, W, o, surf, inc);
    out[g] = o;
}

// ------------------------------------------------------------------------------------
// Device-resident spot pipeline (ort_spot_batch_f64): solve -> aim -> full_trace statistics for
// many spherical prescriptions with NO host round trip between the stages.  These small kernels
// do on the device what ort_system_create / api.py do on the host for one system at a time:
// derive the per-surface records (forward + image row, and the reversed prescription of
// src/RayTracing.jl:267-277), turn first-order results into aiming requests, and aiming results
// into bundle descriptors and axis end points (src/PupilSampling.jl:94-122).
// ------------------------------------------------------------------------------------
// number of coefficients in use for a row: the whole width if any entry is non-zero AFTER the cast to T
// (an all-zero row is the reference's `zero` polynomial), as ort_system_create does on the host
template <typename T>
__device__ __forceinline__ int row_ncoef(const double* __restrict__ c, int ncoef)
{
    int nc = 0;
    for (int j = 0; j < ncoef; ++j) if ((T)c[j] != T(0)) nc = ncoef;
    return nc;
}

// one thread per (system, loop index i): extended skew table [nsys][rows] (+ its polynomial records
// [nsys][rows][kPolyRec]), forward and reversed meridional tables [nsys][rows-1] (+ the reversed coefficient
// rows [nsys][rows][ncoef]; the forward ones are the input), last thicknesses.  K, coef may be null.
// Row i of system s: the extended skew record (+ polynomial record) of loop iteration i and, for i < rows-1, the
// forward and reversed meridional rows (+ the reversed coefficient row).  Any output pointer may be null (not wanted).
// Rs, ts, ns, Ks, cs: the system's own columns (Ks, cs may be null).  rec / poly: this row's slots; mf / mr: this row's
// slots; cr: the reversed coefficient table of the system [rows][ncoef].
template <typename T>
__device__ __forceinline__ void build_table_row(int i, int rows, const double* __restrict__ Rs, const double* __restrict__ ts,
                                                const double* __restrict__ ns, const double* __restrict__ Ks,
                                                const double* __restrict__ cs, int ncoef, double BFD,
                                                SurfRec<T>* __restrict__ rec, T* __restrict__ poly, bool want_poly,
                                                MerSurf* __restrict__ mf, MerSurf* __restrict__ mr, double* __restrict__ cr)
{
    auto tt = [&](int j) { return (j == 0 && !__builtin_isfinite(ts[0])) ? 0.0 : ts[j]; };          // Lens() mutation (Q19)
    // extended system (PupilSampling.jl:111-114): rows+1 rows, loop index i = 0..rows-1 is row i+1
    if (rec) {
        const bool real = i + 1 < rows;                                  // the last iteration is the appended image plane
        const double te = (i == rows - 1) ? BFD : tt(i);                 // t[end-1] = focus
        const double Re = real ? Rs[i + 1] : __builtin_inf();
        const double n1 = ns[i], n2 = real ? ns[i + 1] : 1.0;
        const double Ke = (real && Ks) ? Ks[i + 1] : 0.0;                // K = [surfaces.K; 0.0]  (:112)
        int nc = 0, pcls = 0;
        if (want_poly)                                                   // p = [surfaces.p; zero]  (:113)
            pcls = make_poly_rec<T>(poly, (real && cs) ? cs + (int64_t)(i + 1) * ncoef : nullptr, ncoef, &nc);
        SurfRec<T> r;
        make_rec<T>(r, (T)te, (T)Re, (T)n1, (T)n2, (T)Ke, nc, pcls);
        *rec = r;
    }
    if (i < rows - 1 && mf) {
        MerSurf m;
        m.t = tt(i); m.R = Rs[i + 1]; m.sgn = m.R > 0 ? 1.0 : (m.R < 0 ? -1.0 : m.R); m.K = Ks ? Ks[i + 1] : 0.0;
        m.n1 = ns[i]; m.n2 = ns[i + 1]; m.finite = __builtin_isfinite(m.R) ? 1 : 0;
        m.eta = m.n1 / m.n2; m.invR = m.finite ? 1.0 / m.R : 0.0;
        m.ncoef = cs ? row_ncoef<double>(cs + (int64_t)(i + 1) * ncoef, ncoef) : 0;
        *mf = m;
        // reversed (RayTracing.jl:267-274): rev_R = -[Inf; R[end:-1:2]], rev_t = reverse(t) with rev_t[1] = BFD,
        // K and p plainly reversed (Q17): row j = i + 1 of the reversed system carries K[rows-1-j], p[rows-1-j]
        MerSurf q;
        q.t = (i == 0) ? BFD : tt(rows - 1 - i);
        q.R = -Rs[rows - 1 - i];                                         // rev_R[i+1] = -R[rows-1-i]
        q.sgn = q.R > 0 ? 1.0 : (q.R < 0 ? -1.0 : q.R); q.K = Ks ? Ks[rows - 2 - i] : 0.0;
        q.n1 = ns[rows - 1 - i]; q.n2 = ns[rows - 2 - i]; q.finite = __builtin_isfinite(q.R) ? 1 : 0;
        q.eta = q.n1 / q.n2; q.invR = q.finite ? 1.0 / q.R : 0.0;
        q.ncoef = cs ? row_ncoef<double>(cs + (int64_t)(rows - 2 - i) * ncoef, ncoef) : 0;
        *mr = q;
        if (cr) {
            for (int j = 0; j < ncoef; ++j) cr[(int64_t)(i + 1) * ncoef + j] = cs[(int64_t)(rows - 2 - i) * ncoef + j];
            if (i == 0) for (int j = 0; j < ncoef; ++j) cr[j] = cs[(int64_t)(rows - 1) * ncoef + j];
        }
    }
}

// one thread per (system, loop index i): extended skew table [nsys][rows] (+ its polynomial records
// [nsys][rows][kPolyRec]), forward and reversed meridional tables [nsys][rows-1] (+ the reversed coefficient
// rows [nsys][rows][ncoef]; the forward ones are the input), last thicknesses.  K, coef may be null.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_build_tables(int nsys, int rows, const double* __restrict__ R, const double* __restrict__ t,
                                                         const double* __restrict__ n, const double* __restrict__ K,
                                                         const double* __restrict__ coef, int ncoef,
                                                         const FirstOrderOut* __restrict__ fo,
                                                         SurfRec<T>* __restrict__ rec_ext, T* __restrict__ poly_ext,
                                                         MerSurf* __restrict__ mer_fwd, MerSurf* __restrict__ mer_rev,
                                                         double* __restrict__ crev, double* __restrict__ tl_fwd,
                                                         double* __restrict__ tl_rev)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= (int64_t)nsys * rows) return;
    const int s = (int)(g / rows), i = (int)(g - (int64_t)s * rows);
    const double* ts = t + (int64_t)s * rows;
    const bool mer = i < rows - 1;
    build_table_row<T>(i, rows, R + (int64_t)s * rows, ts, n + (int64_t)s * rows, K ? K + (int64_t)s * rows : nullptr,
                       (coef && ncoef > 0) ? coef + (int64_t)s * rows * ncoef : nullptr, ncoef, fo[s].BFD,
                       rec_ext + (int64_t)s * rows + i, poly_ext ? poly_ext + ((int64_t)s * rows + i) * kPolyRec : nullptr, poly_ext != nullptr,
                       mer ? mer_fwd + (int64_t)s * (rows - 1) + i : nullptr, mer ? mer_rev + (int64_t)s * (rows - 1) + i : nullptr,
                       crev ? crev + (int64_t)s * rows * ncoef : nullptr);
    if (i == 0) {
        tl_fwd[s] = (rows - 1 == 0 && !__builtin_isfinite(ts[0])) ? 0.0 : ts[rows - 1];
        tl_rev[s] = !__builtin_isfinite(ts[0]) ? 0.0 : ts[0];
    }
}

// The aiming request of (system s, field H) from the first-order results (PupilSampling.jl:88-93).
__device__ __forceinline__ AimIn make_aim_in(int s, const FirstOrderOut& o, double a_stop, double H, int layout_fwd)
{
    AimIn q;
    q.system = s; q.stop = o.stop; q.layout_fwd = layout_fwd; q.layout_rev = 1;   // the reversed system is always a Layout (:272-276)
    q.H = fabs(H); q.y_marg = o.y_marg; q.a_stop = a_stop;
    q.chief_y_end = o.chief_y_end; q.chief_u_end = o.chief_u_end; q.f = o.f; q.atol = 1.4901161193847656e-08;
    return q;
}

__global__ __launch_bounds__(kBlock) void k_build_aim(int nsys, int nf, int rows, const FirstOrderOut* __restrict__ fo,
                                                      const double* __restrict__ a, const double* __restrict__ fields,
                                                      int layout_fwd, AimIn* __restrict__ ain)
{
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= nsys * nf) return;
    const int s = g / nf, f = g - s * nf;
    const FirstOrderOut o = fo[s];
    ain[g] = make_aim_in(s, o, a[(int64_t)s * (rows - 1) + o.stop - 1], fields[f], layout_fwd);
}

// The bundle descriptor of aiming result g (PupilSampling.jl:94-99,115): axes of bundle g at g (k_rays + k2).
template <typename T>
__device__ __forceinline__ DevBundle<T> make_dev_bundle(const AimIn& q, const AimOut& o, int g, int k_rays, int k2)
{
    DevBundle<T> d;
    d.system = q.system; d.stop = q.stop - 1;
    d.u = (T)::tan(o.U); d.v = T(0);                                     // PupilSampling.jl:38-39 (V = 0, :115)
    const T nrm = (T)__builtin_sqrt((double)((d.v * d.v + d.u * d.u) + T(1))), inv = T(1) / nrm;
    d.k0 = d.v * inv; d.k1 = d.u * inv; d.k2 = inv;
    d.a_stop = (T)fabs(q.a_stop); d.hprime = (T)o.hprime; d.ybar = T(0); d.z0 = T(1);
    d.yoff = (int64_t)g * (k_rays + k2); d.xoff = d.yoff + k_rays;
    return d;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_build_bundles(int na, int k_rays, int k2, const AimIn* __restrict__ ain,
                                                          const AimOut* __restrict__ aout, DevBundle<T>* __restrict__ bd,
                                                          double* __restrict__ ends, int* __restrict__ fail_flag)
{
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= na) return;
    const AimIn q = ain[g]; const AimOut o = aout[g];
    if (!o.ok) atomicOr(fail_flag, 1);
    bd[g] = make_dev_bundle<T>(q, o, g, k_rays, k2);
    ends[4 * (int64_t)g + 0] = o.y1; ends[4 * (int64_t)g + 1] = o.y2; ends[4 * (int64_t)g + 2] = 0.0; ends[4 * (int64_t)g + 3] = o.y_EP;
}

// ------------------------------------------------------------------------------------
// Small problems (a handful of (system, field) pairs: the reference's own call, full_trace(system, H, 64) = 2,048 rays):
// everything in front of the grid trace in ONE launch, one wave per pair — first-order solve (lane 0, work arrays in
// LDS), tables (lanes over rows; the meridional ones stay in LDS), aiming (aim_group, four lanes), bundle descriptor
// and pupil axes — the work of k_first_order, k_build_tables, k_build_aim, k_aim, k_build_bundles and k_make_axes
// through the same device functions, so the numbers are theirs.  The launches, not the arithmetic, were the cost:
// six dependent launches + a fill ahead of the trace kernel, ~110 us of a 140 us call.
// `fail_flag` must be zero on entry.  rec_ext / poly_ext / fo are written by the pair's first field only.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_small_prepare(int nsys, int nf, int rows, const double* __restrict__ R, const double* __restrict__ t,
                                                      const double* __restrict__ n, const double* __restrict__ K,
                                                      const double* __restrict__ coef, int ncoef, const double* __restrict__ a,
                                                      const double* __restrict__ hp, const double* __restrict__ fields,
                                                      int k_rays, int k2, int layout_fwd, double lam,
                                                      FirstOrderOut* __restrict__ fo, SurfRec<T>* __restrict__ rec_ext,
                                                      T* __restrict__ poly_ext, DevBundle<T>* __restrict__ bd, T* __restrict__ axes,
                                                      int* __restrict__ fail_flag, int fast)
{
    __shared__ double s_work[8][kMaxRows + 2];
    __shared__ double s_in[4][kMaxRows];                                 // the system's R, t, n, a columns
    __shared__ SeidelTerms s_terms[kMaxRows];
    __shared__ MerSurf s_mf[kMaxRows], s_mr[kMaxRows];
    __shared__ double s_crev[kMaxRows * kMaxCoef];
    __shared__ FirstOrderOut s_fo;
    const int g = blockIdx.x, s = g / nf, f = g - s * nf, lane = threadIdx.x;
    const double* Ks = K ? K + (int64_t)s * rows : nullptr;
    const double* cs = (coef && ncoef > 0) ? coef + (int64_t)s * rows * ncoef : nullptr;
    ORT_PHASE(0);
    const double h_s = hp[s], field_f = fields[f];                      // (the inputs may live in host memory: every fetch up front)
    // the serial first-order chain reads its columns dozens of times: one parallel fetch into LDS instead of a
    // global-memory round trip per loop iteration
    for (int i = lane; i < rows; i += 64) {
        s_in[0][i] = R[(int64_t)s * rows + i]; s_in[1][i] = t[(int64_t)s * rows + i]; s_in[2][i] = n[(int64_t)s * rows + i];
        if (i < rows - 1) s_in[3][i] = a[(int64_t)s * (rows - 1) + i];
    }
    __syncthreads();
    const double* Rs = s_in[0]; const double* ts = s_in[1]; const double* ns = s_in[2];
    const FirstOrderWork W = {s_work[0], s_work[1], s_work[2], s_work[3], s_work[4], s_work[5], s_work[6], s_work[7]};
    {
        FirstOrderOut o;
        first_order_paraxial<true>(lane, rows, Rs, ts, ns, s_in[3], h_s, W, o);   // every lane holds the scalars; lane 0's are kept
        if (lane == 0) s_fo = o;
    }
    __syncthreads();
    // Seidel contributions: one lane per surface (independent), summed in surface order by lane 0 — first_order_core's
    // arithmetic, a surface's latency instead of rows - 1 of them
    {
        const double Hl = s_fo.H;
        for (int i = lane; i < rows - 1; i += 64) { double A; s_terms[i] = seidel_terms(i, rows, Rs, ns, nullptr, Hl, lam, W, A); }
    }
    __syncthreads();
    if (lane == 0) {
        double W040 = 0, W131 = 0, W222 = 0, W311 = 0, W220P = 0, W020 = 0, W111 = 0;
        for (int i = 0; i < rows - 1; ++i) {
            const SeidelTerms q = s_terms[i];
            W040 += q.sph; W131 += q.coma; W222 += q.ast; W311 += q.dist; W220P += q.ptz; W020 += q.axl; W111 += q.lat;
        }
        s_fo.W040 = W040; s_fo.W131 = W131; s_fo.W222 = W222; s_fo.W311 = W311; s_fo.W220P = W220P;
        s_fo.W220 = W220P + 0.5 * W222; s_fo.W020 = W020; s_fo.W111 = W111;
        if (f == 0) fo[s] = s_fo;
    }
    __syncthreads();
    ORT_PHASE(1);
    const FirstOrderOut o1 = s_fo;
    for (int i = lane; i < rows; i += 64) {
        const bool mer = i < rows - 1;
        build_table_row<T>(i, rows, Rs, ts, ns, Ks, cs, ncoef, o1.BFD,
                           f == 0 ? rec_ext + (int64_t)s * rows + i : nullptr,
                           (f == 0 && poly_ext) ? poly_ext + ((int64_t)s * rows + i) * kPolyRec : nullptr, poly_ext != nullptr,
                           mer ? s_mf + i : nullptr, mer ? s_mr + i : nullptr, cs ? s_crev : nullptr);
    }
    __syncthreads();
    ORT_PHASE(2);
    const double tlF = ts[rows - 1], tlR = !__builtin_isfinite(ts[0]) ? 0.0 : ts[0];
    const AimIn q = make_aim_in(s, o1, s_in[3][o1.stop - 1], field_f, layout_fwd);
    const AimOut o = aim_group(q, lane & 3, s_mf, cs, tlF, s_mr, cs ? s_crev : nullptr, tlR, rows - 1, ncoef, true, fast != 0);
    ORT_PHASE(6);
    if (lane == 0) {
        if (!o.ok) *fail_flag = 1;                                       // (a plain store: the flag may live in host memory)
        bd[g] = make_dev_bundle<T>(q, o, g, k_rays, k2);
    }
    const int per = k_rays + k2;
    for (int j = lane; j < per; j += 64)                                 // range(y1, y2, k), range(0, y_EP, k / 2)  (:121-122), as k_make_axes
        axes[(int64_t)g * per + j] = (T)((j < k_rays) ? dd_range_elem(o.y1, o.y2, k_rays, j) : dd_range_elem(0.0, o.y_EP, k2, j - k_rays));
    ORT_PHASE(7);
}

// ------------------------------------------------------------------------------------
// wavegrad(eps, lambda) = (eps.x nu / lambda, eps.y nu / lambda)  (src/PupilSampling.jl:165-167) for device-resident
// full_trace results: bundle b's count[b] valid entries of its slab [cap].  Out of place, like the reference's map.
// Grid: nb * chunks blocks of kTile entries.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_wavegrad(int64_t cap, int chunks, const int64_t* __restrict__ count,
                                                     const double* __restrict__ nu, double lambda,
                                                     const T* __restrict__ ex, const T* __restrict__ ey,
                                                     T* __restrict__ gx, T* __restrict__ gy)
{
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int64_t n = count[b] < cap ? count[b] : cap;
    const T nub = (T)nu[b], lam = (T)lambda;
    const int64_t o = (int64_t)b * cap;
    for (int64_t j = (int64_t)chunk * kTile + threadIdx.x; j < n && j < (int64_t)(chunk + 1) * kTile; j += kBlock) {
        gx[o + j] = (ex[o + j] * nub) / lam;                    // getfield(eps, f) * eps.nu / lambda: product, then quotient,
        gy[o + j] = (ey[o + j] * nub) / lam;                    // one IEEE operation each as in the reference's broadcast
    }
}

// ------------------------------------------------------------------------------------
// ABCD, src/TransferMatrix.jl:1-17.  One thread per lens / per vector.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void mm2(const double* A, const double* B, double* C)
{
    const double c0 = A[0] * B[0] + A[1] * B[2];
    const double c1 = A[0] * B[1] + A[1] * B[3];
    const double c2 = A[2] * B[0] + A[3] * B[2];
    const double c3 = A[2] * B[1] + A[3] * B[3];
    C[0] = c0; C[1] = c1; C[2] = c2; C[3] = c3;
}

__global__ __launch_bounds__(kBlock) void k_abcd(int nlens, int k, const double* __restrict__ tau,
                                                 const double* __restrict__ phi, double* __restrict__ M)
{
    const int l = blockIdx.x * kBlock + threadIdx.x;
    if (l >= nlens) return;
    double acc[4] = {1.0, 0.0, 0.0, 1.0};
    for (int i = k - 1; i >= 0; --i) {                            // reverse(axes(M, 1)) (:4)
        const double tq = tau[(int64_t)l * k + i], ph = phi[(int64_t)l * k + i];
        const double Mi[4] = {1.0, tq, -ph, 1.0 - tq * ph};
        if (i == k - 1) { acc[0] = Mi[0]; acc[1] = Mi[1]; acc[2] = Mi[2]; acc[3] = Mi[3]; }
        else { double tmp[4]; mm2(acc, Mi, tmp); acc[0] = tmp[0]; acc[1] = tmp[1]; acc[2] = tmp[2]; acc[3] = tmp[3]; }
    }
    M[(int64_t)l * 4 + 0] = acc[0]; M[(int64_t)l * 4 + 1] = acc[1];
    M[(int64_t)l * 4 + 2] = acc[2]; M[(int64_t)l * 4 + 3] = acc[3];
}

__device__ __forceinline__ void extend2(const double* M, double tau, double tau_p, double* E)
{
    const double L[4] = {1.0, tau_p, 0.0, 1.0};
    const double Rm[4] = {1.0, tau, 0.0, 1.0};
    double tmp[4];
    mm2(L, M, tmp);                                                // :8, left to right
    mm2(tmp, Rm, E);
}

__global__ __launch_bounds__(kBlock) void k_abcd_transfer(const double* __restrict__ M, int64_t nv, const double* __restrict__ v,
                                                          const double* __restrict__ tau, const double* __restrict__ tau_p,
                                                          double* __restrict__ out, int reverse)
{
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= nv) return;
    const double Mm[4] = {M[0], M[1], M[2], M[3]};
    double E[4];
    extend2(Mm, tau[g], tau_p[g], E);
    const double v0 = v[2 * g], v1 = v[2 * g + 1];
    if (!reverse) {                                                // :10
        out[2 * g] = E[0] * v0 + E[1] * v1;
        out[2 * g + 1] = E[2] * v0 + E[3] * v1;
    } else {                                                       // :13, LU with partial pivoting
        double a = E[0], b = E[1], c = E[2], d = E[3], r0 = v0, r1 = v1;
        if (fabs(c) > fabs(a)) { double t; t = a; a = c; c = t; t = b; b = d; d = t; t = r0; r0 = r1; r1 = t; }
        const double l = c / a;
        const double d2 = d - l * b;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / d2;
        const double x0 = (r0 - b * x1) / a;
        out[2 * g] = x0; out[2 * g + 1] = x1;
    }
}

}  // namespace ort

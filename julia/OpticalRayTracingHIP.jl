# OpticalRayTracingHIP.jl — `ccall` shim over libort_hip.so (include/ort.h).
#
# Adds BATCH methods with the reference's names and argument order to OpticalRayTracing.jl,
# and a `full_trace` that replaces only the grid stage (src/PupilSampling.jl:115-146), taking
# the aiming scalars from the unmodified reference code.  Not exercised in the build container
# (no Julia runtime there); the same C ABI is exercised through ctypes by the Python mirror
# (the device-resident calls below: tests/test_gpu_parity.py::test_device_buffers_through_the_c_abi).
#
#   ENV["ORT_HIP_LIB"] = "/path/to/libort_hip.so";  include("OpticalRayTracingHIP.jl")
module OpticalRayTracingHIP

using OpticalRayTracing
import OpticalRayTracing: raytrace, full_trace, Layout, System, RealRay, RealRayError, TransferMatrix, Lens

const LIB = get(ENV, "ORT_HIP_LIB", joinpath(@__DIR__, "..", "opticalraytracing.jl_amd", "csrc", "libort_hip.so"))

const ORT_DEVICE_PTRS = UInt32(1) << 0
const ORT_FAST_MATH = UInt32(1) << 5
const ORT_CLIP = UInt32(1) << 4
const ORT_LAYOUT_INPUT = UInt32(1) << 3

struct OrtBundle            # == ort_bundle (include/ort.h)
    system::Int32; stop::Int32
    U::Float64; V::Float64; a_stop::Float64; hprime::Float64; ybar::Float64; z0::Float64
    yaxis_off::Int64; xaxis_off::Int64
end

lasterr() = unsafe_string(ccall((:ort_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 ? nothing : rc == -2 ? throw(DomainError(rc, lasterr())) : error("ort: " * lasterr())

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ort_ctx_create, LIB), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, r))
        finalizer(c -> ccall((:ort_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.h), new(r[]))
    end
end
const CTX = Ref{Context}()
ctx() = (isassigned(CTX) || (CTX[] = Context()); CTX[])

# ---- device-resident results ------------------------------------------------------------------------
# A host-pointer call brings every ray-sized output back over PCIe (config 2's history: 1.8 GB, ~30 ms against
# 0.3 ms of kernel).  DeviceArray keeps results on the GPU (ort_device_malloc + ORT_DEVICE_PTRS); `download`
# copies back only what the host reads — a row of the history, the image-plane hits, a strided sample.
mutable struct DeviceArray{T}
    p::Ptr{T}; len::Int
    function DeviceArray{T}(len::Integer) where T
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ort_device_malloc, LIB), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx().h, len * sizeof(T), r))
        finalizer(a -> ccall((:ort_device_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx().h, a.p), new{T}(Ptr{T}(r[]), len))
    end
end
function DeviceArray(v::Vector{T}) where T
    d = DeviceArray{T}(length(v))
    GC.@preserve v check(ccall((:ort_device_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx().h, d.p, pointer(v), sizeof(v)))
    return d
end
"download(d, first, n): elements first .. first+n-1 (1-based) of a device array"
function download(d::DeviceArray{T}, first::Integer = 1, n::Integer = d.len - first + 1) where T
    v = Vector{T}(undef, n)
    GC.@preserve v check(ccall((:ort_device_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                               ctx().h, pointer(v), d.p + (first - 1) * sizeof(T), n * sizeof(T)))
    return v
end
synchronize() = check(ccall((:ort_ctx_synchronize, LIB), Cint, (Ptr{Cvoid},), ctx().h))

struct GridOut64            # == ort_grid_out_f64
    xv::Ptr{Float64}; yv::Ptr{Float64}; ld::Int64
    xf::Ptr{Float64}; yf::Ptr{Float64}; xs::Ptr{Float64}; ys::Ptr{Float64}; status::Ptr{Int32}
end

"""
    trace_grid_device(sys, bundles, axes, ny, nx; history = true, flags = 0) -> NamedTuple of DeviceArrays

The hot loop of `full_trace` (src/PupilSampling.jl:121-138) for many bundles with every output LEFT ON THE GPU:
`xv`, `yv` ((rows-1) x N, surface-major: surface s of ray r at s*N + r) when `history`, else the summary
`xf`, `yf`, `status`.  `sys` comes from `upload`, `bundles :: Vector{OrtBundle}`, `axes :: Vector{Float64}` (the
exact Julia `range`s, collected).  The call is asynchronous; `download` / `synchronize()` wait for it.
"""
function trace_grid_device(sys, rows::Integer, bundles::Vector{OrtBundle}, axes::Vector{Float64}, ny::Integer, nx::Integer;
                           history::Bool = true, flags::UInt32 = UInt32(0))
    N = length(bundles) * ny * nx; S = rows - 1
    dax = DeviceArray(axes)
    if history
        xv = DeviceArray{Float64}(S * N); yv = DeviceArray{Float64}(S * N)
        out = Ref(GridOut64(xv.p, yv.p, N, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL)); res = (xv = xv, yv = yv, N = N, S = S)
    else
        xf = DeviceArray{Float64}(N); yf = DeviceArray{Float64}(N); st = DeviceArray{Int32}(N)
        out = Ref(GridOut64(C_NULL, C_NULL, 0, xf.p, yf.p, C_NULL, C_NULL, st.p)); res = (xf = xf, yf = yf, status = st, N = N)
    end
    GC.@preserve bundles out dax check(ccall((:ort_trace_grid_f64, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{OrtBundle}, Ptr{Float64}, Int64, Cint, Cint, Ref{GridOut64}, UInt32),
        ctx().h, sys, length(bundles), bundles, dax.p, length(axes), ny, nx, out, flags | ORT_DEVICE_PTRS))
    synchronize()                                   # `dax` may be collected after this
    return res
end

"""
    full_trace_rms(surfaces::Layout, system, H, k_rays, focus) -> (count, RMS)

`full_trace(...).RMS` and the ray count with NOTHING ray-sized crossing PCIe: the statistics-only route of
`ort_full_trace_f64` (NULL error vectors) — aiming by the unmodified reference, grid stage, stop filter, mirrored
centroid and σ on the GPU; 96 doubles of axes in, 16 bytes out.
"""
function full_trace_rms(surfaces::Layout, system::System, H::Float64, k_rays::Int = 64,
                        focus = system.marginal.z[end] - system.marginal.z[end-1]; coef = nothing, flags::UInt32 = UInt32(0))
    H = abs(H); H ≤ 1.0 || throw(DomainError(H, "Domain: |H| ≤ 1.0"))
    stop = system.stop; a_stop = abs(system.a[stop])
    real_chief = trace_chief_ray(surfaces, system); real_marginal = trace_marginal_ray(surfaces, system)
    EP_t = real_chief.z[1]; U = H * real_chief.u[1]; u = tan(U); y_EP = abs(real_marginal.y[1])
    y1, y2 = OpticalRayTracing.trace_edge_rays(surfaces, y_EP - u * EP_t, -y_EP - u * EP_t, U, stop, a_stop)
    R = [surfaces[:, 1]; Inf]; t = [surfaces[:, 2]; 0.0]; n = [surfaces[:, 3]; 1.0]; t[end-1] = focus
    sys = upload(R, t, n, [surfaces.K; 0.0], coef === nothing ? nothing : [coef; zeros(1, size(coef, 2))])
    k2 = div(k_rays, 2)
    axes = [collect(range(y1, y2, k_rays)); collect(range(0.0, y_EP, k2))]
    b = [OrtBundle(0, stop, U, 0.0, a_stop, u * system.f, 0.0, 1.0, 0, k_rays)]
    cnt = Ref{Int64}(0); rms = Ref{Float64}(0.0)
    GC.@preserve axes b check(ccall((:ort_full_trace_f64, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{OrtBundle}, Ptr{Float64}, Int64, Cint, Cint,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Ref{Float64}, UInt32),
        ctx().h, sys, 1, b, axes, length(axes), k_rays, k2, C_NULL, C_NULL, C_NULL, C_NULL, cnt, rms, flags))
    release(sys)
    return cnt[], rms[]
end

"Power-series coefficients of a row's polynomial; the closure handed to the CPU path is built from the same vector."
poly(c::Vector{Float64}) = OpticalRayTracing.Polynomial(y -> evalpoly(y, c))

function upload(R, t, n, K = nothing, coef = nothing)
    rows = length(R)
    ncoef = coef === nothing ? 0 : size(coef, 2)
    coefT = coef === nothing ? nothing : permutedims(coef)          # [rows][ncoef] row-major, kept alive below
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve R t n K coefT check(ccall((:ort_system_create, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ref{Ptr{Cvoid}}),
        ctx().h, 1, rows, R, t, n, K === nothing ? C_NULL : pointer(K), coefT === nothing ? C_NULL : pointer(coefT), ncoef, r))
    return r[]
end
release(sys) = ccall((:ort_system_destroy, LIB), Cint, (Ptr{Cvoid},), sys)

# raytrace(surfaces, y, x, U, V, Vector{RealRay}) over ray VECTORS -> (xv, yv), each N x (rows-1)
# (column j = surface j: the surface-major layout of the ABI is a Julia N x S matrix)
function raytrace(surfaces::AbstractMatrix, y::Vector{Float64}, x::Vector{Float64},
                  U::Vector{Float64}, V::Vector{Float64}, ::Type{Vector{RealRay}};
                  K = nothing, coef = nothing, flags::UInt32 = UInt32(0))
    M = Matrix{Float64}(surfaces[:, 1:3]); rows = size(M, 1); N = length(y)
    sys = upload(M[:, 1], M[:, 2], M[:, 3], K, coef)
    xv = Matrix{Float64}(undef, N, rows - 1); yv = similar(xv)
    GC.@preserve y x U V xv yv check(ccall((:ort_trace_skew_f64, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, UInt32),
        ctx().h, sys, 0, N, y, x, U, V, xv, yv, N, C_NULL, flags))
    release(sys)
    return xv, yv
end

# raytrace(surfaces, y, U, RealRay) over ray vectors -> (y, U, ts) N x rows; z = cumsum(ts, dims = 2)
function raytrace(surfaces::AbstractMatrix, y::Vector{Float64}, U::Vector{Float64}, ::Type{RealRay};
                  K = nothing, coef = nothing)
    M = Matrix{Float64}(surfaces[:, 1:3]); rows = size(M, 1); N = length(y)
    sys = upload(M[:, 1], M[:, 2], M[:, 3], K, coef)
    yo = Matrix{Float64}(undef, N, rows); Uo = similar(yo); ts = similar(yo)
    flags = surfaces isa Layout{OpticalRayTracing.Aspheric} ? ORT_LAYOUT_INPUT : UInt32(0)
    GC.@preserve y U yo Uo ts check(ccall((:ort_trace_meridional_f64, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, UInt32),
        ctx().h, sys, 0, N, y, U, yo, Uo, ts, N, flags))
    release(sys)
    return yo, Uo, ts
end

# raytrace(lens, y, ω, a; clip) over ray vectors -> (rt_y, rt_ω) N x (k+1)
function raytrace(lens::Lens, y::Vector{Float64}, ω::Vector{Float64},
                  a::AbstractVector = fill(Inf, size(lens, 1)); clip = false)
    τ = lens.M[:, 1]; ϕ = lens.M[:, 2]; k = length(τ); N = length(y); av = Vector{Float64}(a)
    rt_y = Matrix{Float64}(undef, N, k + 1); rt_w = similar(rt_y)
    GC.@preserve τ ϕ av y ω rt_y rt_w check(ccall((:ort_trace_paraxial_f64, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Int64, UInt32),
        ctx().h, 1, k, τ, ϕ, av, N, y, ω, rt_y, rt_w, N, clip ? ORT_CLIP : UInt32(0)))
    return rt_y, rt_w
end

# TransferMatrix(lens) on the device
function transfer_matrix(lens::Lens)
    τ = lens.M[:, 1]; ϕ = lens.M[:, 2]; M = Vector{Float64}(undef, 4)
    GC.@preserve τ ϕ M check(ccall((:ort_abcd_f64, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt32), ctx().h, 1, length(τ), τ, ϕ, M, 0))
    return TransferMatrix(permutedims(reshape(M, 2, 2)))
end

# full_trace: aiming by the UNMODIFIED reference (src/PupilSampling.jl:88-114), grid stage on the GPU
function full_trace(surfaces::Layout, system::System, H::Float64, k_rays::Int, focus, ::Val{:hip};
                    coef = nothing, flags::UInt32 = UInt32(0))
    H = abs(H); H ≤ 1.0 || throw(DomainError(H, "Domain: |H| ≤ 1.0"))
    stop = system.stop; a_stop = abs(system.a[stop])
    real_chief = trace_chief_ray(surfaces, system); real_marginal = trace_marginal_ray(surfaces, system)
    EP_t = real_chief.z[1]; U = H * real_chief.u[1]; u = tan(U); y_EP = abs(real_marginal.y[1])
    y1, y2 = OpticalRayTracing.trace_edge_rays(surfaces, y_EP - u * EP_t, -y_EP - u * EP_t, U, stop, a_stop)
    h′ = u * system.f
    R = [surfaces[:, 1]; Inf]; t = [surfaces[:, 2]; 0.0]; n = [surfaces[:, 3]; 1.0]; t[end-1] = focus
    K = [surfaces.K; 0.0]
    sys = upload(R, t, n, K, coef === nothing ? nothing : [coef; zeros(1, size(coef, 2))])
    k2 = div(k_rays, 2)
    axes = [collect(range(y1, y2, k_rays)); collect(range(0.0, y_EP, k2))]     # exact Julia grid
    b = [OrtBundle(0, stop, U, 0.0, a_stop, h′, 0.0, 1.0, 0, k_rays)]
    cap = 2 * k_rays * k2
    εx = Vector{Float64}(undef, cap); εy = similar(εx); ρ = similar(εx); θ = similar(εx)
    cnt = Ref{Int64}(0); rms = Ref{Float64}(0.0)
    GC.@preserve axes b εx εy ρ θ check(ccall((:ort_full_trace_f64, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{OrtBundle}, Ptr{Float64}, Int64, Cint, Cint,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Ref{Float64}, UInt32),
        ctx().h, sys, 1, b, axes, length(axes), k_rays, k2, εx, εy, ρ, θ, cnt, rms, flags))
    release(sys)
    m = cnt[]
    return RealRayError(εx[1:m], εy[1:m], system.marginal.nu[end], ρ[1:m], θ[1:m], H, rms[])
end

# [full_trace(solve(M, a, h′), H, k_rays).RMS for M in instances, H in fields] as ONE call: solve, aiming,
# pupil axes, trace and spot statistics chained on the device (include/ort.h: ort_spot_batch_f64 / _f32).
# `instances` :: Vector of rows×3 surface matrices of equal size; returns (count, rms) :: nfields × ninst.
function spot_batch(instances::Vector{<:AbstractMatrix}, a::AbstractVector, h′::Float64,
                    fields::Vector{Float64} = [0.0], k_rays::Int = 64; single::Bool = false,
                    flags::UInt32 = UInt32(0))
    ninst = length(instances); rows = size(instances[1], 1)
    col(j) = reduce(vcat, (Float64.(M[:, j]) for M in instances))       # [ninst][rows], instance-major
    R, t, n = col(1), col(2), col(3)
    av = repeat(Float64.(a), ninst); hp = fill(h′, ninst)
    cnt = Matrix{Int64}(undef, length(fields), ninst); rms = Matrix{Float64}(undef, length(fields), ninst)
    GC.@preserve R t n av hp fields cnt rms begin
        rc = single ?
            ccall((:ort_spot_batch_f32, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                  Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, UInt32),
                  ctx().h, ninst, rows, R, t, n, av, hp, length(fields), fields, k_rays, C_NULL, cnt, rms, flags) :
            ccall((:ort_spot_batch_f64, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                  Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, UInt32),
                  ctx().h, ninst, rows, R, t, n, av, hp, length(fields), fields, k_rays, C_NULL, cnt, rms, flags)
        check(rc)
    end
    return cnt, rms
end

# full_trace(system::System{Layout{Aspheric}}, H, k_rays) for a batch of aspheric Layouts in ONE call
# (ort_full_trace_layout_batch_f64): `layouts` :: Vector{Layout} of equal size, `coef` :: rows × ncoef power-series
# table shared by the instances (or `nothing`).  Returns a Vector{RealRayError}, instance-major, field-minor.
function full_trace_batch(layouts::Vector{<:Layout}, a::AbstractVector, h′::Float64, fields::Vector{Float64},
                          k_rays::Int = 64; coef = nothing, flags::UInt32 = UInt32(0))
    ninst = length(layouts); rows = size(layouts[1].M, 1); nf = length(fields)
    col(j) = reduce(vcat, (Float64.(L.M[:, j]) for L in layouts))
    R, t, n, K = col(1), col(2), col(3), reduce(vcat, (Float64.(L.K) for L in layouts))
    ncoef = coef === nothing ? 0 : size(coef, 2)
    coefT = coef === nothing ? nothing : repeat(vec(permutedims(coef)), ninst)          # [ninst][rows][ncoef]
    av = repeat(Float64.(a), ninst); hp = fill(h′, ninst)
    cap = 2 * k_rays * div(k_rays, 2); na = ninst * nf
    εx = Matrix{Float64}(undef, cap, na); εy = similar(εx); ρ = similar(εx); θ = similar(εx)
    cnt = Vector{Int64}(undef, na); rms = Vector{Float64}(undef, na)
    GC.@preserve R t n K coefT av hp fields εx εy ρ θ cnt rms check(ccall((:ort_full_trace_layout_batch_f64, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint,
         Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Cvoid},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, UInt32),
        ctx().h, ninst, rows, R, t, n, K, coefT === nothing ? C_NULL : pointer(coefT), ncoef,
        av, hp, nf, fields, k_rays, C_NULL, εx, εy, ρ, θ, cnt, rms, flags))
    return [RealRayError(εx[1:cnt[b], b], εy[1:cnt[b], b], NaN, ρ[1:cnt[b], b], θ[1:cnt[b], b], fields[(b - 1) % nf + 1], rms[b])
            for b in 1:na]                              # nu (marginal.nu[end]) is in the first-order struct when fo_out is passed
end

end # module

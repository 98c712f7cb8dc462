# WaterLilyHIPNativeExt.jl -- reference-side binding of libwlhip.so (include/wlhip.h, ABI v6).
#
# NOT EXECUTED by this repository's tests: no Julia runtime exists in the build image or on the GPU box.  It is the
# shim a WaterLily maintainer would add as a package extension (compare ext/WaterLilyAMDGPUExt.jl): a device array type
# plus method overrides at function granularity, every one of them a plain `ccall`.  The Python host
# (waterlily_amd/sim.py) binds exactly the same entry points and is what the parity tests exercise.
#
#   using WaterLily, WaterLilyHIPNativeExt
#   sim = Simulation((512,512,512), (1,0,0), 128; ν=128/3700, body=AutoBody(...), T=Float32, mem=HIPArray)
#   sim_step!(sim; remeasure=false)
module WaterLilyHIPNativeExt

using WaterLily
import WaterLily: Flow, Poisson, MultiLevelPoisson, AbstractPoisson, AbstractBody, Simulation, mom_step!, conv_diff!, BDIM!,
                  project!, accelerate!, BC!, exitBC!, perBC!, scale_u!, CFL, set_diag!, mult!, residual!, increment!, Jacobi!,
                  pcg!, L₂, L∞, solver!, restrict!, prolongate!, restrictL!, Vcycle!, update!, measure!, apply!, pressure_force,
                  viscous_force, pressure_moment, BCTuple, nds, loc, inside, time, @log
using StaticArrays
import KernelAbstractions
import LinearAlgebra
using LinearAlgebra: I, inv

const lib = get(ENV, "WLHIP_LIB", "libwlhip.so")

# ---------------------------------------------------------------------------------------------- plumbing
struct WlGrid            # == wl_grid
    D::Int32; n::NTuple{3,Int32}; s::NTuple{3,Int64}; sc::Int64
    nzg::Int32; kz0::Int32; own_lo::Int32; own_hi::Int32; zring::Int32
end
struct WlLevel           # == wl_level_desc
    g::WlGrid; L::Ptr{Cvoid}; D::Ptr{Cvoid}; iD::Ptr{Cvoid}; x::Ptr{Cvoid}; eps::Ptr{Cvoid}; r::Ptr{Cvoid}; z::Ptr{Cvoid}
end
struct WlFlow            # == wl_flow_desc
    g::WlGrid; u::Ptr{Cvoid}; u0::Ptr{Cvoid}; f::Ptr{Cvoid}; p::Ptr{Cvoid}; sigma::Ptr{Cvoid}; V::Ptr{Cvoid}
    mu0::Ptr{Cvoid}; mu1::Ptr{Cvoid}; nu::Cdouble; exitBC::Int32; perdir_mask::Int32
end
chk(rc) = rc == 0 ? nothing : (rc == 10002 ? throw(AssertionError("MultiLevelPoisson requires size=a2ⁿ, where n>2")) :
                               error(unsafe_string(ccall((:wl_last_error, lib), Cstring, ()))))
dtype(::Type{Float32}) = Cint(0)
dtype(::Type{Float64}) = Cint(1)
mask(perdir) = Cint(reduce(|, (1 << (j - 1) for j in perdir); init=0))        # 1-based tuple -> 0-based bit mask
d3(A) = Cdouble[A..., 0, 0][1:3]

# ---------------------------------------------------------------------------------------------- device array
"""Device array owning memory from `wl_malloc`.  To Julia it is the reference's dense column-major array (src/Flow.jl:112-118:
`Array(a)`, `copyto!`, `similar`, `size`); on the device every x-row (first index) is PITCHED: its stride is rounded up to 128
bytes and the allocation is shifted so that the first interior element `a[2,j,k,...]` of every row sits on a 128-byte boundary
-- the layout the library's 16-byte vector kernels run 5 % faster on than on dense rows of `N+2` elements (bench.py,
`layout_dense`).  Host copies are pitched 2-D copies (`wl_h2d_2d` / `wl_d2h_2d`): the padding never reaches Julia."""
mutable struct HIPArray{T,N} <: AbstractArray{T,N}
    base::Ptr{T}              # what wl_malloc returned (freed by the finalizer)
    ptr::Ptr{T}               # element [1,1,...]: base + lead
    dims::NTuple{N,Int}
    pitch::Int                # elements between consecutive x-rows (>= dims[1], a multiple of 128 bytes)
    function HIPArray{T,N}(dims::NTuple{N,Int}) where {T,N}
        al = 128 ÷ sizeof(T)
        pitch = cld(max(dims[1], 1), al) * al
        lead = al - 1                                            # element [2,...] of a row lands on the boundary
        p = Ref{Ptr{Cvoid}}()
        bytes = (lead + pitch * max(1, prod(dims[2:end])) + al) * sizeof(T)
        chk(ccall((:wl_malloc, lib), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, bytes))
        chk(ccall((:wl_memset0, lib), Cint, (Ptr{Cvoid}, Csize_t), p[], bytes))   # (the row padding is never read; keep it defined)
        a = new{T,N}(Ptr{T}(p[]), Ptr{T}(p[]) + lead * sizeof(T), dims, pitch)
        finalizer(x -> ccall((:wl_free, lib), Cint, (Ptr{Cvoid},), x.base), a)
    end
end
rows(a::HIPArray) = max(1, prod(a.dims[2:end]))
HIPArray(h::Array{T,N}) where {T,N} = copyto!(HIPArray{T,N}(size(h)), h)       # `zeros(T,Nd) |> mem` (Flow.jl:114-118)
Base.size(a::HIPArray) = a.dims
Base.similar(a::HIPArray{T}, ::Type{S}=T, dims::Dims=a.dims) where {T,S} = HIPArray{S,length(dims)}(dims)
Base.copyto!(d::HIPArray{T}, h::Array{T}) where T =                                             # dense host rows -> pitched device rows
    (chk(ccall((:wl_h2d_2d, lib), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cvoid}, Csize_t, Csize_t, Csize_t),
               d.ptr, d.pitch * sizeof(T), h, d.dims[1] * sizeof(T), d.dims[1] * sizeof(T), rows(d))); d)
Base.copyto!(h::Array{T}, d::HIPArray{T}) where T =
    (chk(ccall((:wl_d2h_2d, lib), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Cvoid}, Csize_t, Csize_t, Csize_t),
               h, d.dims[1] * sizeof(T), d.ptr, d.pitch * sizeof(T), d.dims[1] * sizeof(T), rows(d))); h)
Base.Array(a::HIPArray{T,N}) where {T,N} = copyto!(Array{T,N}(undef, a.dims), a)
Base.copy(a::HIPArray) = HIPArray(Array(a))
Base.fill!(a::HIPArray{T}, v) where T = iszero(v) ?
    (chk(ccall((:wl_memset0, lib), Cint, (Ptr{Cvoid}, Csize_t), a.ptr, sizeof(T) * a.pitch * rows(a))); a) : copyto!(a, fill(T(v), size(a)))
# element offset of a[I...]: the first index runs along a row, all the others count rows
offset(a::HIPArray{T,N}, I::Vararg{Int,N}) where {T,N} = (I[1] - 1) + a.pitch * (N > 1 ? LinearIndices(a.dims[2:end])[I[2:end]...] - 1 : 0)
offset(a::HIPArray, i::Int) = offset(a, Tuple(CartesianIndices(a.dims)[i])...)               # linear index
# scalar access = one-element transfers: correct, slow -- only what generic serial code (`julia -t 1` @loop bodies,
# tests in GPUArrays.@allowscalar style) falls back to
function Base.getindex(a::HIPArray{T}, I::Vararg{Int}) where T
    r = Ref{T}(); o = offset(a, I...)
    chk(ccall((:wl_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), r, a.ptr + o * sizeof(T), sizeof(T))); r[]
end
function Base.setindex!(a::HIPArray{T}, v, I::Vararg{Int}) where T
    r = Ref{T}(T(v)); o = offset(a, I...)
    chk(ccall((:wl_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), a.ptr + o * sizeof(T), r, sizeof(T))); a
end
Base.getindex(a::HIPArray, I::CartesianIndex) = a[Tuple(I)...]
Base.setindex!(a::HIPArray, v, I::CartesianIndex) = (a[Tuple(I)...] = v)

# ---- Base / LinearAlgebra generics the path calls on arrays (SURVEY.md 8b): device reductions where the library has one.
# wl_dot / wl_sum / wl_max reduce over the WHOLE array, ghost cells included, like the Base generics they stand in for
# (Poisson.jl:94,126-146; Flow.jl:174).  Vector fields (trailing component axis) take the host route.
# (the function name of a `ccall` has to be a constant: one method per entry point, generated here)
for (fn, sym) in ((:lib_sum, :wl_sum), (:lib_max, :wl_max))
    @eval function $fn(a::HIPArray{T}) where T
        o = Ref{Cdouble}()
        chk(ccall(($(QuoteNode(sym)), lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ref{Cdouble}), dtype(T), grid(a), a.ptr, o))
        T(o[])
    end
end
function lib_dot(a::HIPArray{T}, b::HIPArray{T}) where T
    o = Ref{Cdouble}()
    chk(ccall((:wl_dot, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}), dtype(T), grid(a), a.ptr, b.ptr, o))
    T(o[])
end
Base.sum(a::HIPArray{T,N}) where {T,N} = N <= 3 ? lib_sum(a) : sum(Array(a))
Base.maximum(a::HIPArray{T,N}) where {T,N} = N <= 3 ? lib_max(a) : maximum(Array(a))
LinearAlgebra.dot(a::HIPArray{T,N}, b::HIPArray{T,N}) where {T,N} = N <= 3 ? lib_dot(a, b) : LinearAlgebra.dot(Array(a), Array(b))

# ---- broadcast (`.=`, `.*=`, `./=`, Flow.jl:37,139,144,154 -- inside mom_step! these never run: the override below
# replaces the whole step): host fallback, D2H -> Base broadcast -> H2D
struct HIPStyle <: Broadcast.AbstractArrayStyle{Any} end
HIPStyle(::Val) = HIPStyle()
Base.BroadcastStyle(::Type{<:HIPArray}) = HIPStyle()
hostify(x) = x
hostify(x::HIPArray) = Array(x)
hostify(bc::Broadcast.Broadcasted) = Broadcast.broadcasted(bc.f, map(hostify, bc.args)...)
Base.similar(bc::Broadcast.Broadcasted{HIPStyle}, ::Type{T}) where T = HIPArray{T,length(axes(bc))}(map(length, axes(bc)))
Base.copyto!(d::HIPArray, bc::Broadcast.Broadcasted{HIPStyle}) = copyto!(d, Array{eltype(d)}(Broadcast.materialize(hostify(bc))))
Base.copyto!(d::HIPArray{T}, bc::Broadcast.Broadcasted{<:Broadcast.AbstractArrayStyle{0}}) where T = fill!(d, bc[])

# ---- generic @loop / @inside on HIPArray arguments (src/util.jl:119-141): with more than one thread the macro launches a
# KernelAbstractions kernel on `get_backend(first array)`.  This backend stages the call through the host: D2H every
# HIPArray argument, run the SAME kernel on KernelAbstractions' CPU backend (the reference Array path), H2D the arrays
# back.  It makes user code such as `@inside p[I] = ke(I,u)` (Metrics.jl:14-77) or `measure_sdf!` (Body.jl:68) work
# unchanged on device fields; every operator of the hot path has a native override and never comes here.
struct HostStaged <: KernelAbstractions.Backend end
KernelAbstractions.get_backend(::HIPArray) = HostStaged()
KernelAbstractions.synchronize(::HostStaged) = chk(ccall((:wl_sync, lib), Cint, ()))
function (k::KernelAbstractions.Kernel{HostStaged})(args...; ndrange=nothing, workgroupsize=nothing)
    host = map(hostify, args)
    kc = KernelAbstractions.Kernel{KernelAbstractions.CPU,typeof(k).parameters[2],typeof(k).parameters[3],typeof(k.f)}(KernelAbstractions.CPU(), k.f)
    kc(host...; ndrange, workgroupsize)
    KernelAbstractions.synchronize(KernelAbstractions.CPU())
    foreach((d, h) -> d isa HIPArray && copyto!(d, h), args, host)
    nothing
end

# wl_grid of a field whose first D axes are spatial (trailing axes: components, one pitched block of rows each)
grid(a::HIPArray, D=ndims(a)) = (n = (size(a)[1:D]..., ntuple(_ -> 1, 3 - D)...);
    WlGrid(D, Int32.(n), (1, a.pitch, a.pitch * n[2]), a.pitch * n[2] * n[3], 0, 0, 0, 0, 0))

# ---------------------------------------------------------------------------------------------- util.jl
BC!(a::HIPArray{T}, A, saveexit=false, perdir=()) where T =                                  # src/util.jl:192-210
    chk(ccall((:wl_bc_vec, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint),
              dtype(T), grid(a, ndims(a) - 1), a.ptr, d3(A), saveexit, mask(perdir)))
perBC!(a::HIPArray, ::Tuple{}) = nothing
perBC!(a::HIPArray{T}, perdir, N=size(a)) where T =                                           # src/util.jl:227-231
    chk(ccall((:wl_bc_per, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Cint), dtype(T), grid(a), a.ptr, mask(perdir)))
exitBC!(u::HIPArray{T}, u⁰::HIPArray{T}, U, Δt) where T =                                      # src/util.jl:216-222
    chk(ccall((:wl_exit_bc, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cdouble),
              dtype(T), grid(u, ndims(u) - 1), u.ptr, u⁰.ptr, d3(U), Δt))
function L₂(a::HIPArray{T}) where T                                                            # src/util.jl:68
    o = Ref{Cdouble}()
    chk(ccall((:wl_L2_inside, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ref{Cdouble}), dtype(T), grid(a), a.ptr, o)); o[]
end
function apply!(f, c::HIPArray)                                                                # src/util.jl:170-172
    h = Array(c); apply!(f, h); copyto!(c, h)                                                  # user closure: host, then upload
end

# ---------------------------------------------------------------------------------------------- Flow.jl operators
conv_diff!(r::HIPArray{T}, u::HIPArray{T}, Φ::HIPArray{T}; ν=0.1, perdir=()) where T =          # src/Flow.jl:36-60
    chk(ccall((:wl_conv_diff, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint),
              dtype(T), grid(u, ndims(u) - 1), r.ptr, u.ptr, Φ.ptr, ν, mask(perdir)))         # (Φ's top ghost cells: what the scatter form leaves)
BDIM!(a::Flow{N,T,<:HIPArray}) where {N,T} =                                                    # src/Flow.jl:131-135
    chk(ccall((:wl_bdim, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
              Ptr{Cvoid}, Cdouble), dtype(T), grid(a.p), a.u.ptr, a.u⁰.ptr, a.f.ptr, a.V.ptr, a.μ₀.ptr, a.μ₁.ptr, a.Δt[end]))
# accelerate!(r,dt,g,U)  src/Flow.jl:68-73: the host evaluates g(i,t) + dU_i/dt (closures, ForwardDiff), the library adds it
accelerate!(r::HIPArray{T}, dt, g::Function, ::Tuple) where T = accel!(r, i -> g(i, sum(dt)))
accelerate!(r::HIPArray{T}, dt, g::Nothing, U::Function) where T = accel!(r, i -> WaterLily.ForwardDiff.derivative(τ -> U(i, τ), sum(dt)))
accelerate!(r::HIPArray{T}, dt, g::Function, U::Function) where T =
    accel!(r, i -> g(i, sum(dt)) + WaterLily.ForwardDiff.derivative(τ -> U(i, τ), sum(dt)))
accelerate!(r::HIPArray, dt, ::Nothing, ::Tuple) = nothing
accel!(r::HIPArray{T}, f) where T =
    chk(ccall((:wl_accelerate, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cdouble}), dtype(T), grid(r, ndims(r) - 1), r.ptr,
              d3(ntuple(f, ndims(r) - 1))))
scale_u!(a::Flow{N,T,<:HIPArray}, scale) where {N,T} =                                          # src/Flow.jl:170
    chk(ccall((:wl_scale_u, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Cdouble), dtype(T), grid(a.p), a.u.ptr, scale))
function CFL(a::Flow{N,T,<:HIPArray}; Δt_max=10) where {N,T}                                    # src/Flow.jl:172-175
    o = Ref{Cdouble}()
    chk(ccall((:wl_cfl, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ref{Cdouble}),
              dtype(T), grid(a.p), a.σ.ptr, a.u.ptr, a.ν, o)); T(o[])
end

# ---------------------------------------------------------------------------------------------- handles
const MG = IdDict{Any,Ptr{Cvoid}}()
const FL = IdDict{Any,Ptr{Cvoid}}()
level(l) = WlLevel(grid(l.x), l.L.ptr, l.D.ptr, l.iD.ptr, l.x.ptr, l.ϵ.ptr, l.r.ptr, l.z.ptr)
function handle(p::AbstractPoisson{T,<:HIPArray}) where T                                       # Poisson.jl:31 / MultiLevelPoisson.jl:51
    get!(MG, p) do
        lv = p isa MultiLevelPoisson ? [level(l) for l in p.levels] : [level(p)]
        h = Ref{Ptr{Cvoid}}()
        chk(ccall((:wl_mg_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Ptr{WlLevel}, Cint), h, dtype(T), length(lv), lv,
                  mask(p.perdir)))
        h[]
    end
end
function handle(a::Flow{N,T,<:HIPArray}) where {N,T}
    get!(FL, a) do
        d = WlFlow(grid(a.p), a.u.ptr, a.u⁰.ptr, a.f.ptr, a.p.ptr, a.σ.ptr, a.V.ptr, a.μ₀.ptr, a.μ₁.ptr, a.ν, a.exitBC, mask(a.perdir))
        h = Ref{Ptr{Cvoid}}()
        chk(ccall((:wl_flow_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint, Ref{WlFlow}), h, dtype(T), d)); h[]
    end
end
lvl(p, l) = p isa MultiLevelPoisson ? l - 1 : 0

# ---------------------------------------------------------------------------------------------- Poisson.jl / MultiLevelPoisson.jl
set_diag!(D::HIPArray{T}, iD::HIPArray{T}, L::HIPArray{T}) where T =                            # src/Poisson.jl:42-45
    chk(ccall((:wl_set_diag, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), dtype(T), grid(D), D.ptr, iD.ptr, L.ptr))
update!(p::AbstractPoisson{T,<:HIPArray}) where T = chk(ccall((:wl_mg_update, lib), Cint, (Ptr{Cvoid},), handle(p)))
# update!(pois) right after a NATIVE measure!(flow, ::ParametricBody): only the x-rows that measure! rewrote are revisited
# on the finest level (same values; the library falls back to the full update by itself when that is not applicable)
update!(p::AbstractPoisson{T,<:HIPArray}, a::Flow{N,T,<:HIPArray}) where {N,T} =
    chk(ccall((:wl_mg_update_changed, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(p), handle(a)))
mult!(p::AbstractPoisson{T,<:HIPArray}, x::HIPArray) where T =                                  # src/Poisson.jl:62-68
    (chk(ccall((:wl_mg_mult, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), handle(p), 0, x.ptr)); p.z)
residual!(p::Poisson{T,<:HIPArray}) where T = chk(ccall((:wl_mg_residual, lib), Cint, (Ptr{Cvoid}, Cint), handle(p), 0))
increment!(p::Poisson{T,<:HIPArray}) where T = chk(ccall((:wl_mg_increment, lib), Cint, (Ptr{Cvoid}, Cint), handle(p), 0))
Jacobi!(p::Poisson{T,<:HIPArray}; it=1) where T = chk(ccall((:wl_mg_jacobi, lib), Cint, (Ptr{Cvoid}, Cint, Cint), handle(p), 0, it))
pcg!(p::Poisson{T,<:HIPArray}; it=6) where T =
    chk(ccall((:wl_mg_pcg, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}), handle(p), 0, it, C_NULL))
function L₂(p::Poisson{T,<:HIPArray}) where T                                                   # src/Poisson.jl:146
    o = Ref{Cdouble}(); chk(ccall((:wl_mg_L2, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), handle(p), 0, o)); T(o[])
end
function L∞(p::Poisson{T,<:HIPArray}) where T                                                   # src/Poisson.jl:147
    o = Ref{Cdouble}(); chk(ccall((:wl_mg_Linf, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), handle(p), 0, o)); T(o[])
end
L₂(ml::MultiLevelPoisson{T,<:HIPArray}) where T = (o = Ref{Cdouble}();
    chk(ccall((:wl_mg_L2, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), handle(ml), 0, o)); T(o[]))
L∞(ml::MultiLevelPoisson{T,<:HIPArray}) where T = (o = Ref{Cdouble}();
    chk(ccall((:wl_mg_Linf, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), handle(ml), 0, o)); T(o[]))
Vcycle!(ml::MultiLevelPoisson{T,<:HIPArray}; l=1) where T = chk(ccall((:wl_mg_vcycle, lib), Cint, (Ptr{Cvoid}, Cint), handle(ml), l - 1))
restrict!(a::HIPArray{T}, b::HIPArray{T}) where T =                                             # MultiLevelPoisson.jl:33
    chk(ccall((:wl_restrict, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ref{WlGrid}, Ptr{Cvoid}), dtype(T), grid(a), a.ptr, grid(b), b.ptr))
prolongate!(a::HIPArray{T}, b::HIPArray{T}) where T =                                           # MultiLevelPoisson.jl:34
    chk(ccall((:wl_prolongate, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ref{WlGrid}, Ptr{Cvoid}), dtype(T), grid(a), a.ptr, grid(b), b.ptr))
restrictL!(a::HIPArray{T}, b::HIPArray{T}; perdir=()) where T =                                 # MultiLevelPoisson.jl:26-32
    chk(ccall((:wl_restrictL, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ref{WlGrid}, Ptr{Cvoid}, Cint),
              dtype(T), grid(a, ndims(a) - 1), a.ptr, grid(b, ndims(b) - 1), b.ptr, mask(perdir)))
# The pressure-solver log (`WaterLily.logger`, src/util.jl:11-24): when the custom log level is enabled the library records
# {n, L∞(p), L₂(p)} per iteration (wl_mg_log) and the rows are re-emitted through the reference's own @log macro, in the
# reference's format (Poisson.jl:164,167; MultiLevelPoisson.jl:90,94)
logging_on() = Base.CoreLogging.min_enabled_level(Base.CoreLogging.current_logger()) <= WaterLily._psolver
function emit_log(h)
    n = Ref{Cint}()
    chk(ccall((:wl_mg_log_read, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Cint}), h, C_NULL, 0, n))   # cap 0: how many rows wait
    rows = zeros(Cdouble, 3 * max(n[], 1))
    chk(ccall((:wl_mg_log_read, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Cint}), h, rows, n[], n))
    for q in 1:n[]
        @log ", $(Int(rows[3q-2])), $(rows[3q-1]), $(rows[3q])\n"
    end
end
function solver!(p::AbstractPoisson{T,<:HIPArray}; tol=1e-4, itmx=(p isa MultiLevelPoisson ? 32 : 1e3)) where T
    n = Ref{Cint}(); h = handle(p); lg = logging_on()                                           # MultiLevelPoisson.jl:87 / Poisson.jl:162
    chk(ccall((:wl_mg_log, lib), Cint, (Ptr{Cvoid}, Cint), h, lg))
    chk(ccall((:wl_mg_solve, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint, Ref{Cint}), h, tol, Int(itmx), n))
    lg && emit_log(h)
    push!(p.n, n[])
end

# ---------------------------------------------------------------------------------------------- Flow.jl drivers
function accel(a::Flow{N}, dt) where N                                                          # accelerate! (src/Flow.jl:68-73)
    (a.g === nothing && a.U isa Tuple) && return C_NULL
    t = sum(dt)
    g(i) = (a.g === nothing ? 0.0 : a.g(i, t)) + (a.U isa Function ? WaterLily.ForwardDiff.derivative(τ -> a.U(i, τ), t) : 0.0)
    d3(ntuple(g, N))
end
function project!(a::Flow{N,T,<:HIPArray}, b::AbstractPoisson, w=1) where {N,T}                 # src/Flow.jl:137-145
    n = Ref{Cint}()
    chk(ccall((:wl_project, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Cint}), handle(a), handle(b), a.Δt[end], w, n))
    push!(b.n, n[])
end
function mom_step!(a::Flow{N,T,<:HIPArray}, b::AbstractPoisson) where {N,T}                     # src/Flow.jl:153-169
    U = d3(BCTuple(a.U, a.Δt, N))
    gp, gc = accel(a, @view(a.Δt[1:end-1])), accel(a, a.Δt)
    dt, n2 = Ref{Cdouble}(), zeros(Cint, 2)
    hb = handle(b); lg = logging_on()
    chk(ccall((:wl_mg_log, lib), Cint, (Ptr{Cvoid}, Cint), hb, lg))
    chk(ccall((:wl_mom_step, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
              Ref{Cdouble}, Ptr{Cint}), handle(a), hb, a.Δt[end], U, gp, gc, dt, n2))
    if lg   # the rows of the predictor's solve, then the corrector's: each starts with n = 0 (`@log "p"` / `@log "c"`, Flow.jl:158,165)
        n = Ref{Cint}()
        chk(ccall((:wl_mg_log_read, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Cint}), hb, C_NULL, 0, n))   # how many rows wait
        rows = zeros(Cdouble, 3 * max(n[], 1))
        chk(ccall((:wl_mg_log_read, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Cint}), hb, rows, n[], n))
        tag = ("p", "c"); k = 0
        for q in 1:n[]
            rows[3q-2] == 0 && (k += 1; @log tag[min(k, 2)])
            @log ", $(Int(rows[3q-2])), $(rows[3q-1]), $(rows[3q])\n"
        end
    end
    append!(b.n, n2); push!(a.Δt, T(dt[]))
end

# ---------------------------------------------------------------------------------------------- Body.jl / Metrics.jl
function measure!(a::Flow{N,T,<:HIPArray}, body::AbstractBody; t=zero(T), ϵ=1) where {N,T}     # src/Body.jl:31-53
    h = Flow(size(a.p) .- 2, a.U; f=Array, T, perdir=a.perdir, exitBC=a.exitBC)                 # host scratch with the same shape
    measure!(h, body; t, ϵ)                                                                      # user closures + ForwardDiff on the host
    copyto!(a.μ₀, h.μ₀); copyto!(a.μ₁, h.μ₁); copyto!(a.V, h.V); copyto!(a.σ, h.σ)
    chk(ccall((:wl_flow_update, lib), Cint, (Ptr{Cvoid},), handle(a)))                          # rebuild the body-free row flags
end
# ---- parametric bodies: closed-form sdf family + affine map  =>  the whole measure! runs in the library (csrc/wl_measure.h)
struct WlBody            # == wl_body_desc (one leaf; a composite is a Vector{WlBody}: element 1 carries count, the others op)
    family::Int32; identity_map::Int32; p::NTuple{8,Cdouble}
    A::NTuple{9,Cdouble}; b::NTuple{3,Cdouble}; dA::NTuple{9,Cdouble}; db::NTuple{3,Cdouble}; Ainv::NTuple{9,Cdouble}
    op::Int32; count::Int32
end
"""
    ParametricBody(family, params; map=nothing)

`family` ∈ (:sphere, :torus, :plate, :cylinder) with `params` = (c..., radius) | (c₁,c₂,c₃,R,r) | (a, thk) |
(c₁,c₂,c₃,radius,m₁,m₂,m₃) (include/wlhip.h);
`map(t)` returns the affine map ξ = A x + b at time t as `(A, b)` (D×D matrix, D-vector).  Its time derivative comes from
ForwardDiff, like `measure` gets `dot` (src/AutoBody.jl:128).  `sdf`/`measure` fall back to the equivalent `AutoBody`, so
every generic code path (host `measure!`, `nds`, plotting) still works.
"""
struct ParametricBody{F} <: AbstractBody
    family::Symbol; params::Vector{Float64}; map::F; auto::WaterLily.AutoBody
end
function ParametricBody(family::Symbol, params; map=nothing)
    P = Float64.(collect(params))
    sdf = family == :sphere ? (ξ, t) -> √sum(abs2, ξ .- P[1:length(ξ)]) - P[4] :
          family == :torus  ? (ξ, t) -> √((ξ[1] - P[1])^2 + (√((ξ[2] - P[2])^2 + (ξ[3] - P[3])^2) - P[4])^2) - P[5] :
          family == :cylinder ? (ξ, t) -> √sum(ntuple(q -> P[4+q] * (ξ[q] - P[q])^2, length(ξ))) - P[4] :   # m = 1 on the axes of the circle
                              (ξ, t) -> √sum(abs2, ξ .- SVector(clamp(ξ[1], -P[1], P[1]), ntuple(_ -> 0, length(ξ) - 1)...)) - P[2]
    amap = map === nothing ? ((x, t) -> x) : ((x, t) -> ((A, b) = map(t); A * x + b))
    ParametricBody(family, P, map, WaterLily.AutoBody(sdf, amap))
end
WaterLily.sdf(b::ParametricBody, x, t=0; kw...) = WaterLily.sdf(b.auto, x, t; kw...)
WaterLily.measure(b::ParametricBody, x, t; kw...) = WaterLily.measure(b.auto, x, t; kw...)
pad9(M, D) = ntuple(q -> ((r, c) = divrem(q - 1, 3); (r < D && c < D) ? Float64(M[r+1, c+1]) : 0.0), 9)   # row-major 3x3
pad3(v, D) = ntuple(q -> q <= D ? Float64(v[q]) : 0.0, 3)
function desc(body::ParametricBody, t, D)
    fam = Int32(body.family == :sphere ? 0 : body.family == :torus ? 1 : body.family == :plate ? 2 : 3)
    p8 = ntuple(q -> q <= length(body.params) ? body.params[q] : 0.0, 8)
    body.map === nothing && return WlBody(fam, 1, p8, pad9(I(D), D), pad3(zeros(D), D), pad9(zeros(D, D), D), pad3(zeros(D), D), pad9(I(D), D), 0, 1)
    A, b = body.map(t)
    dA = WaterLily.ForwardDiff.derivative(τ -> body.map(τ)[1], t); db = WaterLily.ForwardDiff.derivative(τ -> body.map(τ)[2], t)
    WlBody(fam, 0, p8, pad9(A, D), pad3(b, D), pad9(dA, D), pad3(db, D), pad9(inv(A), D), 0, 1)
end
# `Bodies` of parametric leaves (src/AutoBody.jl:40-110) -> one descriptor array: a maintainer would dispatch
# measure!(::Flow{N,T,<:HIPArray}, ::Bodies) here when every leaf is a ParametricBody, passing Vector{WlBody} to the same entry points
opcode(f) = (f === Base.:+ || f === Base.:∪) ? Int32(0) : f === Base.:- ? Int32(1) : Int32(2)
function desc(leaves::Vector{<:ParametricBody}, ops, t, D)
    v = [desc(b, t, D) for b in leaves]
    [WlBody(d.family, d.identity_map, d.p, d.A, d.b, d.dA, d.db, d.Ainv, l == 1 ? Int32(0) : opcode(ops[l-1]), l == 1 ? Int32(length(v)) : Int32(0))
     for (l, d) in enumerate(v)]
end
const BAND = IdDict{Any,Any}()       # flow => (t, band cells) of the last native measure!
function measure!(a::Flow{N,T,<:HIPArray}, body::ParametricBody; t=zero(T), ϵ=1) where {N,T}     # src/Body.jl:31-53, all on the device
    d = desc(body, t, N); nb = Ref{Int64}()
    chk(ccall((:wl_measure_rows, lib), Cint, (Ptr{Cvoid}, Ref{WlBody}, Cdouble, Ref{Int64}), handle(a), d, ϵ, nb))
    cand = HIPArray{Int64,1}((max(1, nb[]),))
    chk(ccall((:wl_measure_fill, lib), Cint, (Ptr{Cvoid}, Ref{WlBody}, Cdouble, Ptr{Int64}), handle(a), d, ϵ, cand.ptr))
    BAND[a] = (t, cand, nb[])
    nothing
end
# measure!(sim, t)  src/WaterLily.jl:116-119: the native measure! is followed by the changed-rows update!(pois)
# (`Simulation` carries no type parameters, WaterLily.jl:59-65: the method is the reference's own for every other simulation)
function measure!(sim::Simulation, t=sum(sim.flow.Δt))
    measure!(sim.flow, sim.body; t, ϵ=sim.ϵ)
    (sim.flow.p isa HIPArray && sim.body isa ParametricBody) ? update!(sim.pois, sim.flow) : update!(sim.pois)
end
# nds band of a parametric body (Metrics.jl:84-87): wl_body_nds on the band cells measure! listed; returns element offsets + vectors
function band(p::HIPArray, body::ParametricBody, t)
    D = ndims(p); Is = collect(inside(p))
    cells = Int64[LinearIndices(size(p))[I] - 1 for I ∈ Is]            # dense cell indices i + n₀(j + n₁k): what wl_body_nds takes
    offs = Int64[offset(p, Tuple(I)...) for I ∈ Is]                    # element offsets in the PITCHED field: what wl_pforce takes
    cand = HIPArray(cells); v = HIPArray{Float64,1}((D * length(cells),))   # (a maintainer would keep BAND[flow][2]: the |d|<2+ϵ cells)
    chk(ccall((:wl_body_nds, lib), Cint, (Ref{WlGrid}, Ref{WlBody}, Ptr{Int64}, Int64, Ptr{Cdouble}), grid(p), desc(body, t, D),
              cand.ptr, length(cells), v.ptr))
    HIPArray(offs), v
end
function band(p::HIPArray, body, t)                                                             # Metrics.jl:84-87 on the |d|<=1 band
    T = promote_type(Float64, eltype(p)); idx = Int64[]; v = Float64[]
    for I ∈ inside(p)
        n = nds(body, loc(0, I, T), t)
        any(!iszero, n) && (push!(idx, offset(p, Tuple(I)...)); append!(v, n))   # element offset in the pitched field
    end
    HIPArray(idx), HIPArray(v)
end
function pressure_force(p::HIPArray{T}, df, body, t=0, ::Type=Float64) where T                  # src/Metrics.jl:94-100
    idx, v = band(p, body, t); o = zeros(Cdouble, 3)
    chk(ccall((:wl_pforce, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}, Int64, Ptr{Cdouble}),
              dtype(T), grid(p), p.ptr, idx.ptr, v.ptr, length(idx), o)); o[1:ndims(p)]
end
function viscous_force(u::HIPArray{T}, ν, df, body, t=0, ::Type=Float64) where T                # src/Metrics.jl:109-113
    p1 = HIPArray{T,ndims(u)-1}(size(u)[1:end-1]); idx, v = band(p1, body, t); o = zeros(Cdouble, 3)
    chk(ccall((:wl_vforce, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}, Int64, Cdouble, Ptr{Cdouble}),
              dtype(T), grid(u, ndims(u) - 1), u.ptr, idx.ptr, v.ptr, length(idx), ν, o)); o[1:ndims(u)-1]
end
function pressure_moment(x₀, p::HIPArray{T}, df, body, t=0, ::Type=Float64) where T             # src/Metrics.jl:130-134
    idx, v = band(p, body, t); o = zeros(Cdouble, 3)
    chk(ccall((:wl_pmoment, lib), Cint, (Cint, Ref{WlGrid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Cdouble}),
              dtype(T), grid(p), p.ptr, idx.ptr, v.ptr, length(idx), d3(x₀), o)); o[1:ndims(p)]
end

# ---------------------------------------------------------------------------------------------- multi-GPU bootstrap
"""
    init_slabs!(bcast)  ->  (rank, nranks) must already be known to the caller (MPI.jl, Distributed, ...)

One process per GPU.  Rank 0 draws the RCCL unique id, `bcast(::Vector{UInt8})` hands the 128 bytes to every rank (e.g.
`MPI.Bcast!(id, 0, comm)`), then every rank joins the communicator.  Afterwards a Flow / Poisson built on arrays whose
`grid` carries the slab fields (`slab_grid`) runs the z-slab path: halo exchanges (ncclSend/ncclRecv on the library's comm
stream), one ncclAllReduce per dot product, ncclAllGather at the hand-over to the replicated coarse levels.
"""
function init_slabs!(bcast, rank::Integer, nranks::Integer; device=rank, allmin=nothing)
    chk(ccall((:wl_set_device, lib), Cint, (Cint,), device))
    id = zeros(UInt8, 128)
    rank == 0 && chk(ccall((:wl_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    bcast(id)
    chk(ccall((:wl_comm_init_rccl, lib), Cint, (Ptr{UInt8}, Cint, Cint), id, rank, nranks))
    # Scalars (dot products, CFL maximum, force sums) through the shared-memory mailbox instead of one ncclAllReduce each --
    # only when the caller supplies `allmin(x::Int)`, the minimum of x over the ranks (e.g. x -> MPI.Allreduce(x, MPI.MIN, comm)):
    # a true synchronisation AND the all-or-nothing vote.  (A broadcast is neither: its root may return before the others
    # have even entered it.)  Without it the scalars stay on ncclAllReduce.
    allmin === nothing && return
    name = zeros(UInt8, 64)
    ok = 1
    if rank == 0
        nm = "/wlhip-$(getpid())-$(rand(UInt32))"
        copyto!(name, 1, codeunits(nm), 1, ncodeunits(nm))
        ok = ccall((:wl_comm_mailbox, lib), Cint, (Cstring, Cint), nm, 1) == 0 ? 1 : 0
    end
    bcast(name)                                        # the name exists before anybody else looks for it (rank 0 created it first)
    nm = unsafe_string(pointer(name))
    rank == 0 || (ok = ccall((:wl_comm_mailbox, lib), Cint, (Cstring, Cint), nm, 0) == 0 ? 1 : 0)
    ok = allmin(ok)                                    # everybody has tried to map it, and everybody knows whether all succeeded
    rank == 0 && rm("/dev/shm" * nm; force=true)       # (the mappings keep the memory alive)
    if ok == 1                                         # self-test with a short bound before the run depends on it
        keep = Ref{Cint}()
        chk(ccall((:wl_get_option, lib), Cint, (Cint, Ref{Cint}), 26, keep))
        chk(ccall((:wl_set_option, lib), Cint, (Cint, Cint), 26, 10))
        v = Cdouble[rank + 1]
        good = ccall((:wl_allreduce, lib), Cint, (Ptr{Cdouble}, Cint, Cint), v, 1, 0) == 0 && v[1] == nranks * (nranks + 1) / 2
        chk(ccall((:wl_set_option, lib), Cint, (Cint, Cint), 26, keep[]))
        ok = allmin(good ? 1 : 0)
    end
    ok == 1 || chk(ccall((:wl_comm_mailbox_off, lib), Cint, ()))   # all or nothing: every rank takes the same path
    nothing
end
finalize_slabs!() = chk(ccall((:wl_comm_finalize, lib), Cint, ()))
"""wl_grid of rank `r`'s z-slab `a` (local extents `(Ng[1], Ng[2], nz/P + 4)`) of an undecomposed array of extents `Ng`
(ghosts included): nz/P interior planes + 2 halo planes per side (QUICK reads I-2δ..I+δ, src/Flow.jl:6); rank 0 owns the
lower ghost plane, rank P-1 the upper one."""
function slab_grid(a::HIPArray, Ng::NTuple{3,Int}, r, P; ring=false)
    nzl = (Ng[3] - 2) ÷ P; n = (Ng[1], Ng[2], nzl + 4)
    @assert size(a)[1:3] == n
    WlGrid(3, Int32.(n), (1, a.pitch, a.pitch * n[2]), a.pitch * n[2] * n[3], Ng[3], r * nzl + 1 - 2, 2 - ((r == 0 && !ring) ? 1 : 0),
           2 + nzl - 1 + ((r == P - 1 && !ring) ? 1 : 0), ring)
end

export HIPArray, ParametricBody, init_slabs!, finalize_slabs!, slab_grid
end # module

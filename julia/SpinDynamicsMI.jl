# SpinDynamicsMI.jl -- thin `ccall` shim that keeps the user-facing calls of
# javahedi/SpinDynamics.jl (XXZChain / groundstate / time_evolve /
# dynamical_structure_factor, plus the operator seam apply_H!) and routes the
# hot path to libspindyn.so (hand-written HIP for gfx950, include/spindyn.h).
#
# STATUS: written against the C ABI but NOT executed -- there is no Julia
# runtime in the build container or on the GPU box (SURVEY.md 8c).  It is kept
# deliberately thin: every numerical statement lives behind the C ABI, which is
# what the parity tests exercise (through the Python mirror
# spindynamics.jl_amd/, call for call the same entry points).
#
# Reference functions mirrored (file:line in the reference repository):
#   XXZChain, build_model, momenta          src/SpinModel.jl:23-38,63-90,97-99
#   apply_H!, apply_rescaled_H!, Sz_q_vector src/Hamiltonian.jl:211-273,286-301,307-337
#   groundstate, time_evolve, dynamical_structure_factor   src/PublicAPI.jl:25-155
module SpinDynamicsMI

using Random
using Libdl

export Model, build_model, XXZChain, momenta, apply_H!, apply_rescaled_H!, Sz_q_vector, create_spin_operator,
       groundstate, time_evolve, structure_factor, dynamical_structure_factor,
       magnetization_per_site, connected_correlations, structure_factor_Sq,
       domain_wall_state, neel_state, polarized_state, polarized_state_with_flips

const libspindyn = get(ENV, "SPINDYN_LIB", joinpath(@__DIR__, "..", "spindynamics.jl_amd", "libspindyn.so"))

const SD_F64, SD_C128 = Cint(1), Cint(2)
dtype_code(::Type{Float64}) = SD_F64
dtype_code(::Type{ComplexF64}) = SD_C128
dtype_code(::Type{T}) where {T} = throw(ArgumentError("libspindyn supports Float64 and ComplexF64 vectors, got $T"))

# ---- status codes -> the exception types the reference throws ----------------
function check(rc::Cint, ctx::Ptr{Cvoid}=C_NULL)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:sd_last_error, libspindyn), Cstring, (Ptr{Cvoid},), ctx))
    isempty(msg) && (msg = unsafe_string(ccall((:sd_status_string, libspindyn), Cstring, (Cint,), rc)))
    rc == 1 && throw(ArgumentError(msg))          # src/Basis.jl:10-16, src/SpinModel.jl:80, src/PublicAPI.jl:34,87,152
    rc == 2 && throw(DimensionMismatch(msg))      # src/Hamiltonian.jl:63-66,220,289
    rc == 3 && error("starting vector has zero norm")   # src/Lanczos.jl:210-212
    error("libspindyn status $rc: $msg")
end

# ---- context (one per process / per GPU) ---------------------------------------
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=parse(Int, get(ENV, "LOCAL_RANK", "0")))
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:sd_ctx_create, libspindyn), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[])
        finalizer(x -> ccall((:sd_ctx_destroy, libspindyn), Cvoid, (Ptr{Cvoid},), x.h), c)
        return c
    end
end
const _ctx = Ref{Union{Nothing,Context}}(nothing)
default_context() = (_ctx[] === nothing && (_ctx[] = Context()); _ctx[])

# ---- Model: same descriptor fields as SpinModel.Model minus states/idxmap ----------
# Context options (include/spindyn.h): Chebyshev moments two per apply (default) or the reference's one-per-apply loop;
# release of the device vectors a context keeps between calls (staging of sd_apply, pooled work vectors of the recursions).
set_kpm_doubling!(ctx::Context, on::Bool) =
    check(ccall((:sd_ctx_set_kpm_doubling, libspindyn), Cint, (Ptr{Cvoid}, Cint), ctx.h, on ? 1 : 0), ctx.h)
# real psi0: S(q, w) once per pair (q, 2pi - q) (default) or every q on its own as src/KPM_Sqw.jl:218-252 does
set_kpm_pair_q!(ctx::Context, on::Bool) =
    check(ccall((:sd_ctx_set_kpm_pair_q, libspindyn), Cint, (Ptr{Cvoid}, Cint), ctx.h, on ? 1 : 0), ctx.h)
# S(q, w): the momenta's vectors share the launches of their recursions at launch-bound sizes (default), or one momentum at a time
set_q_batch!(ctx::Context, on::Bool) =
    check(ccall((:sd_ctx_set_q_batch, libspindyn), Cint, (Ptr{Cvoid}, Cint), ctx.h, on ? 1 : 0), ctx.h)
# groundstate: full re-orthogonalisation in blocks of 8 columns (default) or column by column as src/Lanczos.jl:116-124
set_gs_blocked!(ctx::Context, on::Bool) =
    check(ccall((:sd_ctx_set_gs_blocked, libspindyn), Cint, (Ptr{Cvoid}, Cint), ctx.h, on ? 1 : 0), ctx.h)
# operator applications the recursion-level calls have queued on this context so far (one per recursion step)
apply_count(ctx::Context) = Int(ccall((:sd_ctx_apply_count, libspindyn), Int64, (Ptr{Cvoid},), ctx.h))
release_scratch!(ctx::Context) = check(ccall((:sd_ctx_release_scratch, libspindyn), Cint, (Ptr{Cvoid},), ctx.h), ctx.h)

mutable struct Model
    L::Int
    nup::Union{Nothing,Int}
    mode::Symbol
    hopping_list::Vector{Tuple{Int,Int,Float64}}
    onsite_field::Vector{Float64}
    zz_list::Vector{Tuple{Int,Int,Float64}}
    ctx::Context
    h::Ptr{Cvoid}
end

function build_model(L::Int; nup::Union{Nothing,Int}=nothing, hopping=[], onsite_field=zeros(L), zz=[])
    ctx = default_context()
    hop = [(Int(i), Int(j), Float64(J)) for (i, j, J) in hopping]
    zzl = [(Int(i), Int(j), Float64(J)) for (i, j, J) in zz]
    hi = Cint[h[1] for h in hop]; hj = Cint[h[2] for h in hop]; hJ = Float64[h[3] for h in hop]
    zi = Cint[z[1] for z in zzl]; zj = Cint[z[2] for z in zzl]; zJ = Float64[z[3] for z in zzl]
    f = Vector{Float64}(onsite_field)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:sd_model_create, libspindyn), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64},
                 Ptr{Float64}, Ref{Ptr{Cvoid}}),
                ctx.h, L, nup === nothing ? -1 : nup, length(hi), hi, hj, hJ, length(zi), zi, zj, zJ, f, r), ctx.h)
    m = Model(L, nup, nup === nothing ? :full : :sector, hop, f, zzl, ctx, r[])
    finalizer(x -> ccall((:sd_model_destroy, libspindyn), Cvoid, (Ptr{Cvoid},), x.h), m)
    return m
end

function XXZChain(L::Int; Jxy::Real=1.0, Jz::Real=1.0, hz::Real=0.0, nup::Union{Nothing,Int}=nothing, boundary::Symbol=:open)
    hopping = [(i, i + 1, Float64(Jxy) / 2) for i in 1:(L - 1)]
    zz = [(i, i + 1, Float64(Jz)) for i in 1:(L - 1)]
    if boundary === :periodic
        if L > 2
            push!(hopping, (L, 1, Float64(Jxy) / 2)); push!(zz, (L, 1, Float64(Jz)))
        end
    elseif boundary !== :open
        throw(ArgumentError("boundary must be :open or :periodic"))
    end
    return build_model(L; nup=nup, hopping=hopping, onsite_field=fill(Float64(hz), L), zz=zz)
end

momenta(model::Model) = 2π .* (0:(model.L - 1)) ./ model.L
Base.length(model::Model) = Int(ccall((:sd_model_dim, libspindyn), Int64, (Ptr{Cvoid},), model.h))

"model.states[start:start+count-1] (1-based start), computed from closed-form unranking"
function states(model::Model, start::Integer=1, count::Integer=length(model))
    out = Vector{UInt64}(undef, count)
    check(ccall((:sd_model_states, libspindyn), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{UInt64}), model.h, start - 1, count, out))
    return out
end

# ---- operator seam: drop-in for Hamiltonian.apply_H! ------------------------------
function apply_H!(out::Vector{T}, ψ::Vector{T}, model::Model) where {T<:Union{Float64,ComplexF64}}
    length(out) == length(ψ) || throw(DimensionMismatch("length(out) != length(ψ)"))
    check(ccall((:sd_apply, libspindyn), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                model.ctx.h, model.h, dtype_code(T), out, ψ, length(ψ)), model.ctx.h)
    return out
end

function apply_rescaled_H!(out::Vector{T}, ψ::Vector{T}, applyH!, model::Model, a::Float64, b::Float64) where {T<:Union{Float64,ComplexF64}}
    length(out) == length(ψ) || throw(DimensionMismatch("length(out) != length(ψ)"))
    if applyH! !== apply_H!
        # any other callable, as the reference takes it (src/Hamiltonian.jl:285-301): H ψ by the caller's operator, then the
        # rescaling pass on the host with the reference's own arithmetic
        applyH!(out, ψ, model)
        @inbounds for i in eachindex(out)
            out[i] = (out[i] - b * ψ[i]) / a
        end
        return out
    end
    # the built-in operator: apply and rescaling fused in one device pass (same arithmetic per element)
    check(ccall((:sd_apply_rescaled, libspindyn), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Float64),
                model.ctx.h, model.h, dtype_code(T), out, ψ, length(ψ), a, b), model.ctx.h)
    return out
end

function Sz_q_vector(model::Model, psi0::AbstractVector{T}, q::Float64) where {T<:Number}
    x = T <: Complex ? Vector{ComplexF64}(psi0) : Vector{Float64}(psi0)
    phi = Vector{ComplexF64}(undef, length(x))
    check(ccall((:sd_szq, libspindyn), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}),
                model.ctx.h, model.h, dtype_code(eltype(x)), x, length(x), q, phi), model.ctx.h)
    return phi
end

# create_spin_operator(site, op_type)  -- src/Hamiltonian.jl:49-136
const _SPIN_OPS = Dict(:z => 0, :plus => 1, :minus => 2, :x => 3, :y => 4)
function create_spin_operator(site::Int, op_type::Symbol)
    site >= 1 || throw(ArgumentError("site must be at least 1"))
    haskey(_SPIN_OPS, op_type) ||
        throw(ArgumentError("unsupported spin operator: $op_type; expected :z, :plus, :minus, :x, or :y"))
    function operator(ψ::AbstractVector{T}, model::Model) where {T}
        x = T <: Complex ? Vector{ComplexF64}(ψ) : Vector{Float64}(ψ)
        out = similar(x)
        check(ccall((:sd_spin_operator, libspindyn), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                    model.ctx.h, model.h, dtype_code(eltype(x)), site, _SPIN_OPS[op_type], x, length(x), out), model.ctx.h)
        return out
    end
    return operator
end

# ---- Observables (src/Observables.jl) and InitialStates (src/InitialStates.jl) ------
function _obs(fname::Symbol, ψ::AbstractVector, model::Model, nout::Int)
    x = eltype(ψ) <: Complex ? Vector{ComplexF64}(ψ) : Vector{Float64}(ψ)
    outs = [Vector{Float64}(undef, model.L) for _ in 1:nout]
    fptr = Libdl.dlsym(Libdl.dlopen(libspindyn), fname)     # a (name, library) pair must be a literal for ccall: resolve the run-time name here
    if nout == 1
        check(ccall(fptr, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Float64}),
                    model.ctx.h, model.h, dtype_code(eltype(x)), x, length(x), outs[1]), model.ctx.h)
    else
        check(ccall(fptr, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}),
                    model.ctx.h, model.h, dtype_code(eltype(x)), x, length(x), outs[1], outs[2]), model.ctx.h)
    end
    return outs
end
magnetization_per_site(ψ::AbstractVector, model::Model) = _obs(:sd_magnetization, ψ, model, 1)[1]
connected_correlations(ψ::AbstractVector, model::Model) = _obs(:sd_connected_correlations, ψ, model, 1)[1]
function structure_factor_Sq(ψ::AbstractVector, model::Model)
    q, S = _obs(:sd_structure_factor, ψ, model, 2)
    return Dict{Float64,Float64}(q[n] => S[n] for n in 1:model.L)       # src/Observables.jl:103-108
end
structure_factor(model::Model, ψ::AbstractVector) = structure_factor_Sq(ψ, model)   # src/PublicAPI.jl:101-106

function _one_hot(model::Model, kind::Integer, flips::Vector{Int}=Int[])
    idx = Ref{Int64}(0)
    f = Cint.(flips)
    check(ccall((:sd_initial_state_index, libspindyn), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Cint, Ref{Int64}),
                model.h, kind, f, length(f), idx), model.ctx.h)      # SD_EARG -> ArgumentError, as the reference throws
    ψ0 = zeros(Float64, length(model))
    ψ0[idx[] + 1] = 1.0
    return ψ0
end
domain_wall_state(model::Model) = _one_hot(model, 0)
neel_state(model::Model) = _one_hot(model, 1)
polarized_state(model::Model; up::Bool=true) = _one_hot(model, up ? 2 : 3)
polarized_state_with_flips(model::Model, flips::Vector{Int}) = _one_hot(model, 4, flips)

# ---- PublicAPI (src/PublicAPI.jl) ---------------------------------------------------
# Start vectors: the reference draws them with randn(rng, T, N) (src/Lanczos.jl:39,99).  With `rng` (default
# Random.default_rng(), as in the reference) the shim draws the same vector in Julia and hands it to the library, so the
# reference's random stream -- and with it every un-converged Lanczos output -- is reproduced.  `seed=k` selects the
# library's counter-based device generator instead (no host vector: the choice for L >= 30); `psi0=v` injects a vector.
function groundstate(model::Model; method::Symbol=:lanczos, lanc_m::Int=100, tol::Float64=1e-12,
                     orthogonalize_tol::Float64=1e-10, rng::AbstractRNG=Random.default_rng(),
                     psi0::Union{Nothing,Vector{Float64}}=nothing, seed::Union{Nothing,Integer}=nothing)
    method === :lanczos || throw(ArgumentError("unsupported ground-state method: $method"))
    N = length(model)
    if psi0 === nothing && seed === nothing
        psi0 = randn(rng, Float64, N)                                   # src/Lanczos.jl:99
    end
    E0 = Ref{Float64}(0.0); mact = Ref{Cint}(0)
    gs = Vector{Float64}(undef, N)
    check(ccall((:sd_lanczos_groundstate, libspindyn), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Ptr{Float64}, UInt64, Ref{Float64}, Ptr{Float64}, Ref{Cint}),
                model.ctx.h, model.h, lanc_m, tol, orthogonalize_tol, psi0 === nothing ? C_NULL : psi0,
                seed === nothing ? 0 : seed, E0, gs, mact), model.ctx.h)
    return E0[], gs
end

# estimate_energy_bounds does not forward rng in the reference either (src/Lanczos.jl:258,267): both Lanczos runs draw
# from Random.default_rng()
function estimate_energy_bounds(model::Model; lanc_m::Int=80, seed::Union{Nothing,Integer}=nothing)
    lo = Ref{Float64}(0.0); hi = Ref{Float64}(0.0)
    N = length(model)
    va = seed === nothing ? randn(Random.default_rng(), ComplexF64, N) : nothing     # src/Lanczos.jl:39
    vb = seed === nothing ? randn(Random.default_rng(), ComplexF64, N) : nothing
    check(ccall((:sd_energy_bounds, libspindyn), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ref{Float64}, Ref{Float64}),
                model.ctx.h, model.h, lanc_m, va === nothing ? C_NULL : va, vb === nothing ? C_NULL : vb,
                seed === nothing ? 0 : seed, lo, hi), model.ctx.h)
    return lo[], hi[]
end

function time_evolve(model::Model, ψ0::AbstractVector, t::Real; method::Symbol=:krylov, Ebounds=nothing,
                     kry_m::Int=30, cheb_n::Int=100, seed::Union{Nothing,Integer}=nothing)
    N = length(ψ0)
    out = Vector{ComplexF64}(undef, N)
    if method === :krylov
        x = eltype(ψ0) <: Complex ? Vector{ComplexF64}(ψ0) : Vector{Float64}(ψ0)
        check(ccall((:sd_krylov_evolve, libspindyn), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Float64, Cint, Ptr{Cvoid}),
                    model.ctx.h, model.h, dtype_code(eltype(x)), x, N, Float64(t), kry_m, out), model.ctx.h)
        return out
    elseif method === :chebyshev
        eltype(ψ0) <: Complex || throw(ArgumentError("chebyshev needs a ComplexF64 ψ0 (src/TimeEvolution/Chebyshev.jl:36,98)"))
        bounds = Ebounds === nothing ? estimate_energy_bounds(model; seed=seed) : Ebounds
        x = Vector{ComplexF64}(ψ0)
        check(ccall((:sd_chebyshev_evolve, libspindyn), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Cint, Float64, Float64, Ptr{Cvoid}),
                    model.ctx.h, model.h, x, N, Float64(t), cheb_n, Float64(bounds[1]), Float64(bounds[2]), out), model.ctx.h)
        return out
    end
    throw(ArgumentError("unsupported time-evolution method: $method"))
end

function dynamical_structure_factor(model::Model, ψ0::AbstractVector, q::AbstractVector, ω::AbstractVector;
                                    method::Symbol=:lanczos, lanc_m::Int=200, eta::Float64=0.05, broaden::Symbol=:lorentz,
                                    a::Union{Nothing,Float64}=nothing, b::Union{Nothing,Float64}=nothing,
                                    kpm_m::Int=200, kernel::Symbol=:jackson, seed::Integer=0)
    q_list = Float64.(q); ω_range = Float64.(ω)
    x = eltype(ψ0) <: Complex ? Vector{ComplexF64}(ψ0) : Vector{Float64}(ψ0)
    S = Matrix{Float64}(undef, length(ω_range), length(q_list))      # C row-major (Qn x W) == Julia (W x Qn) column-major
    if method === :lanczos
        br = broaden === :lorentz ? 0 : broaden === :gauss ? 1 : error("unknown broadening: $broaden")
        check(ccall((:sd_lanczos_sqw, libspindyn), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Float64, Cint, Ptr{Float64}),
                    model.ctx.h, model.h, dtype_code(eltype(x)), x, length(x), q_list, length(q_list), ω_range, length(ω_range),
                    lanc_m, eta, br, S), model.ctx.h)
    elseif method === :kpm
        have = a !== nothing && b !== nothing
        kern = kernel === :jackson ? 0 : kernel === :lorentz ? 1 : 2
        check(ccall((:sd_kpm_sqw, libspindyn), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Float64, Float64,
                     Cint, Cint, UInt64, Ptr{Float64}),
                    model.ctx.h, model.h, dtype_code(eltype(x)), x, length(x), q_list, length(q_list), ω_range, length(ω_range),
                    have, have ? a : 0.0, have ? b : 0.0, kpm_m, kern, seed, S), model.ctx.h)
    else
        throw(ArgumentError("unsupported dynamical structure-factor method: $method"))
    end
    return permutedims(S)                                              # (length(q), length(ω)) as the reference returns
end

end # module

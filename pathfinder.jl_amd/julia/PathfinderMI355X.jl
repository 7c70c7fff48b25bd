# PathfinderMI355X.jl -- thin Julia `ccall` layer over libpfmi.so (include/pfmi.h).
#
# STATUS: written to the C ABI's specification; NOT executed in this repository's CI because no Julia
# toolchain exists in the build image (SURVEY.md hard part H1).  Everything numerical sits behind the C ABI
# and is tested from Python (tests/test_gpu_parity.py); this file is the mechanical binding a maintainer of
# mlcolab/Pathfinder.jl would add.  It keeps the public API (`pathfinder`, `multipathfinder`, `resample`) and
# the LogDensityProblems callback surface in Julia and replaces the four hot-path call sites:
#
#   fit_mvnormals(points, gradients; history_length)              src/singlepath.jl:301-303
#   maximize_elbo(rng, logp, fit_distributions[2:end], N, ntasks) src/singlepath.jl:306-308
#   _compute_psis_result(logp, fit_distributions, draws; ntasks)  src/multipath.jl:221, src/resample.jl:35
#   _resample(rng, draws_per_component, psis_result, ndraws)      src/multipath.jl:225, src/resample.jl:42-44
module PathfinderMI355X

using LinearAlgebra, Random

const libpfmi = get(ENV, "PFMI_LIB", joinpath(@__DIR__, "..", "lib", "libpfmi.so"))

struct PfmiError <: Exception
    code::Int32
    msg::String
end
last_error() = unsafe_string(ccall((:pfmi_last_error, libpfmi), Cstring, ()))
check(rc::Int32) = rc == 0 ? nothing : throw(PfmiError(rc, last_error()))

# ---- context ---------------------------------------------------------------------------------------
mutable struct Context
    ptr::Ptr{Cvoid}
    function Context(device::Integer=0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pfmi_create, libpfmi), Int32, (Int32, Ref{Ptr{Cvoid}}), device, ref))
        ctx = new(ref[])
        finalizer(c -> ccall((:pfmi_destroy, libpfmi), Int32, (Ptr{Cvoid},), c.ptr), ctx)
        return ctx
    end
end

# ---- target: built-in descriptors or an arbitrary Julia closure through @cfunction -------------------
struct CTarget
    kind::Int32; d::Int32; r::Int32; reserved::Int32
    mean::Ptr{Float64}; a::Ptr{Float64}; Wd::Ptr{Float64}; G::Ptr{Float64}
    offset::Float64
    fn::Ptr{Cvoid}; user::Ptr{Cvoid}
end

# logp is called one column at a time, exactly like `logp.(eachcol(ϕ))` (src/elbo.jl:15)
function _logp_trampoline(X::Ptr{Float64}, d::Int32, n::Int64, out::Ptr{Float64}, user::Ptr{Cvoid})::Cvoid
    logp = unsafe_pointer_to_objref(user)[]
    Xm = unsafe_wrap(Array, X, (Int(d), Int(n)))
    o = unsafe_wrap(Array, out, Int(n))
    @inbounds for j in 1:n
        o[j] = logp(view(Xm, :, j))
    end
    return nothing
end

function set_callback_target!(ctx::Context, logp, dim::Integer)
    box = Ref{Any}(logp)                       # keep alive for the duration of the calls (caller holds `box`)
    cfn = @cfunction(_logp_trampoline, Cvoid, (Ptr{Float64}, Int32, Int64, Ptr{Float64}, Ptr{Cvoid}))
    t = Ref(CTarget(2, dim, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0.0, cfn, pointer_from_objref(box)))
    check(ccall((:pfmi_set_target, libpfmi), Int32, (Ptr{Cvoid}, Ref{CTarget}), ctx.ptr, t))
    return box
end

# ---- fit_mvnormals ------------------------------------------------------------------------------------
"""
    fit_batch!(ctx, traces; history_length) -> (status, j_eff, logdet, n_rejected)

`traces` is a vector of (points::Vector{Vector{Float64}}, gradients::Vector{Vector{Float64}}) -- one
`OptimizationTrace` per path (src/optimize.jl:110-114).  Replaces `fit_mvnormals` for all paths at once.
"""
function fit_batch!(ctx::Context, traces; history_length::Int=6, ϵ::Float64=1e-12)
    K = length(traces)
    npts = Int64[length(t[1]) for t in traces]
    d = length(traces[1][1][1])
    theta = reduce(hcat, reduce(vcat, [t[1] for t in traces]))   # d x P, column = point (point-major in memory)
    grad = reduce(hcat, reduce(vcat, [t[2] for t in traces]))
    check(ccall((:pfmi_set_traces, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Int32, Ptr{Float64}, Ptr{Float64}),
                ctx.ptr, K, npts, d, theta, grad))
    check(ccall((:pfmi_fit_batch, libpfmi), Int32, (Ptr{Cvoid}, Int32, Float64), ctx.ptr, history_length, ϵ))
    P = sum(npts)
    status = Vector{Int32}(undef, P); jeff = Vector{Int32}(undef, P)
    logdet = Vector{Float64}(undef, P); nrej = Vector{Int64}(undef, K)
    check(ccall((:pfmi_get_fit_status, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int64}),
                ctx.ptr, status, jeff, logdet, nrej))
    # per-fit failures become PosDefException at materialisation time, as in WoodburyPDMat's constructor
    return status, jeff, logdet, nrej
end

"Materialise fit `p` (0-based) as the pieces of `MvNormal(μ, WoodburyPDMat(A, B, D, F))`."
function get_fit(ctx::Context, p::Integer, d::Int, j::Int)
    m = 2j; k = min(d, m)
    α = Vector{Float64}(undef, d); B = Matrix{Float64}(undef, d, m); D = Matrix{Float64}(undef, m, m)
    qrf = Matrix{Float64}(undef, d, m); T = Matrix{Float64}(undef, k, k); V = Matrix{Float64}(undef, k, k)
    μ = Vector{Float64}(undef, d); ld = Ref{Float64}(NaN)
    check(ccall((:pfmi_get_fit, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ref{Float64}), ctx.ptr, p, α, B, D, qrf, T, V, μ, ld))
    # Pathfinder.WoodburyPDFactorization(U, Q, V) with U = Diagonal(sqrt.(α)), Q = QRCompactWYQ(qrf, T),
    # V = UpperTriangular(V)   (src/woodbury.jl:12-21); Σ.B has size (d, 2j) as test/singlepath.jl:41 expects
    return (; α, B, D, qr_factors=qrf, T, V, μ, logdet=ld[])
end

# ---- maximize_elbo --------------------------------------------------------------------------------------
"""
    elbo_batch!(ctx, ndraws, seeds) -> (elbo, se, best_iter)

`seeds[p]` is the UInt64 of `rand!(rng, UInt64[L])` (src/elbo.jl:2) for point p.  `best_iter[k]` is the 1-based
`iteration_opt` of `maximize_elbo` for path k (0 when the path has no iterations).
"""
function elbo_batch!(ctx::Context, ndraws::Integer, seeds::Vector{UInt64}, K::Integer)
    P = length(seeds)
    elbo = Vector{Float64}(undef, P); se = similar(elbo); best = Vector{Int64}(undef, K)
    check(ccall((:pfmi_elbo_batch, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, Ptr{UInt64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
                ctx.ptr, ndraws, seeds, C_NULL, elbo, se, best))
    return elbo, se, best
end

"ELBOEstimate.draws / rand(rng, fit_distribution, n) regenerated on demand (src/elbo.jl:19, src/singlepath.jl:226-233)"
function draws(ctx::Context, p::Integer, seed::UInt64, n0::Integer, N::Integer, d::Integer)
    X = Matrix{Float64}(undef, d, N); lp = Vector{Float64}(undef, N); lq = similar(lp)
    check(ccall((:pfmi_draws, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.ptr, p, seed, n0, N, C_NULL, X, lp, lq))
    return X, lp, lq
end

# ---- _compute_psis_result + _resample ----------------------------------------------------------------------
"""
    pool_psis_resample!(ctx, ndraws_per_run, points, seeds, ndraws; importance, replace, seed)

Replaces `_compute_psis_result` + `_resample`: pools `ndraws_per_run` draws of fit `points[k]` per path on the
device, runs PSIS on the log ratios, draws `ndraws` indices and gathers the columns.  Returns
(draws, draw_component_ids (1-based), weights, pareto_shape).
"""
function pool_psis_resample!(ctx::Context, N_r::Integer, points::Vector{Int64}, seeds::Vector{UInt64}, ndraws::Integer,
                             d::Integer; importance::Bool=true, replace::Bool=true, seed::UInt64=rand(UInt64))
    K = length(points); S = K * N_r
    check(ccall((:pfmi_pool_build, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{UInt64}), ctx.ptr, N_r, points, seeds))
    w = Vector{Float64}(undef, S); k̂ = Ref{Float64}(NaN); M = Ref{Int64}(0)
    if importance
        dev = Ref{Ptr{Cvoid}}(C_NULL); cnt = Ref{Int64}(0)
        check(ccall((:pfmi_pool_log_ratios_dev, libpfmi), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), ctx.ptr, dev, cnt))
        check(ccall((:pfmi_psis_dev, libpfmi), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int64}),
                    ctx.ptr, dev[], S, w, C_NULL, k̂, M))
    end
    idx = Vector{Int64}(undef, ndraws)
    check(ccall((:pfmi_resample_indices, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, Int64, Int32, Int32, UInt64, Ptr{Float64}, Ptr{Int64}),
                ctx.ptr, S, ndraws, importance, replace, seed, C_NULL, idx))
    X = Matrix{Float64}(undef, d, ndraws)
    check(ccall((:pfmi_pool_gather, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ptr{Float64}),
                ctx.ptr, ndraws, idx, 0, X))
    ids = cld.(idx .+ 1, N_r)                  # draw_component_ids (src/resample.jl:70); idx is 0-based
    return X, ids, (importance ? w : nothing), k̂[]
end

# ---- optional: device L-BFGS for the built-in targets (plays optimize_with_trace, src/optimize.jl:35-59) ------
function optimize_batch!(ctx::Context, x0::Matrix{Float64}; history_length::Int=6, maxiters::Int=1000, g_tol::Float64=1e-8)
    K = size(x0, 2)                              # x0 is d x K column-major == K x d point-major for the C side
    npts = Vector{Int64}(undef, K)
    check(ccall((:pfmi_optimize_batch, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Int32, Float64, Ptr{Int64}),
                ctx.ptr, K, x0, history_length, maxiters, g_tol, npts))
    return npts
end
function get_trace(ctx::Context, k::Integer, npoints::Integer, d::Integer)   # OptimizationTrace, src/optimize.jl:94-100
    θ = Matrix{Float64}(undef, d, npoints); g = similar(θ); lp = Vector{Float64}(undef, npoints)
    check(ccall((:pfmi_get_trace, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx.ptr, k, θ, lp, g))
    return (points=collect(eachcol(θ)), log_densities=lp, gradients=collect(eachcol(g)))
end

# ---- the PDMats surface of a fitted covariance (src/woodbury.jl:326-423), as the HMC extensions use it -------
# (ext/PathfinderAdvancedHMCExt.jl:17-23 builds a metric from Σ; its sampling calls unwhiten!/mul!/quad on it)
struct DeviceWoodbury
    ctx::Context
    point::Int64      # 0-based global trace-point index of the fit
    dim::Int
end
const OP = (unwhiten=Int32(0), whiten=Int32(1), rmul=Int32(2), invunwhiten=Int32(3), mul=Int32(4), solve=Int32(5),
            quad=Int32(6), invquad=Int32(7))

function _apply(W::DeviceWoodbury, op::Int32, x::AbstractVecOrMat{Float64})
    X = Matrix{Float64}(reshape(x, W.dim, :))
    N = size(X, 2)
    out = op >= OP.quad ? Vector{Float64}(undef, N) : similar(X)
    check(ccall((:pfmi_woodbury_apply, libpfmi), Int32, (Ptr{Cvoid}, Int64, Int32, Int64, Ptr{Float64}, Ptr{Float64}),
                W.ctx.ptr, W.point, op, N, X, out))
    return (x isa AbstractVector && op < OP.quad) ? vec(out) : out
end
unwhiten(W::DeviceWoodbury, x) = _apply(W, OP.unwhiten, x)          # PDMats.unwhiten!   src/woodbury.jl:401-406
whiten(W::DeviceWoodbury, x) = _apply(W, OP.whiten, x)              # PDMats.whiten!     src/woodbury.jl:410-415
invunwhiten(W::DeviceWoodbury, x) = _apply(W, OP.invunwhiten, x)    # PDMats.invunwhiten! src/woodbury.jl:417-422
Base.:*(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.mul, x)    # src/woodbury.jl:340-349
Base.:\(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.solve, x)
quad(W::DeviceWoodbury, x) = _apply(W, OP.quad, x)                  # src/woodbury.jl:384-397
invquad(W::DeviceWoodbury, x) = _apply(W, OP.invquad, x)            # src/woodbury.jl:369-382
function diag(W::DeviceWoodbury)                                    # src/woodbury.jl:326-329
    out = Vector{Float64}(undef, W.dim)
    check(ccall((:pfmi_woodbury_diag, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}), W.ctx.ptr, W.point, out))
    return out
end

end # module

# PathfinderMI355X.jl -- Julia host package over libpfmi.so (include/pfmi.h): keeps Pathfinder.jl's `pathfinder()` /
# `multipathfinder()` / `resample()` API and the LogDensityProblems callback surface on the Julia side and replaces the four
# hot-path call sites of the reference (mlcolab/Pathfinder.jl v0.10.7) by batched GPU calls:
#
#   fit_mvnormals(points, gradients; history_length)              src/singlepath.jl:301-303   -> fit_mvnormals(eng, traces; ...)
#   maximize_elbo(rng, logp, fit_distributions[2:end], N, ntasks) src/singlepath.jl:306-308   -> maximize_elbo(eng, rngs, N)
#   _compute_psis_result(logp, fit_distributions, draws; ntasks)  src/multipath.jl:221        -> _compute_psis_result(eng, ...)
#   _resample(rng, draws_per_component, psis_result, ndraws)      src/multipath.jl:225        -> _resample(eng, rng, ...)
#
# STATUS.  No Julia toolchain exists in this repository's build / GPU images (SURVEY.md H1), so this file has never been
# executed here.  It is written to the C ABI's specification, every `ccall` signature is the header's, and the exact order of C
# calls it makes is replayed and checked on the GPU by `examples/julia_sequence.c` (tests/test_gpu_parity_r2.py::
# test_julia_call_sequence_in_c).  Everything numerical sits behind the C ABI and is tested from Python / C.
#
# Round 3.  (i) Targets: besides the host closure (`set_target!(eng, logp, dim)`, kind 2) the built-in device targets are
# reachable -- `GaussTarget(mean, a, Wd, G)` / `FunnelTarget(d)` (kinds 0 / 1: the single-pass scan that never forms x, and the
# device L-BFGS) -- and `DeviceClosureTarget(d, fptr, user)` (kind 3): a `logp` that is itself a GPU kernel launcher (e.g. an
# AMDGPU.jl kernel wrapped in `@cfunction`), evaluated on draws that never leave HBM.  (ii) One Julia process drives several
# GPUs: `multipathfinder(engines::Vector{Engine}, ...)` shards the runs in contiguous blocks; every stage is ENQUEUED on all
# engines before the first wait (`pfmi_*_enqueue`, `pfmi_pool_build_best`, `pfmi_comm_psis_resample`), so one thread keeps all
# GPUs busy, and the pooled stage goes through the RCCL group (`Comm`).
#
# Results.  `multipathfinder(eng, ...)` returns the reference's own `Pathfinder.MultiPathfinderResult` (its fields are
# untyped); per-run results are `DevicePathfinderResult`s -- the reference's `PathfinderResult` requires
# `fit_distributions::Vector{FD}` and an eager `elbo_estimates`, i.e. O(L d N) host memory (89 GB at the headline
# configuration), so the same field names are served lazily instead:
#   fit_distributions[i]   -> MvNormal(mu, WoodburyPDMat(A, B, D, WoodburyPDFactorization(U, Q, V)))   pfmi_get_fit
#   elbo_estimates[i]      -> ELBOEstimate(value, std_err, draws, logp, logq, logr)                    pfmi_draws (regenerated)
#   draws                  -> first ndraws ELBO draws of the winner (+ top-up)                         pfmi_draws
module PathfinderMI355X

using LinearAlgebra, Random
import Distributions, PDMats, Pathfinder, SciMLBase, StatsBase

const libpfmi = get(ENV, "PFMI_LIB", joinpath(@__DIR__, "..", "lib", "libpfmi.so"))

struct PfmiError <: Exception
    code::Int32
    msg::String
end
last_error() = unsafe_string(ccall((:pfmi_last_error, libpfmi), Cstring, ()))
check(rc::Int32) = rc == 0 ? nothing : throw(PfmiError(rc, last_error()))

# ---- context ------------------------------------------------------------------------------------------------------------
"One GPU, one HIP stream (pfmi_ctx).  One Engine per host thread."
mutable struct Engine
    ptr::Ptr{Cvoid}
    device::Int
    generation::Int                 # bumped whenever traces / fits change: lazy handles made earlier refuse to read
    keepalive::Any                  # the boxed logp closure handed to the C side
    comms::Dict{Vector{Ptr{Cvoid}},Any}   # communicators whose FIRST member is this engine, by member list (see comm_for)
    function Engine(device::Integer=0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pfmi_create, libpfmi), Int32, (Int32, Ref{Ptr{Cvoid}}), device, ref))
        eng = new(ref[], device, 0, nothing, Dict{Vector{Ptr{Cvoid}},Any}())
        finalizer(close, eng)
        return eng
    end
end
"""
    close(eng::Engine)

Destroy the context now (idempotent; also the finalizer).  Communicators that hold this engine are closed FIRST: the cached ones on
this side (`eng.comms`), and -- whatever order the GC runs finalizers in -- `pfmi_destroy` itself tears down every live `pfmi_comm`
that borrows the context before freeing it (csrc/comm_rccl.hip: pf_comm_ctx_dying), so a later `pfmi_comm_destroy` never dereferences
a dead `pfmi_ctx`.
"""
function Base.close(eng::Engine)
    eng.ptr == C_NULL && return nothing
    # the communicators cached on this engine go first (those cached on OTHER engines that contain this one are torn down by pfmi_destroy
    # itself on the C side -- pf_comm_ctx_dying -- and replaced at the next comm_for, which checks their members)
    for c in values(eng.comms)
        close(c)
    end
    empty!(eng.comms)
    ccall((:pfmi_destroy, libpfmi), Int32, (Ptr{Cvoid},), eng.ptr)
    eng.ptr = C_NULL
    return nothing
end
struct StaleHandleError <: Exception end
_live(eng::Engine, gen::Int) = eng.generation == gen || throw(StaleHandleError())

# ---- multi-GPU: one Julia process, G engines (pfmi_comm_init_all = ncclCommInitAll), paths in contiguous blocks -------------------------
mutable struct Comm
    ptr::Ptr{Cvoid}
    engines::Vector{Engine}
    function Comm(engines::Vector{Engine})
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pfmi_comm_init_all, libpfmi), Int32, (Int32, Ptr{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}),
                    length(engines), [e.ptr for e in engines], ref))
        c = new(ref[], engines)
        finalizer(close, c)
        return c
    end
end
"destroy the RCCL group now (idempotent; also the finalizer)"
function Base.close(c::Comm)
    c.ptr == C_NULL && return nothing
    ccall((:pfmi_comm_destroy, libpfmi), Int32, (Ptr{Cvoid},), c.ptr)
    c.ptr = C_NULL
    return nothing
end
# One communicator per SET of engines, created at first use and kept (an ncclCommInitAll per multipathfinder call would cost more than
# the pooled stage itself); `close(eng)` removes the entries its engine is part of.  Same policy as the Python host's `_comm_for`.
# The cache lives ON the first engine of the set, not in a global (ADVICE r4: a global Dict was mutated from finalizers without a lock and
# kept every engine ever passed to multipathfinder(engines, ...) alive): Comm -> engines -> comms -> Comm is an ordinary reference cycle,
# collected as a whole once the caller drops the engines.  One Engine belongs to one host thread, so no lock is needed.
function comm_for(engines::Vector{Engine})
    key = [e.ptr for e in engines]
    cache = engines[1].comms
    c = get(cache, key, nothing)
    if c === nothing || c.ptr == C_NULL || any(e.ptr == C_NULL for e in c.engines)
        c = Comm(engines)
        cache[key] = c
    end
    return c::Comm
end

# ---- target: arbitrary Julia closure through @cfunction (the reference's general logp, src/elbo.jl:15) --------------------
struct CTarget
    kind::Int32; d::Int32; r::Int32; reserved::Int32
    mean::Ptr{Float64}; a::Ptr{Float64}; Wd::Ptr{Float64}; G::Ptr{Float64}
    offset::Float64
    fn::Ptr{Cvoid}; user::Ptr{Cvoid}
    dev_fn::Ptr{Cvoid}                                    # kind 3 (pfmi_logp_dev_fn); appended in round 3 -- same layout as pfmi_target
end
# logp is called one column at a time, exactly like `logp.(eachcol(phi))` (src/elbo.jl:15, src/resample.jl:90-92).  `ntasks` (src/elbo.jl:3-6,
# src/utils.jl:33-49; logp must then be thread-safe, src/multipath.jl:104-108): the columns of a staged block are spread over Julia tasks
# HERE, inside the one callback the library makes on the calling thread (pfmi_set_callback_threads stays 1: the library's own threads are
# foreign to the Julia runtime).  The result does not depend on ntasks: one value per column.
const CALLBACK_NTASKS = Ref(1)
function _logp_trampoline(X::Ptr{Float64}, d::Int32, n::Int64, out::Ptr{Float64}, user::Ptr{Cvoid})::Cvoid
    logp = unsafe_pointer_to_objref(user)[]
    Xm = unsafe_wrap(Array, X, (Int(d), Int(n)))
    o = unsafe_wrap(Array, out, Int(n))
    nt = min(CALLBACK_NTASKS[], Int(n))
    if nt <= 1
        @inbounds for j in 1:n
            o[j] = logp(view(Xm, :, j))
        end
    else
        per = cld(Int(n), nt)
        @sync for t in 1:nt
            Threads.@spawn begin
                @inbounds for j in ((t - 1) * per + 1):min(t * per, Int(n))
                    o[j] = logp(view(Xm, :, j))
                end
            end
        end
    end
    return nothing
end
"`ntasks` of the reference for host closures: columns of every staged block over `n` Julia tasks (the closure must be thread-safe)"
set_callback_ntasks!(n::Integer) = (CALLBACK_NTASKS[] = max(1, Int(n)); nothing)
function set_target!(eng::Engine, logp, dim::Integer)
    box = Ref{Any}(logp)
    eng.keepalive = box                                   # alive as long as the engine may call back
    cfn = @cfunction(_logp_trampoline, Cvoid, (Ptr{Float64}, Int32, Int64, Ptr{Float64}, Ptr{Cvoid}))
    t = Ref(CTarget(2, dim, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0.0, cfn, pointer_from_objref(box), C_NULL))
    check(ccall((:pfmi_set_target, libpfmi), Int32, (Ptr{Cvoid}, Ref{CTarget}), eng.ptr, t))
    return nothing
end

# ---- device-resident targets (round 3) ---------------------------------------------------------------------------------------
abstract type DeviceTarget end
"""
    GaussTarget(mean, a[, Wd, G]; offset = 0)      logp(x) = offset - (e'diag(a)e - |G Wd'e|^2) / 2,  e = x - mean

PFMI_TARGET_GAUSS: `a` the diagonal precision part, `Wd = diag(a) W` (d x r, r <= 16) and `G` (r x r lower triangular,
`G'G = inv(I + W'diag(a)W)`) the low-rank correction of `Sigma* = diag(1 ./ a) + W W'`.  Evaluated by the single-pass ELBO scan
without ever forming the draws; also what the device L-BFGS (`optimize_batch!`) differentiates.
"""
struct GaussTarget <: DeviceTarget
    mean::Vector{Float64}; a::Vector{Float64}; Wd::Matrix{Float64}; G::Matrix{Float64}; offset::Float64
end
GaussTarget(mean, a; offset=0.0) = GaussTarget(collect(Float64, mean), collect(Float64, a), zeros(length(mean), 0), zeros(0, 0), offset)
GaussTarget(mean, a, Wd, G; offset=0.0) = GaussTarget(collect(Float64, mean), collect(Float64, a), Matrix{Float64}(Wd), Matrix{Float64}(G), offset)
"`GaussTarget` of `N(mean, diag(sigma2) + W W')` (the parameterisation of SURVEY.md 8d's T_lr)"
function GaussTarget(mean, sigma2, W::AbstractMatrix; offset=0.0)
    a = 1.0 ./ collect(Float64, sigma2)
    Wd = a .* W
    G = inv(cholesky(Symmetric(I + W' * Wd)).L)
    return GaussTarget(collect(Float64, mean), a, Matrix{Float64}(Wd), Matrix{Float64}(G), offset)
end
"PFMI_TARGET_FUNNEL (reference docs/src/examples/quickstart.md:229-234)"
struct FunnelTarget <: DeviceTarget
    d::Int
end
"""
    DeviceClosureTarget(d, fptr, user = C_NULL)

PFMI_TARGET_DEVICE_CALLBACK: `fptr` is a C function pointer `(X_dev::Ptr{Float64}, d::Int32, n::Int64, out_dev::Ptr{Float64},
stream::Ptr{Cvoid}, user::Ptr{Cvoid}) -> Cvoid` that ENQUEUES the caller's kernel on `stream` (the engine's hipStream_t) and
returns: e.g. `@cfunction` of a function that wraps the two device pointers in `AMDGPU.unsafe_wrap(ROCArray, ...)` and launches
an AMDGPU.jl kernel on `HIPStream(stream)`.  The reference's arbitrary `logp` (src/elbo.jl:15) without the PCIe round trip.
"""
struct DeviceClosureTarget <: DeviceTarget
    d::Int; fptr::Ptr{Cvoid}; user::Ptr{Cvoid}
end
DeviceClosureTarget(d::Integer, fptr::Ptr{Cvoid}) = DeviceClosureTarget(d, fptr, C_NULL)
dimension(t::GaussTarget) = length(t.mean)
dimension(t::Union{FunnelTarget,DeviceClosureTarget}) = t.d
function set_target!(eng::Engine, t::GaussTarget)
    r = size(t.Wd, 2)
    eng.keepalive = t
    GC.@preserve t begin
        ct = Ref(CTarget(0, length(t.mean), r, 0, pointer(t.mean), pointer(t.a), r > 0 ? pointer(t.Wd) : C_NULL,
                         r > 0 ? pointer(t.G) : C_NULL, t.offset, C_NULL, C_NULL, C_NULL))
        check(ccall((:pfmi_set_target, libpfmi), Int32, (Ptr{Cvoid}, Ref{CTarget}), eng.ptr, ct))   # the library copies the parameters
    end
    return nothing
end
function set_target!(eng::Engine, t::FunnelTarget)
    ct = Ref(CTarget(1, t.d, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0.0, C_NULL, C_NULL, C_NULL))
    check(ccall((:pfmi_set_target, libpfmi), Int32, (Ptr{Cvoid}, Ref{CTarget}), eng.ptr, ct))
end
function set_target!(eng::Engine, t::DeviceClosureTarget)
    ct = Ref(CTarget(3, t.d, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0.0, C_NULL, t.user, t.fptr))
    check(ccall((:pfmi_set_target, libpfmi), Int32, (Ptr{Cvoid}, Ref{CTarget}), eng.ptr, ct))
end
"host twin of a built-in target (the optimiser's `logp`, result objects): the same formula in Julia"
function logdensity(t::GaussTarget, x)
    e = x .- t.mean
    q = sum(t.a .* e .* e)
    size(t.Wd, 2) > 0 && (q -= sum(abs2, t.G * (t.Wd' * e)))
    return t.offset - q / 2
end
logdensity(t::FunnelTarget, x) = ((x[1] / 3)^2 + (t.d - 1) * x[1] + exp(-x[1]) * sum(abs2, @view x[2:end])) / -2

# ---- per-batch state --------------------------------------------------------------------------------------------------------
struct Batch
    eng::Engine
    gen::Int
    dim::Int
    offsets::Vector{Int}            # 0-based offset of each path's first point, length K + 1
    status::Vector{Int32}           # per point (src/woodbury.jl:202,205: 1 = A not PD, 2 = C not PD)
    jeff::Vector{Int32}             # effective history length per point
    nrej::Vector{Int64}             # rejected BFGS updates per path
end
npaths(b::Batch) = length(b.offsets) - 1

# ---- call site 1: fit_mvnormals ----------------------------------------------------------------------------------------------
"""
    fit_mvnormals(eng, traces; history_length, ϵ) -> Batch

All runs at once (`strict`: see below).  `traces[k]` is the `OptimizationTrace` of run k (`.points`, `.gradients`: vectors of vectors,
src/optimize.jl:94-114).  Replaces `fit_mvnormals` (src/mvnormal.jl:14-21) = `lbfgs_inverse_hessians` + `WoodburyPDMat` +
`muladd(Σ, ∇logp, θ)` for every trace point.
"""
# `Hinit` (src/inverse_hessian.jl:25) as a name: a Julia closure cannot cross the C boundary; the two the reference itself uses can
const HINIT = (gilbert_init=Int32(0), nocedal_wright_scaling=Int32(1))
_hinit_code(h::Symbol) = getproperty(HINIT, h)
_hinit_code(::typeof(Pathfinder.gilbert_init)) = HINIT.gilbert_init
function fit_mvnormals(eng::Engine, traces; history_length::Int=Pathfinder.DEFAULT_HISTORY_LENGTH, ϵ::Float64=1e-12, strict::Bool=true,
                       Hinit=Pathfinder.gilbert_init)
    K = length(traces)
    npts = Int64[length(t.points) for t in traces]
    d = length(first(first(traces).points))
    theta = reduce(hcat, reduce(vcat, [t.points for t in traces]))        # d x P, column = trace point (point-major in memory)
    grad = reduce(hcat, reduce(vcat, [t.gradients for t in traces]))
    eng.generation += 1
    check(ccall((:pfmi_set_traces, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Int32, Ptr{Float64}, Ptr{Float64}),
                eng.ptr, K, npts, d, theta, grad))
    check(ccall((:pfmi_fit_batch_ex, libpfmi), Int32, (Ptr{Cvoid}, Int32, Float64, Int32), eng.ptr, history_length, ϵ, _hinit_code(Hinit)))
    P = sum(npts)
    status = Vector{Int32}(undef, P); jeff = Vector{Int32}(undef, P); nrej = Vector{Int64}(undef, K)
    check(ccall((:pfmi_get_fit_status, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int64}),
                eng.ptr, status, jeff, C_NULL, nrej))
    # the reference builds every WoodburyPDMat eagerly, so a non-positive-definite fit throws here (src/woodbury.jl:202,205);
    # strict = false keeps the library's per-fit status instead (the fit's ELBO is NaN and the argmax skips it)
    strict && any(!=(0), status) && throw(LinearAlgebra.PosDefException(Int(first(filter(!=(0), status)))))
    return Batch(eng, eng.generation, d, vcat(0, cumsum(npts)), status, jeff, nrej)
end

"Materialise fit `p` (0-based global point) as `MvNormal(μ, WoodburyPDMat(A, B, D, F))` (src/mvnormal.jl:18, src/woodbury.jl:12-21,246-257)."
function fit_distribution(b::Batch, p::Integer)
    _live(b.eng, b.gen)
    b.status[p + 1] == 0 || throw(LinearAlgebra.PosDefException(Int(b.status[p + 1])))   # what WoodburyPDMat's constructor throws
    d = b.dim; j = Int(b.jeff[p + 1]); m = 2j; k = min(d, m)
    α = Vector{Float64}(undef, d); B = Matrix{Float64}(undef, d, m); D = Matrix{Float64}(undef, m, m)
    qrf = Matrix{Float64}(undef, d, m); T = Matrix{Float64}(undef, k, k); V = Matrix{Float64}(undef, k, k)
    μ = Vector{Float64}(undef, d); ld = Ref{Float64}(NaN)
    check(ccall((:pfmi_get_fit, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ref{Float64}), b.eng.ptr, p, α, B, D, qrf, T, V, μ, ld))
    # qr(U' \ B) in LAPACK's compact-WY form, exactly what `qr(::Matrix{Float64})` returns (src/woodbury.jl:204)
    Q = LinearAlgebra.QRCompactWYQ(qrf, T)
    F = Pathfinder.WoodburyPDFactorization(Diagonal(sqrt.(α)), Q, UpperTriangular(V))
    Σ = Pathfinder.WoodburyPDMat(Diagonal(α), B, D, F)                    # size(Σ.B) == (d, 2j), test/singlepath.jl:41
    return Distributions.MvNormal(μ, Σ)
end

"`fit_distributions` of run k: built on first access."
struct LazyFitDistributions <: AbstractVector{Any}
    b::Batch
    k::Int
    cache::Dict{Int,Any}
end
Base.size(v::LazyFitDistributions) = (v.b.offsets[v.k + 1] - v.b.offsets[v.k],)
Base.getindex(v::LazyFitDistributions, i::Int) = get!(() -> fit_distribution(v.b, v.b.offsets[v.k] + i - 1), v.cache, i)

# ---- call site 2: maximize_elbo ------------------------------------------------------------------------------------------------
struct ElboBatch
    b::Batch
    ndraws::Int
    seeds::Vector{UInt64}           # per point; seeds = rand!(rng_k, UInt64[L_k]) per run (src/elbo.jl:2)
    value::Vector{Float64}
    std_err::Vector{Float64}
    iteration_opt::Vector{Int64}    # per run, 1-based like the reference's fit_iteration (0: the run has no iterations)
end
"""
    maximize_elbo(eng, batch, rngs, ndraws) -> ElboBatch

`rngs[k]` is run k's own rng (already reseeded from `run_seeds[k]`, src/multipath.jl:190-193); it is advanced exactly like
the reference does: ONE `rand!(rng, UInt64[L_k])` (src/elbo.jl:2).  The NaN-skipping first-max argmax (src/utils.jl:55-72)
runs on the device.
"""
function maximize_elbo(b::Batch, rngs::AbstractVector{<:Random.AbstractRNG}, ndraws::Int;
                       pending=1:npaths(b), run_seeds::Vector{Vector{UInt64}}=[UInt64[] for _ in 1:npaths(b)])
    _live(b.eng, b.gen)
    K = npaths(b); P = b.offsets[end]
    seeds = zeros(UInt64, P)
    for k in 1:K
        L = b.offsets[k + 1] - b.offsets[k] - 1
        # only runs that are (re)tried draw seeds; a finished run keeps its own, so the batched re-evaluation reproduces it bit for bit
        k in pending && (run_seeds[k] = rand!(rngs[k], Vector{UInt64}(undef, L)))
        seeds[(b.offsets[k] + 2):(b.offsets[k] + 1 + L)] .= run_seeds[k]
    end
    elbo = Vector{Float64}(undef, P); se = similar(elbo); best = Vector{Int64}(undef, K)
    check(ccall((:pfmi_elbo_batch, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, Ptr{UInt64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
                b.eng.ptr, ndraws, seeds, C_NULL, elbo, se, best))
    return ElboBatch(b, ndraws, seeds, elbo, se, best)
end

"`rand(rng, dist, n)` / `ELBOEstimate.draws`: draws n0 .. n0+N-1 of fit p, a pure function of (seed, n) (src/elbo.jl:19, src/singlepath.jl:226-233)."
function draws(b::Batch, p::Integer, seed::UInt64, n0::Integer, N::Integer)
    _live(b.eng, b.gen)
    X = Matrix{Float64}(undef, b.dim, N); lp = Vector{Float64}(undef, N); lq = similar(lp)
    check(ccall((:pfmi_draws, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                b.eng.ptr, p, seed, n0, N, C_NULL, X, lp, lq))
    return X, lp, lq
end

"`elbo_estimates` of run k: `Pathfinder.ELBOEstimate`s whose draws are regenerated from the fit's seed on first access."
struct LazyELBOEstimates <: AbstractVector{Any}
    e::ElboBatch
    k::Int
    cache::Dict{Int,Any}
end
Base.size(v::LazyELBOEstimates) = (v.e.b.offsets[v.k + 1] - v.e.b.offsets[v.k] - 1,)
function Base.getindex(v::LazyELBOEstimates, i::Int)
    return get!(v.cache, i) do
        p = v.e.b.offsets[v.k] + i                       # fit_distributions[i + 1]
        ϕ, logpϕ, logqϕ = draws(v.e.b, p, v.e.seeds[p + 1], 0, v.e.ndraws)
        Pathfinder.ELBOEstimate(v.e.value[p + 1], v.e.std_err[p + 1], ϕ, logpϕ, logqϕ, logpϕ - logqϕ)   # src/elbo.jl:22-29
    end
end

# ---- per-run result, field-compatible with Pathfinder.PathfinderResult (src/singlepath.jl:53-70) -----------------------------------
mutable struct DevicePathfinderResult
    input; optimizer; rng; optim_prob; logp
    fit_iteration::Int
    num_tries::Int
    optim_solution; optim_trace
    fit_distributions::LazyFitDistributions
    elbo_estimates::LazyELBOEstimates
    num_bfgs_updates_rejected::Int
    success::Bool
    draw_seed::UInt64
    ndraws::Int
    _draws::Union{Nothing,Matrix{Float64}}
end
function Base.getproperty(r::DevicePathfinderResult, s::Symbol)
    if s === :fit_distribution || s === :fit_distribution_transformed
        return getfield(r, :fit_distributions)[getfield(r, :fit_iteration) + 1]          # src/singlepath.jl:224
    elseif s === :draws || s === :draws_transformed
        if getfield(r, :_draws) === nothing                                             # src/singlepath.jl:226-233
            e = getfield(r, :elbo_estimates).e
            p = e.b.offsets[getfield(r, :elbo_estimates).k] + getfield(r, :fit_iteration)
            setfield!(r, :_draws, draws(e.b, p, getfield(r, :draw_seed), 0, getfield(r, :ndraws))[1])
        end
        return getfield(r, :_draws)
    end
    return getfield(r, s)
end

# ---- PSIS result: the two fields the reference reads (src/resample.jl:64 `.weights`, src/multipath.jl:53 `.pareto_shape`) -----------
struct DevicePSISResult
    weights::Vector{Float64}
    log_weights::Vector{Float64}
    pareto_shape::Float64
    tail_length::Int
end

# ---- call site 3: _compute_psis_result -----------------------------------------------------------------------------------------------
"""
    _compute_psis_result(elbos, fit_points, draw_seeds, ndraws_per_run) -> DevicePSISResult

`draws_per_component = stack(draws)` stays on the device: N_r draws of fit `fit_points[k]` (0-based) with seed
`draw_seeds[k]` per run, log ratios `logp - logpdf(component_k, ·)` in k-major / n-fastest order (src/resample.jl:81-95), then
`PSIS.psis` (src/resample.jl:78).
"""
function _compute_psis_result(b::Batch, fit_points::Vector{Int64}, draw_seeds::Vector{UInt64}, N_r::Int; importance::Bool=true)
    _live(b.eng, b.gen)
    check(ccall((:pfmi_pool_build, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{UInt64}), b.eng.ptr, N_r, fit_points, draw_seeds))
    importance || return nothing
    S = length(fit_points) * N_r
    dev = Ref{Ptr{Cvoid}}(C_NULL); cnt = Ref{Int64}(0)
    check(ccall((:pfmi_pool_log_ratios_dev, libpfmi), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), b.eng.ptr, dev, cnt))
    w = Vector{Float64}(undef, S); lw = similar(w); k̂ = Ref{Float64}(NaN); M = Ref{Int64}(0)
    check(ccall((:pfmi_psis_dev, libpfmi), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int64}),
                b.eng.ptr, dev[], S, w, lw, k̂, M))
    return DevicePSISResult(w, lw, k̂[], Int(M[]))
end

# ---- call site 4: _resample ------------------------------------------------------------------------------------------------------------
"""
    _resample(b, rng, psis_result, N_r, K, ndraws; replace, statsbase=false) -> (draws, draw_component_ids)

Index selection + gather (src/resample.jl:58-72).  `statsbase` chooses who draws the indices:
* `false` (default): the library's own deterministic sampler on the device, keyed by one `rand(rng, UInt64)`;
* `:host`: EXACTLY the reference -- `StatsBase.sample(rng, 1:S, ProbabilityWeights(w, 1), ndraws; replace)` runs in Julia on
  the downloaded PSIS weights (whatever algorithm the installed StatsBase picks: direct, alias, Efraimidis-Spirakis) and only
  the gather of the selected columns happens on the device;
* `true` (weighted, with replacement): the indices are those of `StatsBase.direct_sample!(rng, 1:S, ProbabilityWeights(w, 1), x)`
  -- Julia draws the uniforms ONE SCALAR `rand(rng)` PER DRAW, as that loop does (a bulk `rand(rng, n)` fills from a different
  stream position for Xoshiro / MersenneTwister), the device does the scan (no weight download).  Note that
  `StatsBase.sample` itself only takes `direct_sample!` for small requests (pool < 40 or ndraws < 32 / 64); at real pool sizes it
  switches to `alias_sample!` (version dependent), which nothing here restates: ONLY `:host` reproduces `StatsBase.sample` bit for
  bit at every size.
"""
function _resample(b::Batch, rng::Random.AbstractRNG, psis_result, N_r::Int, K::Int, ndraws::Int; replace::Bool=true,
                   statsbase::Union{Bool,Symbol}=false)
    _live(b.eng, b.gen)
    S = K * N_r
    idx = Vector{Int64}(undef, ndraws)
    if statsbase === :host
        inds = if psis_result === nothing
            StatsBase.sample(rng, 1:S, ndraws; replace)                                                   # src/resample.jl:61
        else
            StatsBase.sample(rng, 1:S, StatsBase.ProbabilityWeights(psis_result.weights, 1.0), ndraws; replace)   # :63-66
        end
        idx .= inds .- 1
    elseif statsbase === true && psis_result !== nothing && replace
        u = Float64[rand(rng) for _ in 1:ndraws]                    # what direct_sample! consumes: one scalar rand(rng) per draw
        check(ccall((:pfmi_resample_indices_direct, libpfmi), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Int64}),
                    b.eng.ptr, S, ndraws, u, idx))
    else
        check(ccall((:pfmi_resample_indices, libpfmi), Int32,
                    (Ptr{Cvoid}, Int64, Int64, Int32, Int32, UInt64, Ptr{Float64}, Ptr{Int64}),
                    b.eng.ptr, S, ndraws, psis_result !== nothing, replace, rand(rng, UInt64), C_NULL, idx))
    end
    X = Matrix{Float64}(undef, b.dim, ndraws)
    check(ccall((:pfmi_pool_gather, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ptr{Float64}),
                b.eng.ptr, ndraws, idx, 0, X))
    return X, cld.(idx .+ 1, N_r)                                   # draw_component_ids (src/resample.jl:70); idx is 0-based
end

# ---- multipathfinder (src/multipath.jl:118-245), batched -------------------------------------------------------------------------------
"""
    multipathfinder(eng::Engine, fun, ndraws; kwargs...) -> Pathfinder.MultiPathfinderResult

Same keywords as the reference.  The K optimisations run on the host exactly as in the reference (each with its own `copy(rng)`
reseeded from `run_seeds[k]`); fits, ELBO scans, pooling, PSIS and resampling are ONE batched GPU call each.
"""
function multipathfinder(eng::Engine, optim_fun::SciMLBase.OptimizationFunction, ndraws::Int;
                         init=nothing, input=optim_fun, dim::Int=-1,
                         nruns::Int=init === nothing ? -1 : length(init),
                         ndraws_elbo::Int=Pathfinder.DEFAULT_NDRAWS_ELBO,
                         ndraws_per_run::Int=max(ndraws_elbo, cld(ndraws, max(nruns, 1))),
                         rng::Random.AbstractRNG=Random.default_rng(),
                         history_length::Int=Pathfinder.DEFAULT_HISTORY_LENGTH,
                         optimizer=Pathfinder.default_optimizer(history_length),
                         importance::Bool=true, ntries::Int=1_000, init_scale=2,
                         init_sampler=Pathfinder.UniformSampler(init_scale), statsbase_indices::Union{Bool,Symbol}=false,
                         ntasks::Int=Threads.nthreads(), ntasks_per_run::Int=1, kwargs...)
    # the reference spreads the runs over `ntasks` tasks and each run's logp evaluations over `ntasks_per_run` (src/multipath.jl:104-108,
    # 190-208); here all runs are one batch, so the closure sees ntasks * ntasks_per_run tasks per staged block of draws
    set_callback_ntasks!(max(1, ntasks) * max(1, ntasks_per_run))
    if init === nothing
        nruns > 0 || throw(ArgumentError("A positive `nruns` must be set or `init` must be provided."))      # :148-150
        dim > 0 || throw(ArgumentError("An initial point `init` or dimension `dim` must be provided."))     # src/singlepath.jl:171
    else
        nruns = length(init)
    end
    d = init === nothing ? dim : length(first(init))
    if ndraws > ndraws_per_run * nruns
        @warn "More draws requested than total number of draws across replicas. Draws will not be unique."
    end
    logp(x) = -optim_fun.f(x, nothing)                                                                      # :159
    set_target!(eng, logp, d)
    # the reference keeps `init = nothing` per run (:151) and each run samples its own start inside `pathfinder` with the run's
    # seeded rng (src/singlepath.jl:167-168, called at :190-193): run_seeds FIRST, then x0[k] from rngs[k] -- as pfmi/api.py does
    run_seeds = rand!(rng, Vector{UInt64}(undef, nruns))                                                    # :162
    rngs = [Random.seed!(copy(rng), s) for s in run_seeds]                                                  # :189-193
    _init = init === nothing ? [init_sampler(rngs[k], Vector{Float64}(undef, d)) for k in 1:nruns] : collect(init)
    probs = [SciMLBase.OptimizationProblem(optim_fun, x0, nothing) for x0 in _init]
    itry = ones(Int, nruns); pending = collect(1:nruns)
    sols = Vector{Any}(undef, nruns); traces = Vector{Any}(undef, nruns)
    local b::Batch, e::ElboBatch
    success = falses(nruns)
    fit_seeds = [UInt64[] for _ in 1:nruns]                           # seeds = rand!(rng_k, UInt64[L_k]) of each run's LAST try
    while true                                                        # the retry loop of src/singlepath.jl:259-283, batched
        for k in pending
            sols[k], traces[k] = Pathfinder.optimize_with_trace(probs[k], deepcopy(optimizer); kwargs...)  # host, src/optimize.jl:35-59
        end
        b = fit_mvnormals(eng, traces; history_length)                # call site 1 (all runs: finished ones are refitted identically)
        e = maximize_elbo(b, rngs, ndraws_elbo; pending, run_seeds=fit_seeds)   # call site 2
        for k in pending
            L = length(traces[k].points) - 1
            it = e.iteration_opt[k]
            v = it > 0 ? e.value[b.offsets[k] + it + 1] : NaN
            success[k] = L > 0 && it > 0 && !isnan(v) && v != -Inf     # src/singlepath.jl:299, 309-314
        end
        pending = [k for k in pending if !success[k] && itry[k] < ntries]
        isempty(pending) && break
        for k in pending
            itry[k] += 1
            probs[k] = SciMLBase.remake(probs[k]; u0=init_sampler(rngs[k], copy(probs[k].u0)))              # :277
        end
    end
    results = Vector{DevicePathfinderResult}(undef, nruns)
    fit_points = Vector{Int64}(undef, nruns); draw_seeds = Vector{UInt64}(undef, nruns)
    for k in 1:nruns
        it = Int(e.iteration_opt[k])
        success[k] || @warn "Pathfinder failed after $(itry[k]) tries. Increase `ntries`, inspect the model for numerical instability, or provide a more suitable `init_sampler`."
        if b.nrej[k] > 0
            perc = round(b.nrej[k] * (100 // length(traces[k].points)); digits=1)
            @warn "$(b.nrej[k]) ($(perc)%) updates to the inverse Hessian estimate were rejected to keep it positive definite."
        end
        fit_points[k] = b.offsets[k] + it                             # fit_distributions[fit_iteration + 1], 0-based point
        # draws = the winner's ELBO draws (+ top-up from the same counter-based stream), or fresh ones after a failure (:226-233)
        draw_seeds[k] = success[k] ? e.seeds[fit_points[k] + 1] : rand(rngs[k], UInt64)
        results[k] = DevicePathfinderResult(input, optimizer, rngs[k], probs[k], logp, it, itry[k], sols[k], traces[k],
                                            LazyFitDistributions(b, k, Dict{Int,Any}()), LazyELBOEstimates(e, k, Dict{Int,Any}()),
                                            Int(b.nrej[k]), success[k], draw_seeds[k], ndraws_per_run, nothing)
    end
    psis_result = _compute_psis_result(b, fit_points, draw_seeds, ndraws_per_run; importance)            # call site 3 (:220-224)
    draws_, ids = _resample(b, rng, psis_result, ndraws_per_run, nruns, ndraws; statsbase=statsbase_indices)   # call site 4 (:225)
    components = [fit_distribution(b, p) for p in fit_points]
    fit_dist = Distributions.MixtureModel(components)                                                      # :215-216
    return Pathfinder.MultiPathfinderResult(input, optimizer, rng, optim_fun, logp, fit_dist, draws_, ids, fit_dist, draws_,
                                            results, psis_result)
end

"`pathfinder(eng, prob; ...)`: a single run through the same batched path (src/singlepath.jl:142-257)."
function pathfinder(eng::Engine, optim_fun::SciMLBase.OptimizationFunction; init, ndraws_elbo::Int=Pathfinder.DEFAULT_NDRAWS_ELBO,
                    ndraws::Int=ndraws_elbo, kwargs...)
    res = multipathfinder(eng, optim_fun, ndraws; init=[init], ndraws_elbo, ndraws_per_run=ndraws, importance=false, kwargs...)
    return res.pathfinder_results[1]
end

# ---- round 3: device-resident targets, several GPUs driven by ONE Julia thread -------------------------------------------------------
# Every stage is enqueued on every engine before the first wait: pfmi_optimize_batch_enqueue / _wait, pfmi_fit_batch (never blocks),
# pfmi_elbo_batch_enqueue, pfmi_pool_build_best (the winners are picked on the device: no host round trip between the scan and the
# pool), pfmi_comm_psis_resample (all-gather + PSIS + index selection + owner gather + all-reduce, ONE synchronisation); the
# per-engine downloads (status, ELBO table, winners) come last, when everything is already in flight.
"contiguous blocks of runs, one per engine (pool order stays k-major, src/resample.jl:93)"
function _blocks(K::Int, G::Int)
    K >= G || throw(ArgumentError("nruns = $K is smaller than the number of engines $G: every engine needs at least one run"))
    base, rem = divrem(K, G)                      # any nruns (src/multipath.jl:131-146): the first K % G blocks are one run longer
    stops = cumsum([base + (g <= rem ? 1 : 0) for g in 1:G])
    return [(g == 1 ? 1 : stops[g - 1] + 1):stops[g] for g in 1:G]
end

function _fit_enqueue!(eng::Engine, history_length::Int, ϵ::Float64)
    check(ccall((:pfmi_fit_batch, libpfmi), Int32, (Ptr{Cvoid}, Int32, Float64), eng.ptr, history_length, ϵ))
end
function _fit_finish(eng::Engine, npts::Vector{Int64}, d::Int)
    K = length(npts); P = sum(npts)
    status = Vector{Int32}(undef, P); jeff = Vector{Int32}(undef, P); nrej = Vector{Int64}(undef, K)
    check(ccall((:pfmi_get_fit_status, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int64}),
                eng.ptr, status, jeff, C_NULL, nrej))
    return Batch(eng, eng.generation, d, vcat(0, cumsum(npts)), status, jeff, nrej)
end
function _elbo_enqueue!(eng::Engine, seeds::Vector{UInt64}, ndraws::Int)
    check(ccall((:pfmi_elbo_batch_enqueue, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{UInt64}, Ptr{Float64}), eng.ptr, ndraws, seeds, C_NULL))
end
function _elbo_wait(b::Batch, seeds::Vector{UInt64}, ndraws::Int)
    P = b.offsets[end]; K = npaths(b)
    elbo = Vector{Float64}(undef, P); se = similar(elbo); best = Vector{Int64}(undef, K)
    check(ccall((:pfmi_elbo_batch_wait, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}), b.eng.ptr, elbo, se, best))
    return ElboBatch(b, ndraws, seeds, elbo, se, best)
end
"pool of the winners picked on the device (src/singlepath.jl:224-233): a failed run draws afresh with `fail_seeds[k]`"
function _pool_build_best!(eng::Engine, N_r::Int, fail_seeds::Vector{UInt64})
    check(ccall((:pfmi_pool_build_best, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{UInt64}), eng.ptr, N_r, fail_seeds))
end
function _pool_winners(eng::Engine, K::Int)
    pts = Vector{Int64}(undef, K); sd = Vector{UInt64}(undef, K); ok = Vector{Int32}(undef, K)
    check(ccall((:pfmi_pool_winners, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt64}, Ptr{Int32}), eng.ptr, pts, sd, ok))
    return pts, sd, ok .!= 0
end

"""
    multipathfinder(engines::Vector{Engine}, target::DeviceTarget, ndraws; nruns | init, optim_fun = nothing, kwargs...)

The reference's `multipathfinder` (src/multipath.jl:118-245) for a device-resident target on one or several GPUs driven by this
Julia thread.  Built-in targets (`GaussTarget`, `FunnelTarget`) are optimised on the device (`pfmi_optimize_batch`, this
repository's own L-BFGS: trajectory parity with Optim.jl is not claimed -- the trace is the hot path's INPUT); a
`DeviceClosureTarget` needs its host twin `optim_fun::SciMLBase.OptimizationFunction` for `Pathfinder.optimize_with_trace`.
The result does not depend on `length(engines)` (test/multipath.jl:107-140 extended to the GPU count).
"""
function multipathfinder(engines::Vector{Engine}, target::DeviceTarget, ndraws::Int;
                         init=nothing, input=target, nruns::Int=init === nothing ? -1 : length(init),
                         ndraws_elbo::Int=Pathfinder.DEFAULT_NDRAWS_ELBO,
                         ndraws_per_run::Int=max(ndraws_elbo, cld(ndraws, max(nruns, 1))),
                         rng::Random.AbstractRNG=Random.default_rng(),
                         history_length::Int=Pathfinder.DEFAULT_HISTORY_LENGTH, importance::Bool=true, ntries::Int=1_000,
                         init_scale=2, init_sampler=Pathfinder.UniformSampler(init_scale), optim_fun=nothing,
                         optimizer=Pathfinder.default_optimizer(history_length), maxiters::Int=1_000, g_tol::Float64=1e-8,
                         comm::Union{Nothing,Comm}=nothing, kwargs...)
    d = dimension(target)
    if init === nothing
        nruns > 0 || throw(ArgumentError("A positive `nruns` must be set or `init` must be provided."))      # :148-150
    else
        nruns = length(init)
    end
    G = length(engines); blocks = _blocks(nruns, G)
    if ndraws > ndraws_per_run * nruns
        @warn "More draws requested than total number of draws across replicas. Draws will not be unique."
    end
    on_device = !(target isa DeviceClosureTarget)
    on_device || optim_fun !== nothing || throw(ArgumentError("a DeviceClosureTarget needs `optim_fun` (its host twin) for the optimiser"))
    logp = on_device ? (x -> logdensity(target, x)) : (x -> -optim_fun.f(x, nothing))                     # :159
    foreach(e -> set_target!(e, target), engines)
    run_seeds = rand!(rng, Vector{UInt64}(undef, nruns))                                                    # :162
    rngs = [Random.seed!(copy(rng), s) for s in run_seeds]                                                  # :189-193
    # each run samples its own start with its seeded rng (src/singlepath.jl:167-168 inside the run of :190-193), after run_seeds
    x0 = init === nothing ? [init_sampler(rngs[k], Vector{Float64}(undef, d)) for k in 1:nruns] : [copy(x) for x in init]
    itry = ones(Int, nruns); pending = collect(1:nruns); success = falses(nruns)
    fit_seeds = [UInt64[] for _ in 1:nruns]; fail_seeds = zeros(UInt64, nruns)
    traces = Vector{Any}(undef, nruns); sols = Vector{Any}(undef, nruns)
    batches = Vector{Batch}(undef, G); elbos = Vector{ElboBatch}(undef, G)
    cm = comm === nothing ? comm_for(engines) : comm               # cached per engine set (ADVICE r3)
    local k̂::Float64, M::Int, idx::Vector{Int64}, draws_::Matrix{Float64}
    rs_seed = rand(rng, UInt64)                                       # _resample's draw from the top-level rng (:225)
    while true                                                        # the retry loop of src/singlepath.jl:259-283, batched
        npts = Vector{Vector{Int64}}(undef, G)
        if on_device                                                  # every run of an engine in one launch; nothing waits yet
            for (g, eng) in enumerate(engines)
                X0 = reduce(hcat, x0[blocks[g]])                      # d x K_g column-major == K_g x d point-major
                eng.generation += 1
                check(ccall((:pfmi_optimize_batch_enqueue, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Int32, Float64),
                            eng.ptr, length(blocks[g]), X0, history_length, maxiters, g_tol))
            end
            for (g, eng) in enumerate(engines)
                npts[g] = Vector{Int64}(undef, length(blocks[g]))
                check(ccall((:pfmi_optimize_batch_wait, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Int64}), eng.ptr, npts[g]))
            end
        else
            for k in pending
                prob = SciMLBase.OptimizationProblem(optim_fun, x0[k], nothing)
                sols[k], traces[k] = Pathfinder.optimize_with_trace(prob, deepcopy(optimizer); kwargs...)   # host, src/optimize.jl:35-59
            end
            for (g, eng) in enumerate(engines)
                tr = traces[blocks[g]]
                npts[g] = Int64[length(t.points) for t in tr]
                theta = reduce(hcat, reduce(vcat, [t.points for t in tr])); grad = reduce(hcat, reduce(vcat, [t.gradients for t in tr]))
                eng.generation += 1
                check(ccall((:pfmi_set_traces, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Int32, Ptr{Float64}, Ptr{Float64}),
                            eng.ptr, length(tr), npts[g], d, theta, grad))
            end
        end
        foreach(e -> _fit_enqueue!(e, history_length, 1e-12), engines)
        seeds = Vector{Vector{UInt64}}(undef, G)
        for (g, eng) in enumerate(engines)
            sd = UInt64[]
            for (j, k) in enumerate(blocks[g])
                L = Int(npts[g][j]) - 1
                if k in pending
                    fit_seeds[k] = rand!(rngs[k], Vector{UInt64}(undef, L))                                 # src/elbo.jl:2
                    fail_seeds[k] = rand(copy(rngs[k]), UInt64)       # what rand(rng_k, fit_distribution, n) would consume after a failure
                end
                push!(sd, UInt64(0)); append!(sd, fit_seeds[k])
            end
            seeds[g] = sd
            _elbo_enqueue!(eng, sd, ndraws_elbo)
            _pool_build_best!(eng, ndraws_per_run, fail_seeds[blocks[g]])
        end
        # draws_per_component = stack(draws) (:217), _compute_psis_result (:221), _resample (:225) -- enqueued behind the scans
        k̂, M, idx, draws_ = psis_resample(cm, d, ndraws; importance, seed=rs_seed)
        for (g, eng) in enumerate(engines)                            # first waits: everything above is in flight on every GPU
            batches[g] = _fit_finish(eng, npts[g], d)
            any(!=(0), batches[g].status) && throw(LinearAlgebra.PosDefException(Int(first(filter(!=(0), batches[g].status)))))
            elbos[g] = _elbo_wait(batches[g], seeds[g], ndraws_elbo)
        end
        for k in pending
            g = findfirst(r -> k in r, blocks); j = k - first(blocks[g]) + 1
            b, e = batches[g], elbos[g]
            L = b.offsets[j + 1] - b.offsets[j] - 1
            it = e.iteration_opt[j]
            v = it > 0 ? e.value[b.offsets[j] + it + 1] : NaN
            success[k] = L > 0 && it > 0 && !isnan(v) && v != -Inf     # src/singlepath.jl:299, 309-314
        end
        pending = [k for k in pending if !success[k] && itry[k] < ntries]
        isempty(pending) && break
        for k in pending
            itry[k] += 1
            x0[k] = init_sampler(rngs[k], copy(x0[k]))                                                     # :277
        end
    end
    results = Vector{DevicePathfinderResult}(undef, nruns)
    components = Vector{Any}(undef, nruns)
    for (g, eng) in enumerate(engines)
        b, e = batches[g], elbos[g]
        for (j, k) in enumerate(blocks[g])
            it = Int(e.iteration_opt[j])
            success[k] || @warn "Pathfinder failed after $(itry[k]) tries. Increase `ntries`, inspect the model for numerical instability, or provide a more suitable `init_sampler`."
            if b.nrej[j] > 0
                perc = round(b.nrej[j] * (100 // (b.offsets[j + 1] - b.offsets[j])); digits=1)
                @warn "$(b.nrej[j]) ($(perc)%) updates to the inverse Hessian estimate were rejected to keep it positive definite."
            end
            p = b.offsets[j] + it
            seed = success[k] ? e.seeds[p + 1] : rand(rngs[k], UInt64)                                     # == fail_seeds[k]
            tr = on_device ? nothing : traces[k]                        # device traces: get_trace(eng, j - 1, npoints, d) on demand
            results[k] = DevicePathfinderResult(input, optimizer, rngs[k], nothing, logp, it, itry[k], on_device ? nothing : sols[k], tr,
                                                LazyFitDistributions(b, j, Dict{Int,Any}()), LazyELBOEstimates(e, j, Dict{Int,Any}()),
                                                Int(b.nrej[j]), success[k], seed, ndraws_per_run, nothing)
            components[k] = fit_distribution(b, p)
        end
    end
    psis_result = nothing
    if importance
        S = nruns * ndraws_per_run
        w = Vector{Float64}(undef, S); lw = similar(w)
        check(ccall((:pfmi_psis_weights, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}), engines[1].ptr, S, w, lw))
        psis_result = DevicePSISResult(w, lw, k̂, M)
    end
    fit_dist = Distributions.MixtureModel(components)                                                      # :215-216
    return Pathfinder.MultiPathfinderResult(input, optimizer, rng, optim_fun, logp, fit_dist, draws_, cld.(idx .+ 1, ndraws_per_run),
                                            fit_dist, draws_, results, psis_result)
end
multipathfinder(eng::Engine, target::DeviceTarget, ndraws::Int; kwargs...) = multipathfinder([eng], target, ndraws; kwargs...)

"""
    resample(eng, result::Pathfinder.MultiPathfinderResult, ndraws; rng, replace, importance, ndraws_per_run)

src/resample.jl:20-46 on the device: stored draws (re-pooled bit-identically from their seeds) or `ndraws_per_run` fresh
candidates per component (seeds = `rand(rng, UInt64, K)`), PSIS, index selection, gather.
"""
function resample(result::Pathfinder.MultiPathfinderResult, ndraws::Int; rng::Random.AbstractRNG=result.rng, replace::Bool=true,
                  importance::Bool=true, ndraws_per_run::Union{Nothing,Int}=nothing)
    runs = result.pathfinder_results
    b = first(runs).fit_distributions.b
    K = length(runs)
    pts = Int64[b.offsets[k] + runs[k].fit_iteration for k in 1:K]
    if ndraws_per_run === nothing                                     # src/resample.jl:97-101
        N_r = first(runs).ndraws; seeds = UInt64[r.draw_seed for r in runs]
    else                                                              # src/resample.jl:102-109
        N_r = ndraws_per_run; seeds = rand(rng, UInt64, K)
    end
    psis_result = _compute_psis_result(b, pts, seeds, N_r; importance)     # (re)pools on the device; for stored draws the same values
    draws_, ids = _resample(b, rng, psis_result, N_r, K, ndraws; replace)
    # stored draws + stored PSIS: the reference hands the SAME object on (src/resample.jl:31-41, test/multipath.jl:158)
    psis_out = (importance && ndraws_per_run === nothing && result.psis_result !== nothing &&
                length(result.psis_result.weights) == K * N_r) ? result.psis_result : psis_result
    return Pathfinder.MultiPathfinderResult(result.input, result.optimizer, result.rng, result.optim_fun, result.logp,
                                            result.fit_distribution, draws_, ids, result.fit_distribution, draws_, runs, psis_out)
end

# ---- multi-GPU collectives (the `Comm` type itself is defined next to `Engine`) ---------------------------------------------------------
"""
    psis_resample(c, dim, ndraws; importance, replace, seed) -> (pareto_shape, tail_length, idx0, draws)

`_compute_psis_result` + `_resample` over every engine's runs (src/multipath.jl:221-225) in ONE call with one synchronisation:
all-gather of the log-ratio shards, PSIS and index selection replicated, owner gather, sum all-reduce.  A `Comm` of one engine
needs no RCCL.  `idx0` is 0-based (global pool columns).
"""
function psis_resample(c::Comm, dim::Int, ndraws::Int; importance::Bool=true, replace::Bool=true, seed::UInt64)
    idx = Vector{Int64}(undef, ndraws); X = Matrix{Float64}(undef, dim, ndraws)
    k̂ = Ref{Float64}(NaN); M = Ref{Int64}(0)
    check(ccall((:pfmi_comm_psis_resample, libpfmi), Int32,
                (Ptr{Cvoid}, Int64, Int32, Int32, UInt64, Ptr{Float64}, Ref{Float64}, Ref{Int64}, Ptr{Int64}, Ptr{Float64}),
                c.ptr, ndraws, importance, replace, seed, C_NULL, k̂, M, idx, X))
    return k̂[], Int(M[]), idx, X
end
"pooled PSIS over every GPU's runs: one RCCL all-gather of the log-ratio shards, PSIS replicated (src/multipath.jl:221)"
function pool_psis(c::Comm)
    k̂ = Ref{Float64}(NaN); M = Ref{Int64}(0)
    check(ccall((:pfmi_comm_pool_psis, libpfmi), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}), c.ptr, k̂, M))
    return k̂[], Int(M[])
end
"replicated index selection, owner gather, one sum all-reduce (src/multipath.jl:225)"
function resample(c::Comm, rng::Random.AbstractRNG, dim::Int, N_r::Int, ndraws::Int; importance::Bool=true, replace::Bool=true)
    idx = Vector{Int64}(undef, ndraws); X = Matrix{Float64}(undef, dim, ndraws)
    check(ccall((:pfmi_comm_resample, libpfmi), Int32, (Ptr{Cvoid}, Int64, Int32, Int32, UInt64, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}),
                c.ptr, ndraws, importance, replace, rand(rng, UInt64), C_NULL, idx, X))
    return X, cld.(idx .+ 1, N_r)
end

# ---- optional: device L-BFGS for the built-in targets (plays optimize_with_trace, src/optimize.jl:35-59) -------------------------------
function optimize_batch!(eng::Engine, x0::Matrix{Float64}; history_length::Int=6, maxiters::Int=1000, g_tol::Float64=1e-8)
    K = size(x0, 2)                              # x0 is d x K column-major == K x d point-major for the C side
    npts = Vector{Int64}(undef, K)
    eng.generation += 1
    check(ccall((:pfmi_optimize_batch, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Int32, Float64, Ptr{Int64}),
                eng.ptr, K, x0, history_length, maxiters, g_tol, npts))
    return npts
end
# ---- streaming pipeline (round 5): optimise + fit + ELBO scan as ONE dataflow the calling task schedules (include/pfmi.h: pfmi_stream_*).
# What src/singlepath.jl:285-325 does per run, for K runs at once, with the fits and scans of the trace points a path has already produced
# running while the paths are still being optimised.  Slot layout: trace point l of run k is slot k * (maxiters + 1) + l (0-based) of every
# per-point table; `seeds` = the runs' predrawn streams, `maxiters + 1` values per run (`rand!(copy(rng_k), Vector{UInt64}(undef, maxiters + 1))`).
#   stream_enqueue!(eng, x0; ...)  ->  stream_seeds!(eng, seeds)  ->  npts = stream_wait!(eng, K)  ->  _pool_build_best! / psis_resample ...
# stream_wait! only schedules (it returns when the last segment is enqueued); several engines: call stream_pump! on each in turn.
function stream_enqueue!(eng::Engine, x0::Matrix{Float64}, ndraws_elbo::Int; history_length::Int=6, maxiters::Int=1000, g_tol::Float64=1e-8,
                         ϵ::Float64=1e-12, seeds::Union{Nothing,Vector{UInt64}}=nothing)
    K = size(x0, 2)
    eng.generation += 1
    check(ccall((:pfmi_stream_enqueue, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Int32, Float64, Float64, Int64, Ptr{UInt64}),
                eng.ptr, K, x0, history_length, maxiters, g_tol, ϵ, ndraws_elbo, seeds === nothing ? C_NULL : seeds))
    return nothing
end
stream_seeds!(eng::Engine, seeds::Vector{UInt64}) = check(ccall((:pfmi_stream_seeds, libpfmi), Int32, (Ptr{Cvoid}, Ptr{UInt64}), eng.ptr, seeds))
function stream_pump!(eng::Engine)
    fin = Ref{Int32}(0)
    check(ccall((:pfmi_stream_pump, libpfmi), Int32, (Ptr{Cvoid}, Ref{Int32}), eng.ptr, fin))
    return fin[] != 0
end
function stream_wait!(eng::Engine, K::Integer)
    npts = Vector{Int64}(undef, K)
    check(ccall((:pfmi_stream_wait, libpfmi), Int32, (Ptr{Cvoid}, Ptr{Int64}), eng.ptr, npts))
    return npts
end
"the engine's default `Hinit` (`:gilbert_init` or `:nocedal_wright_scaling`): what `pfmi_fit_batch` and the streaming pipeline use"
set_hinit!(eng::Engine, h) = check(ccall((:pfmi_set_hinit, libpfmi), Int32, (Ptr{Cvoid}, Int32), eng.ptr, _hinit_code(h)))
"give up an outstanding stream_enqueue! (an exception between enqueue and wait); no-op otherwise"
stream_cancel!(eng::Engine) = check(ccall((:pfmi_stream_cancel, libpfmi), Int32, (Ptr{Cvoid},), eng.ptr))
"1: pfmi_get_fit_status / pfmi_elbo_batch_wait / pfmi_psis_weights only queue their downloads (delivered by the next wait); 0: normal; -1: drop"
defer_downloads!(eng::Engine, mode::Integer) = check(ccall((:pfmi_defer_downloads, libpfmi), Int32, (Ptr{Cvoid}, Int32), eng.ptr, mode))
function psis_resample_enqueue!(c::Comm, ndraws::Int; importance::Bool=true, replace::Bool=true, seed::UInt64)
    check(ccall((:pfmi_comm_psis_resample_enqueue, libpfmi), Int32, (Ptr{Cvoid}, Int64, Int32, Int32, UInt64, Ptr{Float64}),
                c.ptr, ndraws, importance, replace, seed, C_NULL))
end
function psis_resample_wait!(c::Comm, dim::Int, ndraws::Int)
    idx = Vector{Int64}(undef, ndraws); X = Matrix{Float64}(undef, dim, ndraws)
    k̂ = Ref{Float64}(NaN); M = Ref{Int64}(0)
    check(ccall((:pfmi_comm_psis_resample_wait, libpfmi), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}, Ptr{Int64}, Ptr{Float64}), c.ptr, k̂, M, idx, X))
    return k̂[], Int(M[]), idx, X
end
function get_trace(eng::Engine, k::Integer, npoints::Integer, d::Integer)   # OptimizationTrace, src/optimize.jl:94-100
    θ = Matrix{Float64}(undef, d, npoints); g = similar(θ); lp = Vector{Float64}(undef, npoints)
    check(ccall((:pfmi_get_trace, libpfmi), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), eng.ptr, k, θ, lp, g))
    return Pathfinder.OptimizationTrace(collect(eachcol(θ)), lp, collect(eachcol(g)))
end

# ---- the PDMats surface of a fitted covariance on the device (src/woodbury.jl:326-423), as the HMC extensions use it ----------------------
# (ext/PathfinderAdvancedHMCExt.jl:17-23 builds a metric from Σ; its sampling calls unwhiten!/mul!/quad on it)
struct DeviceWoodbury <: PDMats.AbstractPDMat{Float64}
    b::Batch
    point::Int64      # 0-based global trace-point index of the fit
end
const OP = (unwhiten=Int32(0), whiten=Int32(1), rmul=Int32(2), invunwhiten=Int32(3), mul=Int32(4), solve=Int32(5),
            quad=Int32(6), invquad=Int32(7))
function _apply(W::DeviceWoodbury, op::Int32, x::AbstractVecOrMat{Float64})
    _live(W.b.eng, W.b.gen)
    X = Matrix{Float64}(reshape(x, W.b.dim, :))
    N = size(X, 2)
    out = op >= OP.quad ? Vector{Float64}(undef, N) : similar(X)
    check(ccall((:pfmi_woodbury_apply, libpfmi), Int32, (Ptr{Cvoid}, Int64, Int32, Int64, Ptr{Float64}, Ptr{Float64}),
                W.b.eng.ptr, W.point, op, N, X, out))
    return (x isa AbstractVector && op < OP.quad) ? vec(out) : (x isa AbstractVector ? out[1] : out)
end
Base.size(W::DeviceWoodbury) = (W.b.dim, W.b.dim)
PDMats.unwhiten(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.unwhiten, x)          # src/woodbury.jl:401-406
PDMats.whiten(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.whiten, x)              # src/woodbury.jl:410-415
PDMats.invunwhiten(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.invunwhiten, x)    # src/woodbury.jl:417-422
Base.:*(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.mul, x)                       # src/woodbury.jl:340-349
Base.:\(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.solve, x)
PDMats.quad(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.quad, x)                  # src/woodbury.jl:384-397
PDMats.invquad(W::DeviceWoodbury, x::AbstractVecOrMat{Float64}) = _apply(W, OP.invquad, x)            # src/woodbury.jl:369-382
function LinearAlgebra.diag(W::DeviceWoodbury)                                                        # src/woodbury.jl:326-329
    _live(W.b.eng, W.b.gen)
    out = Vector{Float64}(undef, W.b.dim)
    check(ccall((:pfmi_woodbury_diag, libpfmi), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}), W.b.eng.ptr, W.point, out))
    return out
end

# inv(W) and W * c (src/woodbury.jl:317-321, 357-360) build new Pathfinder.WoodburyPDMat objects on the host: materialise the fit
# (pfmi_get_fit -> Pathfinder.WoodburyPDMat with its WoodburyPDFactorization) and let the reference's own methods do the rest.
materialise(W::DeviceWoodbury) = fit_distribution(W.b, W.point).Σ
Base.inv(W::DeviceWoodbury) = inv(materialise(W))
Base.:*(W::DeviceWoodbury, c::Real) = materialise(W) * c
Base.:*(c::Real, W::DeviceWoodbury) = W * c
# the rest of what test/woodbury.jl:228-309 exercises: symmetric, + UniformScaling through the materialised matrix, right division
Base.adjoint(W::DeviceWoodbury) = W
Base.transpose(W::DeviceWoodbury) = W
Base.:+(W::DeviceWoodbury, c::LinearAlgebra.UniformScaling) = materialise(W) + c
Base.:+(c::LinearAlgebra.UniformScaling, W::DeviceWoodbury) = c + materialise(W)
Base.:/(x::AbstractVecOrMat{Float64}, W::DeviceWoodbury) = Matrix(transpose(W \ Matrix(transpose(x))))
PDMats.dim(W::DeviceWoodbury) = W.b.dim

end # module

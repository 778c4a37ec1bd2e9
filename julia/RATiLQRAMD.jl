# RATiLQRAMD.jl -- thin `ccall` glue that keeps the reference's API surface
# (OptimalControlProblem / ILEQGSolver / CrossEntropyBilevelOptimizationSolver / NelderMeadBilevelOptimizationSolver / solve!)
# and sends the iLEQG + Cross-Entropy hot path to libratilqr_hip.so (C ABI: include/ratilqr.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia` binary.  The Python mirror
# (ratilqr.jl_amd/*.py) binds exactly the same entry points and is what the test-suite runs.
#
# Usage (drop-in next to `using RATiLQR`):
#   problem = LQRiskSensitiveProblem(A, B, Q, R, Qf, W, N)             # device model family
#   solver  = AMDCrossEntropyBilevelOptimizationSolver(num_samples=1024, num_elite=100)
#   θ_opt, x_array, l_array, L_array, value, θ_min, θ_max = solve!(solver, problem, x_0, u_array, rng; kl_bound=0.1)
module RATiLQRAMD

using LinearAlgebra, Random

const LIB = get(ENV, "RATILQR_SO", joinpath(@__DIR__, "..", "ratilqr.jl_amd", "csrc", "libratilqr_hip.so"))

abstract type OptimalControlProblem end            # optimal_control_problems.jl:12

"Device model family replacing FiniteHorizonRiskSensitiveOptimalControlProblem(f, c, h, W, N) (:67-73)."
struct LQRiskSensitiveProblem <: OptimalControlProblem
    A::Matrix{Float64}; B::Matrix{Float64}
    Q::Array{Float64}; R::Array{Float64}; P::Array{Float64}      # n×n[×N], m×m[×N], m×n[×N]
    qv::Array{Float64}; rv::Array{Float64}; q0::Vector{Float64}
    Qf::Matrix{Float64}; qvf::Vector{Float64}; q0f::Float64
    kappa::Float64
    W::Array{Float64}                                            # n×n[×N]
    N::Int64
end

# mirrors `struct rat_problem_desc` (include/ratilqr.h)
struct ProblemDesc
    model::Int32; n::Int32; m::Int32; N::Int32; cost_tv::Int32; W_tv::Int32
    A::Ptr{Float64}; B::Ptr{Float64}; Q::Ptr{Float64}; R::Ptr{Float64}; P::Ptr{Float64}
    qv::Ptr{Float64}; rv::Ptr{Float64}; q0::Ptr{Float64}; Qf::Ptr{Float64}; qvf::Ptr{Float64}
    q0f::Float64; kappa::Float64
    pl_a::Float64; pl_b::Float64; pl_p::Float64; pl_pu::Float64; pl_cx::Float64; pl_cu::Float64; pl_h::Float64
    W::Ptr{Float64}
end

# mirrors `struct rat_ileqg_opts` -- the keyword arguments of ILEQGSolver (ileqg.jl:191-194)
struct IleqgOpts
    μ_min::Float64; Δ_0::Float64; λ::Float64; d::Float64; iter_max::Int64
    ϵ_init::Float64; ϵ_min::Float64; adaptive_ϵ_init::Int32
end

# mirrors `struct rat_ce_solver` (cross_entropy_bilevel_optimization.jl:70-98)
mutable struct CeState
    num_samples::Int64; num_elite::Int64; iter_max::Int64; λ::Float64; use_θ_max::Int32
    μ_init::Float64; σ_init::Float64; μ::Float64; σ::Float64; θ_max::Float64; θ_min::Float64
    iter_current::Int64; n_solves::Int64; n_redraws::Int64; n_final_retries::Int64
end

check(rc) = rc == 0 || error("libratilqr_hip: rc=$rc: " * unsafe_string(ccall((:rat_last_error, LIB), Cstring, ())))

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(opts::IleqgOpts, max_batch::Integer, spec_eps::Integer, device::Integer)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:rat_create, LIB), Int32, (Ref{IleqgOpts}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
                    opts, max_batch, spec_eps, device, out))
        h = new(out[])
        finalizer(x -> ccall((:rat_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ptr), h)
    end
end

function problem_set!(h::Handle, p::LQRiskSensitiveProblem)
    n, m = size(p.B)
    GC.@preserve p begin
        d = ProblemDesc(1, n, m, p.N, ndims(p.Q) == 3, ndims(p.W) == 3,
                        pointer(p.A), pointer(p.B), pointer(p.Q), pointer(p.R), pointer(p.P), pointer(p.qv), pointer(p.rv),
                        pointer(p.q0), pointer(p.Qf), pointer(p.qvf), p.q0f, p.kappa, 0, 0, 0, 0, 0, 0, 0, pointer(p.W))
        check(ccall((:rat_problem_set, LIB), Int32, (Ptr{Cvoid}, Ref{ProblemDesc}), h.ptr, d))
    end
end

"ILEQGSolver(problem; kwargs...) -- ileqg.jl:164-208"
mutable struct ILEQGSolver
    opts::IleqgOpts
    h::Handle
    value_current::Float64; iter_current::Int64; ϵ_history::Vector{Tuple{Float64,Float64}}
end
function ILEQGSolver(problem::LQRiskSensitiveProblem; μ_min=1e-6, Δ_0=2.0, λ=0.5, d=1e-2, iter_max=100, ϵ_init=1.0,
                     adaptive_ϵ_init=false, ϵ_min=1e-6, max_batch=1, spec_eps=1, device=0)
    o = IleqgOpts(μ_min, Δ_0, λ, d, iter_max, ϵ_init, ϵ_min, adaptive_ϵ_init)
    h = Handle(o, max_batch, spec_eps, device)          # rat_create validates the @assert ranges of :195-201
    problem_set!(h, problem)
    ILEQGSolver(o, h, Inf, 0, Tuple{Float64,Float64}[])
end

"solve!(ileqg, problem, x_0, u_array; θ) -- ileqg.jl:635-659"
function solve!(s::ILEQGSolver, problem::LQRiskSensitiveProblem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}};
                θ::Float64, verbose=false)
    n, m = size(problem.B); N = problem.N
    u = reduce(hcat, u_array)                            # m×N column-major == time-slowest flat buffer
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    value = Ref(0.0); status = Ref(Int32(0)); iters = Ref(Int32(0)); hn = Ref(Int64(0)); hist = Matrix{Float64}(undef, 2, 4096)
    check(ccall((:rat_ileqg_solve, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64},
                 Ref{Int32}, Ref{Int32}, Ptr{Float64}, Int64, Ref{Int64}),
                s.h.ptr, x_0, u, θ, x, l, L, value, status, iters, hist, 4096, hn))
    status[] in (1, 2) && throw(AssertionError("M: (inv(W) - θ*S) is not PSD"))       # the @assert at :366 / :440
    status[] in (0, 3) || error("iLEQG solve failed with status $(status[])")
    s.value_current = value[]; s.iter_current = iters[]
    s.ϵ_history = [(hist[1, i], hist[2, i]) for i in 1:min(hn[], 4096)]
    return [x[:, t] for t in 1:N+1], [l[:, t] for t in 1:N], [L[:, :, t] for t in 1:N], value[], copy(s.ϵ_history)
end

"""
simulate_dynamics(problem, x_0, u_array, rng) -- ileqg.jl:44-55 -- and simulate_dynamics(problem, x_array, l_array, L_array, rng)
-- ileqg.jl:94-109, K Monte-Carlo rollouts per call.  `z` (n×N×K standard-normal draws, e.g. `randn(rng, n, N, K)`) keeps the
caller's rng in charge of the noise; `z = nothing` uses the device generator keyed by `seed`.  Returns the K state arrays,
(the K control arrays,) and the realised cost of every rollout (integrate_cost, ileqg.jl:115-124).
"""
function simulate_dynamics_noisy(s::ILEQGSolver, problem::LQRiskSensitiveProblem, x_nom, l_array::Vector{Vector{Float64}},
                                 L_array::Union{Nothing,Vector{Matrix{Float64}}}=nothing; K::Integer=1, z=nothing, seed::UInt64=UInt64(0))
    n, m = size(problem.B); N = problem.N
    z === nothing || (K = size(z, 3))
    xn = x_nom isa Vector{Float64} ? x_nom : reduce(hcat, x_nom)
    l = reduce(hcat, l_array)
    L = L_array === nothing ? C_NULL : cat(L_array...; dims=3)
    x = Array{Float64}(undef, n, N + 1, K); u = Array{Float64}(undef, m, N, K); cost = Vector{Float64}(undef, K); dom = Ref(Int32(0))
    check(ccall((:rat_rollout_noisy, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, UInt64, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ref{Int32}),
                s.h.ptr, xn, l, L, K, z === nothing ? C_NULL : z, seed, x, u, cost, dom))
    dom[] != 0 && throw(DomainError(NaN, "simulate_dynamics"))
    xs = [[x[:, t, k] for t in 1:N+1] for k in 1:K]
    return L_array === nothing ? (xs, cost) : (xs, [[u[:, t, k] for t in 1:N] for k in 1:K], cost)
end

"CrossEntropyBilevelOptimizationSolver(; kwargs...) -- cross_entropy_bilevel_optimization.jl:70-127"
mutable struct AMDCrossEntropyBilevelOptimizationSolver
    opts::IleqgOpts
    c::CeState
    spec_eps::Int; device::Int
    h::Union{Nothing,Handle}
end
function AMDCrossEntropyBilevelOptimizationSolver(; μ_min_ileqg=1e-6, Δ_0_ileqg=2.0, λ_ileqg=0.5, d_ileqg=1e-2, iter_max_ileqg=100,
        adaptive_ϵ_init_ileqg=false, ϵ_init_ileqg=1.0, ϵ_min_ileqg=1e-6, μ_init=1.0, σ_init=2.0, num_samples=10, num_elite=3,
        iter_max=5, λ=0.5, use_θ_max=false, spec_eps=1, device=0)
    o = IleqgOpts(μ_min_ileqg, Δ_0_ileqg, λ_ileqg, d_ileqg, iter_max_ileqg, ϵ_init_ileqg, ϵ_min_ileqg, adaptive_ϵ_init_ileqg)
    c = CeState(num_samples, num_elite, iter_max, λ, use_θ_max, μ_init, σ_init, μ_init, σ_init, 0.0, Inf, 0, 0, 0, 0)
    AMDCrossEntropyBilevelOptimizationSolver(o, c, spec_eps, device, nothing)
end

function handle!(s::AMDCrossEntropyBilevelOptimizationSolver, problem)
    if s.h === nothing
        s.h = Handle(s.opts, s.c.num_samples, s.spec_eps, s.device)
        problem_set!(s.h, problem)
    end
    s.h
end

"compute_cost(ce_solver, problem, x, u_array, θ_array, kl_bound) -- :173-195 (replaces the remotecall_fetch fan-out)"
function compute_cost(s::AMDCrossEntropyBilevelOptimizationSolver, problem, x::Vector{Float64}, u_array, θ_array::Vector{Float64}, kl_bound::Float64)
    h = handle!(s, problem); cost = similar(θ_array)
    check(ccall((:rat_ce_compute_cost, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}),
                h.ptr, x, reduce(hcat, u_array), θ_array, length(θ_array), kl_bound, cost))
    cost
end

"set_initial!(ce_solver, problem, x, u_array): uploads the initial state and nominal controls used by the device-pointer entry points"
function set_initial!(s::AMDCrossEntropyBilevelOptimizationSolver, problem, x::Vector{Float64}, u_array)
    h = handle!(s, problem)
    check(ccall((:rat_set_initial, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h.ptr, x, reduce(hcat, u_array)))
end

"""compute_cost with θ and the costs resident in HBM (device pointers, e.g. `pointer(::ROCArray{Float64})` of AMDGPU.jl); the initial
state / controls are the ones of the last `compute_cost` / `set_initial!`.  `enqueue = true` returns after the launch: the batch is
ordered on the handle's HIP stream (`ccall((:rat_stream, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), h.ptr)`)."""
function compute_cost_dev!(s::AMDCrossEntropyBilevelOptimizationSolver, problem, θ_dev::Ptr{Float64}, B::Integer, kl_bound::Float64,
                           cost_dev::Ptr{Float64}; enqueue::Bool=false)
    h = handle!(s, problem)
    if enqueue
        check(ccall((:rat_ce_compute_cost_enqueue, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}), h.ptr, θ_dev, B, kl_bound, cost_dev))
    else
        check(ccall((:rat_ce_compute_cost_dev, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}), h.ptr, θ_dev, B, kl_bound, cost_dev))
    end
    nothing
end

"solve!(ce_solver, problem, x_0, u_array, rng; kl_bound) -- :364-415.  `rng` supplies the N(0,1) stream (randn(rng, k))."
function solve!(s::AMDCrossEntropyBilevelOptimizationSolver, problem::LQRiskSensitiveProblem, x_0::Vector{Float64},
                u_array::Vector{Vector{Float64}}, rng::AbstractRNG; kl_bound::Float64, verbose=false, stream_len=1 << 20)
    @assert kl_bound >= 0 "KL Divergence Bound must be non-negative"
    h = handle!(s, problem); n, m = size(problem.B); N = problem.N
    z = randn(rng, stream_len)                           # θ = μ + σ z, consumed in order (get_positive_samples :233-246)
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    θ = Ref(0.0); val = Ref(0.0); θmin = Ref(0.0); θmax = Ref(0.0)
    GC.@preserve z begin
        check(ccall((:rat_ce_set_stream, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), h.ptr, z, length(z)))
        check(ccall((:rat_ce_solve, LIB), Int32,
                    (Ptr{Cvoid}, Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}, Ptr{Float64}, Ptr{Float64},
                     Ptr{Float64}, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                    h.ptr, s.c, x_0, reduce(hcat, u_array), kl_bound, θ, x, l, L, val, θmin, θmax))
    end
    return θ[], [x[:, t] for t in 1:N+1], [l[:, t] for t in 1:N], [L[:, :, t] for t in 1:N], val[], θmin[], θmax[]
end

# mirrors `struct rat_nm_solver` (nelder_mead_bilevel_optimization.jl:72-128); c_high / c_low persist across solve! calls as in
# the reference (initialize! does not reset them, :164-168)
mutable struct NmState
    α::Float64; β::Float64; γ::Float64; ϵ::Float64; λ::Float64
    iter_max::Int64
    θ_high_init::Float64; θ_low_init::Float64
    iter_current::Int64
    θ_high::Float64; θ_low::Float64
    has_c_high::Int32; has_c_low::Int32
    c_high::Float64; c_low::Float64
    n_solves::Int64; n_batches::Int64
end

"NelderMeadBilevelOptimizationSolver(; kwargs...) -- RAT iLQR++, nelder_mead_bilevel_optimization.jl:72-128"
mutable struct AMDNelderMeadBilevelOptimizationSolver
    opts::IleqgOpts
    c::NmState
    device::Int
    h::Union{Nothing,Handle}
end
function AMDNelderMeadBilevelOptimizationSolver(; μ_min_ileqg=1e-6, Δ_0_ileqg=2.0, λ_ileqg=0.5, d_ileqg=1e-2, iter_max_ileqg=100,
        adaptive_ϵ_init_ileqg=false, ϵ_init_ileqg=1.0, ϵ_min_ileqg=1e-6, α=1.0, β=2.0, γ=0.5, ϵ=1e-2, λ=0.5,
        iter_max=100, θ_high_init=3.0, θ_low_init=1e-8, device=0)
    o = IleqgOpts(μ_min_ileqg, Δ_0_ileqg, λ_ileqg, d_ileqg, iter_max_ileqg, ϵ_init_ileqg, ϵ_min_ileqg, adaptive_ϵ_init_ileqg)
    c = NmState(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    ccall((:rat_nm_default, LIB), Cvoid, (Ref{NmState},), c)            # the constructor defaults of :102-128
    c.α, c.β, c.γ, c.ϵ, c.λ, c.iter_max = α, β, γ, ϵ, λ, iter_max
    c.θ_high_init = c.θ_high = θ_high_init
    c.θ_low_init = c.θ_low = θ_low_init
    AMDNelderMeadBilevelOptimizationSolver(o, c, device, nothing)
end
function handle!(s::AMDNelderMeadBilevelOptimizationSolver, problem)
    if s.h === nothing
        s.h = Handle(s.opts, 6, 1, s.device)                             # one step! asks for at most six vertices
        problem_set!(s.h, problem)
    end
    s.h
end

"solve!(nm_solver, problem, x_0, u_array; kl_bound) -- nelder_mead_bilevel_optimization.jl:276-352"
function solve!(s::AMDNelderMeadBilevelOptimizationSolver, problem::LQRiskSensitiveProblem, x_0::Vector{Float64},
                u_array::Vector{Vector{Float64}}; kl_bound::Float64, verbose=false)
    @assert kl_bound >= 0 "KL Divergence Bound must be non-negative"
    h = handle!(s, problem); n, m = size(problem.B); N = problem.N
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    θ = Ref(0.0); val = Ref(0.0); status = Ref(Int32(0))
    check(ccall((:rat_nm_solve, LIB), Int32,
                (Ptr{Cvoid}, Ref{NmState}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ref{Float64}, Ref{Int32}),
                h.ptr, s.c, x_0, reduce(hcat, u_array), kl_bound, θ, x, l, L, val, status))
    status[] in (1, 2) && throw(AssertionError("M: (inv(W) - θ*S) is not PSD"))
    status[] in (0, 3) || error("final iLEQG solve failed with status $(status[])")
    return θ[], [x[:, t] for t in 1:N+1], [l[:, t] for t in 1:N], [L[:, :, t] for t in 1:N], val[]
end

export OptimalControlProblem, LQRiskSensitiveProblem, ILEQGSolver, AMDCrossEntropyBilevelOptimizationSolver,
       AMDNelderMeadBilevelOptimizationSolver, solve!, compute_cost, simulate_dynamics_noisy
end

# RATiLQRAMD.jl -- Julia host side of the MI355X path: the reference's API surface (src/RATiLQR.jl:20-53) over `ccall` into
# libratilqr_hip.so (C ABI: include/ratilqr.h).  Drop this file next to the reference package and `using .RATiLQRAMD`.
#
# Names are the reference's own (ILEQGSolver, CrossEntropyBilevelOptimizationSolver, NelderMeadBilevelOptimizationSolver,
# CrossEntropyDirectOptimizationSolver, simulate_dynamics, approximate_model, solve_approximate_dp!, ..., solve!); methods dispatch on
# the problem type:
#   * LQRiskSensitiveProblem / PowerLawRiskSensitiveProblem / LQGenerativeProblem -- compiled-in device model families: everything
#     runs on the GPU(s);
#   * any other problem (the reference's closure structs): forwarded to the reference package's own CPU methods when `RATiLQR` is
#     loaded in `Main` (closures cannot cross a C ABI into a kernel; SURVEY.md section 7 "hard parts").
#
# NOT EXECUTED in the build container (no `julia` binary): tests/test_cpu_julia_shim.py checks every `ccall` (name, arity, integer
# widths, pointer-ness, return type) and every mirrored struct (field order and widths) against include/ratilqr.h mechanically.
module RATiLQRAMD

using LinearAlgebra
using Random

const LIB = get(ENV, "RATILQR_HIP_LIB", joinpath(@__DIR__, "..", "ratilqr.jl_amd", "csrc", "libratilqr_hip.so"))

abstract type OptimalControlProblem end
abstract type DeviceRiskSensitiveProblem <: OptimalControlProblem end

"""LQ family (RAT_MODEL_LQ): x' = A x + B u + κ x.^3 + w, w ~ N(0, W(k)); c_k = ½x'Q_k x + ½u'R_k u + u'P_k x + qv_k'x + rv_k'u + q0_k;
h = ½x'Qf x + qvf'x + q0f.  Cost tables are matrices (constant) or 3-d arrays with time last (time-varying); vectors / 2-d likewise."""
struct LQRiskSensitiveProblem <: DeviceRiskSensitiveProblem
    A::Matrix{Float64}; B::Matrix{Float64}
    Q::Array{Float64}; R::Array{Float64}; P::Array{Float64}; qv::Array{Float64}; rv::Array{Float64}; q0::Vector{Float64}
    Qf::Matrix{Float64}; qvf::Vector{Float64}; q0f::Float64; kappa::Float64
    W::Array{Float64}
    N::Int64
end

"""Power-law family (RAT_MODEL_POWERLAW, test/ileqg_test.jl:151-155): f = x.^a + u.^b (n == m), c = cx Σ x.^p + cu Σ u.^pu, h = hconst."""
struct PowerLawRiskSensitiveProblem <: DeviceRiskSensitiveProblem
    n::Int64; N::Int64
    W::Array{Float64}
    a::Float64; b::Float64; p::Float64; pu::Float64; cx::Float64; cu::Float64; hconst::Float64
end
PowerLawRiskSensitiveProblem(n, N, W; a=1.3, b=1.5, p=2.5, pu=p, cx=1.0, cu=1.0, hconst=1.0) =
    PowerLawRiskSensitiveProblem(n, N, W, a, b, p, pu, cx, cu, hconst)

"""Generative family of the PETS path (optimal_control_problems.jl:126-131): f_stochastic = A x + B u + κ x.^3 + w over `lq`'s tables."""
struct LQGenerativeProblem <: OptimalControlProblem
    lq::LQRiskSensitiveProblem
    l1u::Float64
    noise_kind::Int32                      # 0: N(nmean, nchol nchol'), 1: uniform on [nlo, nhi)^n
    nmean::Vector{Float64}; nchol::Matrix{Float64}; nlo::Float64; nhi::Float64
    tw2::Float64; tmean2::Vector{Float64}; tchol2::Matrix{Float64}
end

dims(p::LQRiskSensitiveProblem) = (size(p.B, 1), size(p.B, 2), p.N)
dims(p::PowerLawRiskSensitiveProblem) = (p.n, p.n, p.N)

# ---- plain-data mirrors of the C structs (field order and widths are checked against include/ratilqr.h) -----------------------------
# mirrors `struct rat_problem_desc`
struct ProblemDesc
    model::Int32; n::Int32; m::Int32; N::Int32; cost_tv::Int32; W_tv::Int32
    A::Ptr{Float64}; B::Ptr{Float64}; Q::Ptr{Float64}; R::Ptr{Float64}; P::Ptr{Float64}; qv::Ptr{Float64}; rv::Ptr{Float64}
    q0::Ptr{Float64}; Qf::Ptr{Float64}; qvf::Ptr{Float64}
    q0f::Float64; kappa::Float64
    pl_a::Float64; pl_b::Float64; pl_p::Float64; pl_pu::Float64; pl_cx::Float64; pl_cu::Float64; pl_h::Float64
    W::Ptr{Float64}
end

# mirrors `struct rat_ileqg_opts` -- the keyword arguments of ILEQGSolver (ileqg.jl:191-194)
struct IleqgOpts
    mu_min::Float64; delta_0::Float64; lambda::Float64; d::Float64; iter_max::Int64
    eps_init::Float64; eps_min::Float64; adaptive_eps_init::Int32
end

# mirrors `struct rat_ce_solver` (cross_entropy_bilevel_optimization.jl:70-98)
mutable struct CeState
    num_samples::Int64; num_elite::Int64; iter_max::Int64; lambda::Float64; use_theta_max::Int32
    mu_init::Float64; sigma_init::Float64; mu::Float64; sigma::Float64; theta_max::Float64; theta_min::Float64
    iter_current::Int64; n_solves::Int64; n_redraws::Int64; n_final_retries::Int64
end

# mirrors `struct rat_nm_solver` (nelder_mead_bilevel_optimization.jl:72-128); c_high / c_low persist across solve! calls as in
# the reference (initialize! does not reset them, :164-168)
mutable struct NmState
    alpha::Float64; beta::Float64; gamma::Float64; eps::Float64; lambda::Float64
    iter_max::Int64
    theta_high_init::Float64; theta_low_init::Float64
    iter_current::Int64
    theta_high::Float64; theta_low::Float64
    has_c_high::Int32; has_c_low::Int32
    c_high::Float64; c_low::Float64
    n_solves::Int64; n_batches::Int64
end

# mirrors `struct rat_gen_problem_desc`
struct GenProblemDesc
    lq::ProblemDesc
    l1u::Float64
    noise_kind::Int32
    nmean::Ptr{Float64}; nchol::Ptr{Float64}
    nlo::Float64; nhi::Float64; tw2::Float64
    tmean2::Ptr{Float64}; tchol2::Ptr{Float64}
end

# mirrors `struct rat_pets_solver` (pets.jl:35-50)
mutable struct PetsState
    num_control_samples::Int64; num_trajectory_samples::Int64; num_elite::Int64; iter_max::Int64
    smoothing_factor::Float64
    N::Int64; m::Int64; iter_current::Int64
    mu_init::Ptr{Float64}; Sigma_init::Ptr{Float64}; mu::Ptr{Float64}; Sigma::Ptr{Float64}
end

last_error() = unsafe_string(ccall((:rat_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 || error("libratilqr_hip: rc=$rc: " * last_error())
version() = ccall((:rat_version, LIB), Int32, ())

# ---- handles ------------------------------------------------------------------------------------------------------------------------
"One rat_handle (one device).  `problem` is the problem whose tables the device currently holds: solvers are problem-agnostic in the
reference (`solve!(solver, problem, ...)`), so every entry point re-binds the handle when it is called with another problem object."
mutable struct Handle
    ptr::Ptr{Cvoid}
    problem::Any
    function Handle(opts::IleqgOpts, max_batch::Integer, spec_eps::Integer, device::Integer)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:rat_create, LIB), Int32, (Ref{IleqgOpts}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
                    opts, max_batch, spec_eps, device, out))
        h = new(out[], nothing)
        finalizer(x -> ccall((:rat_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ptr), h)
    end
end

"All GPUs of a node behind one object (rat_create_multi): contiguous θ shards, one RCCL all-gather of the costs per batch."
mutable struct MultiHandle
    ptr::Ptr{Cvoid}
    problem::Any
    function MultiHandle(opts::IleqgOpts, max_batch::Integer, spec_eps::Integer, devices::AbstractVector{<:Integer})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        dev = collect(Int32, devices)
        check(ccall((:rat_create_multi, LIB), Int32, (Ref{IleqgOpts}, Int32, Int32, Int32, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                    opts, max_batch, spec_eps, length(dev), dev, out))
        h = new(out[], nothing)
        finalizer(x -> ccall((:rat_multi_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ptr), h)
    end
end
n_devices(h::MultiHandle) = ccall((:rat_multi_n_devices, LIB), Int32, (Ptr{Cvoid},), h.ptr)
uses_rccl(h::MultiHandle) = ccall((:rat_multi_uses_rccl, LIB), Int32, (Ptr{Cvoid},), h.ptr) != 0
allgathers(h::MultiHandle) = ccall((:rat_multi_allgathers, LIB), Int64, (Ptr{Cvoid},), h.ptr)
device_handle(h::MultiHandle, i::Integer) = ccall((:rat_multi_handle, LIB), Ptr{Cvoid}, (Ptr{Cvoid}, Int32), h.ptr, i)
"true under the test hook RATILQR_MULTI_LOGICAL=1 (several logical devices per GPU, all-gather by device copies)"
is_logical(h::MultiHandle) = ccall((:rat_multi_is_logical, LIB), Int32, (Ptr{Cvoid},), h.ptr) != 0

# execution path of a handle's batched solves (include/ratilqr.h RAT_PATH_*; results are identical on all of them)
const PATH_AUTO, PATH_ROUNDS, PATH_FUSED, PATH_BLOCK = Int32(0), Int32(1), Int32(2), Int32(3)
set_path!(h::Handle, path::Integer) = check(ccall((:rat_set_path, LIB), Int32, (Ptr{Cvoid}, Int32), h.ptr, path))
get_path(h::Handle, B::Integer) = ccall((:rat_get_path, LIB), Int32, (Ptr{Cvoid}, Int64), h.ptr, B)
# execution switches of a handle (include/ratilqr.h lists the keys: "init_share", "materialize", "wdiag", ...): what used to be
# process-global RATILQR_* environment variables is visible to -- and settable by -- the Julia caller
debug_set!(h::Handle, key::AbstractString, value::Integer) = check(ccall((:rat_debug_set, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), h.ptr, key, value))
function debug_get(h::Handle, key::AbstractString)
    v = Ref(Int64(0))
    check(ccall((:rat_debug_get, LIB), Int32, (Ptr{Cvoid}, Cstring, Ref{Int64}), h.ptr, key, v))
    return v[]
end
function shard_bounds(B::Integer, world::Integer, rank::Integer)
    lo = Ref(Int64(0)); hi = Ref(Int64(0))
    check(ccall((:rat_shard_bounds, LIB), Int32, (Int64, Int32, Int32, Ref{Int64}, Ref{Int64}), B, world, rank, lo, hi))
    return lo[], hi[]                                                       # 0-based [lo, hi)
end

# time-slowest flat buffers of the reference's vectors-of-arrays
flat(v::Vector{Vector{Float64}}) = reduce(hcat, v)
flat(v::Vector{Matrix{Float64}}) = cat(v...; dims=3)
flat(v::AbstractArray{Float64}) = v
unflat_vec(a::AbstractMatrix) = [a[:, t] for t in 1:size(a, 2)]
unflat_mat(a::AbstractArray{Float64,3}) = [a[:, :, t] for t in 1:size(a, 3)]

function with_desc(f::Function, p::LQRiskSensitiveProblem)
    n, m, N = dims(p)
    GC.@preserve p begin
        d = ProblemDesc(1, n, m, N, ndims(p.Q) == 3, ndims(p.W) == 3,
                        pointer(p.A), pointer(p.B), pointer(p.Q), pointer(p.R), pointer(p.P), pointer(p.qv), pointer(p.rv),
                        pointer(p.q0), pointer(p.Qf), pointer(p.qvf), p.q0f, p.kappa, 0, 0, 0, 0, 0, 0, 0, pointer(p.W))
        f(d)
    end
end
function with_desc(f::Function, p::PowerLawRiskSensitiveProblem)
    GC.@preserve p begin
        z = Ptr{Float64}(C_NULL)
        d = ProblemDesc(2, p.n, p.n, p.N, 0, ndims(p.W) == 3, z, z, z, z, z, z, z, z, z, z, 0.0, 0.0,
                        p.a, p.b, p.p, p.pu, p.cx, p.cu, p.hconst, pointer(p.W))
        f(d)
    end
end

function problem_set!(h::Handle, p::DeviceRiskSensitiveProblem)
    with_desc(p) do d
        check(ccall((:rat_problem_set, LIB), Int32, (Ptr{Cvoid}, Ref{ProblemDesc}), h.ptr, d))
    end
    h.problem = p
end
function problem_set!(h::MultiHandle, p::DeviceRiskSensitiveProblem)
    with_desc(p) do d
        check(ccall((:rat_multi_problem_set, LIB), Int32, (Ptr{Cvoid}, Ref{ProblemDesc}), h.ptr, d))
    end
    h.problem = p
end
"Re-bind the handle when it is called with a problem other than the one its device tables were built from."
bind!(h::Union{Handle,MultiHandle}, p) = (h.problem === p || problem_set!(h, p); h)

# =====================================================================================================================================
# iLEQG (src/ileqg.jl)
# =====================================================================================================================================
"ApproximationResult -- ileqg.jl:242-252"
struct ApproximationResult
    q_array::Vector{Float64}; q_vec_array::Vector{Vector{Float64}}; Q_array::Vector{Matrix{Float64}}
    r_array::Vector{Vector{Float64}}; R_array::Vector{Matrix{Float64}}; P_array::Vector{Matrix{Float64}}
    A_array::Vector{Matrix{Float64}}; B_array::Vector{Matrix{Float64}}; W_array::Vector{Matrix{Float64}}
end
"DynamicProgrammingResult -- ileqg.jl:328-335"
struct DynamicProgrammingResult
    s_array::Vector{Float64}; s_vec_array::Vector{Vector{Float64}}; S_array::Vector{Matrix{Float64}}
    g_array::Vector{Vector{Float64}}; G_array::Vector{Matrix{Float64}}; H_array::Vector{Matrix{Float64}}
end

"ILEQGSolver(problem; kwargs...) -- ileqg.jl:164-208"
mutable struct ILEQGSolver
    opts::IleqgOpts
    h::Handle
    μ_min::Float64; μ::Float64; Δ_0::Float64; Δ::Float64; λ::Float64; d::Float64; iter_max::Int64
    ϵ_init::Float64; ϵ_init_init::Float64; ϵ_min::Float64; ϵ_init_auto::Bool
    x_array::Vector{Vector{Float64}}; l_array::Vector{Vector{Float64}}; L_array::Vector{Matrix{Float64}}
    value_current::Float64; iter_current::Int64; d_current::Float64; ϵ_history::Vector{Tuple{Float64,Float64}}
end
function ILEQGSolver(problem::DeviceRiskSensitiveProblem; μ_min=1e-6, Δ_0=2.0, λ=0.5, d=1e-2, iter_max=100, ϵ_init=1.0,
                     adaptive_ϵ_init=false, ϵ_min=1e-6, f_returns_jacobian=false, max_batch=1, spec_eps=1, device=0)
    o = IleqgOpts(μ_min, Δ_0, λ, d, iter_max, ϵ_init, ϵ_min, adaptive_ϵ_init)
    h = Handle(o, max_batch, spec_eps, device)          # rat_create validates the @assert ranges of :195-201
    problem_set!(h, problem)
    ILEQGSolver(o, h, μ_min, μ_min, Δ_0, Δ_0, λ, d, iter_max, ϵ_init, ϵ_init, ϵ_min, adaptive_ϵ_init,
                Vector{Float64}[], Vector{Float64}[], Matrix{Float64}[], Inf, 0, Inf, Tuple{Float64,Float64}[])
end

"simulate_dynamics(problem, x_0, u_array) -- ileqg.jl:18-38"
function simulate_dynamics(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}})
    h = bind!(s.h, problem); n, m, N = dims(problem)
    x = Matrix{Float64}(undef, n, N + 1); dom = Ref(Int32(0))
    check(ccall((:rat_rollout_open, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                h.ptr, x_0, flat(u_array), x, dom))
    dom[] != 0 && throw(DomainError(NaN, "simulate_dynamics"))
    unflat_vec(x)
end
"simulate_dynamics(problem, x_array, l_array, L_array) -- ileqg.jl:62-87"
function simulate_dynamics(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_array::Vector{Vector{Float64}},
                           l_array::Vector{Vector{Float64}}, L_array::Vector{Matrix{Float64}})
    h = bind!(s.h, problem); n, m, N = dims(problem)
    xn = Matrix{Float64}(undef, n, N + 1); un = Matrix{Float64}(undef, m, N); dom = Ref(Int32(0))
    check(ccall((:rat_rollout_feedback, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                h.ptr, flat(x_array), flat(l_array), flat(L_array), xn, un, dom))
    dom[] != 0 && throw(DomainError(NaN, "simulate_dynamics"))
    unflat_vec(xn), unflat_vec(un)
end

"""
simulate_dynamics(problem, x_0, u_array, rng) -- ileqg.jl:44-55 -- and simulate_dynamics(problem, x_array, l_array, L_array, rng)
-- ileqg.jl:94-109, K Monte-Carlo rollouts per call.  `z` (n×N×K standard-normal draws, e.g. `randn(rng, n, N, K)`) keeps the
caller's rng in charge of the noise; `z = nothing` uses the device generator keyed by `seed`.  Returns the K state arrays,
(the K control arrays,) and the realised cost of every rollout (integrate_cost, ileqg.jl:115-124).
"""
function simulate_dynamics_noisy(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_nom, l_array::Vector{Vector{Float64}},
                                 L_array::Union{Nothing,Vector{Matrix{Float64}}}=nothing; K::Integer=1, z=nothing, seed::UInt64=UInt64(0))
    h = bind!(s.h, problem); n, m, N = dims(problem)
    z === nothing || (K = size(z, 3))
    xn = x_nom isa Vector{Float64} ? x_nom : flat(x_nom)
    l = flat(l_array)
    L = L_array === nothing ? C_NULL : flat(L_array)
    x = Array{Float64}(undef, n, N + 1, K); u = Array{Float64}(undef, m, N, K); cost = Vector{Float64}(undef, K); dom = Ref(Int32(0))
    check(ccall((:rat_rollout_noisy, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, UInt64, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ref{Int32}),
                h.ptr, xn, l, L, K, z === nothing ? C_NULL : z, seed, x, u, cost, dom))
    dom[] != 0 && throw(DomainError(NaN, "simulate_dynamics"))
    xs = [[x[:, t, k] for t in 1:N+1] for k in 1:K]
    return L_array === nothing ? (xs, cost) : (xs, [[u[:, t, k] for t in 1:N] for k in 1:K], cost)
end

"integrate_cost(problem, x_array, u_array) -- ileqg.jl:115-124"
function integrate_cost(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_array, u_array)
    h = bind!(s.h, problem); c = Ref(0.0)
    check(ccall((:rat_integrate_cost, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}), h.ptr, flat(x_array), flat(u_array), c))
    c[]
end

"approximate_model(problem, u_array, x_array) -- ileqg.jl:258-322 (analytic derivatives of the model family on the device)"
function approximate_model(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, u_array, x_array)
    h = bind!(s.h, problem); n, m, N = dims(problem)
    q = Vector{Float64}(undef, N + 1); qv = Matrix{Float64}(undef, n, N + 1); Q = Array{Float64}(undef, n, n, N + 1)
    r = Matrix{Float64}(undef, m, N); R = Array{Float64}(undef, m, m, N); P = Array{Float64}(undef, m, n, N)
    A = Array{Float64}(undef, n, n, N); B = Array{Float64}(undef, n, m, N); W = Array{Float64}(undef, n, n, N); dom = Ref(Int32(0))
    check(ccall((:rat_approximate_model, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                h.ptr, flat(u_array), flat(x_array), q, qv, Q, r, R, P, A, B, W, dom))
    dom[] != 0 && throw(DomainError(NaN, "approximate_model"))
    ApproximationResult(q, unflat_vec(qv), unflat_mat(Q), unflat_vec(r), unflat_mat(R), unflat_mat(P), unflat_mat(A), unflat_mat(B), unflat_mat(W))
end

function dp_buffers(n, m, N)
    (Vector{Float64}(undef, N + 1), Matrix{Float64}(undef, n, N + 1), Array{Float64}(undef, n, n, N + 1),
     Matrix{Float64}(undef, m, N), Array{Float64}(undef, m, n, N), Array{Float64}(undef, m, m, N))
end
dp_result(b) = DynamicProgrammingResult(b[1], unflat_vec(b[2]), unflat_mat(b[3]), unflat_vec(b[4]), unflat_mat(b[5]), unflat_mat(b[6]))

"solve_approximate_dp!(ileqg, approx_result; θ) -- ileqg.jl:341-406: writes ileqg.L_array, updates μ, Δ; returns (dp_result, dl_array)"
function solve_approximate_dp!(s::ILEQGSolver, ap::ApproximationResult; θ::Float64=0.0, verbose=false)
    n, m = size(ap.B_array[1]); N = length(ap.B_array)
    L = Array{Float64}(undef, m, n, N); dl = Matrix{Float64}(undef, m, N); b = dp_buffers(n, m, N)
    mu = Ref(s.μ); de = Ref(s.Δ); st = Ref(Int32(0))
    check(ccall((:rat_dp_gain_sweep, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Float64, Ref{Float64}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                s.h.ptr, ap.q_array, flat(ap.q_vec_array), flat(ap.Q_array), flat(ap.r_array), flat(ap.R_array), flat(ap.P_array),
                flat(ap.A_array), flat(ap.B_array), θ, mu, de, L, dl, st, b[1], b[2], b[3], b[4], b[5], b[6]))
    s.μ, s.Δ = mu[], de[]
    st[] == 2 && throw(AssertionError("M: (inv(W) - θ*S) is not PSD"))              # the @assert at :366
    st[] == 0 || error("solve_approximate_dp!: status $(st[])")
    s.L_array = unflat_mat(L)
    dp_result(b), unflat_vec(dl)
end

"solve_approximate_dp(approx_result, L_array, dl_array=nothing; θ, μ) -- ileqg.jl:412-465 (W(k) comes from the solver's problem)"
function solve_approximate_dp(s::ILEQGSolver, ap::ApproximationResult, L_array::Vector{Matrix{Float64}},
                              dl_array::Union{Nothing,Vector{Vector{Float64}}}=nothing; θ::Float64=0.0, μ::Float64=0.0)
    n, m = size(ap.B_array[1]); N = length(ap.B_array)
    b = dp_buffers(n, m, N); st = Ref(Int32(0))
    check(ccall((:rat_dp_policy_eval, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ref{Int32},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                s.h.ptr, ap.q_array, flat(ap.q_vec_array), flat(ap.Q_array), flat(ap.r_array), flat(ap.R_array), flat(ap.P_array),
                flat(ap.A_array), flat(ap.B_array), flat(L_array), dl_array === nothing ? C_NULL : flat(dl_array), θ, μ, st,
                b[1], b[2], b[3], b[4], b[5], b[6]))
    st[] == 0 || throw(AssertionError("M: (inv(W) - θ*S) is not PSD"))              # the @assert at :440
    dp_result(b)
end

# ---- batched sweeps on host-built tiles: the CE batch path of problems the host linearises itself (closures + ForwardDiff) ----------
stack_ap(aps::Vector{ApproximationResult}) = (reduce(vcat, [a.q_array for a in aps]), cat([flat(a.q_vec_array) for a in aps]...; dims=3),
    cat([flat(a.Q_array) for a in aps]...; dims=4), cat([flat(a.r_array) for a in aps]...; dims=3), cat([flat(a.R_array) for a in aps]...; dims=4),
    cat([flat(a.P_array) for a in aps]...; dims=4), cat([flat(a.A_array) for a in aps]...; dims=4), cat([flat(a.B_array) for a in aps]...; dims=4))

"solve_approximate_dp! for B samples in one launch (tiles from `approximate_model` of each sample): returns (L_arrays, dl_arrays, μ, Δ, status)"
function solve_approximate_dp_batch!(s::ILEQGSolver, aps::Vector{ApproximationResult}, θ::Vector{Float64}, μ::Vector{Float64}, Δ::Vector{Float64})
    B = length(aps); n, m = size(aps[1].B_array[1]); N = length(aps[1].B_array)
    t = stack_ap(aps); mu = copy(μ); de = copy(Δ)
    L = Array{Float64}(undef, m, n, N, B); dl = Array{Float64}(undef, m, N, B); st = Vector{Int32}(undef, B)
    check(ccall((:rat_dp_gain_sweep_batch, LIB), Int32,
                (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                s.h.ptr, B, t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], θ, mu, de, L, dl, st))
    [unflat_mat(L[:, :, :, b]) for b in 1:B], [unflat_vec(dl[:, :, b]) for b in 1:B], mu, de, st
end
"solve_approximate_dp (dl = nothing) for B samples in one launch: returns (s_array[1] per sample, Inf where M is not PD; status)"
function solve_approximate_dp_batch(s::ILEQGSolver, aps::Vector{ApproximationResult}, L_arrays::Vector{Vector{Matrix{Float64}}},
                                    θ::Vector{Float64}, μ::Vector{Float64})
    B = length(aps); t = stack_ap(aps)
    L = cat([flat(Lb) for Lb in L_arrays]...; dims=4); value = Vector{Float64}(undef, B); st = Vector{Int32}(undef, B)
    check(ccall((:rat_dp_policy_eval_batch, LIB), Int32,
                (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                s.h.ptr, B, t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], L, θ, μ, value, st))
    value, st
end

"increase_μ_and_Δ! -- ileqg.jl:471-474"
function increase_μ_and_Δ!(s::ILEQGSolver)
    s.Δ = max(s.Δ_0, s.Δ * s.Δ_0)
    s.μ = max(s.μ_min, s.μ * s.Δ)
end
"decrease_μ_and_Δ! -- ileqg.jl:480-488"
function decrease_μ_and_Δ!(s::ILEQGSolver)
    s.Δ = min(1 / s.Δ_0, s.Δ / s.Δ_0)
    s.μ = s.μ * s.Δ >= s.μ_min ? s.μ * s.Δ : 0.0
end

"initialize!(ileqg, problem, x_0, u_array, θ) -- ileqg.jl:214-236, composed from the operator entry points"
function initialize!(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}}, θ::Float64)
    n, m, N = dims(problem)
    s.μ, s.Δ = 0.0, s.Δ_0
    s.d_current, s.iter_current, s.ϵ_init = Inf, 0, s.ϵ_init_init
    empty!(s.ϵ_history)
    s.x_array = simulate_dynamics(s, problem, x_0, u_array)
    s.l_array = copy(u_array)
    s.L_array = [zeros(m, n) for _ in 1:N]
    ap = approximate_model(s, problem, s.l_array, s.x_array)
    s.value_current = solve_approximate_dp(s, ap, s.L_array; θ=θ, μ=s.μ).s_array[1]
end

"line_search!(ileqg, problem, dl_array_new, θ) -- ileqg.jl:494-592, composed from the operator entry points"
function line_search!(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, dl_array_new::Vector{Vector{Float64}}, θ::Float64, verbose=false)
    cur = s.value_current; ϵ = s.ϵ_init; count = 0
    while true
        count += 1
        l_new = [s.l_array[t] + ϵ * dl_array_new[t] for t in 1:problem.N]                         # :509
        x_new, u_new = simulate_dynamics(s, problem, s.x_array, l_new, s.L_array)                 # :517
        ap_new = approximate_model(s, problem, u_new, x_new)                                      # :520
        new = try
            solve_approximate_dp(s, ap_new, s.L_array; θ=θ, μ=s.μ).s_array[1]                     # :522-528
        catch
            ϵ *= s.λ                                                                              # :529-535
            continue
        end
        push!(s.ϵ_history, (ϵ, new - cur))                                                        # :537
        if !(isapprox(new, cur) || new < cur)                                                     # :538
            ϵ *= s.λ                                                                              # :557
            ϵ < s.ϵ_min || continue                                                               # :558 forced accept below ϵ_min
        end
        s.d_current = maximum(norm.(s.l_array .- u_new))                                          # :539 / :559
        s.value_current, s.x_array, s.l_array = new, x_new, u_new
        break
    end
    if s.ϵ_init_auto                                                                              # :582-591
        if count == 1
            s.ϵ_init = min(s.ϵ_init_init, ϵ / s.λ)
        else
            while ϵ < s.ϵ_min
                ϵ = ϵ / s.λ
            end
            s.ϵ_init = ϵ
        end
    end
end

"step!(ileqg, problem, θ) -- ileqg.jl:598-613"
function step!(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, θ::Float64, verbose=false)
    s.iter_current += 1
    ap = approximate_model(s, problem, s.l_array, s.x_array)                                      # :604
    _, dl = solve_approximate_dp!(s, ap; θ=θ)                                                     # :610-611
    line_search!(s, problem, dl, θ, verbose)                                                      # :612
end

"solve!(ileqg, problem, x_0, u_array; θ) -- ileqg.jl:635-659: the whole solve in ONE kernel launch on the device"
function solve!(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}};
                θ::Float64, verbose=false)
    h = bind!(s.h, problem); n, m, N = dims(problem)
    u = flat(u_array)                                    # m×N column-major == time-slowest flat buffer
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    value = Ref(0.0); status = Ref(Int32(0)); iters = Ref(Int32(0)); hn = Ref(Int64(0))
    cap = 256
    hist = Matrix{Float64}(undef, 2, cap)
    while true                                           # ϵ_history is unbounded in the reference: grow and re-run (deterministic) if it did not fit
        check(ccall((:rat_ileqg_solve, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64},
                     Ref{Int32}, Ref{Int32}, Ptr{Float64}, Int64, Ref{Int64}),
                    h.ptr, x_0, u, θ, x, l, L, value, status, iters, hist, cap, hn))
        hn[] <= cap && break
        cap = Int(hn[]); hist = Matrix{Float64}(undef, 2, cap)
    end
    status[] in (1, 2) && throw(AssertionError("M: (inv(W) - θ*S) is not PSD"))       # the @assert at :366 / :440
    status[] in (0, 3) || error("iLEQG solve failed with status $(status[])")
    s.value_current = value[]; s.iter_current = iters[]
    s.ϵ_history = [(hist[1, i], hist[2, i]) for i in 1:hn[]]
    s.x_array, s.l_array, s.L_array = unflat_vec(x), unflat_vec(l), unflat_mat(L)
    return copy(s.x_array), copy(s.l_array), copy(s.L_array), value[], copy(s.ϵ_history)
end

"Batched solve! for many θ at once (what compute_cost fans out, cross_entropy...jl:144-167): value (Inf on failure), status, iterations, line-search evaluations"
function solve_batch(s::ILEQGSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64}, u_array, θ_array::Vector{Float64})
    h = bind!(s.h, problem); B = length(θ_array)
    value = Vector{Float64}(undef, B); st = Vector{Int32}(undef, B); it = Vector{Int32}(undef, B); ls = Vector{Int32}(undef, B)
    check(ccall((:rat_ileqg_solve_batch, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                h.ptr, x_0, flat(u_array), θ_array, B, value, st, it, ls))
    value, st, it, ls
end

# =====================================================================================================================================
# RAT iLQR: Cross-Entropy over θ (src/cross_entropy_bilevel_optimization.jl)
# =====================================================================================================================================
"CrossEntropyBilevelOptimizationSolver(; kwargs...) -- cross_entropy_bilevel_optimization.jl:70-127.  `devices = 0:7` shards every CE batch
over those GPUs (rat_create_multi); the default is one device."
mutable struct CrossEntropyBilevelOptimizationSolver
    opts::IleqgOpts
    c::CeState
    spec_eps::Int; devices::Vector{Int32}
    h::Union{Nothing,Handle,MultiHandle}
    z::Vector{Float64}                      # the N(0,1) stream currently registered with the handle (kept alive)
    carrier::Any                            # closure problems: the carrier ILEQGSolver whose device handle runs the batched sweeps ...
    carrier_key::Any                        # ... and the (n, m, N, batch, problem, hash of W(k), opts) it was built for: one handle per solver, not one per compute_cost
end
function CrossEntropyBilevelOptimizationSolver(; μ_min_ileqg=1e-6, Δ_0_ileqg=2.0, λ_ileqg=0.5, d_ileqg=1e-2, iter_max_ileqg=100,
        adaptive_ϵ_init_ileqg=false, ϵ_init_ileqg=1.0, ϵ_min_ileqg=1e-6, μ_init=1.0, σ_init=2.0, num_samples=10, num_elite=3,
        iter_max=5, λ=0.5, f_returns_jacobian=false, use_θ_max=false, spec_eps=1, device=0, devices=[device])
    o = IleqgOpts(μ_min_ileqg, Δ_0_ileqg, λ_ileqg, d_ileqg, iter_max_ileqg, ϵ_init_ileqg, ϵ_min_ileqg, adaptive_ϵ_init_ileqg)
    c = CeState(num_samples, num_elite, iter_max, λ, use_θ_max, μ_init, σ_init, μ_init, σ_init, 0.0, Inf, 0, 0, 0, 0)
    CrossEntropyBilevelOptimizationSolver(o, c, spec_eps, collect(Int32, devices), nothing, Float64[], nothing, nothing)
end

"The solver's device context for `problem`: created on first use, re-bound when the problem object changes (receding-horizon callers
rebuild the problem every control step), recreated when the batch no longer fits."
function handle!(s::CrossEntropyBilevelOptimizationSolver, problem, batch::Integer=s.c.num_samples)
    if s.h === nothing || s.h.problem === nothing
        s.h = length(s.devices) > 1 ? MultiHandle(s.opts, max(batch, s.c.num_samples), s.spec_eps, s.devices) :
                                      Handle(s.opts, max(batch, s.c.num_samples), s.spec_eps, s.devices[1])
    end
    bind!(s.h, problem)
end
single(h::Handle) = h.ptr
single(h::MultiHandle) = device_handle(h, 0)

"initialize!(ce_solver) -- :133-138"
initialize!(s::CrossEntropyBilevelOptimizationSolver) = ccall((:rat_ce_initialize, LIB), Cvoid, (Ref{CeState},), s.c)

"compute_cost(ce_solver, problem, x, u_array, θ_array, kl_bound) -- :173-195 (replaces the remotecall_fetch fan-out)"
function compute_cost(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x::Vector{Float64}, u_array,
                      θ_array::Vector{Float64}, kl_bound::Float64)
    h = handle!(s, problem, length(θ_array)); cost = similar(θ_array)
    if h isa MultiHandle
        check(ccall((:rat_multi_ce_compute_cost, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}),
                    h.ptr, x, flat(u_array), θ_array, length(θ_array), kl_bound, cost))
    else
        check(ccall((:rat_ce_compute_cost, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}),
                    h.ptr, x, flat(u_array), θ_array, length(θ_array), kl_bound, cost))
    end
    cost
end
"compute_cost with the per-sample solver statistics of every shard: (cost, status, iterations, line-search evaluations).  On several
devices the four arrays travel in the one all-gather of the batch (rat_multi_ce_compute_cost_ex)."
function compute_cost_detail(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x::Vector{Float64}, u_array,
                             θ_array::Vector{Float64}, kl_bound::Float64)
    B = length(θ_array)
    h = handle!(s, problem, B); cost = similar(θ_array)
    st = Vector{Int32}(undef, B); it = Vector{Int32}(undef, B); ls = Vector{Int32}(undef, B)
    if h isa MultiHandle
        check(ccall((:rat_multi_ce_compute_cost_ex, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    h.ptr, x, flat(u_array), θ_array, B, kl_bound, cost, st, it, ls))
    else
        check(ccall((:rat_ileqg_solve_batch, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    h.ptr, x, flat(u_array), θ_array, B, cost, st, it, ls))
        cost .+= kl_bound ./ θ_array                                                     # :193
    end
    cost, st, it, ls
end
"Batched solve! over the solver's devices: value (Inf on failure), status, iterations, line-search evaluations (compute_value_worker over a batch, :144-167)"
function solve_batch(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64}, u_array, θ_array::Vector{Float64})
    B = length(θ_array)
    h = handle!(s, problem, B)
    value = Vector{Float64}(undef, B); st = Vector{Int32}(undef, B); it = Vector{Int32}(undef, B); ls = Vector{Int32}(undef, B)
    if h isa MultiHandle
        check(ccall((:rat_multi_ileqg_solve_batch, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    h.ptr, x_0, flat(u_array), θ_array, B, value, st, it, ls))
    else
        check(ccall((:rat_ileqg_solve_batch, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                    h.ptr, x_0, flat(u_array), θ_array, B, value, st, it, ls))
    end
    value, st, it, ls
end
"compute_cost_serial -- :198-227: one solve per call, the reference's debugging twin of compute_cost"
function compute_cost_serial(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x, u_array, θ_array::Vector{Float64}, kl_bound::Float64)
    @assert length(θ_array) == s.c.num_samples                                       # :204
    [compute_cost(s, problem, x, u_array, [θ], kl_bound)[1] for θ in θ_array]
end
"compute_value_worker(ce_solver, problem, x, u_array, θ) -- :144-167: one fresh solve, Inf where the reference catches an exception"
compute_value_worker(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x, u_array, θ::Float64) =
    compute_cost(s, problem, x, u_array, [θ], 0.0)[1]

"get_positive_samples(μ, σ, num_samples, rng) -- :233-246 (host arithmetic; the same rejection rule the library applies to its stream)"
function get_positive_samples(μ::Float64, σ::Float64, num_samples::Integer, rng::AbstractRNG)
    out = Float64[]
    while length(out) < num_samples
        θ = μ + σ * randn(rng)
        θ > 0.0 && push!(out, θ)
    end
    out
end

"set_initial!(ce_solver, problem, x, u_array): uploads the initial state and nominal controls used by the device-pointer entry points"
function set_initial!(s::CrossEntropyBilevelOptimizationSolver, problem, x::Vector{Float64}, u_array)
    h = handle!(s, problem)
    if h isa MultiHandle
        check(ccall((:rat_multi_set_initial, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h.ptr, x, flat(u_array)))
    else
        check(ccall((:rat_set_initial, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h.ptr, x, flat(u_array)))
    end
end

"""compute_cost with θ and the costs resident in HBM (device pointers, e.g. `pointer(::ROCArray{Float64})` of AMDGPU.jl); the initial
state / controls are the ones of the last `compute_cost` / `set_initial!`.  `enqueue = true` returns after the launch: the batch is
ordered on the handle's HIP stream (`hip_stream(ce_solver)`).  Single-device solvers only."""
function compute_cost_dev!(s::CrossEntropyBilevelOptimizationSolver, problem, θ_dev::Ptr{Float64}, B::Integer, kl_bound::Float64,
                           cost_dev::Ptr{Float64}; enqueue::Bool=false)
    h = handle!(s, problem, B)
    if enqueue
        check(ccall((:rat_ce_compute_cost_enqueue, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}), single(h), θ_dev, B, kl_bound, cost_dev))
    else
        check(ccall((:rat_ce_compute_cost_dev, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}), single(h), θ_dev, B, kl_bound, cost_dev))
    end
    nothing
end
"compute_cost_dev!(...; enqueue = true) with the per-sample status / iteration / line-search counts written beside the costs (device
pointers to Int32 arrays, any may be C_NULL): a rank's contribution to the cost + status all-gather of a sharded CE batch"
function compute_cost_enqueue_ex!(s::CrossEntropyBilevelOptimizationSolver, problem, θ_dev::Ptr{Float64}, B::Integer, kl_bound::Float64,
                                  cost_dev::Ptr{Float64}, status_dev::Ptr{Int32}, iters_dev::Ptr{Int32}, ls_dev::Ptr{Int32})
    h = handle!(s, problem, B)
    check(ccall((:rat_ce_compute_cost_enqueue_ex, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                single(h), θ_dev, B, kl_bound, cost_dev, status_dev, iters_dev, ls_dev))
    nothing
end
hip_stream(s::CrossEntropyBilevelOptimizationSolver) = ccall((:rat_stream, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), single(s.h))

# The reference's step! / solve! consume `rng` exactly as far as get_positive_samples needs (:233-246): a hand-driven loop of step! with
# one MersenneTwister, or anything the caller draws from it afterwards, must see the same generator state.  So the draws are made HERE
# with the caller's rng (the reference's own sequence), the batch goes to the device (compute_cost), and the bookkeeping of :291-334 is
# the library's host arithmetic (rat_ce_begin_step / rat_ce_update) -- no pre-drawn stream, nothing consumed that the reference leaves.
function ce_round!(s::CrossEntropyBilevelOptimizationSolver, problem, x::Vector{Float64}, u_array, kl_bound::Float64, rng::AbstractRNG)
    handle!(s, problem); B = s.c.num_samples
    check(ccall((:rat_ce_begin_step, LIB), Int32, (Ref{CeState},), s.c))                 # :259
    redraws = 0
    while true
        redraws > 1000 && error("CE redraw loop cut after 1000 redraws (the reference would spin, :266-305)")
        μ, σ = s.c.iter_current == 1 ? (s.c.mu_init, s.c.sigma_init) : (s.c.mu, s.c.sigma)   # :266-279
        θ = get_positive_samples(μ, σ, B, rng)
        cost = compute_cost(s, problem, x, u_array, θ, kl_bound)
        s.c.n_solves += B
        redraws > 0 && (s.c.n_redraws += 1)
        redraw = Ref{Int32}(0)
        check(ccall((:rat_ce_update, LIB), Int32, (Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}), s.c, θ, cost, redraw))   # :291-334
        redraw[] == 0 && return θ, cost
        redraws += 1
    end
end

"step!(ce_solver, problem, x, u_array, kl_bound, rng) -- :252-335.  Returns (θ_array, cost_array) of the accepted batch."
function step!(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x::Vector{Float64}, u_array, kl_bound::Float64,
               rng::AbstractRNG, verbose=false, serial=false)
    ce_round!(s, problem, x, u_array, kl_bound, rng)
end

"solve!(ce_solver, problem, x_0, u_array, rng; kl_bound) -- :364-415."
function solve!(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64},
                u_array::Vector{Vector{Float64}}, rng::AbstractRNG; kl_bound::Float64, verbose=false, serial=false)
    @assert kl_bound >= 0 "KL Divergence Bound must be non-negative"
    h = handle!(s, problem); n, m, N = dims(problem)
    initialize!(s)                                                                        # :370
    s.c.n_final_retries = 0
    θ_opt, θ_min, θ_max = 0.0, 0.0, 0.0
    if kl_bound > 0.0
        while s.c.iter_current < s.c.iter_max                                             # :371-373
            ce_round!(s, problem, x_0, u_array, kl_bound, rng)
        end
        θ_min, θ_max = s.c.theta_min, s.c.theta_max
        θ_opt = s.c.use_theta_max != 0 ? θ_max : s.c.mu                                   # :375-382
    end
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    val = Ref(0.0); st = Ref{Int32}(0)
    tries = 0
    while true                                                                            # :390-414: final solve, retried at max(0, θ_opt - σ)
        tries > 10000 && error("final-solve retry loop cut (the reference would spin, :410-413)")
        check(ccall((:rat_ileqg_solve, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int32},
                     Ptr{Int32}, Ptr{Float64}, Int64, Ptr{Int64}),
                    single(h), x_0, flat(u_array), θ_opt, x, l, L, val, st, C_NULL, C_NULL, 0, C_NULL))
        st[] in (0, 3) && break
        θ_opt = max(0.0, θ_opt - s.c.sigma)                                               # :412
        s.c.n_final_retries += 1
        tries += 1
    end
    value = kl_bound > 0.0 ? val[] + kl_bound / θ_opt : val[]                             # :406 / :408
    return θ_opt, unflat_vec(x), unflat_vec(l), unflat_mat(L), value, θ_min, θ_max
end

# One-ccall variants on an injected N(0,1) stream `z` (θ = μ + σ z, consumed in order, θ <= 0 rejected: get_positive_samples :233-246): the
# whole CE step / solve runs behind the C ABI -- draws, batches on the device(s), updates, final solve -- which is the lowest-latency way to
# run RAT iLQR in a receding-horizon loop (2.8 ms per solve at 1024 samples x 5 iterations on one MI355X).  They do not touch a Julia rng;
# `stream_pos(ce_solver)` tells how many draws were consumed.
function set_stream!(s::CrossEntropyBilevelOptimizationSolver, h, z::Vector{Float64})
    s.z = z                                              # kept alive: the handle reads it in place
    check(ccall((:rat_ce_set_stream, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), single(h), s.z, length(s.z)))
end
stream_pos(s::CrossEntropyBilevelOptimizationSolver) = ccall((:rat_ce_stream_pos, LIB), Int64, (Ptr{Cvoid},), single(s.h))
function step_stream!(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x::Vector{Float64}, u_array, kl_bound::Float64,
                      z::Vector{Float64})
    h = handle!(s, problem); B = s.c.num_samples
    z === s.z || set_stream!(s, h, z)                    # the same stream object continues where the last call stopped
    θ = Vector{Float64}(undef, B); cost = Vector{Float64}(undef, B)
    GC.@preserve s begin
        if h isa MultiHandle
            check(ccall((:rat_multi_ce_step, LIB), Int32, (Ptr{Cvoid}, Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
                        h.ptr, s.c, x, flat(u_array), kl_bound, θ, cost))
        else
            check(ccall((:rat_ce_step, LIB), Int32, (Ptr{Cvoid}, Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
                        h.ptr, s.c, x, flat(u_array), kl_bound, θ, cost))
        end
    end
    θ, cost
end
function solve_stream!(s::CrossEntropyBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64},
                       u_array::Vector{Vector{Float64}}, z::Vector{Float64}; kl_bound::Float64)
    @assert kl_bound >= 0 "KL Divergence Bound must be non-negative"
    h = handle!(s, problem); n, m, N = dims(problem)
    z === s.z || set_stream!(s, h, z)
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    θ = Ref(0.0); val = Ref(0.0); θmin = Ref(0.0); θmax = Ref(0.0)
    GC.@preserve s begin
        if h isa MultiHandle
            check(ccall((:rat_multi_ce_solve, LIB), Int32,
                        (Ptr{Cvoid}, Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}, Ptr{Float64}, Ptr{Float64},
                         Ptr{Float64}, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                        h.ptr, s.c, x_0, flat(u_array), kl_bound, θ, x, l, L, val, θmin, θmax))
        else
            check(ccall((:rat_ce_solve, LIB), Int32,
                        (Ptr{Cvoid}, Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}, Ptr{Float64}, Ptr{Float64},
                         Ptr{Float64}, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                        h.ptr, s.c, x_0, flat(u_array), kl_bound, θ, x, l, L, val, θmin, θmax))
        end
    end
    return θ[], unflat_vec(x), unflat_vec(l), unflat_mat(L), val[], θmin[], θmax[]
end

# =====================================================================================================================================
# RAT iLQR++: Nelder-Mead over θ (src/nelder_mead_bilevel_optimization.jl)
# =====================================================================================================================================
"NelderMeadBilevelOptimizationSolver(; kwargs...) -- nelder_mead_bilevel_optimization.jl:72-128"
mutable struct NelderMeadBilevelOptimizationSolver
    opts::IleqgOpts
    c::NmState
    device::Int
    h::Union{Nothing,Handle}
end
function NelderMeadBilevelOptimizationSolver(; μ_min_ileqg=1e-6, Δ_0_ileqg=2.0, λ_ileqg=0.5, d_ileqg=1e-2, iter_max_ileqg=100,
        adaptive_ϵ_init_ileqg=false, ϵ_init_ileqg=1.0, ϵ_min_ileqg=1e-6, α=1.0, β=2.0, γ=0.5, ϵ=1e-2, λ=0.5,
        iter_max=100, θ_high_init=3.0, θ_low_init=1e-8, f_returns_jacobian=false, device=0)
    o = IleqgOpts(μ_min_ileqg, Δ_0_ileqg, λ_ileqg, d_ileqg, iter_max_ileqg, ϵ_init_ileqg, ϵ_min_ileqg, adaptive_ϵ_init_ileqg)
    c = NmState(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    ccall((:rat_nm_default, LIB), Cvoid, (Ref{NmState},), c)            # the constructor defaults of :102-128
    c.alpha, c.beta, c.gamma, c.eps, c.lambda, c.iter_max = α, β, γ, ϵ, λ, iter_max
    c.theta_high_init = c.theta_high = θ_high_init
    c.theta_low_init = c.theta_low = θ_low_init
    NelderMeadBilevelOptimizationSolver(o, c, device, nothing)
end
function handle!(s::NelderMeadBilevelOptimizationSolver, problem)
    s.h === nothing && (s.h = Handle(s.opts, 1024, 1, s.device))         # both initial vertices + three iterations' vertices in the first device call
    bind!(s.h, problem)
end
"initialize!(nm_solver) -- :164-168 (c_high / c_low are left alone, as in the reference)"
initialize!(s::NelderMeadBilevelOptimizationSolver) = ccall((:rat_nm_initialize, LIB), Cvoid, (Ref{NmState},), s.c)
"compute_cost_worker(nm_solver, problem, x, u_array, θ, kl_bound) -- :134-158"
function compute_cost_worker(s::NelderMeadBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x::Vector{Float64}, u_array, θ::Float64, kl_bound::Float64)
    h = handle!(s, problem); c = Ref(0.0)
    check(ccall((:rat_nm_compute_cost, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ref{Float64}), h.ptr, x, flat(u_array), θ, kl_bound, c))
    c[]
end
"step!(nm_solver, problem, x, u_array, kl_bound) -- :174-252"
function step!(s::NelderMeadBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x::Vector{Float64}, u_array, kl_bound::Float64, verbose=false)
    h = handle!(s, problem)
    GC.@preserve s check(ccall((:rat_nm_step, LIB), Int32, (Ptr{Cvoid}, Ref{NmState}, Ptr{Float64}, Ptr{Float64}, Float64), h.ptr, s.c, x, flat(u_array), kl_bound))
end
"solve!(nm_solver, problem, x_0, u_array; kl_bound) -- nelder_mead_bilevel_optimization.jl:276-352"
function solve!(s::NelderMeadBilevelOptimizationSolver, problem::DeviceRiskSensitiveProblem, x_0::Vector{Float64},
                u_array::Vector{Vector{Float64}}; kl_bound::Float64, verbose=false)
    @assert kl_bound >= 0 "KL Divergence Bound must be non-negative"
    h = handle!(s, problem); n, m, N = dims(problem)
    x = Matrix{Float64}(undef, n, N + 1); l = Matrix{Float64}(undef, m, N); L = Array{Float64}(undef, m, n, N)
    θ = Ref(0.0); val = Ref(0.0); status = Ref(Int32(0))
    GC.@preserve s check(ccall((:rat_nm_solve, LIB), Int32,
                (Ptr{Cvoid}, Ref{NmState}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}, Ref{Float64}, Ref{Int32}),
                h.ptr, s.c, x_0, flat(u_array), kl_bound, θ, x, l, L, val, status))
    status[] in (1, 2) && throw(AssertionError("M: (inv(W) - θ*S) is not PSD"))
    status[] in (0, 3) || error("final iLEQG solve failed with status $(status[])")
    return θ[], unflat_vec(x), unflat_vec(l), unflat_mat(L), val[]
end

# =====================================================================================================================================
# PETS: cross-entropy over control sequences (src/pets.jl)
# =====================================================================================================================================
"CrossEntropyDirectOptimizationSolver(μ_init_array, Σ_init_array; kwargs...) -- pets.jl:35-68"
mutable struct CrossEntropyDirectOptimizationSolver
    num_control_samples::Int64; num_trajectory_samples::Int64; num_elite::Int64; iter_max::Int64; smoothing_factor::Float64
    μ_init_array::Vector{Vector{Float64}}; Σ_init_array::Vector{Matrix{Float64}}
    μ_array::Vector{Vector{Float64}}; Σ_array::Vector{Matrix{Float64}}
    N::Int64; iter_current::Int64
    device::Int
    h::Union{Nothing,Handle}
    devices::Vector{Int32}                 # more than one: compute_cost shards the control samples over these GPUs (rat_multi_pets_*)
    mh::Union{Nothing,MultiHandle}
end
function CrossEntropyDirectOptimizationSolver(μ_init_array::Vector{Vector{Float64}}, Σ_init_array::Vector{Matrix{Float64}};
        num_control_samples=10, num_trajectory_samples=10, num_elite=3, iter_max=5, smoothing_factor=0.1, device=0, devices=[device])
    @assert length(μ_init_array) == length(Σ_init_array)
    CrossEntropyDirectOptimizationSolver(num_control_samples, num_trajectory_samples, num_elite, iter_max, smoothing_factor,
                                         μ_init_array, Σ_init_array, copy(μ_init_array), copy(Σ_init_array), length(μ_init_array), 0, Int(devices[1]), nothing,
                                         collect(Int32, devices), nothing)
end
"initialize!(direct_solver) -- pets.jl:70-74"
function initialize!(s::CrossEntropyDirectOptimizationSolver)
    s.iter_current = 0
    s.μ_array = copy(s.μ_init_array); s.Σ_array = copy(s.Σ_init_array)
end
function with_gen_desc(f::Function, problem::LQGenerativeProblem)
    p = problem.lq; n, m, N = dims(p)
    GC.@preserve problem begin
        lq = ProblemDesc(1, n, m, N, ndims(p.Q) == 3, 0, pointer(p.A), pointer(p.B), pointer(p.Q), pointer(p.R), pointer(p.P), pointer(p.qv),
                         pointer(p.rv), pointer(p.q0), pointer(p.Qf), pointer(p.qvf), p.q0f, p.kappa, 0, 0, 0, 0, 0, 0, 0, Ptr{Float64}(C_NULL))
        f(GenProblemDesc(lq, problem.l1u, problem.noise_kind, pointer(problem.nmean), pointer(problem.nchol), problem.nlo, problem.nhi,
                         problem.tw2, pointer(problem.tmean2), pointer(problem.tchol2)))
    end
end
function handle!(s::CrossEntropyDirectOptimizationSolver, problem::LQGenerativeProblem)
    if s.h === nothing || s.h.problem !== problem
        s.h === nothing && (s.h = Handle(IleqgOpts(1e-6, 2.0, 0.5, 1e-2, 100, 1.0, 1e-6, 0), 1, 1, s.device))
        with_gen_desc(problem) do d
            check(ccall((:rat_pets_problem_set, LIB), Int32, (Ptr{Cvoid}, Ref{GenProblemDesc}), s.h.ptr, d))
        end
        s.h.problem = problem
    end
    s.h
end
function multi_handle!(s::CrossEntropyDirectOptimizationSolver, problem::LQGenerativeProblem)
    if s.mh === nothing || s.mh.problem !== problem
        s.mh === nothing && (s.mh = MultiHandle(IleqgOpts(1e-6, 2.0, 0.5, 1e-2, 100, 1.0, 1e-6, 0), 1, 1, s.devices))
        with_gen_desc(problem) do d
            check(ccall((:rat_multi_pets_problem_set, LIB), Int32, (Ptr{Cvoid}, Ref{GenProblemDesc}), s.mh.ptr, d))
        end
        s.mh.problem = problem
    end
    s.mh
end

"""compute_cost(direct_solver, problem, x, control_sequence_array, rng, use_true_model) -- pets.jl:100-126 (all S × K stochastic rollouts in
one launch; noise from the device generator keyed by a seed drawn from `rng`)"""
function compute_cost(s::CrossEntropyDirectOptimizationSolver, problem::LQGenerativeProblem, x::Vector{Float64},
                      control_sequence_array::Vector{Vector{Vector{Float64}}}, rng::AbstractRNG, use_true_model=false)
    S = length(control_sequence_array)
    ctrl = cat([flat(c) for c in control_sequence_array]...; dims=3)           # m × N × S
    cost = Vector{Float64}(undef, S)
    if length(s.devices) > 1                                                   # control samples sharded over the GPUs (pets.jl:108-124)
        mh = multi_handle!(s, problem)
        check(ccall((:rat_multi_pets_compute_cost, LIB), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int32, Ptr{Float64}, Ptr{Float64}, UInt64, Ptr{Float64}),
                    mh.ptr, x, ctrl, S, s.num_trajectory_samples, use_true_model, C_NULL, C_NULL, rand(rng, UInt64), cost))
        return cost
    end
    h = handle!(s, problem)
    check(ccall((:rat_pets_compute_cost, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int32, Ptr{Float64}, Ptr{Float64}, UInt64, Ptr{Float64}),
                h.ptr, x, ctrl, S, s.num_trajectory_samples, use_true_model, C_NULL, C_NULL, rand(rng, UInt64), cost))
    cost
end

function with_pets_state(f::Function, s::CrossEntropyDirectOptimizationSolver)
    m = length(s.μ_array[1])
    mi, si, mu, sg = flat(s.μ_init_array), flat(s.Σ_init_array), flat(s.μ_array), flat(s.Σ_array)
    GC.@preserve mi si mu sg begin
        st = PetsState(s.num_control_samples, s.num_trajectory_samples, s.num_elite, s.iter_max, s.smoothing_factor, s.N, m, s.iter_current,
                       pointer(mi), pointer(si), pointer(mu), pointer(sg))
        f(st)
        s.iter_current = st.iter_current
    end
    s.μ_array, s.Σ_array = unflat_vec(mu), unflat_mat(sg)
end

"step!(direct_solver, problem, x, rng, use_true_model) -- pets.jl:193-245"
function step!(s::CrossEntropyDirectOptimizationSolver, problem::LQGenerativeProblem, x::Vector{Float64}, rng::AbstractRNG,
               use_true_model=false, verbose=false, serial=false)
    h = handle!(s, problem); m = length(s.μ_array[1])
    zc = randn(rng, m, s.N, s.num_control_samples)
    with_pets_state(s) do st
        check(ccall((:rat_pets_step, LIB), Int32,
                    (Ptr{Cvoid}, Ref{PetsState}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt64, Ptr{Float64}, Ptr{Float64}),
                    h.ptr, st, x, use_true_model, zc, C_NULL, C_NULL, rand(rng, UInt64), C_NULL, C_NULL))
    end
end

"solve!(direct_solver, problem, x_0, rng; use_true_model) -- pets.jl:270-281.  Returns (μ_array, Σ_array)."
function solve!(s::CrossEntropyDirectOptimizationSolver, problem::LQGenerativeProblem, x_0::Vector{Float64}, rng::AbstractRNG;
                use_true_model=false, verbose=false, serial=false)
    h = handle!(s, problem); m = length(s.μ_array[1])
    zc = randn(rng, m, s.N, s.num_control_samples, s.iter_max)
    with_pets_state(s) do st
        check(ccall((:rat_pets_solve, LIB), Int32,
                    (Ptr{Cvoid}, Ref{PetsState}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt64),
                    h.ptr, st, x_0, use_true_model, zc, C_NULL, C_NULL, rand(rng, UInt64)))
    end
    copy(s.μ_array), copy(s.Σ_array)
end

# =====================================================================================================================================
# Generic closures: host closures (the reference package's own functions) + batched device sweeps
# =====================================================================================================================================
# A problem that is not one of the device model families (the reference's FiniteHorizonRiskSensitiveOptimalControlProblem(f, c, h, W, N) with
# arbitrary closures) needs the reference package loaded: its closures are evaluated by the reference's own simulate_dynamics /
# approximate_model.  A single iLEQG solve! and Nelder-Mead are handed to the reference whole; the Cross-Entropy solver -- the one with
# B independent solves per batch -- runs their Riccati sweeps on the device, one launch per round (below).
reference_module() = isdefined(Main, :RATiLQR) ? getfield(Main, :RATiLQR) :
    error("this problem is not a device model family (LQRiskSensitiveProblem, PowerLawRiskSensitiveProblem, LQGenerativeProblem) and the " *
          "reference package RATiLQR is not loaded: `using RATiLQR` to run generic closures on the CPU")

function solve!(s::ILEQGSolver, problem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}}; θ::Float64, verbose=false)
    R = reference_module()
    ref = R.ILEQGSolver(problem; μ_min=s.μ_min, Δ_0=s.Δ_0, λ=s.λ, d=s.d, iter_max=s.iter_max, ϵ_init=s.ϵ_init_init,
                        adaptive_ϵ_init=s.ϵ_init_auto, ϵ_min=s.ϵ_min)
    R.solve!(ref, problem, x_0, u_array; θ=θ, verbose=verbose)
end
# ---- closure problems through the batched device sweeps (SURVEY 8f #3; the loop of ratilqr.jl_amd/generic.py: solve_closure_batch) -----
# A closure problem cannot cross the C ABI, but its Riccati sweeps can: the host evaluates the user's closures with the REFERENCE'S OWN
# functions -- `simulate_dynamics` (rollouts, with the Jacobians `f` returns when f_returns_jacobian) and `approximate_model` (ForwardDiff,
# ileqg.jl:265-273, or the user's A / B, :302-311) -- and every solve_approximate_dp! / solve_approximate_dp of the B samples of a
# compute_cost call runs in ONE device launch (rat_dp_gain_sweep_batch / rat_dp_policy_eval_batch).  The B per-sample solve! sequences
# (initialize!, then step! = approximate_model -> solve_approximate_dp! -> line_search!, :598-613, until :642-653) advance in lockstep
# rounds only so that their sweeps share launches; every decision is the reference's (accept rule :538, ε_min :558, adaptive ε_init
# :582-591), so results equal B separate reference solves.  `closure_device[] = false` forwards to the reference package instead.
const closure_device = Ref(true)

"A device handle that only carries W(k), n, m, N of a closure problem (the sweeps read their tiles from the host-built ApproximationResult)."
function carrier_problem(problem, n::Integer, m::Integer)
    N = problem.N
    Wt = cat([Matrix{Float64}(problem.W(k)) for k in 0:N-1]...; dims=3)                  # W(k), k 0-based (optimal_control_problems.jl:67-73)
    LQRiskSensitiveProblem(zeros(n, n), zeros(n, m), Matrix{Float64}(I(n)), Matrix{Float64}(I(m)), zeros(m, n), zeros(n), zeros(m), [0.0],
                           Matrix{Float64}(I(n)), zeros(n), 0.0, 0.0, Wt, N)
end
ref_ap(a) = ApproximationResult(a.q_array, a.q_vec_array, [Matrix{Float64}(Q) for Q in a.Q_array], a.r_array, [Matrix{Float64}(Rm) for Rm in a.R_array],
                                a.P_array, a.A_array, a.B_array, a.W_array)

"""
    solve_closure_batch(opts, problem, x_0, u_array, θ_array; f_returns_jacobian=false, device=0, max_batch=length(θ_array))

B complete `solve!`s (ileqg.jl:635-659) of a closure problem, one per θ: `(value, status, iterations, line-search evaluations)`, value = Inf
where the reference would throw (compute_value_worker :163-165).  Host: the reference's `simulate_dynamics` / `approximate_model`; device:
all sweeps of a round in one launch.
"""
function solve_closure_batch(o::IleqgOpts, problem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}}, θ::Vector{Float64};
                             f_returns_jacobian::Bool=false, device::Integer=0, carrier::Union{Nothing,ILEQGSolver}=nothing)
    R = reference_module()
    B = length(θ); N = problem.N; n = length(x_0); m = length(u_array[1])
    cs = carrier === nothing ? ILEQGSolver(carrier_problem(problem, n, m); μ_min=o.mu_min, Δ_0=o.delta_0, λ=o.lambda, d=o.d, iter_max=o.iter_max,
                                           ϵ_init=o.eps_init, ϵ_min=o.eps_min, adaptive_ϵ_init=o.adaptive_eps_init != 0, max_batch=B, device=device) : carrier
    rollout(xb, l, L) = f_returns_jacobian ? R.simulate_dynamics(problem, xb, l, L, f_returns_jacobian=true) :
                                             (R.simulate_dynamics(problem, xb, l, L, f_returns_jacobian=false)..., nothing, nothing)
    lin(u, x, A, Bm) = ref_ap(A === nothing ? R.approximate_model(problem, u, x) : R.approximate_model(problem, u, x, A, Bm))
    status = fill(Int32(-1), B); value = fill(Inf, B); iters = zeros(Int32, B); ls_evals = zeros(Int32, B)
    μ = zeros(B); Δ = fill(o.delta_0, B); d_cur = fill(Inf, B); ϵ_init = fill(o.eps_init, B)
    # initialize! (:214-236): one rollout / linearisation serves every sample -- θ only enters the sweep
    x0s, A0, B0 = f_returns_jacobian ? R.simulate_dynamics(problem, x_0, u_array, f_returns_jacobian=true) :
                                       (R.simulate_dynamics(problem, x_0, u_array, f_returns_jacobian=false), nothing, nothing)
    ap0 = lin(u_array, x0s, A0, B0)
    x = [deepcopy(x0s) for _ in 1:B]; l = [deepcopy(u_array) for _ in 1:B]; L = [[zeros(m, n) for _ in 1:N] for _ in 1:B]
    AB = Any[(A0, B0) for _ in 1:B]
    v, st = solve_approximate_dp_batch(cs, fill(ap0, B), L, θ, μ)
    for b in 1:B
        st[b] != 0 ? (status[b] = Int32(1)) : (value[b] = v[b])                          # the uncaught @assert of initialize! (:234) -> Inf
    end
    dl = Vector{Any}(nothing, B); ϵ = copy(ϵ_init); count = zeros(Int, B); in_ls = falses(B)
    while true
        live = findall(==(Int32(-1)), status)
        isempty(live) && break
        need = [b for b in live if !in_ls[b]]
        if !isempty(need)                                                                # step! (:598-613): approximate_model + solve_approximate_dp!
            iters[need] .+= 1
            aps = [lin(l[b], x[b], AB[b]...) for b in need]
            Lg, dlg, μg, Δg, stg = solve_approximate_dp_batch!(cs, aps, θ[need], μ[need], Δ[need])
            for (i, b) in enumerate(need)
                μ[b], Δ[b] = μg[i], Δg[i]
                if stg[i] != 0
                    status[b] = stg[i]; value[b] = Inf; continue
                end
                L[b], dl[b] = Lg[i], dlg[i]
                ϵ[b], count[b], in_ls[b] = ϵ_init[b], 0, true
            end
        end
        cand = [b for b in findall(==(Int32(-1)), status) if in_ls[b]]
        isempty(cand) && continue
        trial = Dict{Int,Any}()
        for b in cand                                                                    # a line_search! candidate (:504-521)
            count[b] += 1; ls_evals[b] += 1
            xn, un, An, Bn = rollout(x[b], l[b] .+ ϵ[b] .* dl[b], L[b])
            trial[b] = (xn, un, lin(un, xn, An, Bn), An, Bn)
        end
        vn, stn = solve_approximate_dp_batch(cs, [trial[b][3] for b in cand], [L[b] for b in cand], θ[cand], μ[cand])
        for (i, b) in enumerate(cand)
            if count[b] > 4000
                status[b] = Int32(7); value[b] = Inf; in_ls[b] = false; continue
            end
            if stn[i] != 0                                                               # :522-535: no ε_min test, no history entry
                ϵ[b] *= o.lambda; continue
            end
            new, cur = vn[i], value[b]
            if !(new ≈ cur || new < cur)                                                 # :538
                ϵ[b] *= o.lambda                                                         # :557
                ϵ[b] < o.eps_min || continue                                             # :558: forced accept of the rejected candidate
            end
            xn, un, _, An, Bn = trial[b]
            d_cur[b] = maximum(norm.(l[b] .- un))                                        # :539 / :559
            value[b], x[b], l[b], in_ls[b] = new, xn, un, false
            AB[b] = (An, Bn)
            if o.adaptive_eps_init != 0                                                  # :582-591
                if count[b] == 1
                    ϵ_init[b] = min(o.eps_init, ϵ[b] / o.lambda)
                else
                    e = ϵ[b]
                    while e < o.eps_min; e /= o.lambda; end
                    ϵ_init[b] = e
                end
            end
            if o.d > d_cur[b] && μ[b] <= o.mu_min                                        # :642
                status[b] = Int32(0)
            elseif iters[b] == o.iter_max                                                # :648
                status[b] = Int32(3)
            end
        end
    end
    [(status[b] == 0 || status[b] == 3) ? value[b] : Inf for b in 1:B], status, iters, ls_evals
end

"The carrier solver of a closure problem (its device handle: stream, events, pinned buffers, B-sample state), kept on the CE solver and
rebuilt only when the sizes, the batch or the problem object change -- not once per compute_cost call (ADVICE r04)."
function closure_carrier!(s::CrossEntropyBilevelOptimizationSolver, problem, n::Integer, m::Integer, B::Integer)
    # the carrier bakes in W(k) as evaluated now and s.opts (its device gain sweeps restart with mu_min / delta_0): both are part of the key,
    # so a reassigned s.opts or a W closure that reads mutable state rebuilds it instead of running stale sweeps (ADVICE r05)
    o = s.opts
    wh = hash([Matrix{Float64}(problem.W(k)) for k in 0:problem.N-1])
    key = (n, m, problem.N, B, objectid(problem), wh,
           (o.mu_min, o.delta_0, o.lambda, o.d, o.iter_max, o.eps_init, o.eps_min, o.adaptive_eps_init))
    if s.carrier === nothing || s.carrier_key != key
        o = s.opts
        s.carrier = ILEQGSolver(carrier_problem(problem, n, m); μ_min=o.mu_min, Δ_0=o.delta_0, λ=o.lambda, d=o.d, iter_max=o.iter_max, ϵ_init=o.eps_init,
                                ϵ_min=o.eps_min, adaptive_ϵ_init=o.adaptive_eps_init != 0, max_batch=B, device=s.devices[1])
        s.carrier_key = key
    end
    s.carrier
end

"compute_cost (:173-195) of a closure problem: the B solves of the batch share their device sweeps"
function compute_cost(s::CrossEntropyBilevelOptimizationSolver, problem, x::Vector{Float64}, u_array::Vector{Vector{Float64}},
                      θ_array::Vector{Float64}, kl_bound::Float64; f_returns_jacobian::Bool=false)
    cs = closure_carrier!(s, problem, length(x), length(u_array[1]), length(θ_array))
    value, _, _, _ = solve_closure_batch(s.opts, problem, x, u_array, θ_array; f_returns_jacobian=f_returns_jacobian, device=s.devices[1], carrier=cs)
    value .+ kl_bound ./ θ_array                                                         # :193
end

"solve! (:364-415) of a closure problem: the CE loop of the reference with compute_cost on the batched device sweeps"
function solve!(s::CrossEntropyBilevelOptimizationSolver, problem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}}, rng::AbstractRNG;
                kl_bound::Float64, verbose=false, serial=false, f_returns_jacobian::Bool=false)
    R = reference_module(); o = s.opts; c = s.c
    if !closure_device[] || serial
        ref = R.CrossEntropyBilevelOptimizationSolver(μ_min_ileqg=o.mu_min, Δ_0_ileqg=o.delta_0, λ_ileqg=o.lambda, d_ileqg=o.d, iter_max_ileqg=o.iter_max,
                  adaptive_ϵ_init_ileqg=o.adaptive_eps_init != 0, ϵ_init_ileqg=o.eps_init, ϵ_min_ileqg=o.eps_min, μ_init=c.mu_init, σ_init=c.sigma_init,
                  num_samples=c.num_samples, num_elite=c.num_elite, iter_max=c.iter_max, λ=c.lambda, use_θ_max=c.use_theta_max != 0)
        out = R.solve!(ref, problem, x_0, u_array, rng; kl_bound=kl_bound, verbose=verbose, serial=serial)
        c.mu_init, c.sigma_init = ref.μ_init, ref.σ_init                 # the reference mutates and keeps these across solve! calls (App. B.10)
        return out
    end
    @assert kl_bound >= 0 "KL Divergence Bound must be non-negative"
    initialize!(s)                                                                       # :369
    θ_opt = 0.0
    if kl_bound > 0
        while c.iter_current < c.iter_max                                                # :371-373, step! :252-335 with the device-swept compute_cost
            ccall((:rat_ce_begin_step, LIB), Int32, (Ref{CeState},), c)
            while true
                θs = c.iter_current == 1 ? get_positive_samples(c.mu_init, c.sigma_init, Int(c.num_samples), rng) :
                                           get_positive_samples(c.mu, c.sigma, Int(c.num_samples), rng)
                cost = compute_cost(s, problem, x_0, u_array, θs, kl_bound; f_returns_jacobian=f_returns_jacobian)
                redraw = Ref(Int32(0))
                check(ccall((:rat_ce_update, LIB), Int32, (Ref{CeState}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}), c, θs, cost, redraw))
                redraw[] == 0 && break
            end
        end
        θ_opt = c.use_theta_max != 0 ? c.theta_max : c.mu                                # :375-382
    end
    cs1 = closure_carrier!(s, problem, length(x_0), length(u_array[1]), max(1, Int(c.num_samples)))     # (a batch of one fits the batch's carrier)
    while true                                                                           # :390-414: final solve with the retry on failure
        value, st, _, _ = solve_closure_batch(o, problem, x_0, u_array, [θ_opt]; f_returns_jacobian=f_returns_jacobian, device=s.devices[1], carrier=cs1)
        if st[1] == 0 || st[1] == 3
            ref = R.ILEQGSolver(problem; μ_min=o.mu_min, Δ_0=o.delta_0, λ=o.lambda, d=o.d, iter_max=o.iter_max, ϵ_init=o.eps_init,
                                adaptive_ϵ_init=o.adaptive_eps_init != 0, ϵ_min=o.eps_min, f_returns_jacobian=f_returns_jacobian)
            # the trajectory and gains of the accepted θ: the reference's own solve!.  An exception there (a borderline isposdef on which device
            # and CPU differ) feeds the same retry as a failed device solve (:410-413) instead of escaping the loop (ADVICE r04)
            out = try
                R.solve!(ref, problem, x_0, u_array; θ=θ_opt, verbose=false)
            catch
                nothing
            end
            if out !== nothing
                xa, la, La, val, _ = out
                return kl_bound > 0 ? (θ_opt, xa, la, La, val + kl_bound / θ_opt, c.theta_min, c.theta_max) : (θ_opt, xa, la, La, val, 0.0, 0.0)
            end
        end
        θ_opt = max(0.0, θ_opt - c.sigma)                                                # :412
    end
end
function solve!(s::NelderMeadBilevelOptimizationSolver, problem, x_0::Vector{Float64}, u_array::Vector{Vector{Float64}}; kl_bound::Float64, verbose=false)
    R = reference_module(); o = s.opts; c = s.c
    ref = R.NelderMeadBilevelOptimizationSolver(μ_min_ileqg=o.mu_min, Δ_0_ileqg=o.delta_0, λ_ileqg=o.lambda, d_ileqg=o.d, iter_max_ileqg=o.iter_max,
              adaptive_ϵ_init_ileqg=o.adaptive_eps_init != 0, ϵ_init_ileqg=o.eps_init, ϵ_min_ileqg=o.eps_min, α=c.alpha, β=c.beta, γ=c.gamma, ϵ=c.eps,
              λ=c.lambda, iter_max=c.iter_max, θ_high_init=c.theta_high_init, θ_low_init=c.theta_low_init)
    R.solve!(ref, problem, x_0, u_array; kl_bound=kl_bound, verbose=verbose)
end

export OptimalControlProblem, LQRiskSensitiveProblem, PowerLawRiskSensitiveProblem, LQGenerativeProblem,
       simulate_dynamics, simulate_dynamics_noisy, integrate_cost, ILEQGSolver, initialize!, ApproximationResult, approximate_model,
       DynamicProgrammingResult, solve_approximate_dp!, solve_approximate_dp, increase_μ_and_Δ!, decrease_μ_and_Δ!, line_search!, step!, solve!,
       solve_batch, solve_approximate_dp_batch!, solve_approximate_dp_batch, solve_closure_batch, closure_device, CrossEntropyBilevelOptimizationSolver, compute_value_worker, compute_cost, compute_cost_serial, get_positive_samples,
       set_initial!, compute_cost_dev!, NelderMeadBilevelOptimizationSolver, compute_cost_worker, CrossEntropyDirectOptimizationSolver,
       shard_bounds, compute_cost_detail, step_stream!, solve_stream!, stream_pos, set_path!, get_path, is_logical, debug_set!, debug_get
end

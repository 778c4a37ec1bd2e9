# runtests.jl -- one command for a maintainer with Julia and an MI355X:
#
#     julia --project=. julia/runtests.jl            (RATILQR_LIB=/path/to/libratilqr_hip.so if the library is not at its in-tree place)
#
# Part 1 restates the reference's own unit tests on the device model families of RATiLQRAMD (the closures of those tests, written as
# tables): test/ileqg_test.jl:20-174 and test/cross_entropy_bilevel_optimization_test.jl:27-41 -- same assertions, same tolerances.
# Part 2 runs when the reference package is loadable (`using RATiLQR`): the SURVEY section 8(d) workload through the reference's
# `compute_cost_serial` on the host cores and through the device, side by side -- the "Julia CPU reference timed in the same run" of
# BASELINE.md, which bench.py cannot provide on a box without Julia (it reports the C port of the reference instead and says so).
# This file cannot be executed where it was written (no julia binary in the build image): tests/test_cpu_julia_shim.py checks
# mechanically that every RATiLQRAMD name it uses exists with the arity used here.
using LinearAlgebra
using Random
using Test

include(joinpath(@__DIR__, "RATiLQRAMD.jl"))
using .RATiLQRAMD

const I2 = Matrix(1.0I, 2, 2)

# c(k, x, u) = cq/2 x'x + cr/2 u'u + cp x'u + k * ck (k = 0-based time), h(x) = hq/2 x'x + h0, f(x, u) = x + u, W(k) = w I, as LQ tables
function lq_test_problem(N; cq=0.0, cr=0.0, cp=0.0, ck=0.0, hq=0.0, h0=0.0, w=1.0)
    Q = cat([cq * I2 for _ in 1:N]...; dims=3); R = cat([cr * I2 for _ in 1:N]...; dims=3); P = cat([cp * I2 for _ in 1:N]...; dims=3)
    LQRiskSensitiveProblem(copy(I2), copy(I2), Q, R, P, zeros(2, N), zeros(2, N), [ck * (k - 1) for k in 1:N], hq * I2, zeros(2), h0, 0.0,
                           w * I2, N)
end

@testset "iLEQG on the device (test/ileqg_test.jl:20-174)" begin
    N = 10
    prob = lq_test_problem(N; ck=1.0, h0=1.0)                                  # f = x + u, c = k, h = 1, W = I        (:12-16)
    solver = ILEQGSolver(prob)
    u_array = [ones(2) for _ in 1:N]
    x_array = simulate_dynamics(solver, prob, zeros(2), u_array)
    @test x_array[1] == zeros(2)                                               # :23
    @test all([x_array[ii + 1] == x_array[ii] + u_array[ii] for ii in 1:N-1])  # :24
    L_array = [ones(2, 2) for _ in 1:N]
    x_new, u_new = simulate_dynamics(solver, prob, x_array, u_array, L_array)
    @test all([u_new[ii] == u_array[ii] for ii in 1:N])                        # :28
    @test all([x_new[ii] == x_array[ii] for ii in 1:N])                        # :29
    @test integrate_cost(solver, prob, x_array, u_array) ≈ sum(0:N-1) + 1.0    # :32-33

    initialize!(solver, prob, zeros(2), u_array, 0.0)                          # :37-49
    @test solver.l_array == u_array
    @test solver.L_array == [zeros(2, 2) for _ in 1:N]
    @test solver.x_array == x_array
    @test solver.μ == 0.0
    @test solver.Δ == solver.Δ_0
    @test solver.d_current == Inf
    @test solver.iter_current == 0
    @test solver.ϵ_history == Tuple{Float64,Float64}[]
    dp_init = solve_approximate_dp(solver, approximate_model(solver, prob, u_array, x_array), [zeros(2, 2) for _ in 1:N]; θ=0.0, μ=0.0)
    @test solver.value_current ≈ dp_init.s_array[1]

    probq = lq_test_problem(N; cq=1.0, cr=2.0, cp=1.0, hq=1.0)                 # c = x'x/2 + u'u + x'u, h = x'x/2     (:52-53)
    sq = ILEQGSolver(probq)
    ap = approximate_model(sq, probq, u_array, x_array)
    @test all([isapprox(ap.q_array[ii], 0.5 * (2 * (ii - 1)^2) + 1.0 * 2 + 2 * (ii - 1)) for ii in 1:N])     # :56
    @test isapprox(ap.q_array[end], 0.5 * dot(x_array[end], x_array[end]))     # :57
    @test all([isapprox(ap.q_vec_array[ii], x_array[ii] + ones(2)) for ii in 1:N])                           # :59
    @test isapprox(ap.q_vec_array[end], x_array[end])
    @test all([isapprox(ap.Q_array[ii], I2) for ii in 1:N+1])
    @test all([isapprox(ap.r_array[ii], x_array[ii] + 2.0 * ones(2)) for ii in 1:N])
    @test all([isapprox(ap.R_array[ii], 2.0 * I2) for ii in 1:N])
    @test all([isapprox(ap.P_array[ii], I2) for ii in 1:N])
    @test all([ap.W_array[ii] == I2 for ii in 1:N])                            # :66

    probr = lq_test_problem(N; cq=1.0, cr=2.0, hq=1.0)                         # c = x'x/2 + u'u                      (:68-69)
    sr = ILEQGSolver(probr)
    initialize!(sr, probr, zeros(2), u_array, 0.0)
    ap = approximate_model(sr, probr, u_array, x_array)
    dp, dl = solve_approximate_dp!(sr, ap; θ=0.0)                              # :72
    @test length(dp.s_array) == N + 1 && length(dp.s_vec_array) == N + 1 && length(dp.S_array) == N + 1
    @test all([issymmetric(S) for S in dp.S_array]) && all([isposdef(S) for S in dp.S_array])                # :78-79
    @test length(dp.g_array) == N && length(dp.G_array) == N && length(dp.H_array) == N
    let S = ap.Q_array[N + 1], Llqr = Vector{Matrix{Float64}}(undef, N)                                     # gains == LQR (:86-105)
        for ii in N:-1:1
            A, B, Q, R = ap.A_array[ii], ap.B_array[ii], ap.Q_array[ii], ap.R_array[ii]
            Llqr[ii] = -(R + B' * S * B) \ (B' * S * A)
            S = Q + A' * S * A - A' * S * B / (R + B' * S * B) * B' * S * A
        end
        @test all([all(Llqr[ii] .≈ sr.L_array[ii]) for ii in 1:N])
    end
    @test all(isapprox.(norm.(u_array .+ dl .- sr.L_array .* x_array[1:end-1]), 0.0, atol=1e-8))            # :108
    dp2, dl2 = solve_approximate_dp!(sr, ap; θ=1e-8)                           # :110
    @test isapprox(dp.s_array[1], dp2.s_array[1], rtol=1e-5)                   # :124
    @test all([isapprox(dl[ii], dl2[ii]) for ii in 1:N])                       # :125
    solve_approximate_dp!(sr, ap; θ=0.0)
    dp3 = solve_approximate_dp(sr, ap, sr.L_array, dl; θ=0.0, μ=0.0)
    @test dp3.s_array ≈ dp.s_array rtol = 1e-12   # (`==` in the reference, :130: there both calls are the same Julia code; here two kernels)
    line_search!(sr, probr, dl, 0.0, false)
    @test sr.value_current ≈ dp.s_array[1]                                     # :134

    s3 = ILEQGSolver(probr); initialize!(s3, probr, zeros(2), u_array, 0.0); increase_μ_and_Δ!(s3)
    @test s3.Δ == 4.0 && s3.μ == 1e-6                                          # :137-141
    s4 = ILEQGSolver(probr); initialize!(s4, probr, zeros(2), u_array, 0.0); decrease_μ_and_Δ!(s4)
    @test s4.Δ == 0.5 && s4.μ == 0.0                                           # :144-148

    pl = PowerLawRiskSensitiveProblem(2, N, 0.01 * I2)                         # f = x.^1.3 + u.^1.5, c = Σ x.^2.5 + u.^2.5, h = 1   (:151-155)
    u01 = [0.1 * ones(2) for _ in 1:N]
    sp = ILEQGSolver(pl)
    θ = 0.5
    initialize!(sp, pl, zeros(2), u01, θ)
    apn = approximate_model(sp, pl, sp.l_array, sp.x_array)
    _, dln = solve_approximate_dp!(sp, apn; θ=θ)
    line_search!(sp, pl, dln, θ, false)
    @test length(sp.ϵ_history) == 1 && sp.ϵ_history[1][1] == 1.0 && sp.ϵ_history[1][2] < 0.0               # :168-170
    xs, _, _, _, _ = solve!(sp, pl, zeros(2), u01; θ=0.0)
    @test all([all(isapprox.(xs[ii], zeros(2), atol=1e-4)) for ii in 1:N+1])   # :172-174
end

@testset "Cross Entropy Bilevel Optimization on the device (test/cross_entropy_bilevel_optimization_test.jl:27-41)" begin
    N = 10
    pl = PowerLawRiskSensitiveProblem(2, N, 0.01 * I2)
    x_0 = zeros(2); u_array = [0.1 * ones(2) for _ in 1:N]
    solver = CrossEntropyBilevelOptimizationSolver(num_samples=3)
    initialize!(solver)
    θ_array = [0.1, 0.3, 0.43]; kl_bound = 1.0
    costs = compute_cost(solver, pl, x_0, u_array, θ_array, kl_bound)
    @test all(isapprox(costs, compute_cost_serial(solver, pl, x_0, u_array, θ_array, kl_bound)))            # :30-32
    cd, st, it, ls = compute_cost_detail(solver, pl, x_0, u_array, θ_array, kl_bound)
    @test cd == costs && all(st .== 0) && all(it .== 4) && all(ls .== 4)       # SURVEY App. C: 4 iterations, 4 line-search evaluations
    θs = get_positive_samples(0.0, 1.0, 10, MersenneTwister(123))
    @test all(θs .> 0.0) && length(θs) == 10                                   # :34-35
    rng = MersenneTwister(12344)
    θ_opt, _, _, _, c_opt = solve!(solver, pl, x_0, u_array, rng, kl_bound=kl_bound, verbose=false)
    @test !isinf(c_opt) && !isnan(θ_opt)                                       # :37-41
    # the rng is consumed exactly as far as get_positive_samples needs: a second generator replayed by hand ends in the same state
    rng1, rng2 = MersenneTwister(7), MersenneTwister(7)
    s2 = CrossEntropyBilevelOptimizationSolver(num_samples=3); initialize!(s2)
    θ1, _ = step!(s2, pl, x_0, u_array, kl_bound, rng1)
    if s2.c.n_redraws == 0
        @test θ1 == get_positive_samples(1.0, 2.0, 3, rng2) && rand(rng1) == rand(rng2)
    end
end

@testset "getting-started example of the reference's documentation (docs/source/getting-started.md:40-118)" begin
    # 2-D single integrator x' = x + dt u, c = x'Qx/2 + u'Ru/2, h = x'Qx/2, W = 0.1 dt I, N = 10: an LQ-family problem as it stands
    dt, N = 0.1, 10
    problem = LQRiskSensitiveProblem(copy(I2), dt * I2, copy(I2), 0.01 * I2, zeros(2, 2), zeros(2), zeros(2), [0.0], copy(I2), zeros(2), 0.0, 0.0,
                                     0.1 * dt * I2, N)
    solver = CrossEntropyBilevelOptimizationSolver()                               # the defaults of the example
    rng = MersenneTwister(12345)
    x_0 = [5.0, 5.0]; u_array = [zeros(2) for _ in 1:N]
    θ_opt, x_array, l_array, L_array, value, θ_min, θ_max = solve!(solver, problem, x_0, u_array, rng, kl_bound=0.1)
    @test isfinite(value) && θ_opt > 0 && θ_min <= θ_opt <= θ_max
    @test norm(x_array[end]) < 0.2 * norm(x_0)                                      # the policy steers the state to the origin
    @test all([norm(x_array[ii + 1]) < norm(x_array[ii]) for ii in 1:N])
    @test length(L_array) == N && size(L_array[1]) == (2, 2) && maximum(abs.(L_array[1])) > 0.1
    # with RATiLQR loaded, the reference's own solve! from the same generator state draws the same θ and must agree
    if isdefined(Main, :RATiLQR)
        f(x, u) = x + dt * u
        ref_problem = Main.RATiLQR.FiniteHorizonRiskSensitiveOptimalControlProblem(f, (k, x, u) -> 0.5 * dot(x, x) + 0.005 * dot(u, u), x -> 0.5 * dot(x, x),
                                                                                   k -> Matrix(0.1 * dt * I, 2, 2), N)
        ref_solver = Main.RATiLQR.CrossEntropyBilevelOptimizationSolver()
        θ_ref, x_ref, _, _, value_ref, _, _ = Main.RATiLQR.solve!(ref_solver, ref_problem, x_0, u_array, MersenneTwister(12345), kl_bound=0.1, verbose=false)
        @test isapprox(θ_opt, θ_ref, rtol=1e-9) && isapprox(value, value_ref, rtol=1e-9)
        @test maximum(norm.(x_array .- x_ref)) < 1e-9
    end
end

# ---- Part 2: the reference on the host cores beside the device, same problem, same θ (SURVEY section 8d) ---------------------------------
function survey_problem_tables(; n=12, m=4, N=50, w=1e-3, seed=0)
    rng = MersenneTwister(seed)                        # (any seeded draw serves the comparison: both sides get the SAME tables)
    A = 0.9 * Matrix(qr(randn(rng, n, n)).Q); B = randn(rng, n, m) / sqrt(n); x_0 = randn(rng, n)
    A, B, x_0, [zeros(m) for _ in 1:N], w
end

if get(ENV, "RATILQR_SKIP_REFERENCE", "0") != "1"
    have_ref = try
        @eval using RATiLQR
        true
    catch
        false
    end
    if have_ref
        @testset "reference (CPU) vs device on the SURVEY 8(d) workload" begin
            n, m, N = 12, 4, 50
            A, B, x_0, u_array, w = survey_problem_tables()
            ref_prob = RATiLQR.FiniteHorizonRiskSensitiveOptimalControlProblem((x, u) -> A * x + B * u, (k, x, u) -> 0.5 * dot(x, x) + 0.05 * dot(u, u),
                                                                               x -> 0.5 * dot(x, x), k -> Matrix(w * I, n, n), N)
            dev_prob = LQRiskSensitiveProblem(A, B, Matrix(1.0I, n, n), Matrix(0.1I, m, m), zeros(m, n), zeros(n), zeros(m), [0.0],
                                              Matrix(1.0I, n, n), zeros(n), 0.0, 0.0, Matrix(w * I, n, n), N)
            Bs = parse(Int, get(ENV, "RATILQR_REF_SAMPLES", "32"))
            θ_array = get_positive_samples(1.0, 2.0, Bs, MersenneTwister(1000))
            ref_solver = RATiLQR.CrossEntropyBilevelOptimizationSolver(num_samples=Bs)
            RATiLQR.compute_cost_serial(ref_solver, ref_prob, x_0, u_array, θ_array, 0.1)            # warm-up (compilation), untimed
            t_ref = @elapsed ref_cost = RATiLQR.compute_cost_serial(ref_solver, ref_prob, x_0, u_array, θ_array, 0.1)
            dev_solver = CrossEntropyBilevelOptimizationSolver(num_samples=Bs)
            compute_cost(dev_solver, dev_prob, x_0, u_array, θ_array, 0.1)
            t_dev = @elapsed dev_cost = compute_cost(dev_solver, dev_prob, x_0, u_array, θ_array, 0.1)
            fin = isfinite.(ref_cost)
            @test fin == isfinite.(dev_cost)
            @test maximum(abs.(ref_cost[fin] .- dev_cost[fin]) ./ abs.(ref_cost[fin])) < 1e-9        # SURVEY section 8c: value rel. <= 1e-9
            println("reference compute_cost_serial: $(round(Bs / t_ref, digits=1)) solves/s on 1 Julia thread; device: $(round(Bs / t_dev, digits=1)) solves/s ",
                    "at a batch of $Bs (the device's metric batch is 1024: python bench.py)")
        end
    else
        println("Julia reference not runnable: the package RATiLQR is not loadable in this environment (Part 2 skipped)")
    end
end

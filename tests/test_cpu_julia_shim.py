"""julia/RATiLQRAMD.jl cannot be executed here (no `julia` binary in the image), so its C-ABI surface is checked mechanically against
include/ratilqr.h -- the way tests/test_cpu_abi.py checks the Python mirror against the loaded library:

  * every `ccall((:rat_x, LIB), Ret, (ArgTypes...), ...)`: the symbol is declared in the header, arity matches, every argument and the
    return type have the declared width / signedness / pointer-ness / pointee (struct pointers by the `# mirrors` comments);
  * every `# mirrors \\`struct rat_x\\`` Julia struct: same field names, order and types as the C struct;
  * which header functions the shim binds (a short, explicit list of host-only helpers and measurement hooks is not bound)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "ratilqr.h")).read()
JL = open(os.path.join(ROOT, "julia", "RATiLQRAMD.jl")).read()

SCALARS_C = {"int32_t": "i32", "rat_rc": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "void"}
HANDLES = {"rat_handle", "rat_multi"}            # opaque pointers (typedef struct x_s *x)
SCALARS_JL = {"Int32": "i32", "Int64": "i64", "UInt64": "u64", "Float64": "f64", "Cvoid": "void", "Cstring": "cstr"}


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return "\n".join(ln for ln in s.split("\n") if not ln.lstrip().startswith("#"))        # preprocessor lines


def c_type(t):
    """canonical form of a C type string such as 'const double *', 'rat_handle *', 'const rat_ce_solver *'"""
    t = t.replace("const", " ").strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    assert len(base) == 1, t
    base = base[0]
    if base == "char" and stars == 1:
        return "cstr"
    if base in HANDLES:
        kind = "p:void"
    elif base in SCALARS_C:
        kind = SCALARS_C[base]
    else:
        kind = "struct:" + base
    return "p:" * stars + kind


def parse_header():
    h = strip_comments(HDR)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", h, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first, *rest = [p.strip() for p in decl.split(",")]
            mm = re.match(r"(.*?)(\**)\s*(\w+)$", first)
            base, stars, name = mm.group(1).strip(), mm.group(2), mm.group(3)
            fields.append((name, c_type(base + " " + stars)))
            for r in rest:
                mm2 = re.match(r"(\**)\s*(\w+)$", r)
                fields.append((mm2.group(2), c_type(base + " " + mm2.group(1))))
        structs[m.group(3)] = fields
    h_nostruct = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(rat_\w+)\s*\(([^()]*)\)\s*;", h_nostruct):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef") or not ret:
            continue
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)                     # drop the parameter name
                argt.append(c_type(mm.group(1)))
        protos[name] = (c_type(ret), argt)
    return structs, protos


def split_top(s):
    """split on top-level commas (ignoring commas inside {} () [])"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def jl_struct_map():
    """Julia struct name -> (C struct name, [(field, julia type)]) from the `# mirrors` comments"""
    out = {}
    for m in re.finditer(r"# mirrors `struct (\w+)`[^\n]*\n(?:#[^\n]*\n)*(?:mutable )?struct (\w+)\n(.*?)\nend", JL, flags=re.S):
        cname, jname, body = m.group(1), m.group(2), m.group(3)
        fields = []
        for line in body.split("\n"):
            line = line.split("#")[0]
            for f in line.split(";"):
                f = f.strip()
                if f:
                    name, typ = f.split("::")
                    fields.append((name.strip(), typ.strip()))
        out[jname] = (cname, fields)
    return out


def jl_type(t, smap):
    t = t.strip()
    if t in SCALARS_JL:
        return SCALARS_JL[t]
    m = re.match(r"(Ptr|Ref)\{(.*)\}$", t)
    if m:
        inner = m.group(2).strip()
        if inner == "Cvoid":
            return "p:void"
        return "p:" + jl_type(inner, smap)
    if t in smap:
        return "struct:" + smap[t][0]
    raise AssertionError(f"unknown Julia type in a ccall: {t}")


def parse_ccalls():
    calls = []
    for m in re.finditer(r"ccall\(\(:(rat_\w+), LIB\),", JL):
        i = m.end()
        depth, j = 0, i
        parts, cur = [], ""
        while True:                                                  # return type, then the argument-type tuple
            ch = JL[j]
            if ch in "({[":
                depth += 1
            elif ch in ")}]":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur.strip())
                cur = ""
                if len(parts) == 2:
                    break
            elif depth < 0:                                          # `ccall(..., Ret, ())` closes right after the tuple
                parts.append(cur.strip())
                break
            else:
                cur += ch
            j += 1
        ret, tup = parts[0], parts[1]
        assert tup.startswith("(") and tup.endswith(")"), (m.group(1), tup)
        line = JL.count("\n", 0, m.start()) + 1
        calls.append((m.group(1), ret, split_top(tup[1:-1]), line))
    return calls


STRUCTS, PROTOS = parse_header()
SMAP = jl_struct_map()


def test_header_parser_sees_the_whole_abi():
    from ratilqr.jl_amd import _native as nv
    assert set(PROTOS) == set(nv.EXPORTS)                            # the regex parser and the export list agree
    assert PROTOS["rat_create"] == ("i32", ["p:struct:rat_ileqg_opts", "i32", "i32", "i32", "p:p:void"])
    assert PROTOS["rat_last_error"] == ("cstr", []) and PROTOS["rat_stream"] == ("p:void", ["p:void"])
    assert PROTOS["rat_rollout_noisy"][1][4:7] == ["i64", "p:f64", "u64"]
    assert [f[0] for f in STRUCTS["rat_ce_solver"]][-4:] == ["iter_current", "n_solves", "n_redraws", "n_final_retries"]


def test_every_ccall_matches_its_prototype():
    calls = parse_ccalls()
    assert len(calls) >= 40
    for name, ret, args, line in calls:
        assert name in PROTOS, f"julia/RATiLQRAMD.jl:{line}: {name} is not declared in include/ratilqr.h"
        c_ret, c_args = PROTOS[name]
        assert jl_type(ret, SMAP) == c_ret, f"julia/RATiLQRAMD.jl:{line}: {name} returns {c_ret}, ccall says {ret}"
        assert len(args) == len(c_args), f"julia/RATiLQRAMD.jl:{line}: {name} takes {len(c_args)} arguments, ccall passes {len(args)}"
        for k, (a, c) in enumerate(zip(args, c_args)):
            assert jl_type(a, SMAP) == c, f"julia/RATiLQRAMD.jl:{line}: {name} argument {k + 1} is {c}, ccall says {a}"


def test_mirrored_structs_have_the_c_layout():
    want = {"rat_problem_desc", "rat_ileqg_opts", "rat_ce_solver", "rat_nm_solver", "rat_gen_problem_desc", "rat_pets_solver"}
    assert {c for c, _ in SMAP.values()} == want == set(STRUCTS)
    for jname, (cname, jfields) in SMAP.items():
        cfields = STRUCTS[cname]
        assert [f[0] for f in jfields] == [f[0] for f in cfields], f"{jname} vs {cname}: field names / order differ"
        for (fn, jt), (_, ct) in zip(jfields, cfields):
            assert jl_type(jt, SMAP) == ct, f"{jname}.{fn}: Julia {jt}, C {ct}"


def test_positional_struct_constructors_pass_every_field():
    """CeState(...), NmState(...), IleqgOpts(...), PetsState(...) are built positionally: the argument count must equal the field count."""
    for jname in ("CeState", "NmState", "IleqgOpts", "PetsState", "GenProblemDesc", "ProblemDesc"):
        nfields = len(SMAP[jname][1])
        for m in re.finditer(r"(?<![\w{.:])" + jname + r"\(", JL):
            if JL[max(0, m.start() - 7): m.start()].endswith("struct "):
                continue
            depth, j = 1, m.end()
            while depth:
                depth += JL[j] in "({["
                depth -= JL[j] in ")}]"
                j += 1
            args = split_top(JL[m.end(): j - 1])
            line = JL.count("\n", 0, m.start()) + 1
            assert len(args) == nfields, f"julia/RATiLQRAMD.jl:{line}: {jname}(...) has {len(args)} arguments, the struct has {nfields} fields"


def test_bound_surface():
    bound = {c[0] for c in parse_ccalls()}
    not_bound = set(PROTOS) - bound
    allowed = {"rat_default_ileqg_opts", "rat_set_ileqg_opts", "rat_ileqg_solve_batch_dev", "rat_ce_default", "rat_ce_seed",
               "rat_ce_get_positive_samples", "rat_ce_draw", "rat_ce_draw_stream", "rat_pets_initialize",
               "rat_pets_sample_controls", "rat_pets_update", "rat_profile_enable", "rat_profile_reset", "rat_profile_get", "rat_layout_info",
               "rat_ce_update_dev"}
    assert not_bound <= allowed, sorted(not_bound - allowed)
    # the reference's exported names (src/RATiLQR.jl:20-53) exist under their own names
    for name in ("simulate_dynamics", "integrate_cost", "ILEQGSolver", "initialize!", "ApproximationResult", "approximate_model",
                 "DynamicProgrammingResult", "solve_approximate_dp!", "solve_approximate_dp", "increase_μ_and_Δ!", "decrease_μ_and_Δ!",
                 "line_search!", "step!", "solve!", "CrossEntropyBilevelOptimizationSolver", "compute_value_worker", "compute_cost",
                 "compute_cost_serial", "get_positive_samples", "NelderMeadBilevelOptimizationSolver", "compute_cost_worker",
                 "CrossEntropyDirectOptimizationSolver", "OptimalControlProblem"):
        assert re.search(r"(function |struct |abstract type |\n)" + re.escape(name) + r"[\s({]", JL), name
        assert re.search(r"\nexport\b.*[\s,]" + re.escape(name) + r"[,\n]", JL, flags=re.S), name


def test_solvers_rebind_their_handle_to_the_problem_argument():
    """every entry point that takes a problem goes through bind!/handle! (no stale device tables)."""
    for m in re.finditer(r"\nfunction (\w+!?)\((s::\w+), problem(::\w+)?[^\n]*\n(.*?)\nend", JL, flags=re.S):
        name, body = m.group(1), m.group(4)
        if "reference_module()" in body or name in ("handle!", "multi_handle!") or (name in ("initialize!", "line_search!", "step!") and "ILEQGSolver" in m.group(2)):
            continue                                                 # forwarded to the reference / composed from operators that bind
        if "ccall" in body:
            assert "bind!(" in body or "handle!(" in body, f"{name}({m.group(2)}, problem, ...) calls the library without binding the problem"


def _jl_signatures():
    """name -> list of (min_positional, max_positional) over the method definitions of the shim (`function name(...)` and `name(...) = ...`)."""
    sigs = {}
    for m in re.finditer(r"(?m)^(?:function\s+)?([A-Za-z_μΔϵθ][\w!μΔϵθ]*)\(", JL):
        name = m.group(1)
        if JL[max(0, m.start() - 1): m.start()] not in ("", "\n"):
            continue
        depth, j = 1, m.end()
        while depth:
            depth += JL[j] in "({["
            depth -= JL[j] in ")}]"
            j += 1
        if not (JL[m.start():].startswith("function") or re.match(r"\s*(where\b[^=\n]*)?=[^=]", JL[j:j + 40])):
            continue
        inner = JL[m.end(): j - 1]
        pos = split_top(inner.split(";")[0]) if inner.strip() else []
        pos = [a for a in pos if a.strip()]
        nmax = len(pos)
        nmin = len([a for a in pos if "=" not in a])
        sigs.setdefault(name, []).append((nmin, nmax))
    return sigs


def test_runtests_jl_uses_only_names_and_arities_the_shim_defines():
    """julia/runtests.jl (the reference's tests restated on the device families, plus the reference-vs-device timing) cannot run here:
    every exported RATiLQRAMD function it calls must exist with a method of that positional arity, every struct it builds with that
    many fields, and every ILEQGSolver field it reads must be a field of the shim's struct."""
    RT = open(os.path.join(ROOT, "julia", "runtests.jl")).read()
    RT = "\n".join(ln.split("#")[0] if not ln.lstrip().startswith("#") else "" for ln in RT.split("\n"))
    exports = set(re.findall(r"[\w!μΔϵθ]+", re.search(r"\nexport\b(.*?)\nend", JL, flags=re.S).group(1)))
    sigs = _jl_signatures()
    structs = {m.group(1): len([f for f in re.split(r"[;\n]", m.group(2)) if re.match(r"\s*\w[\w!μΔϵθ]*::", f)])
               for m in re.finditer(r"\n(?:mutable )?struct (\w+)(?: <: \w+)?\n(.*?)\nend", JL, flags=re.S)}
    used = 0
    for m in re.finditer(r"(?<![\w.!])([A-Za-z_μΔϵθ][\w!μΔϵθ]*)\(", RT):
        name = m.group(1)
        if name not in exports or RT[max(0, m.start() - 8): m.start()].endswith("RATiLQR."):
            continue
        depth, j = 1, m.end()
        while depth:
            depth += RT[j] in "({["
            depth -= RT[j] in ")}]"
            j += 1
        inner = RT[m.end(): j - 1]
        npos = len([a for a in split_top(inner.split(";")[0]) if a.strip() and not re.match(r"\s*[\wμΔϵθ]+\s*=[^=]", a)]) if inner.strip() else 0
        line = RT.count("\n", 0, m.start()) + 1
        used += 1
        if name in sigs:
            assert any(lo <= npos <= hi for lo, hi in sigs[name]), f"julia/runtests.jl:{line}: {name} called with {npos} positional arguments; the shim defines {sigs[name]}"
        else:
            assert name in structs, f"julia/runtests.jl:{line}: {name} is exported but has no method or struct in the shim"
            if name in ("LQRiskSensitiveProblem",):
                assert npos == structs[name], f"julia/runtests.jl:{line}: {name}(...) has {npos} arguments, the struct has {structs[name]} fields"
    assert used > 40
    fields = set(re.findall(r"(\w[\w!μΔϵθ_]*)::", re.search(r"\nmutable struct ILEQGSolver\n(.*?)\nend", JL, flags=re.S).group(1)))
    for f in re.findall(r"\b(?:solver|sr|sp|s3|s4)\.([\w!μΔϵθ_]+)", RT):
        assert f in fields, f"julia/runtests.jl reads ILEQGSolver.{f}, which the shim's struct does not have"
    for f in re.findall(r"\bs2\.c\.(\w+)", RT):
        assert f in re.search(r"\nmutable struct CeState\n(.*?)\nend", JL, flags=re.S).group(1)


def _function_body(name, first_arg_pat):
    m = re.search(r"\nfunction " + re.escape(name) + r"\(" + first_arg_pat + r".*?\n(.*?)\nend\n", JL, flags=re.S)
    assert m, name
    return m.group(1)


def test_closure_problems_reach_the_batched_device_sweeps():
    """VERDICT r03 next #7 / SURVEY 8f #3: a closure problem's CE batch does not fall through to the reference's CPU solve -- the host
    evaluates closures with the reference's own simulate_dynamics / approximate_model and every sweep of a round is ONE device launch."""
    body = _function_body("solve_closure_batch", r"o::IleqgOpts, problem")
    # host side: the reference's functions, both the ForwardDiff and the user-Jacobian forms (ileqg.jl:265-273, 302-311)
    assert "R.simulate_dynamics(problem" in body and "f_returns_jacobian=true" in body and "R.approximate_model(problem, u, x, A, Bm)" in body
    # device side: the two batch entry points, nothing per sample
    assert "solve_approximate_dp_batch!(cs" in body and "solve_approximate_dp_batch(cs" in body
    assert "rat_dp_gain_sweep\"" not in body and "R.solve!" not in body
    # the reference's decisions, by line: accept rule, forced accept below eps_min, convergence / iter_max
    for frag in ("new ≈ cur || new < cur", "ϵ[b] < o.eps_min || continue", "o.d > d_cur[b] && μ[b] <= o.mu_min", "iters[b] == o.iter_max"):
        assert frag in body, frag
    # the batch wrappers bind the ABI entry points with the header's prototypes (checked call by call in test_every_ccall_matches_its_prototype)
    bound = {c[0] for c in parse_ccalls()}
    assert {"rat_dp_gain_sweep_batch", "rat_dp_policy_eval_batch", "rat_ce_update", "rat_ce_begin_step"} <= bound
    # compute_cost / solve! of the CE solver on an untyped (closure) problem go through it; forwarding to the reference is the opt-out
    cc = _function_body("compute_cost", r"s::CrossEntropyBilevelOptimizationSolver, problem, x::Vector\{Float64\}, u_array::Vector\{Vector\{Float64\}\}")
    assert "solve_closure_batch(" in cc and "kl_bound ./ θ_array" in cc
    sv = _function_body("solve!", r"s::CrossEntropyBilevelOptimizationSolver, problem, x_0")
    assert "closure_device[]" in sv and "compute_cost(s, problem" in sv and "rat_ce_update" in sv and "θ_opt - c.sigma" in sv
    assert re.search(r"if !closure_device\[\] \|\| serial\n.*?R\.solve!\(ref", sv, flags=re.S)
    # the carrier handle only contributes W(k), 0-based k as in optimal_control_problems.jl:67-73
    assert "problem.W(k)) for k in 0:N-1" in JL

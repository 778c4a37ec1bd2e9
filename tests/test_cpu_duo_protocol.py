"""Model checks of the protocol between the TWO WORKGROUPS of a sample in solve_block_psw_kernel (csrc/kernels.hip, round 6) -- no GPU needed.

(1) The check-in: role B announces itself with a compare-and-swap on the pair word, role A decides ONCE whether its partner is there ("duo")
    or not ("solo").  Every interleaving of the two roles' atomic steps is enumerated (role B may start arbitrarily late, or never): the
    outcomes must be (duo, duo) or (solo, leaves) -- never one role waiting for the other, never a role B that carries on beside a solo role A
    -- and a stale word of an earlier launch must read as "nobody there".
(2) The hand-overs: role A posts once per loop iteration (a round's rollout, or "plain gain sweep wanted") and once on exit; role B posts once per
    sweep; both count posts in sequence numbers.  The two control flows are restated as coroutines over a shared memory and run under random
    schedules for random decision sequences (rejected candidates, "accepting would end the solve", failures, the round guard): they must both
    terminate, role B must have swept exactly what role A's state asked for, and no wait may be left hanging.

These are restatements of the kernel's control flow (the functions below name the lines they mirror), not the kernel: the race hunt and the
parity tests on the device (tests/test_gpu_psweep.py, tools/stress_block.py) hold the real thing."""
import itertools
import random

STALE, B_HERE, SOLO, DUO = "stale", 1, 2, 4


# ---- (1) the check-in ------------------------------------------------------------------------------------------------------------------
def role_a_steps(grace):
    """Role A after initialize!'s copy (kernels.hip: `if (fa.duo_stride > 0) { ... }`): `grace` polls for a partner that has checked in, then a
    CAS stale -> SOLO; a CAS that fails because the partner has just arrived decides as if it had been seen."""
    def prog(mem, same_xcd):
        for _ in range(grace):
            w = yield ("load",)
            if w == B_HERE:
                code = DUO if same_xcd else SOLO
                yield ("store", code)
                return code
        while True:
            w = yield ("load",)
            if w == B_HERE:
                code = DUO if same_xcd else SOLO
                yield ("store", code)
                return code
            ok = yield ("cas", w, SOLO)
            if ok:
                yield ("store", SOLO)
                return SOLO
    return prog


def role_b_steps(polls):
    """Role B at its start: CAS anything-that-is-not-this-launch's -> B_HERE (a word of this launch can only be SOLO: role A was first), then
    polls until role A has decided (bounded in the kernel; role A is resident, so the decision comes)."""
    def prog(mem, same_xcd):
        code = None
        while code is None:
            w = yield ("load",)
            if w in (SOLO, DUO):
                code = w
                break
            ok = yield ("cas", w, B_HERE)
            if ok:
                code = B_HERE
        for _ in range(polls):
            if code != B_HERE:
                break
            w = yield ("load",)
            if w != B_HERE:
                code = w
        return code
    return prog


def run_interleaving(order, grace, same_xcd, b_present=True):
    mem = {"w": STALE}
    progs = {"A": role_a_steps(grace)(mem, same_xcd)}
    if b_present:
        progs["B"] = role_b_steps(64)(mem, same_xcd)
    pending, result = {}, {}
    for k, g in progs.items():
        pending[k] = next(g)
    it = iter(order)
    while pending:
        who = next(it, None)
        if who is None or who not in pending:
            who = sorted(pending)[0] if who is None else (sorted(pending)[0] if who not in pending else who)
        op = pending[who]
        if op[0] == "load":
            val = mem["w"]
        elif op[0] == "store":
            mem["w"] = op[1]
            val = None
        else:
            val = mem["w"] == op[1]
            if val:
                mem["w"] = op[2]
        try:
            pending[who] = progs[who].send(val)
        except StopIteration as e:
            result[who] = e.value
            del pending[who]
    return result


def test_check_in_every_interleaving():
    outcomes = set()
    for grace in (0, 1, 2):
        for same in (True, False):
            for order in itertools.product("AB", repeat=9):
                r = run_interleaving(order, grace, same)
                a, b = r["A"], r["B"]
                assert (a, b) in ((DUO, DUO), (SOLO, SOLO)), (order, grace, same, r)
                assert same or a == SOLO                          # partners in different XCDs never pair
                outcomes.add((a, b))
            assert run_interleaving((), grace, same, b_present=False) == {"A": SOLO}      # a partner that never starts: role A does not wait
    assert outcomes == {(DUO, DUO), (SOLO, SOLO)}


# ---- (2) the hand-overs ----------------------------------------------------------------------------------------------------------------
# Role A's post = (sequence number << 2) | request: 0 nothing, 1 the plain gain sweep (mode 0), 2 the speculative one (mode 4), 3 over.
def role_a(mem, script, max_rounds, log):
    """Role A's loop in duo mode (kernels.hip, `for (guard ...)`): script[k] = (act, ends, outcome) of iteration k with outcome in
    {"accept", "reject", "fail"}; `act` False = no valid speculative sweep: role B runs the plain gain sweep (mode 0)."""
    seqA = seqB = 0
    yield from wait(mem, "B", seqB := seqB + 1)                    # initialize!'s evaluation beside role B's first gain sweep (mode 5)
    running, k = True, 0
    for _ in range(max_rounds):
        if not running or k >= len(script):
            break
        act, ends, outcome = script[k]
        k += 1
        mem["round"] = k                                            # (the sample's state words: overwritten every round)
        if not act:
            seqA += 1; mem["A"] = (seqA << 2) | 1; log.append(("A asks", 0, k))
            yield from wait(mem, "B", seqB := seqB + 1)
            continue
        seqA += 1; mem["A"] = (seqA << 2) | (0 if ends else 2)
        if not ends:
            log.append(("A asks", 4, k))
        yield                                                                              # (the evaluation)
        if not ends:
            yield from wait(mem, "B", seqB := seqB + 1)
        if outcome == "fail" or (outcome == "accept" and ends):
            running = False
    seqA += 1; mem["A"] = (seqA << 2) | 3
    return "done"


def role_b(mem, max_rounds, log):
    """Role B's loop (kernels.hip, `if (role == 1) { ... }`): the request comes with the post; the state words are read by the sweep only."""
    seqA = seqB = 0
    mode = 5
    for _ in range(max_rounds + 1):
        log.append(("B sweeps", mode, mem.get("round", 0)))
        yield                                                                              # (the sweep)
        seqB += 1; mem["B"] = seqB
        while True:
            seqA += 1
            got = yield from wait(mem, "A", seqA << 2)
            code = got & 3
            if code == 3:
                return "done"
            if (got >> 2) > seqA or code == 0:
                continue
            mode = 0 if code == 1 else 4
            break
    return "guard"


def wait(mem, who, want):
    polls = 0
    while mem[who] < want:
        polls += 1
        assert polls < 10000, f"a wait on {who} >= {want} was left hanging"
        yield
    return mem[who]


def test_hand_overs_random_scripts_and_schedules():
    rng = random.Random(7)
    for trial in range(3000):
        n = rng.randint(0, 12)
        script = []
        for _ in range(n):
            act = rng.random() < 0.8
            ends = act and rng.random() < 0.4
            script.append((act, ends, rng.choice(["accept", "accept", "reject", "fail"]) if act else "reject"))
        max_rounds = rng.choice([len(script) + 3, max(1, len(script) - 2)])               # (the round guard may cut the loop short)
        mem, log = {"A": 0, "B": 0}, []
        a, b = role_a(mem, script, max_rounds, log), role_b(mem, max_rounds + 4, log)
        live = {"A": a, "B": b}
        done = {}
        steps = 0
        slow = rng.choice(["A", "B", None])                                               # (one role may be much slower than the other)
        while live:
            who = rng.choice(sorted(live))
            if slow in live and who == slow and rng.random() < 0.8 and len(live) > 1:
                continue
            try:
                next(live[who])
            except StopIteration as e:
                done[who] = e.value
                del live[who]
            steps += 1
            assert steps < 400000
        assert done == {"A": "done", "B": "done"}, (script, done)
        # role B swept exactly what role A asked for, in order, each sweep on the state words of the round that asked for it
        asked = [(m, k) for what, m, k in log if what == "A asks"]
        swept = [(m, k) for what, m, k in log if what == "B sweeps"]
        assert swept[0][0] == 5 and swept[1:] == asked, (script, asked, swept)

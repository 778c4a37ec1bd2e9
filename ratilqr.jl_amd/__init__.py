"""ratilqr.jl_amd -- MI355X-native iLEQG + Cross-Entropy (RAT iLQR) hot path.

Host-side mirror of the reference's exported API (src/RATiLQR.jl:20-53) over the C ABI of
``csrc/libratilqr_hip.so`` (declared in include/ratilqr.h).  Import as ``import ratilqr.jl_amd``.
"""
from .problems import (  # noqa: F401
    OptimalControlProblem,
    FiniteHorizonRiskSensitiveOptimalControlProblem,
    LQRiskSensitiveProblem,
    PowerLawRiskSensitiveProblem,
    synthetic_lq_problem,
)
from ._native import RatError, SO_PATH  # noqa: F401,E402
from . import _native as native  # noqa: F401,E402
from .ileqg import (  # noqa: F401,E402
    Context,
    ILEQGSolver,
    ApproximationResult,
    DynamicProgrammingResult,
    simulate_dynamics,
    simulate_dynamics_noisy,
    integrate_cost,
    approximate_model,
    solve_approximate_dp,
    solve_approximate_dp_,
    increase_mu_and_delta_,
    decrease_mu_and_delta_,
    line_search_,
    step_,
    solve_,
    solve_stepwise_,
)
from .ileqg import initialize_ as initialize_ileqg_  # noqa: F401,E402
from . import ileqg, cross_entropy  # noqa: F401,E402
from .generic import GenericRiskSensitiveProblem, GenericContext, solve_closure_batch  # noqa: F401,E402
from .cross_entropy import (  # noqa: F401,E402
    CrossEntropyBilevelOptimizationSolver,
    compute_cost,
    compute_cost_serial,
    compute_value_worker,
    get_positive_samples,
)
from . import multi  # noqa: F401,E402
from .multi import MultiContext  # noqa: F401,E402
from . import nelder_mead  # noqa: F401,E402
from .nelder_mead import NelderMeadBilevelOptimizationSolver, compute_cost_worker  # noqa: F401,E402
from .problems import FiniteHorizonGenerativeOptimalControlProblem, LQGenerativeProblem  # noqa: F401,E402
from . import pets  # noqa: F401,E402
from .pets import CrossEntropyDirectOptimizationSolver  # noqa: F401,E402

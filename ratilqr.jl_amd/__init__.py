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

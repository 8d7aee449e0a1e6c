"""
brancher_amd — MI355X-native stochastic variational inference engine behind Brancher's API.

Scope: the ELBO-gradient hot path of LucaAmbrogioni/Brancher
(`inference.perform_inference -> ProbabilisticModel.estimate_log_model_evidence`),
re-implemented as hand-written HIP kernels for gfx950 behind a C ABI (include/bsvi.h).
See DESIGN.md.
"""
__version__ = "0.1.0"

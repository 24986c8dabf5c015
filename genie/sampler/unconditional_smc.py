from genie2_amd.smc import TwistedSampler as SMCSampler, systematic_resampling, compute_ess, compute_ess_from_log_w, normalize_weights, normalize_log_weights  # noqa: F401

"""SGA temperature schedule (reference common/latent_rvs_utils.py:90-103)."""
import math


def sga_schedule_at_step(t, r, ub, lb=1e-8, t0=200.0):
    """tau(t) = clamp(ub * exp(-r (t - t0)), lb, ub)."""
    tau = ub * math.exp(-r * (float(t) - t0))
    return min(max(tau, lb), ub)

"""MI355X-native Monte Carlo race-simulation engine (hot path of dan-lee-gh/monte-carlo-gp).

Public surface mirrors reference src/simulation.py: CarState, RaceConfig, RaceSimulator; run_monte_carlo_batch runs several
races in one launch.
The compute path is the HIP library libmcgp_hip.so (C ABI: include/mcgp.h); there is no
CPU fallback.
"""
from .simulation import CarState, RaceConfig, RaceSimulator, histogram_to_probs, run_monte_carlo_batch  # noqa: F401
from . import config  # noqa: F401

__all__ = ['CarState', 'RaceConfig', 'RaceSimulator', 'histogram_to_probs', 'run_monte_carlo_batch', 'config']

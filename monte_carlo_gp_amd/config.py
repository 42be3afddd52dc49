"""Per-season constants that feed the simulation path.

The tables live in data/season_constants.json (values of the reference's data tables,
src/config.py:7-51,54-78, and of the RaceConfig constants hard-coded at src/predictor.py:59-61,64);
this module only exposes them under the reference's names.  The host-side feature glue of the
reference's config module (penalty types beyond the grid shift, track types) is out of scope
(SURVEY.md section 2, row 2).  Dict order is the file's order (it is the canonical driver order).
"""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'season_constants.json')) as _f:
    _C = json.load(_f)

DRIVER_TEAMS: dict = _C['driver_teams']
DEFAULT_DNF_RATES: dict = _C['default_dnf_rates']
TIRE_COMPOUNDS: dict = _C['tire_compounds']
CIRCUITS: dict = _C['circuits']
SC_PROBABILITY: float = _C['sc_probability']
VSC_PROBABILITY: float = _C['vsc_probability']
RED_FLAG_PROBABILITY: float = _C['red_flag_probability']
DRS_DELTA: float = _C['drs_delta']

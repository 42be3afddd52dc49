"""Per-season constants that feed the simulation path.

Values are the data tables of reference src/config.py:7-51,54-78 (driver/team
map, per-team DNF rates, tyre compounds, circuits) and the RaceConfig constants
hard-coded at reference src/predictor.py:59-61,64.  Data only; the host-side
feature glue of the reference's config (penalty types, track types) is out of
scope.
"""

DRIVER_TEAMS = {
    'VER': 'Red Bull', 'LAW': 'Red Bull', 'NOR': 'McLaren', 'PIA': 'McLaren',
    'LEC': 'Ferrari', 'HAM': 'Ferrari', 'RUS': 'Mercedes', 'ANT': 'Mercedes',
    'ALO': 'Aston Martin', 'STR': 'Aston Martin', 'GAS': 'Alpine', 'DOO': 'Alpine',
    'TSU': 'Racing Bulls', 'HAD': 'Racing Bulls', 'ALB': 'Williams', 'SAI': 'Williams',
    'HUL': 'Sauber', 'BOR': 'Sauber', 'OCO': 'Haas', 'BEA': 'Haas',
}

DEFAULT_DNF_RATES = {
    'Red Bull': 0.0015, 'McLaren': 0.0012, 'Ferrari': 0.0018, 'Mercedes': 0.0010,
    'Aston Martin': 0.0020, 'Alpine': 0.0025, 'Racing Bulls': 0.0022, 'Williams': 0.0025,
    'Sauber': 0.0028, 'Haas': 0.0025,
}

TIRE_COMPOUNDS = {
    'SOFT': {'pace_delta': -0.8, 'deg_rate': 0.08, 'optimal_laps': 15},
    'MEDIUM': {'pace_delta': 0.0, 'deg_rate': 0.05, 'optimal_laps': 25},
    'HARD': {'pace_delta': 0.6, 'deg_rate': 0.03, 'optimal_laps': 40},
    'INTERMEDIATE': {'pace_delta': 5.0, 'deg_rate': 0.02, 'optimal_laps': 30},
    'WET': {'pace_delta': 10.0, 'deg_rate': 0.01, 'optimal_laps': 50},
}

CIRCUITS = {
    'Bahrain': {'laps': 57, 'pit_loss': 21.0, 'drs_zones': 3, 'overtake_delta': 0.6},
    'Saudi Arabia': {'laps': 50, 'pit_loss': 20.0, 'drs_zones': 3, 'overtake_delta': 0.7},
    'Australia': {'laps': 58, 'pit_loss': 22.0, 'drs_zones': 4, 'overtake_delta': 0.5},
    'Japan': {'laps': 53, 'pit_loss': 23.0, 'drs_zones': 1, 'overtake_delta': 1.0},
    'China': {'laps': 56, 'pit_loss': 22.0, 'drs_zones': 2, 'overtake_delta': 0.6},
    'Miami': {'laps': 57, 'pit_loss': 21.0, 'drs_zones': 3, 'overtake_delta': 0.7},
    'Monaco': {'laps': 78, 'pit_loss': 24.0, 'drs_zones': 1, 'overtake_delta': 1.5},
    'Canada': {'laps': 70, 'pit_loss': 22.0, 'drs_zones': 2, 'overtake_delta': 0.6},
    'Spain': {'laps': 66, 'pit_loss': 21.0, 'drs_zones': 2, 'overtake_delta': 0.8},
    'Austria': {'laps': 71, 'pit_loss': 20.0, 'drs_zones': 3, 'overtake_delta': 0.5},
    'Great Britain': {'laps': 52, 'pit_loss': 21.0, 'drs_zones': 2, 'overtake_delta': 0.7},
    'Hungary': {'laps': 70, 'pit_loss': 22.0, 'drs_zones': 1, 'overtake_delta': 1.2},
    'Belgium': {'laps': 44, 'pit_loss': 23.0, 'drs_zones': 2, 'overtake_delta': 0.5},
    'Netherlands': {'laps': 72, 'pit_loss': 20.0, 'drs_zones': 2, 'overtake_delta': 1.0},
    'Italy': {'laps': 53, 'pit_loss': 26.0, 'drs_zones': 2, 'overtake_delta': 0.4},
    'Azerbaijan': {'laps': 51, 'pit_loss': 24.0, 'drs_zones': 2, 'overtake_delta': 0.5},
    'Singapore': {'laps': 62, 'pit_loss': 30.0, 'drs_zones': 3, 'overtake_delta': 1.1},
    'United States': {'laps': 56, 'pit_loss': 21.0, 'drs_zones': 2, 'overtake_delta': 0.7},
    'Mexico': {'laps': 71, 'pit_loss': 22.0, 'drs_zones': 3, 'overtake_delta': 0.6},
    'Brazil': {'laps': 71, 'pit_loss': 21.0, 'drs_zones': 2, 'overtake_delta': 0.5},
    'Las Vegas': {'laps': 50, 'pit_loss': 21.0, 'drs_zones': 2, 'overtake_delta': 0.6},
    'Qatar': {'laps': 57, 'pit_loss': 21.0, 'drs_zones': 2, 'overtake_delta': 0.8},
    'Abu Dhabi': {'laps': 58, 'pit_loss': 22.0, 'drs_zones': 2, 'overtake_delta': 0.7},
}

# RaceConfig constants of reference src/predictor.py:59-61,64
SC_PROBABILITY = 0.01
VSC_PROBABILITY = 0.015
RED_FLAG_PROBABILITY = 0.002
DRS_DELTA = 0.3

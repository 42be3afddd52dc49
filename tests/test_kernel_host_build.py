"""The register kernel's SOURCE (csrc/race_kernel_reg.hip.h), compiled for the host by tools/emu and run one
thread at a time, against the CPU oracle's Philox back-end: bit-exact finishing orders and histograms.

This is a CPU-side regression net for the kernel's race logic (it runs without a GPU, in seconds); the parity
tests proper are the -m gpu tests, which run the gfx950 build through the C ABI.  The host build is test
infrastructure: nothing under monte_carlo_gp_amd/ can reach it and the product has no CPU path."""
import json

import numpy as np
import pytest

import kernel_host_build as K
import oracle_py as O


@pytest.mark.parametrize('name', ['S60', 'S78', 'S50', 'EVT', 'HET', 'DMP', 'WET', 'N10'])
def test_golden_cases(name):
    case = O.load_case(name)
    n_sims = 600
    ref = O.Problem(case).run(n_sims, rng=O.RNG_PHILOX, seed=42, want_orders=True)
    hist, orders = K.run(case, n_sims, 42)
    bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
    assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
    assert np.array_equal(hist, ref['hist'])


def test_fuzzed_configurations():
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        cases = json.load(f)
    skipped = []
    for name, c in cases.items():
        if not (1 <= len(c["grid_probs"]) <= 32):
            continue                                   # field sizes served by the generic LDS kernel
        ref = O.Problem(c).run(300, rng=O.RNG_PHILOX, seed=c['seed'], want_orders=True)
        try:
            hist, orders = K.run(c, 300, c['seed'])
        except K.NotServed:
            skipped.append(name)               # a DNF probability >= 1: the generic kernel's case (GPU fuzz test)
            continue
        bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
        assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
        assert np.array_equal(hist, ref['hist']), name
    assert len(skipped) <= 3, skipped


# S60, seed 42: the two simulations of the first 10^9 in which a GRID draw is the word 0, i.e. u = 0.0 exactly (found by
# tests/test_gpu_scale.py's 10^9 run, whose Latin-square check caught a driver placed twice: a cdf entry of 0 -- a
# driver already placed -- had been taken for "above u")
ZERO_GRID_DRAWS = (33884187, 666225793)


@pytest.mark.parametrize('sim', ZERO_GRID_DRAWS)
def test_grid_draw_of_exactly_zero(sim):
    case = O.load_case('S60')
    ref = O.Problem(case).run(3, rng=O.RNG_PHILOX, seed=42, sim_offset=sim - 1, want_orders=True)
    hist, orders = K.run(case, 3, 42, sim_offset=sim - 1)
    assert sorted(orders[1].tolist()) == list(range(20))
    assert np.array_equal(orders, ref['orders'])
    assert np.array_equal(hist, ref['hist'])


def test_offsets_and_64bit_seeds():
    case = O.load_case('S50')
    seed, base = 0xDEADBEEFCAFEF00D, (1 << 40) + 12345
    ref = O.Problem(case).run(400, rng=O.RNG_PHILOX, seed=seed, sim_offset=base, want_orders=True)
    hist, orders = K.run(case, 400, seed, sim_offset=base)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


def test_inverse_normal_transform_matches_the_oracle_word_for_word():
    """race_common.hip.h: normal_from_u32 against the oracle's, on the cells parity runs never reach (the 16
    smallest tail cells of either sign have probability 7e-9 per draw), every segment boundary, and a random sample."""
    import ctypes as C
    L = K.lib()
    L.emu_normal_from_u32.restype = C.c_float
    L.emu_normal_from_u32.argtypes = [C.c_uint32]
    Lo = O.lib()
    Lo.orc_normal_from_u32.restype = C.c_float
    Lo.orc_normal_from_u32.argtypes = [C.c_uint32]
    rng = np.random.default_rng(3)
    words = list(range(0, 64)) + [0x80000000 | i for i in range(64)]
    for b in range(4, 32):
        for d in (-2, -1, 0, 1, 2):
            words += [((1 << b) + d) & 0xffffffff, ((1 << b) + d + (1 << (b - 1))) & 0xffffffff]
    words += [0x7fffffff, 0xffffffff, 0x7ffffff0, 0xfffffff0] + [int(x) for x in rng.integers(0, 2 ** 32, 20000)]
    for w in words:
        a, b = L.emu_normal_from_u32(w), Lo.orc_normal_from_u32(w)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), hex(w)


def _field_of(n):
    """An n-car field with S60's parameters, 30 laps, an all-zero grid column (uniform fallback, reference :126-129) and
    per-driver spreads."""
    rng = np.random.default_rng(n)
    drivers = [f'D{i:02d}' for i in range(n)]
    base = O.load_case('S60')
    case = dict(base)
    case['config'] = dict(base['config'], total_laps=30,
                          driver_teams={d: list(base['config']['dnf_rates'])[i % 10] for i, d in enumerate(drivers)})
    g = rng.random((n, n))
    g[:, n // 2] = 0.0
    case['grid_probs'] = {d: [float(x) for x in g[i]] for i, d in enumerate(drivers)}
    case['base_pace'] = {d: 90.0 + 0.15 * i for i, d in enumerate(drivers)}
    case['tire_deg'] = {d: 0.03 + 0.002 * i for i, d in enumerate(drivers)}
    case['driver_variance'] = {d: 0.2 for d in drivers}
    case['driver_dnf_rates'] = {d: 0.01 for d in drivers}
    return case


@pytest.mark.parametrize('n', [1, 2, 5, 9, 13, 17, 18, 22, 25, 26, 28, 32])
def test_field_sizes_up_to_the_abi_maximum(n):
    """Every register instantiation is the same source; sizes beyond the golden / fuzz cases, with an all-zero
    grid column (uniform fallback, reference :126-129) and per-driver spreads."""
    case = _field_of(n)
    ref = O.Problem(case).run(300, rng=O.RNG_PHILOX, seed=5, want_orders=True)
    hist, orders = K.run(case, 300, 5)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


@pytest.mark.parametrize('name', ['S60', 'HET', 'N10'])
def test_sample_grid_exact_path_equals_the_fast_path(name):
    """_sample_grid decides `cdf[d] / cdf[-1] > u` without dividing when the draw is outside a 2^-40 band around the
    threshold, and falls back to the reference's divisions otherwise: the build that divides for EVERY draw
    (MCGP_GRID_EXACT) must give the same races, and both equal the oracle (reference :119-137)."""
    case = O.load_case(name)
    ref = O.Problem(case).run(400, rng=O.RNG_PHILOX, seed=7, want_orders=True)
    _, fast = K.run(case, 400, 7)
    _, exact = K.run(case, 400, 7, variant='grid_exact')
    assert np.array_equal(fast, exact) and np.array_equal(exact, ref['orders'])


def test_more_than_eight_attempts_in_a_pass():
    """overtake_delta = 0: every pair with a pace advantage attempts, so most passes have a lane with more than the
    eight attempts the W plane holds and take the general path, eight at a time (reference :516-524)."""
    import copy
    case = copy.deepcopy(O.load_case('S60'))
    case['config']['overtake_delta'] = 0.0
    ref = O.Problem(case).run(300, rng=O.RNG_PHILOX, seed=5, want_orders=True)
    hist, orders = K.run(case, 300, 5)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


@pytest.mark.parametrize('name', ['S60', 'S78', 'S50', 'EVT', 'HET', 'DMP', 'WET', 'N10'])
def test_reference_width_deviates_match_the_oracle(name):
    """deviates = 53 (reg_simulate<N, true>): 53-bit uniforms and binary64 normals, the reference's width (reference
    :137,194,302,330,524), against the oracle's PHILOX53 back-end -- same words, same companion blocks, same table."""
    case = O.load_case(name)
    n_sims = 400
    ref = O.Problem(case).run(n_sims, rng=O.RNG_PHILOX53, seed=42, want_orders=True)
    hist, orders = K.run(case, n_sims, 42, deviates=53)
    bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
    assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
    assert np.array_equal(hist, ref['hist'])


def test_register_kernel_domain():
    """reg_kernel_serves: lap times that are not safely positive (reg_time_floor < 8 s) and a negative overtake_delta are
    the generic kernel's; 20-second laps are served, with the oracle's results."""
    import copy
    case = copy.deepcopy(O.load_case('S60'))
    case['base_pace'] = {d: 5.0 + 0.1 * i for i, d in enumerate(case['base_pace'])}
    with pytest.raises(K.NotServed):
        K.run(case, 10, 1)
    case['base_pace'] = {d: 20.0 + 0.1 * i for i, d in enumerate(case['base_pace'])}
    ref = O.Problem(case).run(300, rng=O.RNG_PHILOX, seed=11, want_orders=True)
    hist, orders = K.run(case, 300, 11)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])
    case['config']['overtake_delta'] = -0.5
    with pytest.raises(K.NotServed):
        K.run(case, 10, 1)


def test_fuzzed_configurations_at_reference_width():
    """The host build of reg_simulate<N, true> against the oracle's PHILOX53 back-end on every fuzz configuration the
    register kernel takes, whatever its field size (the GPU run of the same comparison: tests/test_gpu_fuzz.py)."""
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        cases = json.load(f)
    done, sizes = 0, set()
    for name, c in cases.items():
        if c['config']['overtake_delta'] < 0:
            continue
        ref = O.Problem(c).run(120, rng=O.RNG_PHILOX53, seed=c['seed'], want_orders=True)
        try:
            hist, orders = K.run(c, 120, c['seed'], deviates=53)
        except K.NotServed:
            continue
        assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist']), name
        done += 1
        sizes.add(len(c['grid_probs']))
    assert done >= 70 and len(sizes) >= 12, (done, sorted(sizes))


@pytest.mark.parametrize('n', [1, 2, 5, 18, 19, 22, 23, 28, 29, 32])
def test_reference_width_at_other_field_sizes(n):
    """deviates = 53 is built for every field size: 19-22-car sessions, and the sizes where the block's
    LDS holds all 512 / some / none of the table rows it wants (RegGeo::kNorm53Rows: 28 cars 96 rows, 29+ none)."""
    case = _field_of(n)
    ref = O.Problem(case).run(150, rng=O.RNG_PHILOX53, seed=9, want_orders=True)
    hist, orders = K.run(case, 150, 9, deviates=53)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


@pytest.mark.parametrize('name', ['S60', 'EVT', 'HET', 'N10'])
def test_reference_width_exact_paths_equal_the_fast_paths(name):
    """The reference-width build decides a Bernoulli draw by the leading word alone unless the word EQUALS the leading
    word of its threshold (one draw in 2^32), and reads a deviate's table row from LDS unless the block does not hold
    it (one in 2^32 at 20 cars).  The `wide_exact` host build takes the exact / device-memory path for every draw:
    same finishing orders as the default host build and as the oracle."""
    case = O.load_case(name)
    ref = O.Problem(case).run(300, rng=O.RNG_PHILOX53, seed=42, want_orders=True)
    hist, orders = K.run(case, 300, 42, deviates=53, variant='wide_exact')
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])
    hist2, orders2 = K.run(case, 300, 42, deviates=53)
    assert np.array_equal(orders2, orders) and np.array_equal(hist2, hist)
    # ... and a build in which a few per cent of the draws count as ties (a word within 2^27 of its threshold's leading
    # word): the exact path entered by simulations that need it and by their neighbours that do not
    hist3, orders3 = K.run(case, 300, 42, deviates=53, variant='wide_near_ties')
    assert np.array_equal(orders3, orders) and np.array_equal(hist3, hist)


def test_reference_width_threshold_ties_take_the_exact_path():
    """Probabilities chosen so that draw words DO equal the leading word of their threshold: an event probability of 1
    (T = 2^53: every word is below), of 0 (T = 0: a word of 0 ties) and pace gaps beyond the 0.5 cap (thr = 2^31)."""
    import copy
    case = copy.deepcopy(O.load_case('EVT'))
    for key, val in (('sc_probability', 1.0), ('vsc_probability', 0.0), ('red_flag_probability', 2.0 ** -40)):
        c = copy.deepcopy(case)
        c['config'][key] = val
        ref = O.Problem(c).run(200, rng=O.RNG_PHILOX53, seed=3, want_orders=True)
        hist, orders = K.run(c, 200, 3, deviates=53)
        assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist']), key


def test_thousand_lap_race():
    """MCGP_MAX_LAPS: the retirement keys (lap << 5 | driver, 15 bits) and the age field of pk at their largest."""
    import copy
    case = copy.deepcopy(O.load_case('N10'))
    case['config']['total_laps'] = 1000
    case['driver_dnf_rates'] = {d: 0.002 for d in case['base_pace']}          # most cars out somewhere in 1000 laps
    ref = O.Problem(case).run(24, rng=O.RNG_PHILOX, seed=3, want_orders=True)
    hist, orders = K.run(case, 24, 3)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])
    # ... and at reference width, where the block's LDS has no room for a 1000-lap table of retirement thresholds (the
    # chain is then worked out per wave) -- and a 400-lap race, whose table still fits a 10-car block
    for laps, n_sims in ((1000, 16), (400, 24)):
        case['config']['total_laps'] = laps
        ref = O.Problem(case).run(n_sims, rng=O.RNG_PHILOX53, seed=3, want_orders=True)
        hist, orders = K.run(case, n_sims, 3, deviates=53)
        assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist']), laps

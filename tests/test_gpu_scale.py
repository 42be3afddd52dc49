"""Full-size checks (BASELINE.json sizes) through size-independent properties, and the
statistical link to the reference's own Mersenne-Twister runs."""
import numpy as np
import pytest

import oracle_py as O
from helpers import product_run

pytestmark = pytest.mark.gpu


def _check_latin(hist, n_sims):
    # every simulation classifies every driver exactly once: rows and columns sum to N
    assert (hist.sum(axis=0) == n_sims).all() and (hist.sum(axis=1) == n_sims).all()


@pytest.mark.parametrize('name,n_sims', [('S60', 10_000_000), ('S78', 10_000_000)])
def test_full_size_run_properties(require_gpu, name, n_sims):
    case = O.load_case(name)
    hist, probs, _ = product_run(case, n_sims, 42)
    _check_latin(hist, n_sims)
    # checksum of checksums: the same run as 3 ragged pieces, bit-identical histogram
    parts = [0, n_sims // 3, n_sims // 3 + n_sims // 4 + 1, n_sims]
    acc = np.zeros_like(hist)
    for a, b in zip(parts[:-1], parts[1:]):
        acc += product_run(case, b - a, 42, sim_offset=a)[0]
    assert np.array_equal(acc, hist)
    # spot-check a slice deep inside the range against the oracle
    off = n_sims - 4096
    _, _, orders = product_run(case, 4096, 42, sim_offset=off, orders=True)
    ref = O.Problem(case).run(4096, rng=O.RNG_PHILOX, seed=42, sim_offset=off, want_orders=True)
    assert np.array_equal(orders, ref['orders'])


@pytest.mark.parametrize('name', ['S60', 'S78'])
def test_statistical_link_to_reference_mt(require_gpu, name):
    """Link 3: HIP-Philox vs the REFERENCE's own MT runs (tests/golden/ref_stat_*.npz, 2e5 / 1e5 sims).

    Tolerance: every histogram cell within 4.5 binomial standard errors of the reference sample
    (+1e-4 absolute for cells with a handful of counts); win probabilities within the same band.
    """
    ref = np.load(O.GOLDEN_DIR + f'/ref_stat_{name}.npz')
    n_ref = int(ref['n_each']) * len(ref['seeds'])
    p_ref = ref['hist'] / n_ref
    n_gpu = 20_000_000
    hist, probs, _ = product_run(O.load_case(name), n_gpu, 2025)
    p_gpu = hist / n_gpu
    se = np.sqrt(np.maximum(p_gpu * (1 - p_gpu), 1e-12) * (1 / n_ref + 1 / n_gpu))
    z = np.abs(p_gpu - p_ref) / (se + 1e-12)
    bad = np.argwhere(np.abs(p_gpu - p_ref) > 4.5 * se + 1e-4)
    assert bad.size == 0, f'{name}: cells {bad[:5].tolist()} off, max z {z.max():.2f}'
    # chi-square over all n*n cells with enough counts: a global shape check
    mask = p_ref * n_ref >= 50
    chi2 = float(np.sum(((p_ref - p_gpu) ** 2 / (se ** 2))[mask]))
    dof = int(mask.sum())
    assert chi2 < dof + 6 * np.sqrt(2 * dof), (chi2, dof)


def test_statistical_link_oracle_mt_large(require_gpu):
    """A tighter version of link 3 through the pinned oracle: 4e5 MT simulations on the CPU
    against 4e7 Philox simulations on the GPU, S60."""
    from concurrent.futures import ThreadPoolExecutor
    case = O.load_case('S60')

    def one(seed):
        return O.Problem(case).run(50_000, rng=O.RNG_MT, seed=seed)['hist']
    with ThreadPoolExecutor(8) as ex:          # ctypes releases the GIL
        h_mt = sum(ex.map(one, range(1000, 1008)))
    n_mt, n_gpu = 400_000, 40_000_000
    hist, _, _ = product_run(case, n_gpu, 99)
    p_mt, p_gpu = h_mt / n_mt, hist / n_gpu
    se = np.sqrt(np.maximum(p_gpu * (1 - p_gpu), 1e-12) * (1 / n_mt + 1 / n_gpu))
    assert np.all(np.abs(p_mt - p_gpu) <= 4.5 * se + 5e-5), float(np.max(np.abs(p_mt - p_gpu) / (se + 1e-12)))
    assert abs(p_mt[0, 0] - p_gpu[0, 0]) < 4.5 * se[0, 0]          # VER win probability


@pytest.mark.parametrize('name', ['S50', 'EVT', 'HET', 'DMP', 'N10'])
def test_statistical_link_other_cases(require_gpu, name):
    """Link 3 on the remaining golden configurations (event storm, heterogeneous 21-car field,
    rain, 10-car field, the set.pop()-sensitive 50-lap race): 2e5 oracle-MT simulations (8 seeds)
    against 1e7 HIP-Philox simulations, every cell within 4.5 SE + 1e-4."""
    from concurrent.futures import ThreadPoolExecutor
    case = O.load_case(name)

    def one(seed):
        return O.Problem(case).run(25_000, rng=O.RNG_MT, seed=seed)['hist']
    with ThreadPoolExecutor(8) as ex:
        h_mt = sum(ex.map(one, range(500, 508)))
    n_mt, n_gpu = 200_000, 10_000_000
    hist, _, _ = product_run(case, n_gpu, 31337)
    _check_latin(hist, n_gpu)
    p_mt, p_gpu = h_mt / n_mt, hist / n_gpu
    se = np.sqrt(np.maximum(p_gpu * (1 - p_gpu), 1e-12) * (1 / n_mt + 1 / n_gpu))
    worst = float(np.max(np.abs(p_mt - p_gpu) / (se + 1e-12)))
    assert np.all(np.abs(p_mt - p_gpu) <= 4.5 * se + 1e-4), (name, worst)

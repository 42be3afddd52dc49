#!/usr/bin/env python3
"""What the product's 32-bit deviates change, measured ON THE GPU by common random numbers, at 10^9 simulations.

The library's two modes run the same simulation ids under the same seed: deviates = 32 (uniforms w / 2^32, normals from a
binary32 cubic table: the product's fast path) and deviates = 53 (the reference's width: 53-bit uniforms, binary64
normals -- reference src/simulation.py:137,194,302,330,524 --, every draw keeping the 32-bit mode's word as its leading
bits).  A simulation that finishes in the same order under both is unaffected by the substitution; the rest bound the
change of the histogram.  Both modes are bit-identical to the CPU oracle's PHILOX / PHILOX53 back-ends
(tests/test_gpu_parity.py), which tools/deviate_bias.py compares on the host cores at 10^7.

    python tools/deviate_bias_gpu.py [--sims 1000000000] [--cases S60 S78] [--out profiles/r4_deviate_bias.txt]

Finishing orders stay on the device (mcgp_run_device into torch buffers, chunks of 10^7) and are compared there."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sims', type=int, default=1_000_000_000)
    ap.add_argument('--chunk', type=int, default=10_000_000)
    ap.add_argument('--cases', nargs='+', default=['S60', 'S78'])
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r4_deviate_bias.txt'))
    args = ap.parse_args()
    import torch
    import oracle_py as O
    from monte_carlo_gp_amd import RaceConfig, _native as N
    from monte_carlo_gp_amd.simulation import RaceSimulator, _Problem, _dptr
    lib = N.lib()
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream(dev)
    lines = ['# Effect of the 32-bit deviates on results, by common random numbers ON THE GPU (tools/deviate_bias_gpu.py):',
             '# library mode deviates = 32 (the product) vs deviates = 53 (53-bit uniforms, binary64 inverse normal CDF) on the',
             f'# same (seed, simulation id, lap, purpose, index) words; library build {N.build_hash()}.', '']
    for name in args.cases:
        case = O.load_case(name)
        drivers = list(case['grid_probs'])
        n = len(drivers)
        probs = {}
        for dv in (32, 53):
            probs[dv] = _Problem(RaceConfig(**case['config']), drivers, case['base_pace'], case['tire_deg'],
                                 case['driver_variance'], case['driver_dnf_rates'], case['track_condition'],
                                 O.load_cases()['set_pop'], dv)
        g = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
        hist = {dv: torch.zeros(n * n, dtype=torch.int64, device=dev) for dv in (32, 53)}
        orders = {dv: torch.empty(args.chunk * n, dtype=torch.uint8, device=dev) for dv in (32, 53)}
        differ = winner = 0
        ms = {32: 0.0, 53: 0.0}
        examples = []
        t0 = time.time()
        for off in range(0, args.sims, args.chunk):
            m = min(args.chunk, args.sims - off)
            for dv in (32, 53):
                p = probs[dv]
                N.check(lib.mcgp_run_device(C.byref(p.cfg), C.byref(p.drv), _dptr(g), n, m, off, args.seed, 0,
                                            C.c_void_p(stream.cuda_stream), C.c_void_p(hist[dv].data_ptr()),
                                            C.c_void_p(orders[dv].data_ptr())))
                t = C.c_float()
                N.check(lib.mcgp_last_kernel_ms(0, C.byref(t)))
                ms[dv] += t.value
            a = orders[32][:m * n].view(m, n)
            b = orders[53][:m * n].view(m, n)
            d = (a != b).any(dim=1)
            differ += int(d.sum())
            winner += int((a[:, 0] != b[:, 0]).sum())
            if len(examples) < 6 and bool(d.any()):
                for i in torch.nonzero(d)[:6 - len(examples)].flatten().tolist():
                    examples.append((off + i, int((a[i] != b[i]).sum())))
        h32 = hist[32].cpu().numpy().reshape(n, n)
        h53 = hist[53].cpu().numpy().reshape(n, n)
        N_ = args.sims
        delta = h53 - h32
        p = h32 / N_
        se = np.sqrt(np.maximum(p * (1 - p), 1e-300) / N_)
        win = delta[:, 0] / N_
        lines += [
            f'== {name}: {N_} simulations, seed {args.seed}, both modes on the same Philox words ({time.time() - t0:.0f} s; kernel time '
            f'{ms[32] / 1e3:.1f} s at 32 bits, {ms[53] / 1e3:.1f} s at 53: x{ms[53] / ms[32]:.2f})',
            f'simulations whose finishing order differs: {differ}  ({differ / N_:.3e} of all)',
            f'simulations whose WINNER differs:           {winner}  ({winner / N_:.3e})',
            f'histogram cells that differ: {int((delta != 0).sum())} of {n * n}; max |count delta| {int(np.abs(delta).max())} = '
            f'{np.abs(delta).max() / N_:.3e} in probability; sum |delta| / 2N = {np.abs(delta).sum() / 2 / N_:.3e}',
            f'win-probability delta per driver (53-bit minus 32-bit), max |.| = {np.abs(win).max():.3e}: '
            + ' '.join(f'{x:+.1e}' for x in win),
            f'largest |delta| in units of the binomial standard error of a {N_}-simulation run: '
            f'{float(np.max(np.abs(delta / N_) / se)):.3f}',
            'first differing simulation ids (id, positions changed): ' + ', '.join(str(e) for e in examples), '']
        print('\n'.join(lines[-9:]), flush=True)
    with open(args.out, 'w') as f:
        f.write('\n'.join(lines))


if __name__ == '__main__':
    main()

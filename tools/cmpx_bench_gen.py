#!/usr/bin/env python3
"""Emit tools/cmpx_bodies.inc for tools/cmpx_bench.hip: the loop bodies of the comparator microbenchmark on fixed registers
(slot i: time in v[10 + 2 i : 11 + 2 i], payload in v(50 + i)).  Not product code."""
def t(i): return f'v[{10 + 2 * i}:{11 + 2 * i}]'
def lo(i): return f'v{10 + 2 * i}'
def hi(i): return f'v{11 + 2 * i}'
def p(i): return f'v{50 + i}'

def comparator(form, a, b):
    if form == 'A':      # the kernel's form
        return [f'v_cmp_gt_f64 vcc, {t(a)}, {t(b)}', f'v_min_f64 v[72:73], {t(a)}, {t(b)}', f'v_max_f64 {t(b)}, {t(a)}, {t(b)}',
                f'v_cndmask_b32 v74, {p(a)}, {p(b)}, vcc', f'v_cndmask_b32 {p(b)}, {p(b)}, {p(a)}, vcc',
                # (the product writes the minimum in place; here two moves stand in for the renaming the compiler does for free:
                #  they are counted out below by the variant A0, which has them without the comparator)
                f'v_mov_b32 {lo(a)}, v72', f'v_mov_b32 {hi(a)}, v73', f'v_mov_b32 {p(a)}, v74']
    if form == 'A0':
        return [f'v_mov_b32 {lo(a)}, v72', f'v_mov_b32 {hi(a)}, v73', f'v_mov_b32 {p(a)}, v74']
    if form == 'B':
        return [f'v_cmpx_gt_f64 vcc, {t(a)}, {t(b)}', f'v_swap_b32 {lo(a)}, {lo(b)}', f'v_swap_b32 {hi(a)}, {hi(b)}',
                f'v_swap_b32 {p(a)}, {p(b)}', 's_mov_b64 exec, s[20:21]']
    if form == 'C':
        return [f'v_cmp_gt_f64 vcc, {t(a)}, {t(b)}', 's_and_b64 exec, s[20:21], vcc', f'v_swap_b32 {lo(a)}, {lo(b)}',
                f'v_swap_b32 {hi(a)}, {hi(b)}', f'v_swap_b32 {p(a)}, {p(b)}', 's_mov_b64 exec, s[20:21]']
    if form == 'D':      # compare + min/max under the full mask, payload by swap under EXEC
        return [f'v_cmpx_gt_f64 vcc, {t(a)}, {t(b)}', f'v_swap_b32 {p(a)}, {p(b)}', 's_mov_b64 exec, s[20:21]',
                f'v_min_f64 v[72:73], {t(a)}, {t(b)}', f'v_max_f64 {t(b)}, {t(a)}, {t(b)}', f'v_mov_b32 {lo(a)}, v72', f'v_mov_b32 {hi(a)}, v73']
    raise ValueError(form)

clob = ','.join(f'"v{i}"' for i in list(range(10, 42)) + list(range(50, 66)) + list(range(70, 76))) + ',"vcc","s20","s21","s22","memory"'
for form in ('A', 'A0', 'B', 'C', 'D'):
    for shape in ('layer', 'chain'):
        ins = ['v_mbcnt_lo_u32_b32 v70, -1, 0', 'v_mbcnt_hi_u32_b32 v70, -1, v70', 'v_mov_b32 v72, 0', 'v_mov_b32 v73, 0', 'v_mov_b32 v74, 0']
        for i in range(16):
            ins += [f'v_mul_u32_u24 v71, 13, v70', f'v_add_u32 v71, {i * 7 % 64}, v71', 'v_and_b32 v71, 31, v71', f'v_cvt_f64_u32 {t(i)}, v71', f'v_mov_b32 {p(i)}, {i}']
        ins += ['s_mov_b64 s[20:21], exec', 's_mov_b32 s22, %0', '1:']
        pairs = [(i, i + 1) for i in range(15)] if shape == 'chain' else [(i, i + 8) for i in range(8)] + [(2 * i, 2 * i + 1) for i in range(8)]
        for a, b in pairs:
            ins += comparator(form, a, b)
        ins += [f'v_add_f64 {t(0)}, {t(0)}, 4.0', f'v_add_f64 {t(9)}, {t(9)}, -2.0', 's_sub_u32 s22, s22, 1', 's_cmp_lg_u32 s22, 0', 's_cbranch_scc1 1b']
        ins += ['v_mov_b32 v75, v50'] + [f'v_mad_u32_u24 v75, v75, 31, {p(i)}' for i in range(1, 16)]
        ins += ['v_cvt_f64_u32 v[72:73], v75', f'v_add_f64 v[72:73], v[72:73], {t(3)}', f'v_add_f64 v[72:73], v[72:73], {t(12)}',
                'v_lshlrev_b32 v70, 3, v70', 'global_store_dwordx2 v70, v[72:73], %1', 's_waitcnt vmcnt(0)']
        print(f'__global__ void __launch_bounds__(256) k_{form}_{shape}(uint32_t iters, double *out)\n{{\n    asm volatile(')
        for x in ins:
            print(f'        "{x}\\n\\t"')
        print(f'        : : "s"(iters), "s"(out) : {clob});\n}}')

#!/bin/bash
cd "$(dirname "$0")/.."
for v in n32_d3x n32_d1x n32_d0x; do echo "== $v"; MCGP_LIB=$PWD/abl/libmcgp_$v.so python tools/dbg_wide_n.py 32 2>&1 | tail -1; done
for n in 31 30; do echo "== n${n}_d3x"; MCGP_LIB=$PWD/abl/libmcgp_n${n}_d3x.so python tools/dbg_wide_n.py $n 2>&1 | tail -1; done

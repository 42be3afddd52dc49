#!/bin/bash
cd "$(dirname "$0")/.."
for lib in abl/libmcgp_bcall0.so abl/libmcgp_bcall1.so abl/libmcgp_bcall0.so abl/libmcgp_bcall1.so; do MCGP_LIB=$PWD/$lib python tools/batch_time.py 10000 100000 1000000 2>&1 | tail -3; done

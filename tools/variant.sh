#!/bin/bash
# Build a named diagnostic variant of the library (field sizes in SIZES, default 20) with extra compiler flags:
#   tools/variant.sh <name> "<flags>"      -> abl/libmcgp_<name>.so      (abl/ is git-ignored but travels with gpurun)
# Compare on one GPU box with tools/ab.sh [workload] abl/libmcgp_a.so abl/libmcgp_b.so ...
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
sizes=${SIZES:-20}
only=""
[ "$sizes" = "20" ] && only="-DMCGP_ONLY_N20"
mkdir -p abl
root=$PWD
( cd monte_carlo_gp_amd/csrc && make -s -j8 BUILD=/tmp/abl_$name OUT=$root/abl/libmcgp_$name.so REG_SIZES="$sizes" \
    HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $only $flags" \
    $root/abl/libmcgp_$name.so )
ls -la abl/libmcgp_$name.so

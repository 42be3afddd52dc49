#!/bin/bash
# Register / scratch / LDS footprint of every register instantiation (compile only):  tools/resources.sh [extra flags]
cd "$(dirname "$0")/../monte_carlo_gp_amd/csrc"
one() {
  n=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DMCGP_INST_N=$n "$@" -S --cuda-device-only \
    -o /tmp/res_$n.s reg_inst.hip -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk -v n=$n '/ VGPRs:/{v=$(NF-1)} /ScratchSize/{s=$(NF-1)} /VGPRs Spill/{vs=$(NF-1)} /SGPRs Spill/{ss=$(NF-1)} /Occupancy/{o=$(NF-1)} END{printf "N=%2d vgpr=%3d scratch=%4d vgpr_spill=%3d sgpr_spill=%3d occupancy=%d\n", n, v, s, vs, ss, o}'
}
export -f one
seq 1 32 | xargs -P 8 -I{} bash -c "one {} $*" | sort -t= -k2 -n

#!/bin/bash
cd "$(dirname "$0")/.."
export DEV=53 SIMS=4000000
specs=()
for pair in "N10 10" "N16 16" "N18 18" "N19 19" "S60 20" "S78 20" "HET 21" "N22 22"; do set -- $pair; specs+=("$1 abl/libmcgp_wn$2_l0.so abl/libmcgp_wn$2_l1.so"); done
tools/ab_multi.sh "${specs[@]}"

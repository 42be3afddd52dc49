#!/bin/bash
cd "$(dirname "$0")/.."
tools/ab_multi.sh "S60 abl/libmcgp_dlean0.so abl/libmcgp_dlean1.so" "S78 abl/libmcgp_dlean0.so abl/libmcgp_dlean1.so"

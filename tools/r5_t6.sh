#!/bin/bash
cd "$(dirname "$0")/.."
DEV=53 tools/ab.sh S60 abl/libmcgp_wnow.so abl/libmcgp_wlean2.so abl/libmcgp_wlean3.so abl/libmcgp_wlean3b.so

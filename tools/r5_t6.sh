#!/bin/bash
cd "$(dirname "$0")/.."
DEV=53 tools/ab.sh S60 abl/libmcgp_wbase.so abl/libmcgp_wcall2.so abl/libmcgp_wcall3.so

#!/bin/bash
cd "$(dirname "$0")/.."
SFX=_wide tools/profile.sh r5 S60 --deviates 53

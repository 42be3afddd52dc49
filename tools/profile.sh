#!/bin/bash
# rocprofv3 passes for one workload of bench.py on the GPU box (run through gpurun); raw output goes to
# gpurun_out/<tag>_<workload>_<pass>/, tools/summarize_profiles.py turns it into the tracked files under profiles/.
#   tools/profile.sh r4 S60 [extra bench.py flags, e.g. --orders]
# Passes (counters never share a run with tracing; FETCH_SIZE / WRITE_SIZE / SQ sets in separate runs as
# MI355X_MICROARCH.md prescribes; the profiled program is python3 itself, directly after `--`):
#   stats  --kernel-trace --stats                    kernel durations
#   sq1    SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
#   sq2    SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU
#   grbm   GRBM_GUI_ACTIVE
#   fetch  FETCH_SIZE          write  WRITE_SIZE
set -e
cd "$(dirname "$0")/.."
tag=$1; wl=$2; shift 2
extra="$*"
sfx=${SFX:-}
export TMPDIR=/tmp
mkdir -p gpurun_out
args="bench.py --workload $wl --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-extras $extra"
run() {   # name, rocprofv3 options...
  name=$1; shift
  d=gpurun_out/${tag}_${wl}${sfx}_$name
  rm -rf $d
  rocprofv3 "$@" --output-format csv -d $d -- python3 $args > $d.json 2> $d.log || { echo "pass $name failed"; tail -5 $d.log; exit 1; }
  echo "pass $name done"
}
# build (or find up to date) both libraries BEFORE any profiled pass: under rocprofv3 the GPU is initialised before
# python starts, and a GPU-initialised process must not fork + exec make
python3 -c "import sys; sys.path.insert(0, 'tests')
from monte_carlo_gp_amd import _native as N; import oracle_py as O
N.build(); O.build(); print(N.build_hash())" > gpurun_out/${tag}_${wl}${sfx}_hash.txt
run stats --kernel-trace --stats
run sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU
run grbm --pmc GRBM_GUI_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE

#!/bin/bash
# round 5, first GPU call: profile the reference-width kernel + the counter list of this box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
(rocprofv3 --list-avail > gpurun_out/r5_avail.txt 2>&1 || rocprofv3-avail list > gpurun_out/r5_avail.txt 2>&1) ; echo "avail: $(wc -l < gpurun_out/r5_avail.txt) lines"
SFX=_wide tools/profile.sh r5 S60 --deviates 53 || exit 1
args="bench.py --workload S60 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --deviates 53"
for set in "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  name=cache_$(echo $set | cut -d' ' -f1)
  d=gpurun_out/r5_S60_wide_$name
  rm -rf $d
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 $args > $d.json 2> $d.log && echo "pass $name done" || { echo "pass $name failed"; tail -3 $d.log; }
done
# the default kernel on the same box, for the ratio
python3 bench.py --workload S60 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r5_first_default.json 2> gpurun_out/r5_first_default.log
python3 -c "import json; d=json.load(open('gpurun_out/r5_first_default.json')); print('default kernel ms', d['roofline']['kernel_ms_avg'])"
python3 -c "import json; d=json.loads(open('gpurun_out/r5_S60_wide_stats.json').read().strip().splitlines()[-1]); print('wide kernel ms', d['roofline']['kernel_ms_avg'], d['roofline']['kernel'])"

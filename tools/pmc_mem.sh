#!/bin/bash
# Vector-memory pipeline counters of one BASELINE config (is the walker bound by divergent address processing in the TA / L1,
# by L2 / fabric latency, or by VALU?).  Counters only, one small group per pass, program directly after `--`.
# usage: tools/pmc_mem.sh <cfg4|cfg5> <out-dir-under-gpurun_out> [ENV=VALUE ...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
CFG=$1; O=gpurun_out/$2; shift 2
for kv in "$@"; do export "$kv"; done
mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -oE "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TD_[A-Z0-9_]+|SQ_[A-Z0-9_]*(VMEM|WAIT|INST_LEVEL|IFETCH|BUSY)[A-Z0-9_]*)\b" $O/avail.txt | sort -u > $O/avail_mem_counters.txt
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_GATE_EN1_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/m$i -- python3 tools/perf_configs.py $CFG > $O/m$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 $O/m$i.log; }
done
python3 tools/pmc_summary.py $O/mem_summary.json $(find $O/m* -name "*counter_collection.csv") > $O/mem_summary.txt 2>&1
rm -rf $O/m1 $O/m2 $O/m3 $O/m4 $O/m5 $O/m6 $O/m7
head -12 $O/mem_summary.txt | cut -c1-1500

#!/bin/bash
# PMC passes (counters only, no trace domains) + kernel stats of one BASELINE config on the GPU box.
# usage: tools/pmc_cfg.sh <cfg4|cfg5> <out-dir-under-gpurun_out> [ENV=VALUE ...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
CFG=$1; O=gpurun_out/$2; shift 2
for kv in "$@"; do export "$kv"; done
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/perf_configs.py $CFG > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; }
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc1 -- python3 tools/perf_configs.py $CFG > $O/pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $O/pmc1.log; }
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc2 -- python3 tools/perf_configs.py $CFG > $O/pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $O/pmc2.log; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc3 -- python3 tools/perf_configs.py $CFG > $O/pmc3.log 2>&1 || { echo pmc3 failed; tail -5 $O/pmc3.log; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- python3 tools/perf_configs.py $CFG > $O/pmc4.log 2>&1 || { echo pmc4 failed; tail -5 $O/pmc4.log; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5 -- python3 tools/perf_configs.py $CFG > $O/pmc5.log 2>&1 || { echo pmc5 failed; tail -5 $O/pmc5.log; }
python3 tools/pmc_summary.py $O/pmc_summary.json $(find $O/pmc* -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/stats   # raw per-dispatch tables are large; the summaries stay
head -8 $O/kernel_stats.csv | cut -c1-160; head -6 $O/pmc_summary.txt | cut -c1-1100

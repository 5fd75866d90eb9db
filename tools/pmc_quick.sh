#!/bin/bash
# Two counter passes (issue / lane utilisation / LDS) of one config: a quick look at what a kernel change did.
# usage: tools/pmc_quick.sh <cfg4|cfg5|feature scene> <out-dir-under-gpurun_out> [ENV=VALUE ...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
CFG=$1; O=gpurun_out/$2; shift 2
for kv in "$@"; do export "$kv"; done
mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc1 -- python3 tools/perf_configs.py $CFG > $O/pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $O/pmc1.log; }
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc2 -- python3 tools/perf_configs.py $CFG > $O/pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $O/pmc2.log; }
python3 tools/pmc_summary.py $O/pmc_summary.json $(find $O/pmc* -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc1 $O/pmc2
head -12 $O/pmc_summary.txt | cut -c1-1100

#!/bin/bash
# usage (GPU box): bash tools/shade_coherence_probe.sh <out-dir-under-gpurun_out>
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-shade_probe}
mkdir -p $O
for v in as_is one_material no_delta all_spheres; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p_$v -- python3 tools/shade_coherence_probe.py $v > $O/$v.log 2>&1 || { echo "$v failed"; tail -3 $O/$v.log; }
  python3 tools/pmc_summary.py $O/$v.json $(find $O/p_$v -name "*counter_collection.csv") | grep "k_shade" | cut -c1-330 > $O/$v.txt
  rm -rf $O/p_$v
  echo "== $v: $(tail -1 $O/$v.log)"; cat $O/$v.txt
done

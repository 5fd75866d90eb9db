#!/bin/bash
# Where does k_shade<1>'s time go?  Measurement-only builds of libspt_hip.so with one region of the shade kernel
# compiled out (the films are WRONG; only the bounce-0 launch is comparable, its input does not depend on the region):
#   tools/shade_attribution.sh build     here (no GPU): three extra library directories lib_exp{0,A,B}
#   gpurun -- bash tools/shade_attribution.sh run     on the GPU box: cfg4 with each of them
set -e
cd "$(dirname "$0")/.."
BASE="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Iinclude"
if [ "$1" = build ]; then
  for v in "0:" "A:-DSPT_EXP_NO_LIGHT" "B:-DSPT_EXP_CHEAP_SAMPLE" "C:-DSPT_EXP_NO_LIGHT -DSPT_EXP_CHEAP_SAMPLE"; do
    tag=${v%%:*}; flags=${v#*:}
    make -s -j8 hip-plain LIBDIR=simple-path-tracer_amd/lib_exp$tag OBJDIR=build/hip_exp$tag HIPFLAGS="$BASE $flags" 2>&1 | grep -v shadows || true
    cp simple-path-tracer_amd/lib/libspt_host.so simple-path-tracer_amd/lib/libspt_hip_bez.so simple-path-tracer_amd/lib_exp$tag/
  done
elif [ "$1" = pmc ]; then
  # lane attribution (round 3): VALU instructions and lane utilisation of the shade kernels per variant (counters only, the
  # program directly after `--`; SPT_LIB_DIR is read by the binding inside the process)
  export TMPDIR=/tmp
  O=gpurun_out/${2:-shade_attr}
  mkdir -p $O
  for tag in 0 A B C; do
    export SPT_LIB_DIR=$PWD/simple-path-tracer_amd/lib_exp$tag
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/v$tag -- python3 tools/perf_configs.py cfg4 > $O/v$tag.log 2>&1 || { echo "variant $tag failed"; tail -3 $O/v$tag.log; }
    python3 tools/pmc_summary.py $O/variant_$tag.json $(find $O/v$tag -name "*counter_collection.csv") | grep "k_shade" > $O/variant_$tag.txt
    rm -rf $O/v$tag
    echo "== variant $tag"; cut -c1-400 $O/variant_$tag.txt
  done
else
  for tag in 0 A B C; do
    echo "== variant $tag"
    SPT_LIB_DIR=$PWD/simple-path-tracer_amd/lib_exp$tag timeout -k 10 300 python3 tools/perf_configs.py cfg4 2>&1 | grep -o '"ms": [0-9.]*\|"kernel_ms": {[^}]*}'
  done
fi

#!/bin/bash
# Runs the GPU parity suite under every A/B switch of libspt_hip.so ("Debug / A-B switches"): each switch
# changes which kernels run, none may change a film.  On the GPU box: gpurun -- bash tools/switch_sweep.sh
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for sw in SPT_NO_FUSED SPT_NO_LDS_TABLES SPT_NO_LDS_GEO SPT_NO_PIXEL_CULL SPT_NO_OVERLAP SPT_NO_DYN_SHADOW SPT_NO_DYN_EXTEND "SPT_PRIMARY_CHUNKS=1" "SPT_PRIMARY_CHUNKS=5" "SPT_BOX_BAND_BYTES=200000" "SPT_BVH_MAX_LEAF=2"; do
  case "$sw" in *=*) assign="$sw";; *) assign="$sw=1";; esac
  if env "$assign" timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "not full_size" > gpurun_out/sweep.log 2>&1; then
    echo "$assign: $(tail -1 gpurun_out/sweep.log)"
  else
    echo "$assign: FAILED"; tail -15 gpurun_out/sweep.log; exit 1
  fi
done

# How the film's D2H copy is issued decides whether the next frame's kernels run beside it (kernel trace: the runtime's blit copy
# kernel and the k_primary that meets it serialise: 580 -> 798 us).  Runtime switches that change the copy path, on the bench line.
set -e
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/ab_copy_$name.json 2> gpurun_out/ab_copy_$name.err || echo "$name failed"; }
for i in 1 2; do
  run base_$i SPT_DUMMY=0
  run sdma_$i HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0
  for n in 1 4 8 16 32; do run wg${n}_$i DEBUG_CLR_LIMIT_BLIT_WG=$n; done
done

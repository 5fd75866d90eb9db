# A/B of the samples rendered per wavefront pass on the headline workload (default: 128 M samples per pass = 128 spp at 1024^2)
set -e
for i in 1 2; do
  for s in 128 64 32 16 128 64; do
    timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-secondary --no-cpu-baseline --samples-per-pass $s > gpurun_out/ab_pass${s}_$i.json 2> gpurun_out/ab_pass${s}_$i.err
  done
done

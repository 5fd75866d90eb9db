#!/bin/bash
# Round-end measurement refresh on the GPU box (run through gpurun from the repo root): bench line,
# rocprofv3 kernel stats of the same command, PMC traffic / instruction counters (few-dispatch runs,
# one counter group per pass, no trace domains mixed in), 2-rank rehearsal of the N > 1 path.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/refresh
mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
echo "bench: $(python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'])")"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 5 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --profile-steps 0 --no-cpu-baseline --no-secondary > $O/pmc_$c.log 2>&1 || { echo pmc $c failed; tail -5 $O/pmc_$c.log; exit 1; }
done
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python bench.py --steps 1 --warmup 0 --profile-steps 0 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1 || { echo pmc sq failed; tail -5 $O/pmc_sq.log; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --single-device --no-cpu-baseline --no-secondary --profile-steps 0 > $O/bench_2rank_one_gpu.json 2> $O/bench_2rank.err || { echo 2rank failed; tail -8 $O/bench_2rank.err; }
echo "2rank: $(tail -1 $O/bench_2rank_one_gpu.json | cut -c1-200)"
find $O -name "*.csv" | head -20

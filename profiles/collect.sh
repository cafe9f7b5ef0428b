#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash profiles/collect.sh'): rocprofv3 evidence for bench.py's numbers.
#   1. --kernel-trace --stats of the default bench command (fp32) and of --dtype bf16, each also with --no-overlap
#      (wgrad on the main stream: kernels run one at a time, so per-kernel averages are not stretched by co-running)
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) of a short run
# Raw output goes to gpurun_out/prof/ (scratch); profiles/summarize.py turns it into the committed summaries.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$dt -o s -- python3 $R/bench.py --steps 10 --warmup 3 --dtype $dt --no-secondary > $O/bench_$dt.json 2> $O/bench_$dt.err
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$dt -o s -- python3 $R/bench.py --steps 10 --warmup 3 --dtype $dt --no-secondary --no-cpu-baseline --no-overlap > $O/serial_$dt.json 2> $O/serial_$dt.err
  echo "stats $dt done"
done
for dt in fp32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_$dt -o p -- python3 $R/bench.py --steps 1 --warmup 1 --dtype $dt --no-secondary --no-cpu-baseline > $O/pmc_${c}_$dt.json 2> $O/pmc_${c}_$dt.err
    echo "pmc $c $dt done"
  done
done
# the other single-GPU configurations of BASELINE.json (configs[3] EfficientNet-B0 B=512, configs[4] DeepLabv3+ 513x513 B=16): serial kernel
# table + the two PMC passes, so that their bench lines carry a roofline object with measured traffic as the headline does
for m in efficientnet_b0 deeplabv3plus; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_${m}_bf16 -o s -- python3 $R/bench.py --model $m --dtype bf16 --steps 10 --warmup 3 --no-overlap > $O/serial_${m}_bf16.json 2> $O/serial_${m}_bf16.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_${m}_bf16 -o p -- python3 $R/bench.py --model $m --dtype bf16 --steps 1 --warmup 1 > $O/pmc_${c}_${m}_bf16.json 2> $O/pmc_${c}_${m}_bf16.err
  done
  echo "$m done"
done
cd $R && python3 profiles/summarize.py $O ${ROUND:-round5}

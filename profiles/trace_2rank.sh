#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash profiles/trace_2rank.sh'): rocprofv3 kernel + memory-copy trace of the N = 2 bench path.
# Two ranks share the one GPU of the box (MCN_BENCH_DEVICE=0) and all-reduce over gloo — the data-parallel code path of
# `bench.py --gpus 2` (bucketed all-reduce enqueued from the backward launch list on the wgrad stream) with the host doing
# the reduction; each rank is its own `rocprofv3 -- python3 bench.py` (no launcher between the profiler and the program).
# Raw traces: gpurun_out/trace2/r{0,1}; profiles/overlap_summary.py turns them into profiles/<round>_overlap_2rank.json.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/trace2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MCN_BENCH_DEVICE=0 MCN_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2
DT=${DT:-bf16}
ARGS="--gpus 2 --steps 4 --warmup 2 --dtype $DT --no-secondary --no-cpu-baseline"
RANK=0 LOCAL_RANK=0 timeout -k 10 420 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r0 -o t -- python3 $R/bench.py $ARGS > $O/r0.json 2> $O/r0.err &
P0=$!
RANK=1 LOCAL_RANK=1 timeout -k 10 420 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r1 -o t -- python3 $R/bench.py $ARGS > $O/r1.json 2> $O/r1.err
wait $P0
cat $O/r0.json
cd $R && python3 profiles/overlap_summary.py $O ${ROUND:-round2}

#!/bin/bash
# rocprofv3 passes for the bench (run on the GPU box through gpurun; build the libraries first).
# Kernel trace + stats in one run; PMC counters each in their own run (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes; never combined with sys/runtime tracing).
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export DGMI_SKIP_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-variants > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $OUT/pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $OUT/pmc_l2.err
echo "l2 done"
find $OUT -name "*.csv" | head -40

#!/bin/bash
# rocprofv3 passes for round 4 (run on the GPU box through gpurun; build the libraries first).
# (1) bench.py: kernel trace + stats in one run, PMC counters each in their own run (separate --pmc passes, as
#     MI355X_MICROARCH.md prescribes; never combined with sys/runtime tracing; the program itself after `--`).
# (2) the edge-dropped step (tools/dropped_step_profile.py): the same passes for the products a training step runs —
#     compaction kernels + the plain kernels over the compacted layouts — and, for comparison, the on-the-fly KEEP kernels.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export DGMI_SKIP_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-variants > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo "trace done"
PASSES=("fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" "rd_dram:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "wr_dram:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum")
for pass in "${PASSES[@]}"; do
  name=${pass%%:*}
  ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $OUT/pmc_$name.err
  echo "$name done"
done
for form in 0 1; do
  D=$OUT/dropped$form
  mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $REPO/tools/dropped_step_profile.py 20 $form > $D/step.log 2> $D/trace.err
  for pass in "${PASSES[@]:0:3}"; do
    name=${pass%%:*}
    ctrs=${pass#*:}
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $D/pmc_$name -- python3 $REPO/tools/dropped_step_profile.py 3 $form > /dev/null 2> $D/pmc_$name.err
  done
  echo "dropped step form $form done: $(tail -1 $D/step.log)"
done
find $OUT -name "*.csv" | wc -l

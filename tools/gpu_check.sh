#!/bin/bash
# full GPU test suite (+ optionally the default bench line), logs under gpurun_out/check/
mkdir -p gpurun_out/check
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/check/gputest.log
cat gpurun_out/check/gputest.log
if [ "$1" = "bench" ]; then
  python bench.py > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err
  tail -c 6000 gpurun_out/check/bench.json
  tail -5 gpurun_out/check/bench.err
fi

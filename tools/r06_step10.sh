#!/bin/bash
# round 6 step 10: the full GPU suite on the final tree (node BATCH plane's per-launch events included) + the default bench line
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r06_s10_gputests.log 2>&1
python bench.py > gpurun_out/r06_s10_bench.json 2> gpurun_out/r06_s10_bench.err
tail -3 gpurun_out/r06_s10_gputests.log; cut -c1-600 gpurun_out/r06_s10_bench.json

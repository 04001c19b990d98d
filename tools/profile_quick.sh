#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats only of a short bench.py run.   tools/profile_quick.sh <tag> [bench args]
set -u
TAG=${1:-q}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
cd - > /dev/null
python3 tools/summarize_profile.py $OUT $OUT/summary | head -45

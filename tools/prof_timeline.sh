#!/bin/bash
# kernel timeline of one steady-state train step -> gpurun_out/step_timeline[_TAG].csv ; usage: prof_timeline.sh [tag]  (env passes through)
tag=${1:+_$1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pb; rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-detect --timing-steps 1 > /tmp/pb.log 2>&1
mkdir -p gpurun_out; python tools/step_timeline.py /tmp/pb/*/*kernel_trace.csv gpurun_out/step_timeline$tag.csv 1

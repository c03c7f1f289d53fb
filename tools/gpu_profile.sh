#!/bin/bash
# rocprofv3 kernel trace + stats of one bench step; summaries land in gpurun_out/prof_<tag>/
# usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
ls -R $out | head -30

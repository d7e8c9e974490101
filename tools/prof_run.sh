#!/bin/bash
# usage: tools/prof_run.sh <tag> <python script> [args...]  -> gpurun_out/prof_<tag>/<tag>_results.db (rocprofv3 kernel trace)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o $tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
echo "rocprofv3 exit $?"

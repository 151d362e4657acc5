#!/bin/bash
# One raw SQ-counter pass over a probe; prints per-kernel counter sums per call.
# usage: tools/pmc_raw.sh <tag> "<counters>" <script> [args...]
set -u
tag=$1; shift
counters=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcraw_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $counters -d $out/p --output-format csv -- python3 $script "$@" > $out/run.log 2>&1 || exit 1
PMC_RAW=1 python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/p > $GRAFT_REPO_ROOT/gpurun_out/pmcraw_$tag.md
rm -rf $out/p

#!/bin/bash
# Three counter passes (SQ / TCC read / TCC write) over one probe command; each pass is its own run.
# usage: tools/pmc_run.sh <tag> <python script and args...>
set -u
tag=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $out/sq --output-format csv -- python3 $script "$@" > $out/sq.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $out/rd --output-format csv -- python3 $script "$@" > $out/rd.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $out/wr --output-format csv -- python3 $script "$@" > $out/wr.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/sq $out/rd $out/wr > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.md
# raw per-dispatch CSVs are large: keep the summary only
rm -rf $out/sq $out/rd $out/wr

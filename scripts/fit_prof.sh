#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel table of 41 fit iterations.  usage: fit_prof.sh D B method tag
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fitprof_$4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/scripts/factor_prof.py $1 $2 $3 > $OUT/log.txt 2>&1
python3 $ROOT/scripts/prof_top.py $OUT 22

#!/bin/bash
# One GPU validation cycle of round 4 (runs ON the GPU box through gpurun): selected tests, the c4 A/B bench, c4 kernel tables.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
python -m pytest ${TESTS:-tests/test_gpu_bam.py tests/test_gpu_aux.py tests/test_gpu_fit.py tests/test_gpu_factor.py tests/test_gpu_dist.py tests/test_abi.py} -q -m gpu --maxfail=10 2>&1 | tail -${TAILN:-60} > $OUT/tests.log
python scripts/c4_update_bench.py > $OUT/c4_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/c4prof $OUT/c4prof_nsfuse
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4prof -- python3 $ROOT/scripts/c4_update_bench.py prof > $OUT/c4prof.log 2>&1
C4_NSFUSE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4prof_nsfuse -- python3 $ROOT/scripts/c4_update_bench.py prof > $OUT/c4prof_nsfuse.log 2>&1
cd $ROOT
python scripts/prof_top.py $OUT/c4prof 26 > $OUT/c4_kernels.txt 2>&1
python scripts/prof_top.py $OUT/c4prof_nsfuse 8 > $OUT/c4_kernels_nsfuse.txt 2>&1
cat $OUT/tests.log $OUT/c4_ab.txt $OUT/c4_kernels.txt $OUT/c4_kernels_nsfuse.txt

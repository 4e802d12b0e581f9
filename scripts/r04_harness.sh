#!/bin/bash
# chol64_blk harness variants on the GPU box: full output of each build (profiles/r04/chol64b_variants.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
: > $OUT/chol64b_variants.txt
for v in "" "-DCHOLB_TEST_REPLICA_DELAY=2" "-DCHOLB_TEST_FORCE_ORDER" "-DCHOLB_TEST_FORCE_ORDER -DCHOLB_TEST_OLD_WRITEBACK" "-DCHOLB_TEST_REPLICA_DELAY=2 -DCHOLB_TEST_OLD_WRITEBACK" "-DCHOLB_TEST_CORRUPT_REPLICA"; do
  echo "=== build flags: [$v]" >> $OUT/chol64b_variants.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I gsm-vi_amd/csrc scripts/chol64b_test.hip -o /tmp/cbt $v 2>/dev/null
  timeout 60 /tmp/cbt >> $OUT/chol64b_variants.txt 2>&1
  echo "exit code $?" >> $OUT/chol64b_variants.txt
done
cat $OUT/chol64b_variants.txt | grep -v "shader clock\|half [01]" | cut -c1-170

#!/bin/bash
# (re)build the diagnostic variant libraries of the chol64_blk race analysis (csrc/Makefile, target `variant`)
set -e
cd "$(dirname "$0")/../gsm-vi_amd/csrc"
make -j8 variant VARIANT=oldwb VFLAGS=-DCHOLB_TEST_OLD_WRITEBACK > /dev/null
make -j8 variant VARIANT=delay VFLAGS=-DCHOLB_TEST_REPLICA_DELAY=1 > /dev/null
make -j8 variant VARIANT=oldwb_delay VFLAGS="-DCHOLB_TEST_OLD_WRITEBACK -DCHOLB_TEST_REPLICA_DELAY=1" > /dev/null
ls -la ../libgsmvi_hip_*.so

#!/bin/bash
# SQ wait / LDS / VMEM counters of the bf16 step, three separate --pmc passes:  bash profiles/experiments/pmc_sq.sh <tag> [bench args]
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-probe --no-fwd-bwd-rate $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/p1 -- python3 $A > $out/p1.log 2>&1 && echo p1 ok
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $out/p2 -- python3 $A > $out/p2.log 2>&1 && echo p2 ok
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $out/p3 -- python3 $A > $out/p3.log 2>&1 && echo p3 ok
for p in p1 p2 p3; do python3 profiles/pmc_by_kernel.py $out/$p ${PMC_FILTER:-conv} > $out/$p.txt; done

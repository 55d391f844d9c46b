#!/bin/bash
# PMC passes over the split-bf16 forward kernel (LSNF_MATH=bf16x3); separate passes, no tracing domains beside --pmc.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LSNF_MATH=bf16x3
P=$GRAFT_REPO_ROOT/tools/run_fwd.py
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 $P 5 > $OUT/p1.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/p2 -- python3 $P 5 > $OUT/p2.log 2>&1 && \
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/p3 -- python3 $P 5 > $OUT/p3.log 2>&1
echo pmc_exit=$?

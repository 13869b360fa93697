#!/bin/bash
# PMC passes for one command (each pass its own rocprofv3 run: --pmc + --kernel-trace only).
# usage: tools/pmc.sh <outdir> <python script and args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$out/$n -- python3 "${CMD[@]}" > $R/gpurun_out/$out/$n.log 2>&1; }
mkdir -p $R/gpurun_out/$out
CMD=("$@")
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE
# p5 (round 6): do vector and matrix instructions co-execute?  (SQ_VALU_MFMA_COEXEC_CYCLES, with the two busy counters it is read against
# in the SAME pass; quad-cycle units except _MFMA_BUSY_, MI355X_MICROARCH.md) + the transcendental instruction counts
run p5 SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_FLOPS_FP16_TRANS SQ_INSTS_VALU_FLOPS_FP32_TRANS

#!/bin/bash
# training-step SQ counters per launch (MFMA-busy, LDS conflicts, waits)
T=gpurun_out/r04q; mkdir -p $T
timeout -k 10 1000 python tools/pmc_per_launch.py r04q train "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" > $T/pmc.log 2>&1; echo "pmc rc $?"; tail -5 $T/pmc.log

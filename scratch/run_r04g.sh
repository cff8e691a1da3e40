#!/bin/bash
# round 4: counter passes -- per-launch fabric traffic of the training step, TCP/TCC/LDS/SQ counters of the inference
# step for the 1x1 family (item 4)
T=gpurun_out/r04g; mkdir -p $T
timeout -k 10 900 python tools/pmc_per_launch.py r04g train "FETCH_SIZE" "WRITE_SIZE" > $T/train_traffic.log 2>&1; echo "train traffic rc $?"; tail -3 $T/train_traffic.log
timeout -k 10 1100 python tools/pmc_per_launch.py r04g infer "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" > $T/infer_counters.log 2>&1; echo "infer counters rc $?"; tail -3 $T/infer_counters.log
ls gpurun_out/pmc_launch_r04g_* 

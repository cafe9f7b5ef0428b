#!/bin/bash
# counter passes over the plain fp32 1x1 forward launches (profiles/probes/pmc_fwd_k64.py); 3-4 counters per pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_ANY" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SPI_RA_RES_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_CSN_BUSY SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -o p -- python3 $R/profiles/probes/pmc_fwd_k64.py > $O/pass$i.log 2>&1 && echo "pass $i ok"
done
cd $R && python3 profiles/probes/pmc_layers_summary.py $O > $O/summary.txt 2>&1; tail -5 $O/summary.txt | cut -c1-300

#!/bin/bash
# usage: tools/pmc_fwd_only.sh OUTDIR -- issue / wait counters of the forward-only tile kernel (three blocks per CU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
cd /tmp; export TMPDIR=/tmp
export DW_GRAD=0
mkdir -p $OUT
run() { timeout -s KILL 90 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/tools/run_tile_only.py 3 > $OUT/$1.log 2>&1; }
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run p2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run p4 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE"
python3 $R/tools/pmc_summary.py $OUT --kernel photo_tile

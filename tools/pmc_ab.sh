#!/bin/bash
# usage: tools/pmc_ab.sh OUTDIR LIB [LIB ...] -- two counter passes over the tile kernel for each library build (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
cd /tmp; export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so)
  export DEPTHWARP_LIB=$R/$so
  mkdir -p $OUT/$n
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/$n/p1 -- python3 $R/tools/run_tile_only.py 3 > $OUT/$n/p1.log 2>&1 || exit 1
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$n/p2 -- python3 $R/tools/run_tile_only.py 3 > $OUT/$n/p2.log 2>&1 || exit 1
  echo "== $n"
  python3 $R/tools/pmc_summary.py $OUT/$n --kernel photo_tile
done

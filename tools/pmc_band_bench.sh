#!/bin/bash
# usage: pmc_run.sh <tag> <binary>
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/$1; mkdir -p $out
bin=$root/$2
run() { # name counters
  (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $out/tmp_$1 -- $bin > $out/$1.log 2>&1)
  f=$(find $out/tmp_$1 -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp $f $out/$1.csv
  rm -rf $out/tmp_$1
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
run b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES"
run c "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
python $root/tools/pmc_summary.py $out/pmc.json $out/a.csv $out/b.csv $out/c.csv

#!/bin/bash
# Developer tool (GPU box): SQ counter passes over tools/kbench.py for one library build.
#   gpurun -- 'bash tools/sq_profile.sh OUTDIR [kbench args]'  then  python tools/sq_table.py gpurun_out/OUTDIR
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
O=gpurun_out/$1; shift
rm -rf $O; mkdir -p $O
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  n=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$n -- python3 tools/kbench.py --steps 3 "$@" > $O/$n.log 2>&1
done
ls $O

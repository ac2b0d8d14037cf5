#!/bin/bash
# Collects what profiles/ holds, on a GPU box, from the repo root:
#   gpurun -- 'bash tools/profile.sh r02'   then   python tools/summarize_pmc.py gpurun_out/prof_r02 r02
#                                                  python tools/summarize_sq.py  gpurun_out/prof_r02_sq r02
# Kernel stats and every PMC group are separate rocprofv3 runs (counters never together with other trace domains).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
[ -x build/exp/hbm_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o build/exp/hbm_calib tools/hbm_calib.hip
TAG=${1:-r02}
O=gpurun_out/prof_$TAG
B="--no-cpu-baseline --no-stirred --no-second-scaling --no-secondary"
mkdir -p $O
# 1. per-kernel durations of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 $B > $O.bench.log 2>&1
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in their own passes, plus the calibration copy (8-B lanes, known bytes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 2 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 2 $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- ./build/exp/hbm_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- ./build/exp/hbm_calib > /dev/null 2>&1
# 3. SQ counters, three per pass
S=gpurun_out/prof_${TAG}_sq
rm -rf $S; mkdir -p $S
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $S/$n -- python3 bench.py --steps 2 --warmup 1 $B > $S/$n.log 2>&1
done
# 4. the other laws of the path (K3 differs): kernel stats of tools/kbench.py, the program directly after --
for LAW in hencky dp; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$LAW -- python3 tools/kbench.py --law $LAW > $O.kbench_$LAW.log 2>&1
done
tail -1 $O.bench.log | cut -c1-300

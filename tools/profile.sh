#!/bin/bash
# Collects what profiles/ holds, on a GPU box, from the repo root:
#   gpurun -- 'bash tools/profile.sh r04 step residual twod dp'      (any subset of the parts; default: step)
# then here:  python tools/summarize_pmc.py gpurun_out/prof_r04 r04 [parts]   and   python tools/summarize_sq.py ... (see there)
# Kernel stats and every PMC group are separate rocprofv3 runs (counters never together with other trace domains); the
# profiled program stands directly after `--`.
#   step      bench.py (3-D, 1 M particles, Neo-Hookean explicit step): stats, FETCH_SIZE / WRITE_SIZE, five SQ groups
#   residual  bench.py --workload residual (nlps_gpu_lagrangian_evaluation, fused and separate stages, NH and D-P): stats,
#             FETCH_SIZE / WRITE_SIZE, two SQ groups
#   twod      bench.py --workload step2d (2-D, 1 M particles): stats, FETCH_SIZE / WRITE_SIZE, two SQ groups
#   dp        tools/kbench.py --law dp | hencky: stats, and the SQ groups for Drucker-Prager
#   tangent   bench.py --workload tangent: the plain run (-> profiles/<tag>_tangent_bench.jsonl) and kernel stats
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p build/exp
[ -x build/exp/hbm_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o build/exp/hbm_calib tools/hbm_calib.hip
TAG=${1:-r04}; shift
PARTS=${@:-step}
O=gpurun_out/prof_$TAG
mkdir -p $O
SQ5=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
SQ2=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM")
# the calibration copy (8-B lanes, known bytes): what FETCH_SIZE / WRITE_SIZE read for this code's access width on THIS box
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- ./build/exp/hbm_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- ./build/exp/hbm_calib > /dev/null 2>&1
pmc_passes() {  # name, sq-group array name, program + arguments ...
  local name=$1 groups=$2; shift 2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -- "$@" > /dev/null 2>&1
  local -n G=$groups
  for C in "${G[@]}"; do
    n=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${name}_sq/$n -- "$@" > $O/${name}_sq_$n.log 2>&1
  done
  echo "[profile] $name: counter passes done"
}
for PART in $PARTS; do
  case $PART in
  step)
    B="--no-cpu-baseline --no-stirred --no-second-scaling --no-secondary --no-implicit"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_stats -- python3 bench.py --steps 20 --warmup 5 $B > $O.step.log 2>&1
    pmc_passes step SQ5 python3 bench.py --steps 3 --warmup 2 $B
    tail -1 $O.step.log | cut -c1-200 ;;
  residual)
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/residual_stats -- python3 bench.py --workload residual --no-cpu-baseline --steps 10 --warmup 3 > $O.residual.log 2>&1
    pmc_passes residual SQ2 python3 bench.py --workload residual --no-cpu-baseline --steps 3 --warmup 1
    tail -2 $O.residual.log | cut -c1-200 ;;
  twod)
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/twod_stats -- python3 bench.py --workload step2d --steps 20 --warmup 5 > $O.twod.log 2>&1
    pmc_passes twod SQ2 python3 bench.py --workload step2d --steps 3 --warmup 2
    tail -1 $O.twod.log | cut -c1-200 ;;
  tangent)
    python3 bench.py --workload tangent > $O.tangent_bench.jsonl 2> /dev/null
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/tangent_stats -- python3 bench.py --workload tangent --no-cpu-baseline > $O.tangent.log 2>&1
    tail -1 $O.tangent_bench.jsonl | cut -c1-200 ;;
  dp)
    for LAW in hencky dp; do
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/kbench_${LAW}_stats -- python3 tools/kbench.py --law $LAW > $O.kbench_$LAW.log 2>&1
    done
    pmc_passes kbench_dp SQ5 python3 tools/kbench.py --law dp --steps 3
    tail -1 $O.kbench_dp.log ;;
  esac
done
ls $O

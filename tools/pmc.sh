#!/bin/bash
# PMC passes for bench.py (one rocprofv3 run per counter set; see MI355X guide: max 8 SQ counters per pass).
# usage: tools/pmc.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd $root
run() { n=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$n -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BARGS > gpurun_out/pmc_${tag}_$n.log 2>&1 || exit 1; }
BARGS="$*"
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run c GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
# matrix-pipe counters in a pass of their own (tolerated if a name is unknown to this rocprofv3)
( run d SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES ) || echo "pass d (MFMA counters) failed"
python3 tools/pmc_report.py $tag

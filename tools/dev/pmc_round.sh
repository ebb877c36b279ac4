#!/bin/bash
# PMC passes over tools/dev/prof_step.py (steady state, last 40 full-size dispatches), one counter group per pass.  The raw rocprofv3 output
# stays under /tmp on the box (gpurun merges at most 64 MiB of gpurun_out/ back); only the exported per-counter averages are kept.
# usage: pmc_round.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04}; W=/tmp/pmc_$T; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
pass() { tag=$1; shift; rocprofv3 --pmc "$@" -d $W/$tag -o p -- python3 $R/tools/dev/prof_step.py > $W/$tag.log 2>&1 && \
         python3 $R/tools/dev/rocpd_export.py pmc $W/$tag/p_results.db $O/${T}_pmc_$tag.txt || { echo "pass $tag failed"; tail -5 $W/$tag.log; return 1; }; }
pass fetch FETCH_SIZE && pass write WRITE_SIZE && \
pass issue SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES && \
pass issue2 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE && \
pass fp SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 ; \
pass lds SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE ; \
pass mem SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM ; \
pass fp2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F32 ; \
ls $O | grep ${T}_pmc

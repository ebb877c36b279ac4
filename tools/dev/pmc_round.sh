#!/bin/bash
# PMC passes over tools/dev/prof_step.py (steady state, last 40 full-size dispatches), one counter group per pass.
# usage: pmc_round.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/${T}_counters_list.txt 2>&1
pass() { tag=$1; shift; rocprofv3 --pmc "$@" -d $O/${T}_pmc_$tag -o p -- python3 $R/tools/dev/prof_step.py > $O/${T}_pmc_$tag.log 2>&1 && \
         python3 $R/tools/dev/rocpd_export.py pmc $O/${T}_pmc_$tag/p_results.db $O/${T}_pmc_$tag.txt; }
pass fetch FETCH_SIZE && pass write WRITE_SIZE && \
pass issue SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES && \
pass issue2 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE && \
pass fp SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 ; \
pass lds SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE ; \
pass mem SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM ; \
pass fp2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F32 ; \
grep -c . $O/${T}_counters_list.txt; grep -i "VALU" $O/${T}_counters_list.txt | head -40

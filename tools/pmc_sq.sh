#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): SQ-level PMC passes (wait / issue / LDS / MFMA counters) of one probe program, reduced to CSV.
#   tools/pmc_sq.sh <name> <python script> [args...]   ->  gpurun_out/pmc_sq/<name>_{sq1,sq2,mfma}.csv
# Counters in their own passes with --kernel-trace only (no --stats / sys-trace next to --pmc on this pool).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/pmc_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/ps1_$NAME -o p -- python3 "$@" > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/ps1_$NAME/p_results.db --pmc --top 0 --csv $OUT/${NAME}_sq1.csv
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM --kernel-trace -d /tmp/ps2_$NAME -o p -- python3 "$@" > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/ps2_$NAME/p_results.db --pmc --top 0 --csv $OUT/${NAME}_sq2.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/ps3_$NAME -o p -- python3 "$@" > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/ps3_$NAME/p_results.db --pmc --top 0 --csv $OUT/${NAME}_mfma.csv
ls $OUT

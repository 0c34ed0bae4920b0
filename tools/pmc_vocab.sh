set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_vocab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/pv1 -o p -- python3 $R/tools/vocab_probe.py > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/pv1/p_results.db --pmc --top 0 --csv $OUT/sq1.csv
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM --kernel-trace -d /tmp/pv2 -o p -- python3 $R/tools/vocab_probe.py > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/pv2/p_results.db --pmc --top 0 --csv $OUT/sq2.csv
ls $OUT

# SQ counters of the fp32 fused 10^3 Winograd kernel (k_conv_wf) inside DenseCPD-synth:  gpurun -- 'bash tools/jobs/wf_pmc.sh'
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); OUT=$ROOT/gpurun_out/wf/pmc; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/tools/plan_report.py --measure densecpd"
cd /tmp
run() { tag=$1; shift; TH_GUARD=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag -o $tag -- $CMD > $OUT/$tag.log 2>&1; DB=$(find $OUT/$tag -name '*.db' | head -1); [ -n "$DB" ] && python $ROOT/tools/pmc_raw.py --match "k_conv_wf<" $DB 2>&1 | head -2; }
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES
run b GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run d GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU
rm -rf $OUT/*/*.db $OUT/*/*/*.db

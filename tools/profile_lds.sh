cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_wf_lds; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/lds -o lds -- python $GRAFT_REPO_ROOT/tools/bench_layer.py 10 64 16 3 8192 > $OUT/lds.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py $(find $OUT -name '*.db' | head -1) 2>&1 | head -12
rm -rf $OUT/*/*.db $OUT/*/*/*.db

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_BRANCH"; do
rm -rf /tmp/p
timeout 240 rocprofv3 --kernel-trace --pmc $G -d /tmp/p -o o -- python $R/tools/_h5prof.py > /tmp/o.txt 2>&1
echo "rocprof rc=$?"
f=$(find /tmp/p -name "*.db" | head -1)
if [ -n "$f" ]; then python $R/tools/_pmc_inflate.py "$f" < /dev/null | cut -c1-400; else tail -5 /tmp/o.txt; fi
done

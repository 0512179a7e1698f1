cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/p -o o -- python $R/tools/_h5prof.py > /tmp/o.txt 2>&1
echo "rocprof rc=$?"
f=$(find /tmp/p -name "*.db" | head -1)
if [ -n "$f" ]; then (cd $R && python tools/rocpd_summary.py --kernel-trace "$f" < /dev/null | grep -E "k_lz|k_inflate|k_place" | cut -c1-200); else tail -5 /tmp/o.txt; fi

set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest $R/tests/test_gpu_inflate.py -x -q 2>&1 | grep -E "passed|failed|Error" | head -5
for S in 0 8 16 32 64; do
  export TH_LZ_SHORT=$S
  rocprofv3 --kernel-trace --stats -d /tmp/p$S -o o -- python $R/tools/bench_h5_decode.py 8192 4096 > /tmp/o$S.txt 2>&1 || true
  f=$(find /tmp/p$S -name "*kernel_stats.csv" | head -1)
  echo "SHORT=$S: $(grep -E 'k_lz_resolve|k_inflate_tokens' $f | cut -d, -f1-4 | tr '\n' ' ')"
done

# the bench value against the chunk (frames per launch):  gpurun -- 'bash tools/jobs/chunk_sweep.sh 2048 4096 6144 8192'
for c in "$@"; do
  echo "== chunk $c"
  timeout 300 python bench.py --chunk $c --no-cpu-baseline --no-profile --no-extras --no-pmc 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done

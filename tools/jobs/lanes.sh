# the bench value with one and two lanes (chunks in flight on two streams):  gpurun -- 'bash tools/jobs/lanes.sh'
for v in TH_LANES=1 TH_LANES=2 "TH_LANES=2 TH_LANE_LAG=3"; do
  echo "== $v"
  env $v timeout 300 python bench.py --no-cpu-baseline --no-profile --no-extras --no-pmc 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('knobs'))"
done

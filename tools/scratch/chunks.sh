for t in timed densecpd; do for c in 2048 4096 8192; do
  python bench.py --topology $t --chunk $c --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', $c, round(d['value']))"
done; done

for d in 0 1 64; do
  TH_CONV_DBG=$d python bench.py --topology densecpd --no-cpu-baseline --steps 1 --frames 40960 > gpurun_out/dbg_$d.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/dbg_$d.json').read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if x['label'].startswith('conv3d_2:') or x['label'].startswith('conv3d_11:')]
print($d, [(x['label'][:9], round(x['ms_total'],2)) for x in k])
PY
done

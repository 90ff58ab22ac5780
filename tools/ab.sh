#!/bin/bash
# A/B harness: runs bench.py once per library build given on the command line (RT_LIB=<path>)
for lib in "$@"; do
  RT_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-workloads --repeats 5 > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; continue; }
  python - "$lib" <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print(sys.argv[1], 'Mrays/s %.1f ms/spp %.2f'%(d['value'], d['ms_per_step']), {k:round(v,1) for k,v in d['kernel_ms'].items() if v>1},
      'occ ext %.3f con %.3f' % (d['roofline'].get('lane_occupancy', 0), d['connect'].get('lane_occupancy', 0)),
      'nodes/tris per ray %.1f %.1f' % (d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray']),
      'unshared ms', {k: round(v, 1) for k, v in (d.get('unshared') or {}).get('kernel_ms', {}).items() if v > 1}, 'nonfinite', d['config']['film_nonfinite_values'])
PY
done

#!/bin/bash
# first look at a new walk: structure + parity tests, then the traversal microbench and the frame bench
out=$PWD/gpurun_out/first; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_bvh_build_gpu.py tests/test_intersect_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1
rc=$?; tail -15 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/bench_intersect.py > $out/isect.log 2>&1 || { tail -5 $out/isect.log; exit 1; }
cat $out/isect.log
timeout -k 10 300 python -u bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/first/bench.json'))
print('Mrays/s %.1f ms/spp %.2f'%(d['value'], d['ms_per_step']), {k:round(v,1) for k,v in d['kernel_ms'].items() if v>1},
      'nodes/tris per ray %.1f %.1f' % (d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray']),
      'unshared ms', {k: round(v, 1) for k, v in (d.get('unshared') or {}).get('kernel_ms', {}).items() if v > 1}, 'nonfinite', d['config']['film_nonfinite_values'])
PY

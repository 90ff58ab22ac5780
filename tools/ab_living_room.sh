#!/bin/bash
for lib in "$@"; do
  RT_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --scene living-room --no-cpu-baseline --no-workloads --no-unshared-pass --repeats 5 > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; continue; }
  python -c "
import json,sys
d=json.load(open('/tmp/ab.json')); print('living-room', sys.argv[1], 'Mrays/s %.1f'%d['value'])" $lib
done

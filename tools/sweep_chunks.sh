#!/bin/bash
# decode / encode throughput vs sub-batch size on concurrent streams
for c in 0 12 9 6; do
  echo -n "chunk $c: "
  python bench.py --steps 12 --no-cpu-baseline --chunk $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['regions']; print(d['config']['launch'], 'decode', r['decode']['mpixels_per_s'], 'encode', r['encode']['mpixels_per_s'], 'e2e', r['encode_decode_score']['mpixels_per_s'])"
done

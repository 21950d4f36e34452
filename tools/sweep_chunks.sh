#!/bin/bash
# decode throughput vs sub-batch size on concurrent streams
for c in 0 9 6 3; do
  echo -n "chunk $c: "
  python bench.py --decode-only --steps 30 --chunk $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['launch'], d['value'], d['ms_per_step'])"
done

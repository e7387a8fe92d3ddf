#!/bin/bash
# experiment: Schur job granularity (SVI_SCHUR_DIV) vs phase times
for D in 1536 3072 6144 12288; do
  SVI_SCHUR_DIV=$D python bench.py --no-cpu-baseline --no-matcher --no-frontend 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['phases_ms_per_call']
print('div', $D, 'it/s %.1f' % d['value'], 'schur %.3f assemble %.3f' % (p['schur'], p['assemble']))"
done

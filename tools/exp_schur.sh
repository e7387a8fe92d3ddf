#!/bin/bash
for j in 7000 8192 3500; do
echo "== qj $j"; SVI_SCHUR_QJ=$j python bench.py --no-cpu-baseline --no-matcher --no-frontend 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_per_call']; print('%.1f it/s  schur %.1f assemble %.1f'%(d['value'], 1e3*p['schur'], 1e3*p['assemble']))"
done

#!/bin/bash
# bench value against warm-up steps / timed steps / lanes: tools/wk_sweep.sh "W K LANES" ...
for cfg in "$@"; do
  set -- $cfg
  python bench.py --warmup $1 --steps $2 --lanes $3 --no-cpu-baseline --no-host-rate 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('W', d['warmup'], 'K', d['steps'], 'lanes', d['config'].get('lanes', '$3'), d['value'], d['ms_per_step'])"
done

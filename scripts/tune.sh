#!/bin/bash
# usage: scripts/tune.sh "<env assignments>" <tile> -> one summary line
out=$(env $1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tile $2 2>/dev/null | grep '^{' | tail -1)
python3 - "$1" "$2" <<PY
import json,sys
d=json.loads('''$out''')
print(sys.argv[1].ljust(34), "tile", sys.argv[2].rjust(4), "op_us %.2f iter_us %.2f it/s %.0f elem/s %.3g iters %d"%(d['roofline']['us_per_launch'], d['phases_ms']['ms_cg']*1e3/d['cg_iterations'], d['cg_iters_per_sec'], d['value'], d['cg_iterations']))
PY

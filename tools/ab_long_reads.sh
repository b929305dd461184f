#!/bin/bash
# same-box A/B (tools/ab_cmd.sh) of the long-read sub-record: prints value / ms / golden count
python bench.py --sub-configs long_reads --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
s=d['sub_records']['configs[3]']
print(json.dumps({'gcups': s['value'], 'ms': s['ms'], 'golden': s['windows_equal_to_oracle_golden'], 'differ': s['windows_differing_from_golden']}))"

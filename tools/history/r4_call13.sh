#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('$1', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_repeats']['median'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if v}, d['config'].get('table_keys_touched'))
"; }
python bench.py --zipf 1.1 --no-cpu-baseline --signal-keys 0 --batches 8 --repeats 3 --no-fm-leg 2>/dev/null | line zipf_nosignal
python bench.py --zipf 1.1 --no-cpu-baseline --batches 8 --repeats 3 --no-fm-leg 2>/dev/null | line zipf_signal
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --keys-per-gpu 125000000 --capacity 64000000 --no-cpu-baseline --repeats 3 --batches 8 --signal-keys 0 2>/dev/null | line cfg4_nosignal
python bench.py --model fm --k 64 --optimizer ftrl --zipf 1.1 --keys-per-gpu 125000000 --capacity 64000000 --no-cpu-baseline --repeats 3 --batches 8 2>/dev/null | line cfg4_signal

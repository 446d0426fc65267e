#!/bin/bash
# round 6, call 4: where the first-epoch leg's milliseconds go — kernel + HIP API stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
for pct in 2 30; do
  rm -rf /tmp/ft
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d /tmp/ft -- \
    python "$GRAFT_REPO_ROOT/tools/r6/fresh_probe.py" 10000000 14 $pct > /tmp/ft.out 2> /tmp/ft.err)
  tail -1 /tmp/ft.out | cut -c1-700
  for kind in kernel_stats hip_api_stats; do
    f=$(find /tmp/ft -name "*${kind}.csv" | head -1)
    echo "== pct $pct $kind $f"
    [ -n "$f" ] && cp "$f" gpurun_out/r6/fresh_pct${pct}_${kind}.csv && python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    name = r.get("Name", "")
    m = re.search(r"k_\w+(<[^>]*>)?", name)
    print("%-52s calls %6s total %10.1f us avg %9.1f us %6s%%" % ((m.group(0) if m else name)[:52], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  done
done
tail -3 /tmp/ft.err

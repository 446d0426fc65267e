#!/bin/bash
# The reference's run_ps_local.sh, for this build: N workers = N GPUs of this node.
#   ./run_local.sh [num_workers] [model: 0 LR | 1 FM] [epochs]
root=$(cd "$(dirname "$0")" && pwd)
n=${1:-2}; model=${2:-0}; epochs=${3:-100}
(cd "$root" && python -m xflow_amd.build > /dev/null) || exit 1
# the reference ships three identical train shards (data/small_train-0000{0,1,2}); the repo
# keeps one copy as a test fixture
d=$(mktemp -d)
for ((r = 0; r < n; ++r)); do
    cp "$root/tests/golden/small_train-00000" "$d/small_train-$(printf %05d $r)"
done
cp "$root/tests/golden/small_test-00000" "$d/small_test-00000"
"$root/scripts/local.sh" "$n" "$n" "$root/xflow_amd/lib/xflow_lr" \
    "$d/small_train" "$d/small_test" "$model" "$epochs"
rc=$?
rm -rf "$d"
exit $rc

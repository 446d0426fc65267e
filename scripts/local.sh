#!/bin/bash
# Local multi-GPU run, with the interface of the reference's scripts/local.sh:
#
#   scripts/local.sh num_servers num_workers bin [args..]
#   e.g. scripts/local.sh 2 2 xflow_amd/lib/xflow_lr data/small_train data/small_test 0 10
#
# One worker process per GPU.  There are no server or scheduler processes to start: the
# "servers" are the key-range shards of the table in the workers' HBM, the "scheduler" is the
# worker that owns DMLC_PS_ROOT_PORT (rank 0).  num_servers is accepted and ignored.  The
# workers find each other through the same environment ps-lite uses: DMLC_NUM_WORKER,
# DMLC_PS_ROOT_URI, DMLC_PS_ROOT_PORT; ranks are handed out in arrival order, worker r trains
# on <train>-0000r and rank 0 scores <test>-00000 (lr_worker.cc:208-215).
if [ $# -lt 3 ]; then
    echo "usage: $0 num_servers num_workers bin [args..]"
    exit 2
fi
export DMLC_NUM_SERVER=$1; shift
export DMLC_NUM_WORKER=$1; shift
bin=$1; shift

export DMLC_PS_ROOT_URI=${DMLC_PS_ROOT_URI:-127.0.0.1}
export DMLC_PS_ROOT_PORT=${DMLC_PS_ROOT_PORT:-8000}
export DMLC_ROLE=worker
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}

pids=()
for ((i = 0; i < DMLC_NUM_WORKER; ++i)); do
    "$bin" "$@" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do
    wait "$p" || rc=$?
done
exit $rc

#!/bin/bash
# Runs GPU steps one after another on the gpurun box: `step <name> <timeout-seconds> <command...>`.
# An ordinary failure is recorded and the next step runs; a step that TIMES OUT (or is killed) ends the whole call --
# nothing further is started on a GPU that may be hung.
step() {
    local name=$1 limit=$2
    shift 2
    timeout -k 10 "$limit" "$@"
    local rc=$?
    echo "[step] $name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[step] $name timed out / was killed: stopping here"
        exit $rc
    fi
    return 0
}

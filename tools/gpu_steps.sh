#!/bin/bash
# Runs a list of GPU steps on the box.  A step that fails in the ordinary way (tests red, non-zero exit) does not
# stop the list; a step that TIMES OUT or is killed does (no further GPU step after a hang).
#   step <name> <timeout seconds> <command ...>     output -> gpurun_out/<name>.log
mkdir -p gpurun_out
step() {
    local name=$1 limit=$2
    shift 2
    echo "[step] $name: $*"
    timeout -k 10 "$limit" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "[step] $name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[step] $name timed out / was killed: stopping"
        exit 1
    fi
    return 0
}

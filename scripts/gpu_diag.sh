#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python "$@" > gpurun_out/diag.log 2>&1; echo rc=$?; tail -40 gpurun_out/diag.log

#!/bin/bash
# builder convenience: one gpurun call from the repo root, log under gpurun_out/<name>.log:  bash tools/gpu.sh NAME TIMEOUT_S 'command'
cd /root/repo || exit 1
mkdir -p gpurun_out/r06
/usr/local/graft/bin/gpurun --timeout "$2" -- "mkdir -p gpurun_out/r06; $3" > "gpurun_out/$1.log" 2>&1
echo done >> "gpurun_out/$1.log"

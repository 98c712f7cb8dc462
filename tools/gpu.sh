#!/bin/bash
# build (fails loudly), then run one command on a GPU box: tools/gpu.sh <timeout-s> '<command>'
set -e
make -C /root/repo/waterlily_amd/csrc > /tmp/wl_build.log 2>&1 || { tail -30 /tmp/wl_build.log; echo BUILD FAILED; exit 1; }
make -C /root/repo/oracle > /tmp/wlo_build.log 2>&1 || { tail -30 /tmp/wlo_build.log; echo ORACLE BUILD FAILED; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"

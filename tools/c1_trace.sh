#!/bin/bash
# bash tools/c1_trace.sh <config prefix> : HIP-event class times (one stream, two streams) + kernel timeline of the last frame
export TMPDIR=/tmp
CFG=${1:-"C1 Cornell"}
PTX_OVERLAP_SHADOW=0 timeout -k 10 120 python3 tools/frame_breakdown.py "$CFG" uniform 2>&1 | tail -2
timeout -k 10 120 python3 tools/frame_breakdown.py "$CFG" uniform 2>&1 | tail -2
rm -rf /tmp/tlprof
timeout -s KILL 200 rocprofv3 --kernel-trace -d /tmp/tlprof --output-format csv -- python3 tools/frame_breakdown.py "$CFG" uniform > /tmp/tl.log 2>&1
python3 tools/trace_timeline.py /tmp/tlprof ${2:-70}

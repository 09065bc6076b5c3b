#!/bin/bash
# bash tools/cfg_timeline.sh <config prefix> <n_last_dispatches> : kernel timeline (one stream unless PTX_OVERLAP_SHADOW=1 is exported) of the last frame_breakdown frame
export TMPDIR=/tmp
export PTX_OVERLAP_SHADOW=${PTX_OVERLAP_SHADOW:-0}
rm -rf /tmp/tlprof
timeout -s KILL 280 rocprofv3 --kernel-trace -d /tmp/tlprof --output-format csv -- python3 tools/frame_breakdown.py "$1" uniform > /tmp/tl.log 2>&1
python3 tools/trace_timeline.py /tmp/tlprof ${2:-2000}

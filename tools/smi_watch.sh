#!/bin/bash
# dev tool (gpurun): shader clock / power / temperature sampled every 0.5 s while a command runs.   usage: tools/smi_watch.sh <log> <cmd...>
LOG=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | tr -d '\n' >> $LOG; echo >> $LOG; sleep 0.5; done ) &
W=$!
"$@"
kill $W

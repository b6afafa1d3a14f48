#!/bin/bash
# board power while a micro-benchmark binary loops:  bash scripts/micro/power_run.sh <binary> <iters>   (run on the GPU box)
# prints the binary's own lines, then the rocm-smi samples taken while it ran (W, sclk MHz)
BIN=$1; IT=${2:-6000}
TMP=$(mktemp)
( sleep 1.2; for i in 1 2 3 4 5 6 7 8; do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk clock level" | sed -E 's/.*Power \(W\): *([0-9.]+).*/\1 W/; s/.*sclk clock level:.*\(([0-9]+)Mhz\).*/\1 MHz/' | tr '\n' ' '; echo; sleep 0.25; done ) > $TMP &
S=$!
$BIN $IT | tail -1
wait $S
echo "  power samples while it ran: $(cat $TMP | tr '\n' '|')"
rm -f $TMP

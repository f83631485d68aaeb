#!/bin/bash
# Board power (rocm-smi) while each stress mode of power_modes runs for 4 s; idle first.
cd "$(dirname "$0")"
echo "idle: $(rocm-smi --showpower 2>/dev/null | grep -i 'Package Power' | sed 's/.*: //') W"
for mode in mfma mfmaz mfma32 mfma16 mfmaf16 lds l2 valu; do
  ./power_modes $mode 4 > /tmp/pm_$mode.txt 2>&1 &
  pid=$!
  sleep 1.5
  p1=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'Package Power\|sclk' | sed 's/.*: //' | tr '\n' ' ')
  sleep 1
  p2=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'Package Power\|sclk' | sed 's/.*: //' | tr '\n' ' ')
  wait $pid
  echo "$(cat /tmp/pm_$mode.txt) | smi: $p1 | $p2"
done

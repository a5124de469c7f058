#!/bin/bash
# same-box A/B of round 4's train-step changes: the narrow 3x3 forward kernel and the one-launch flip of the data-gradient weights off / on
for m in off on off on; do
  if [ $m = off ]; then extra="--set biasact.NARROW_FWD=False,biasact.DGRAD_WEIGHT_TABLE=False"; else extra=""; fi
  echo "$m $(python bench.py --no-predict --no-cpu-baseline --steps 30 --warmup 10 $extra 2>&1 | grep '^{"metric' | tail -1 | cut -c60-175)"
done

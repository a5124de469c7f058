#!/bin/bash
# same-box A/B of two builds of libretinanet_hip.so: ab_lib.sh A.so B.so [bench args]  -> ms_per_step of A B A B (the B build is left installed)
A=$1; B=$2; shift 2
for lib in $A $B $A $B; do
  cp $lib pytorch_retinanet_amd/libretinanet_hip.so
  echo "$lib $(python bench.py --no-predict --no-cpu-baseline --steps 30 --warmup 10 "$@" 2>&1 | grep '^{"metric' | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done

#!/bin/bash
# same-box A/B of two builds of libretinanet_hip.so: ab_lib.sh A.so B.so [bench args]  -> ms_per_step of A B A B.
# The installed library is backed up first and restored on exit (also when interrupted): later runs on the box measure the packaged build.
A="$1"; B="$2"; shift 2
INSTALLED=pytorch_retinanet_amd/libretinanet_hip.so
BACKUP="$(mktemp "${TMPDIR:-/tmp}/libretinanet_hip.XXXXXX.so")"
cp "$INSTALLED" "$BACKUP"
trap 'cp "$BACKUP" "$INSTALLED"; rm -f "$BACKUP"' EXIT
for lib in "$A" "$B" "$A" "$B"; do
  cp "$lib" "$INSTALLED"
  echo "$lib $(python bench.py --no-predict --no-cpu-baseline --steps 30 --warmup 10 "$@" 2>&1 | grep '^{"metric' | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done

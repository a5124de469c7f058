#!/bin/bash
# same-box A/B of predict_e2e (R101 predict at B = 16) over several builds of the library: tools/ab_predict.sh lib1.so lib2.so ...  (two interleaved rounds)
INSTALLED=pytorch_retinanet_amd/libretinanet_hip.so
BACKUP="$(mktemp "${TMPDIR:-/tmp}/libretinanet_hip.XXXXXX.so")"
cp "$INSTALLED" "$BACKUP"
trap 'cp "$BACKUP" "$INSTALLED"; rm -f "$BACKUP"' EXIT
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" "$INSTALLED"
    echo "$lib $(python bench.py --no-cpu-baseline --steps 3 --warmup 2 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c 'import sys, json; d = json.loads(sys.stdin.read()); p = d["roofline_other"]["predict_e2e"]; print("predict_e2e", p["value"], "img/s", p["ms_per_batch"], "ms/batch; step", d["ms_per_step"], "ms")')"
  done
done

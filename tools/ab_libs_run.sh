#!/bin/bash
# same-box A/B of several builds of the library on one command: tools/ab_libs_run.sh "python tools/x.py args" lib1.so lib2.so ...  (two interleaved rounds)
CMD="$1"; shift
INSTALLED=pytorch_retinanet_amd/libretinanet_hip.so
BACKUP="$(mktemp "${TMPDIR:-/tmp}/libretinanet_hip.XXXXXX.so")"
cp "$INSTALLED" "$BACKUP"
trap 'cp "$BACKUP" "$INSTALLED"; rm -f "$BACKUP"' EXIT
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" "$INSTALLED"
    echo "== $lib"
    $CMD 2>/dev/null | tail -${AB_TAIL:-3}
  done
done

#!/bin/bash
# same-box A/B of K3 over several builds of the library: tools/k3_ab_libs.sh lib1.so lib2.so ...  (runs tools/k3_rp_ab.py --in-kernel-only twice per build, interleaved)
INSTALLED=pytorch_retinanet_amd/libretinanet_hip.so
BACKUP="$(mktemp "${TMPDIR:-/tmp}/libretinanet_hip.XXXXXX.so")"
cp "$INSTALLED" "$BACKUP"
trap 'cp "$BACKUP" "$INSTALLED"; rm -f "$BACKUP"' EXIT
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" "$INSTALLED"
    echo "== $lib"
    python tools/k3_rp_ab.py --in-kernel-only 2>/dev/null | grep '^{'
  done
done

#!/bin/bash
# scratch/libretinanet_hip_pwstamps.so: the library with the stamped copy of csrc/pw.hip in place of pw.o (csrc/ must be built: make -C pytorch_retinanet_amd/csrc)
set -e
cd "$(dirname "$0")/../.."
python tools/probes/make_stamped_pw.py
C=pytorch_retinanet_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Iinclude -I$C -ffp-contract=fast -c scratch/pw_stamps.hip -o scratch/pw_stamps.o
OBJS=$(cd $C && ls *.o | grep -v '^pw.o$' | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libretinanet_hip_pwstamps.so $OBJS scratch/pw_stamps.o
echo "built scratch/libretinanet_hip_pwstamps.so"

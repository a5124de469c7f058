#!/bin/bash
# builds tools/probes/libnms_stamped.so: the library with scratch/nms_stamped.hip (written by make_nms_stamped.py: csrc/nms.hip + stamps) in place of csrc/nms.hip
python3 "$(dirname "$0")/make_nms_stamped.py" || exit 1
cd "$(dirname "$0")/../../pytorch_retinanet_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -I../../include -I. -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -c ../../scratch/nms_stamped.hip -o /tmp/nms_stamped.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libnms_stamped.so api.o anchors.o match.o loss.o detect.o /tmp/nms_stamped.o norm.o transform.o conv.o pool.o optim.o pw.o stem.o wgrad3x3.o narrow3x3.o

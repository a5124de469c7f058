#!/bin/bash
# development probe: time nms_mask_kernel cut off after phase k (1 keys loaded, 2 rank sort, 3 boxes gathered, 4 matrix)
cd $GRAFT_REPO_ROOT/pytorch_retinanet_amd/csrc
for k in 1 2 3 4 0; do
  if [ $k = 0 ]; then D=""; else D="-DRN_NMS_STOP=$k"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -I../../include -I. -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $D -c nms.hip -o nms.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libretinanet_hip.so api.o anchors.o match.o loss.o detect.o nms.o norm.o transform.o conv.o pool.o optim.o
  cd $GRAFT_REPO_ROOT; echo "stop=$k: $(tools/prof_kernels.sh detect 2>&1 | grep nms_mask)"; cd pytorch_retinanet_amd/csrc
done

#!/bin/bash
# same-box off / on A/B of round 5's train-step switches (the build-time placements -- PW_WGRAD_XCD_MAP, CONV_XCD_MAP, WGRAD_XCD_MAP, BAND_NARROW_FWD --
# and the tile kernel's prefetching epilogue have no switch: tools/ab_lib.sh with two builds)
OFF="pwconv.FUSE_CONV3_BWD=False,pwconv.FUSE_CHAIN=False,pwconv.FUSE_BWD_CHAIN=False,pwconv.CONV3_FWD_WALKER=False,pwconv.STEM_WGRAD_BN=False,pwconv.STEM_POOL_BN_SUMS=False"
OFF="$OFF,biasact.DENSE_SPLITK=False,biasact.BOX_OUTPUT_WGRAD_NARROW=False,biasact.DENSE_BAND=False,biasact.DENSE_BAND_STATS=False,biasact.TOWER_SUM2=False"
OFF="$OFF,pwconv.DOWN_WGRAD_PW=False,pwconv.STRIDED_WGRAD_PW=False,pwconv.BIAS_1X1_MM=False,pwconv.MANY_ROWS_MM=False"
for m in off on off on; do
  if [ $m = off ]; then extra="--set $OFF"; else extra=""; fi
  echo "$m $(python bench.py --no-predict --no-cpu-baseline --steps 30 --warmup 10 $extra 2>&1 | grep '^{"metric' | tail -1 | grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 30, "warmup": 10, "ms_per_step": [0-9.]*')"
done

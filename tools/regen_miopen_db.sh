#!/bin/bash
# Refresh the shipped MIOpen find-db (pytorch_retinanet_amd/miopen_db) for the conv shapes of the headline
# config: start from the shipped files, let cudnn.benchmark's find append what is missing, hand the result
# back through gpurun_out/miopen_db (copy it over pytorch_retinanet_amd/miopen_db afterwards).
set -e
db=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
rm -rf $db; mkdir -p $db
cp pytorch_retinanet_amd/miopen_db/* $db/
export MIOPEN_USER_DB_PATH=$db
t0=$(date +%s)
python bench.py --steps 3 --warmup 2 --no-cpu-baseline | tail -1 | cut -c1-200
echo "find + bench took $(( $(date +%s) - t0 )) s"
ls -la $db

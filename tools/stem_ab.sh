#!/bin/bash
# A/B of the stem + pool kernel at 640 frames: the tile kernel against the column-strip kernel (the default).
export ONLY=OP_STEM_POOL
run() { echo "== $*"; env "$@" timeout -k 10 100 python -u tools/time_plan_ops.py resnet18_mean 640 20 2>&1 | grep STEM || exit 1; }
run ASL_STEM_TILES=1 ASL_PLANAR_STEM=0
run ASL_PLANAR_STEM=0
run ASL_PLANAR_STEM=1

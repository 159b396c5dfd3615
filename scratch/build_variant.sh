#!/bin/bash
# A build of the library with extra compiler flags, for A/B runs on one box (scratch/ab_libs.sh):
#   scratch/build_variant.sh scratch/libs/lean.so -DMOCAP_BOX_GROUP=3 -DMOCAP_BOX_SU=7 -DMOCAP_BOX_WAVES=4
out=$1; shift
mkdir -p "$(dirname "$out")"
cd "$(dirname "$0")/../mocapv2_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed "$@" -o "../../$out" abi.hip blob_filter.hip blob_boxes.hip blob_contours.hip bayer_gray.hip geom.hip -ldl

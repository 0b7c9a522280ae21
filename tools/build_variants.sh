#!/bin/bash
# tools/build_variants.sh NAME "FLAGS" [NAME "FLAGS" ...] -- A/B builds of librto_hip.so under build/variants/ (git-ignored,
# built with -DRTO_DEV_KNOBS: only these builds read the RTO_* A/B environment variables; the shipped library reads none,
# shipped to the GPU box); select one with RTO_HIP_LIB=build/variants/librto_hip_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/build/variants"
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -DRTO_DEV_KNOBS $F "$R/ray_tracing_octrees_amd/csrc/rto_api.hip" -o "$R/build/variants/librto_hip_$N.so" &
done
wait
ls -la "$R/build/variants"

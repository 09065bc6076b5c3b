#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]  ->  rust-pathtracer_amd/lib/variants/NAME.so (experiments only)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rust-pathtracer_amd"
mkdir -p lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -Wall -Wno-unused-function \
  -I../include "$@" csrc/device/pt_core.hip -o lib/variants/$NAME.so

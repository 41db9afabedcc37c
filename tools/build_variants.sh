#!/bin/bash
# Experiment builds of libhlm_hip.so for same-box A/B runs (tools/dense_variant_probe.py):
#   tools/build_variants.sh name1:"-DFLAG ..." name2:"..."   ->  tiger_hlm_gpu_amd/lib/variants/libhlm_<name>.so
cd "$(dirname "$0")/../tiger_hlm_gpu_amd/csrc" || exit 1
mkdir -p ../lib/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  echo "building $name with [$flags]"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -I../../include -I. $flags -shared \
      -o ../lib/variants/libhlm_$name.so hlm_abi.hip || exit 1
done

#!/bin/bash
# Resource usage of the built kernels (after `make -C weekend-raytracer-wgpu_amd/csrc asm`):
#   tools/kres.sh [name-pattern]     e.g. tools/kres.sh 'pool_kernelILj256ELj112ELj6ELb0ELb0'
f="$(dirname "$0")/../weekend-raytracer-wgpu_amd/csrc/build/resource_usage.txt"
awk -v pat="${1:-.}" '/Function Name/{name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name); show=(name ~ pat); if(show) printf "%s\n", name}
  show && /(TotalSGPRs|VGPRs:|ScratchSize|Occupancy|Spill)/{l=$0; sub(/^remark: [^ ]+ +/,"",l); sub(/ \[-R.*/,"",l); printf "    %s\n", l}' "$f"

#!/bin/bash
# usage: tools/kres.sh file.hip  -> compact per-kernel resource table (VGPR/AGPR/scratch/LDS/occupancy)
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: //; s/\t/ | /g'

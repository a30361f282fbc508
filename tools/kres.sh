#!/bin/bash
# usage: tools/kres.sh file.hip  -> one line per kernel: name vgpr agpr scratch occupancy spill lds
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 \
 | grep -E "Function Name|Name:|VGPRs:|AGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size" \
 | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Name:/{if(n)print n,v,a,s,o,sp,l; n=$NF} /^VGPRs:/{v="v="$2} /^AGPRs:/{a="a="$2} /ScratchSize/{s="scr="$NF} /Occupancy/{o="occ="$NF} /VGPRs Spill/{sp="spill="$NF} /LDS Size/{l="lds="$NF} END{print n,v,a,s,o,sp,l}' | c++filt

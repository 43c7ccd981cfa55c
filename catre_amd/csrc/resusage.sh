#!/bin/bash
# Print a compact per-kernel resource table (VGPR/AGPR/SGPR/spills/LDS/occupancy) for the gfx950 build.
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage catre_kernels.hip -o /tmp/_catre_res.so 2>&1 \
 | awk '/Function Name/{name=$5} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /TotalSGPRs/{s=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /Occupancy/{o=$(NF-1)} /VGPRs Spill/{vs=$(NF-1)} /LDS Size/{printf "%-34s vgpr=%-4s agpr=%-4s sgpr=%-4s scratch=%-3s vspill=%-3s lds=%-7s occ=%s\n", substr(name,1,34), v, a, s, sc, vs, $(NF-1), o}' 

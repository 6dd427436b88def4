#!/bin/bash
# register / scratch usage of every kernel in one HIP source: tools/regs.sh vacnic_amd/csrc/gemm_t256.hip
f=$1
cd $(dirname $f)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $(basename $f) -o /tmp/regs_$$.o 2>&1 | \
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPR Spill/ {sp=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {print "VGPR", v, "AGPR", a, "scratch", s, "spill", sp, "occ", o, name}' | c++filt | cut -c1-170
rm -f /tmp/regs_$$.o

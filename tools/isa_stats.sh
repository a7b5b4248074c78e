#!/bin/bash
# Per-kernel ISA statistics of the built wildfire object: instruction count, registers, LDS, spills.
# usage: tools/isa_stats.sh [object] [name filter]
set -e
OBJ=${1:-/root/repo/free-range-zoo_amd/csrc/wildfire.o}
FILT=${2:-wf_step_kernelILi6ELi3}
T=$(mktemp -d)
cp "$OBJ" $T/o.o
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o >/dev/null 2>&1)
CO=$(ls $T/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $CO > $T/k.s
awk '/^[0-9a-f]+ <.*>:/{name=$2} /^\t[a-z]/{c[name]++} END{for(n in c) print c[n], n}' $T/k.s | sort -n | grep "$FILT" || true
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | grep -E "\.name:|vgpr_count|sgpr_count|spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - - | grep "$FILT" | sed -E 's/ +/ /g; s/\t/ /g; s/\.(group_segment_fixed_size|private_segment_fixed_size|sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count)/\1/g' | awk '{print $0}' | sed -E 's/group_segment_fixed_size/lds/; s/private_segment_fixed_size/scratch/' | cut -c1-250
cp $T/k.s /tmp/isa/k_all.s 2>/dev/null || true
rm -rf $T

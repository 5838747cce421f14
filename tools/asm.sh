#!/bin/bash
# usage: tools/asm.sh <file.hip> <mangled-kernel-prefix>   -> /tmp/asm/k.s with that kernel's ISA + resource summary
set -e
mkdir -p /tmp/asm
cd /root/repo/graphnet_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/asm/all.s "$1" 2>&1 | grep -E "error" || true
awk "/^$2/,/\.end_amdhsa_kernel/" /tmp/asm/all.s > /tmp/asm/k.s
grep -n "scratch_\|private_segment_fixed\|next_free_vgpr" /tmp/asm/k.s || true
echo "--- waits/barriers"
grep -n "s_barrier\|s_waitcnt vmcnt" /tmp/asm/k.s || true

#!/bin/bash
# usage: tools/ab_kstats.sh A.so B.so [bench args]   per-kernel stats of bench.py with each library in place (same box)
A=$1; B=$2; shift 2
for L in $A $B; do cp $L fov-3dgs_amd/libfovraster_hip.so; echo "== $L"; tools/kstats_bench.sh "$@" 2>&1 | grep -v "k_activate\|k_l1_ssim\|k_pack"; done

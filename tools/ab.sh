#!/bin/bash
# usage: tools/ab.sh A.so B.so -- cmd...   runs cmd alternately with each library in place (same box, ABAB)
A=$1; B=$2; shift 3
for r in 1 2; do for L in $A $B; do cp $L fov-3dgs_amd/libfovraster_hip.so; echo "== $L"; "$@" 2>/dev/null | tail -1 | cut -c1-140; done; done

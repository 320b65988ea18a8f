#!/bin/bash
# usage (on the GPU box): tools/ab_run.sh ROUNDS NAME1 NAME2 ... -- cmd...
# Runs cmd with each of fov-3dgs_amd/ab/NAME.so in place of the library, interleaved ROUNDS times (same box), and prints
# the last line of each run. "base" = the library as built by make.
R=$1; shift
names=(); while [ "$1" != "--" ]; do names+=("$1"); shift; done; shift
cp fov-3dgs_amd/libfovraster_hip.so fov-3dgs_amd/ab/base.so
for r in $(seq $R); do for n in "${names[@]}"; do
	cp fov-3dgs_amd/ab/$n.so fov-3dgs_amd/libfovraster_hip.so
	echo "== $n: $("$@" 2>/dev/null | tail -1 | cut -c1-400)"
done; done
cp fov-3dgs_amd/ab/base.so fov-3dgs_amd/libfovraster_hip.so

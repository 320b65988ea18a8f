#!/bin/bash
# usage (on the GPU box): tools/ab_run.sh ROUNDS NAME1 NAME2 ... -- cmd...
# Runs cmd with each of fov-3dgs_amd/ab/NAME.so (built by tools/ab_build.sh) as the library, interleaved ROUNDS times on the same
# box, and prints the last line of each run. "base" = the library as built by make. The library is selected through the
# FOVRASTER_LIB environment variable (fov-3dgs_amd/_native.py): nothing is copied over the build.
if [ $# -lt 4 ]; then echo "usage: $0 ROUNDS NAME... -- cmd..." >&2; exit 2; fi
R=$1; shift
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
if [ "$1" != "--" ]; then echo "$0: missing -- before the command" >&2; exit 2; fi
shift
for r in $(seq $R); do for n in "${names[@]}"; do
	L=$PWD/fov-3dgs_amd/ab/$n.so; [ "$n" == "base" ] && L=$PWD/fov-3dgs_amd/libfovraster_hip.so
	echo "== $n: $(FOVRASTER_LIB=$L "$@" 2>/dev/null | tail -1 | cut -c1-1200)"
done; done

#!/bin/bash
# usage (on the GPU box): tools/ab_env.sh ROUNDS "ENV1=.. ENV2=.." "ENVB=.." ... -- cmd...
# Runs cmd under each environment setting in turn ("-" = nothing set), interleaved ROUNDS times on the same box, and prints the
# last line of each run (A / B comparisons of run-time switches: FOVRASTER_FUSE_SCAN=0 against the default, ...).
if [ $# -lt 4 ]; then echo "usage: $0 ROUNDS ENV... -- cmd..." >&2; exit 2; fi
R=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
for r in $(seq $R); do for e in "${envs[@]}"; do
	if [ "$e" == "-" ]; then out=$("$@" 2>/dev/null | tail -${AB_TAIL:-1} | cut -c1-400); else out=$(env $e "$@" 2>/dev/null | tail -${AB_TAIL:-1} | cut -c1-400); fi
	echo "== [$e] $out"
done; done

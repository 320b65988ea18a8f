#!/bin/bash
# usage: tools/ab_build.sh NAME [file.hip ...] [-- -DFLAG ...]
# Builds fov-3dgs_amd/ab/NAME.so: the current objects of csrc/ with the named sources recompiled with the extra flags.
# (An experiment build for tools/ab_run.sh; *.so travel to the GPU box but stay out of git.)
set -e
cd "$(dirname "$0")/../fov-3dgs_amd/csrc"
name=$1; shift
files=(); flags=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; flags=("$@"); break; fi; files+=("$1"); shift; done
make -j8 >/dev/null
objs=()
for s in api preprocess binning render backward loss activations; do
	hit=0; for f in "${files[@]}"; do [ "$f" == "$s.hip" ] && hit=1; done
	if [ $hit == 1 ]; then
		/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "${flags[@]}" -c $s.hip -o ../ab/$name.$s.o &
		objs+=(../ab/$name.$s.o)
	else objs+=($s.o); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/$name.so "${objs[@]}"
rm -f ../ab/$name.*.o
echo "built fov-3dgs_amd/ab/$name.so (${files[*]} with ${flags[*]})"

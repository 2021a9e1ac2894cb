#!/bin/bash
# An alternative libdynmm_hip.so for whole-step A/B runs (scratch/r6/ab_lib.sh): the product's objects with the versions of the
# named kernel files taken from an older commit.
#   bash scratch/r6/build_old_lib.sh <commit> <out.so> file1.hip [file2.hip ...]
# e.g. the A side of profiles/r06_ab_runs.md "short kernels" table (round-5 tail.hip + pointwise.hip):
#   bash scratch/r6/build_old_lib.sh 53688b6 scratch/r6/libdynmm_old_tail.so tail.hip pointwise.hip
set -e
cd "$(dirname "$0")/../.."
commit=$1; out=$2; shift 2
make -C dynmm_amd/csrc > /dev/null
objs=$(ls dynmm_amd/csrc/build/*.o)
tmp=$(mktemp -d)
for f in "$@"; do
  stem=${f%.hip}
  git show "$commit:dynmm_amd/csrc/$f" > dynmm_amd/csrc/_old_$f
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Idynmm_amd/csrc -Wno-unused-function -Wno-inline-asm -c dynmm_amd/csrc/_old_$f -o $tmp/$stem.o
  rm dynmm_amd/csrc/_old_$f
  objs=$(echo "$objs" | grep -v "/$stem.o")
  objs="$objs $tmp/$stem.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out"
rm -rf $tmp
ls -la "$out"

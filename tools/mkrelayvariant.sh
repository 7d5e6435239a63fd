#!/bin/bash
# usage: mkrelayvariant.sh <name> [extra hipcc flags ...]   -> .ab/lib<name>.so
# A variant of env_relay_kernel only: cavoid_relay.hip recompiled with the given flags (from $SRC if set: a patched copy of csrc/), every other
# translation unit taken from the product's object files (rl_collision_avoidance_amd/build/*.o: run build() first).  Seconds instead of minutes.
set -e
name=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
src=${SRC:-$repo/rl_collision_avoidance_amd/csrc}
obj=$repo/rl_collision_avoidance_amd/build
mkdir -p $repo/.ab /tmp/relayvar_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$repo/include -I$src -mllvm -disable-machine-licm "$@" \
      -c $src/cavoid_relay.hip -o /tmp/relayvar_$name/cavoid_relay.o
others=$(ls $obj/*.o | grep -v "\.\(fault\|ulp[0-9]\|trace\)\.o$" | grep -v "/cavoid_relay\.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/relayvar_$name/cavoid_relay.o -ldl -o $repo/.ab/lib$name.so
ls -la $repo/.ab/lib$name.so

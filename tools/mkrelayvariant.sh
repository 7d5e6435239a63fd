#!/bin/bash
# usage: [TU=cavoid_multistep] mkrelayvariant.sh <name> [extra hipcc flags ...]   -> .ab/lib<name>.so
# A variant of ONE translation unit (default cavoid_relay: env_relay_kernel): that .hip recompiled with the given flags (from $SRC if set: a patched copy
# of csrc/), every other unit taken from the product's object files (rl_collision_avoidance_amd/build/*.o: run build() first).  Seconds to a minute instead of minutes.
set -e
name=$1; shift
tu=${TU:-cavoid_relay}
repo=$(cd "$(dirname "$0")/.." && pwd)
src=${SRC:-$repo/rl_collision_avoidance_amd/csrc}
obj=$repo/rl_collision_avoidance_amd/build
mkdir -p $repo/.ab /tmp/relayvar_$name
licm=""; case $tu in cavoid_multistep|cavoid_rvo|cavoid_relay|cavoid_relay_rvo|cavoid_actor|cavoid_actor_rvo|cavoid_actor_frozen) licm="-mllvm -disable-machine-licm";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$repo/include -I$src $licm "$@" -c $src/$tu.hip -o /tmp/relayvar_$name/$tu.o
others=$(ls $obj/*.o | grep -v "\.\(fault\|ulp[0-9]\|trace\)\.o$" | grep -v "/$tu\.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/relayvar_$name/$tu.o -ldl -o $repo/.ab/lib$name.so
ls -la $repo/.ab/lib$name.so

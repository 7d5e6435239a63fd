#!/bin/bash
# usage: mkvariant.sh <name> <python patch file or ''>   -> /root/repo/.ab/lib<name>.so (dev-only N = 4, 10)
set -e
name=$1; patch=$2
d=/tmp/var_$name; rm -rf $d; mkdir -p $d
cp ${SRC:-/root/repo/rl_collision_avoidance_amd/csrc}/* $d/
if [ -n "$patch" ]; then (cd $d && python3 $patch); fi
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I/root/repo/include -I$d -DCAVOID_DEV_ONLY_N ${XFLAGS}"
pids=""
for tu in cavoid_capi cavoid_multistep cavoid_rvo cavoid_relay cavoid_relay_rvo cavoid_quad cavoid_actor cavoid_actor_rvo cavoid_actor_frozen cavoid_rollout_capi cavoid_policy_capi cavoid_comm_capi; do
  extra=""; case $tu in cavoid_multistep|cavoid_rvo|cavoid_relay|cavoid_relay_rvo|cavoid_actor|cavoid_actor_rvo|cavoid_actor_frozen) extra="-mllvm -disable-machine-licm";; esac
  hipcc $F $extra -c $d/$tu.hip -o $d/$tu.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
mkdir -p /root/repo/.ab
hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -ldl -o /root/repo/.ab/lib$name.so
ls -la /root/repo/.ab/lib$name.so

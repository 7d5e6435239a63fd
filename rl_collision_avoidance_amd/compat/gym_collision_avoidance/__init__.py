"""Import-path shim: put ``rl_collision_avoidance_amd/compat`` on PYTHONPATH and the reference's
unmodified GA3C modules resolve their ``gym_collision_avoidance`` imports to the MI355X-native
implementation."""

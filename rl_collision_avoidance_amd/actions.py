"""``Actions`` -- the discrete action table of the GA3C-CADRL policy, as the reference imports it
(``from gym_collision_avoidance.envs.policies.GA3C_CADRL.network import Actions``,
/root/reference/ga3c/GA3C/Server.py:36; used at Server.py:51-52 ``.num_actions`` and
Regression.py:157-160 ``.actions[:,0]`` speed / ``.actions[:,1]`` heading change)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class Actions(object):
    def __init__(self):
        table = ((C.c_double * 2) * _lib.MAX_ACTIONS)()
        n = C.c_int32(0)
        _lib.check(_lib.lib().cavoid_default_actions(C.cast(table, C.c_void_p), C.byref(n)), "cavoid_default_actions")
        self.actions = np.array([[table[r][0], table[r][1]] for r in range(n.value)], dtype=np.float64)
        self.num_actions = len(self.actions)

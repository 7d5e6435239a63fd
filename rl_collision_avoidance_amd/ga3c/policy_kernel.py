"""``FusedPolicy`` -- the actors' ``predict_p_and_v`` (+ ``select_action``) as one MFMA kernel launch.

Host mirror of ``cavoid_policy_*`` (include/cavoid.h): takes a ``NetworkVP_rnn`` (arch 'rnn'), hands its
parameters to the library in the reference checkpoint's layout, and is then callable like
``NetworkVPCore.predict_p_and_v`` (/root/reference/ga3c/GA3C/NetworkVPCore.py:175-176).  The network module stays
the single owner of the weights (the trainer updates it); call ``refresh()`` after an optimiser step.
There is no fallback: without the HIP library / a GPU this raises."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from .. import _lib
from .network import NetworkVP_rnn


class FusedPolicy(object):
    accepts_strided_obs = True          # BatchedRollout hands over the env's obs tensor itself, no slice copy

    def __init__(self, net: NetworkVP_rnn, seed: int = 0, forget_bias: float = 1.0):
        if net.arch != "rnn":
            raise ValueError("FusedPolicy implements MULTI_AGENT_ARCH 'rnn' (the recorded configuration)")
        dev = net.layer1_kernel.device
        if dev.type != "cuda":
            raise ValueError("FusedPolicy needs the network on the GPU")
        self.net, self.device = net, dev
        self.num_actions, self.max_others, self.input_size = net.num_actions, net.max_others, net.input_size
        self.forget_bias = float(forget_bias)
        self._lib = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._lib.cavoid_policy_create(self.max_others, self.num_actions, dev.index or 0, C.byref(h)), "cavoid_policy_create")
        self._h = h
        self.seed(seed)
        self.refresh()

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            torch.cuda.synchronize(self.device)
            self._lib.cavoid_policy_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def seed(self, seed: int) -> None:
        _lib.check(self._lib.cavoid_policy_seed(self._h, C.c_uint64(int(seed) & (2 ** 64 - 1)), self._stream()), "cavoid_policy_seed")

    def refresh(self) -> None:
        """Re-pack the module's current parameters (after a trainer step / checkpoint load)."""
        n = self.net
        w = _lib.CavoidPolicyWeights()
        w.struct_size = C.sizeof(_lib.CavoidPolicyWeights)
        w.min_policy, w.forget_bias = float(n.min_policy), self.forget_bias
        ptr = lambda t: C.c_void_p(self._f32(t).data_ptr())
        self._keep = []                  # tensors that had to be made contiguous stay alive until the next refresh
        if n.normalize:
            w.avg, w.std = ptr(n.avg), ptr(n.std)
        for name in ("lstm_kernel", "lstm_bias", "layer1_kernel", "layer1_bias", "layer2_kernel", "layer2_bias",
                     "fc1_kernel", "fc1_bias", "p_kernel", "p_bias", "v_kernel", "v_bias"):
            setattr(w, name, ptr(getattr(n, name)))
        _lib.check(self._lib.cavoid_policy_load(self._h, C.byref(w), self._stream()), "cavoid_policy_load")

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            self._keep.append(t)
        return t

    def forward(self, x: torch.Tensor, sample: Optional[bool] = None, greedy: bool = False
                ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """x float32 [B, input_size] (rows may be strided: a column slice of the env's obs tensor) ->
        (p [B, A], v [B], actions int32 [B] or None)."""
        if x.dim() != 2 or x.shape[1] != self.input_size or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError("x must be float32 [B, %d] on %s" % (self.input_size, self.device))
        if x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < self.input_size):
            x = x.contiguous()
        B = x.shape[0]
        stride = x.stride(0) if B > 1 else self.input_size
        p = torch.empty((B, self.num_actions), dtype=torch.float32, device=self.device)
        v = torch.empty((B,), dtype=torch.float32, device=self.device)
        want_actions = greedy if sample is None else (sample or greedy)
        a = torch.empty((B,), dtype=torch.int32, device=self.device) if want_actions else None
        _lib.check(self._lib.cavoid_policy_forward(self._h, C.c_void_p(x.data_ptr()), B, stride, C.c_void_p(p.data_ptr()),
                                                   C.c_void_p(v.data_ptr()), C.c_void_p(a.data_ptr()) if a is not None else None,
                                                   1 if greedy else 0, self._stream()), "cavoid_policy_forward")
        return p, v, a

    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        p, v, _ = self.forward(x, sample=False)
        return p, v

    def act(self, x: torch.Tensor, greedy: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """predict + select_action (ProcessAgent.py:89-103,128-144): (actions int32 [B], p, v)."""
        p, v, a = self.forward(x, sample=True, greedy=greedy)
        return a, p, v

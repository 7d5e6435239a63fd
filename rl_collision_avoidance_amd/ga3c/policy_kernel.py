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
from .network import NetworkVP_rnn, split_k_factor


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
        # the inference form is fixed at creation (CAVOID_POLICY_F32 / CAVOID_POLICY_PRODUCTS are read by cavoid_policy_create only)
        use_split, products = C.c_int32(), C.c_int32()
        _lib.check(self._lib.cavoid_policy_info(h, None, C.byref(use_split), C.byref(products), None), "cavoid_policy_info")
        self.inference_form = ("split", int(products.value)) if use_split.value else ("f32", 0)
        self.seed(seed)
        self.refresh(check_range=True)

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

    def clamped_weights(self) -> int:
        """How many weights of the last ``refresh`` the default float16-split form had to clamp to +-65504 (include/cavoid.h,
        cavoid_policy_info; always 0 for the bf16 / float32 forms).  Synchronises the current stream."""
        n = C.c_int32()
        _lib.check(self._lib.cavoid_policy_info(self._h, self._stream(), None, None, C.byref(n)), "cavoid_policy_info")
        return int(n.value)

    def refresh(self, with_backward: bool = False, check_range: bool = False) -> None:
        """Re-pack the module's current parameters (after a trainer step / checkpoint load).  ``with_backward`` also
        packs the transposed copies the fused trainer pass needs.  ``check_range`` (the constructor and checkpoint loads use it; it
        costs a stream synchronisation): fail loudly when a weight lies beyond the float16 split's +-65504 instead of running a
        network that silently differs from the reference's float32 predictor."""
        n = self.net
        w = _lib.CavoidPolicyWeights()
        w.struct_size = C.sizeof(_lib.CavoidPolicyWeights)
        w.min_policy, w.forget_bias = float(n.min_policy), self.forget_bias
        w.with_backward = 1 if with_backward else 0
        ptr = lambda t: C.c_void_p(self._f32(t).data_ptr())
        self._keep = []                  # tensors that had to be made contiguous stay alive until the next refresh
        if n.normalize:
            w.avg, w.std = ptr(n.avg), ptr(n.std)
        for name in ("lstm_kernel", "lstm_bias", "layer1_kernel", "layer1_bias", "layer2_kernel", "layer2_bias",
                     "fc1_kernel", "fc1_bias", "p_kernel", "p_bias", "v_kernel", "v_bias"):
            setattr(w, name, ptr(getattr(n, name)))
        _lib.check(self._lib.cavoid_policy_load(self._h, C.byref(w), self._stream()), "cavoid_policy_load")
        if check_range and self.inference_form == ("split", 16):
            bad = self.clamped_weights()
            if bad:
                raise ValueError("%d weight(s) lie beyond +-65504 (after the LSTM gates' log2 e scale): the default float16-split inference "
                                 "form would clamp them.  Create the policy with CAVOID_POLICY_PRODUCTS=3 (bf16 pieces, float32's range) or "
                                 "CAVOID_POLICY_F32=1 in the environment." % bad)

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            self._keep.append(t)
        return t

    def forward(self, x: torch.Tensor, sample: Optional[bool] = None, greedy: bool = False,
                rows: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, actions_out: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """x float32 [B, input_size] (rows may be strided: a column slice of the env's obs tensor) ->
        (p [B, A], v [B], actions int32 [B] or None).  ``rows`` = (row_index int32 [B], row_count int32 [1]) on the
        device restricts the pass to the listed rows; the others' outputs are zero."""
        if x.dim() != 2 or x.shape[1] != self.input_size or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError("x must be float32 [B, %d] on %s" % (self.input_size, self.device))
        if x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < self.input_size):
            x = x.contiguous()
        B = x.shape[0]
        stride = x.stride(0) if B > 1 else self.input_size
        new = torch.zeros if rows is not None else torch.empty
        p = new((B, self.num_actions), dtype=torch.float32, device=self.device)
        v = new((B,), dtype=torch.float32, device=self.device)
        want_actions = greedy if sample is None else (sample or greedy)
        a = new((B,), dtype=torch.int32, device=self.device) if want_actions else None
        if actions_out is not None:                        # with a row list: ONLY the listed rows' actions are overwritten
            if actions_out.dtype != torch.int32 or actions_out.numel() != B or not actions_out.is_contiguous():
                raise ValueError("actions_out must be a contiguous int32 tensor of B elements")
            a = actions_out
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if rows is None:
            _lib.check(self._lib.cavoid_policy_forward(self._h, ptr(x), B, stride, ptr(p), ptr(v), ptr(a), 1 if greedy else 0,
                                                       self._stream()), "cavoid_policy_forward")
        else:
            _lib.check(self._lib.cavoid_policy_forward_rows(self._h, ptr(x), B, stride, ptr(rows[0]), ptr(rows[1]), ptr(p), ptr(v),
                                                            ptr(a), 1 if greedy else 0, self._stream()), "cavoid_policy_forward_rows")
        return p, v, a

    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        p, v, _ = self.forward(x, sample=False)
        return p, v

    def act(self, x: torch.Tensor, greedy: bool = False, rows=None, actions_out=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """predict + select_action (ProcessAgent.py:89-103,128-144): (actions int32 [B], p, v)."""
        p, v, a = self.forward(x, sample=True, greedy=greedy, rows=rows, actions_out=actions_out)
        return a, p, v


class FusedA3CTrainer(object):
    """``Server.train_model`` (/root/reference/ga3c/GA3C/Server.py:114-124) with the network's forward pass, the A3C loss
    (NetworkVPCore.py:71-100) and the row-local half of the backward pass as two MFMA kernel launches
    (``cavoid_policy_train``); the weight gradients are the library GEMMs ``X^T G`` over all rows (split-K batched), the
    optimiser is the same fused Adam as ``A3CTrainer``.  Gradients equal PyTorch autograd's on ``NetworkVP_rnn.loss`` to
    float32 rounding (tests/test_gpu_policy.py)."""

    def __init__(self, net: NetworkVP_rnn, policy: Optional[FusedPolicy] = None, learning_rate: float = 2e-5, group=None,
                 distributed: Optional[bool] = None):
        from .network import A3CTrainer
        self.net = net
        self.policy = policy if policy is not None else FusedPolicy(net)
        self._base = A3CTrainer(net, learning_rate=learning_rate, group=group, distributed=distributed)
        self.opt = self._base.opt
        self.device = self.policy.device
        self._buffers = {}
        H, A = net.HIDDEN, net.num_actions
        # packed gate column k = 64w + 16 gate + u  <->  checkpoint column c = 64 gate + 16w + u
        c = torch.arange(4 * H, device=self.device)
        gate, w, u = c // H, (c % H) // 16, c % 16
        self._k_of_c = 64 * w + 16 * gate + u
        # packed LSTM gradient [72 rows: 64 hidden, 7 inputs, pad] x [256 packed columns] -> checkpoint layout [7 + 64, 256]
        rows = torch.cat([torch.arange(H, H + net.OTHER), torch.arange(0, H)]).to(self.device)
        self._lstm_flat = (rows.unsqueeze(1) * (4 * H) + self._k_of_c.unsqueeze(0)).reshape(-1)
        self._l1_rows = torch.cat([torch.arange(H, H + net.HOST), torch.arange(0, H)]).to(self.device)
        self.policy.refresh(with_backward=True)

    training_step = property(lambda self: self._base.training_step,
                             lambda self, v: setattr(self._base, "training_step", v))
    frame_counter = property(lambda self: self._base.frame_counter)

    def _scratch(self, rows64: int):
        b = self._buffers.get(rows64)
        if b is None:
            M, dev = self.net.max_others, self.device
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            t = {"z1": f(rows64, 256), "z2": f(rows64, 256), "z3": f(rows64, 256), "l1_in": f(rows64, 72), "h_in": f(M, rows64, 72),
                 "save": f(rows64 // 64, M, 16, 256, 8), "gh": f(rows64, 16), "loss": f(2), "g1": f(rows64, 256), "g2": f(rows64, 256),
                 "g3": f(rows64, 256), "gl": f(M, rows64, 256), "db": f(1040)}
            c = _lib.CavoidPolicyTrainBuffers()
            c.struct_size, c.capacity_rows = C.sizeof(_lib.CavoidPolicyTrainBuffers), rows64
            for k, v in t.items():
                setattr(c, k, C.c_void_p(v.data_ptr()))
            b = self._buffers[rows64] = (t, c)
            if len(self._buffers) > 4:                     # keep the cache small: minibatch size + a remainder or two
                self._buffers.pop(next(iter(self._buffers)))
        return b

    @staticmethod
    def _xtg(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        """x^T g over all rows, split-K (see network._SplitKLinear)."""
        R = x.shape[0]
        S = split_k_factor(R)
        if S == 1:
            return x.t() @ g
        return torch.bmm(x.view(S, R // S, -1).transpose(1, 2), g.view(S, R // S, -1)).sum(dim=0)

    def train(self, x: torch.Tensor, y_r: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        """One optimiser step on the batch.  ``a``: action indices [n] or the reference's one-hot float [n, A].
        Returns the loss (cost_p + cost_v) as a 0-d device tensor: nothing here waits for the GPU."""
        net, pol = self.net, self.policy
        n = int(x.shape[0])
        if n == 0:
            # (multi-GPU padding step: this rank has no rows, but the all-reduced gradients of the others still move its
            #  parameters -- the packed MFMA weights must follow, or this replica's actors keep acting on stale weights)
            loss = torch.as_tensor(self._base.train(x, y_r, a if a.dim() == 2 else
                                                    torch.nn.functional.one_hot(a.long(), net.num_actions).float()), device=self.device)
            pol.refresh(with_backward=True)
            return loss
        x = x.to(torch.float32).contiguous()
        y_r = y_r.to(torch.float32).contiguous()
        a_idx = (a.argmax(dim=1) if a.dim() == 2 else a).to(torch.int32).contiguous()
        # buffer rows: a multiple of 2048 (the split-K slice of the weight-gradient GEMMs) once the batch is that large;
        # the kernels write every buffer row, rows past n with zero gradients
        rows64 = (n + 2047) // 2048 * 2048 if n >= 2048 else (n + 63) // 64 * 64
        t, cbuf = self._scratch(rows64)
        ptr = lambda v: C.c_void_p(v.data_ptr())
        _lib.check(pol._lib.cavoid_policy_train(pol._h, ptr(x), n, x.stride(0), ptr(y_r), ptr(a_idx), float(net.beta),
                                                float(net.log_epsilon), C.byref(cbuf), pol._stream()), "cavoid_policy_train")
        A, H, M = net.num_actions, net.HIDDEN, net.max_others
        xtg = self._xtg
        d_head = xtg(t["z3"], t["gh"])
        net.p_kernel.grad, net.v_kernel.grad = d_head[:, :A].contiguous(), d_head[:, A:A + 1].contiguous()
        db = t["db"]                                       # packed bias order: lstm 256 | layer1 | layer2 | fc1 | heads 16
        loss = t["loss"].sum()
        net.p_bias.grad, net.v_bias.grad = db[1024:1024 + A], db[1024 + A:1025 + A]
        net.fc1_kernel.grad, net.fc1_bias.grad = xtg(t["z2"], t["g3"]), db[768:1024]
        net.layer2_kernel.grad, net.layer2_bias.grad = xtg(t["z1"], t["g2"]), db[512:768]
        d_l1 = xtg(t["l1_in"], t["g1"])                                     # rows: 64 hidden, 4 host, 4 padding
        net.layer1_kernel.grad = d_l1.index_select(0, self._l1_rows)
        net.layer1_bias.grad = db[256:512]
        gl = t["gl"].view(M * rows64, 4 * H)
        d_lstm = xtg(t["h_in"].view(M * rows64, 72), gl)                   # rows: 64 hidden, 7 inputs, 1 padding; packed gate columns
        net.lstm_kernel.grad = d_lstm.reshape(-1).index_select(0, self._lstm_flat).view(H + net.OTHER, 4 * H)
        net.lstm_bias.grad = db[:256].index_select(0, self._k_of_c)
        if self._base.distributed:
            self._base._allreduce_grads()
        self.opt.step()
        self._base.training_step += 1
        self._base.frame_counter += n
        pol.refresh(with_backward=True)                    # actors and the next training pass see the new weights
        return loss

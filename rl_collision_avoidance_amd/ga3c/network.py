"""``NetworkVP_rnn`` re-expressed in PyTorch-ROCm, so that the policy can sit
on the same device as the batched env (BASELINE configs[4]).

Graph (all citations /root/reference/ga3c/GA3C):
  x [B, NN_INPUT_SIZE] -> (x - avg) / std                                   NetworkVP_rnn.py:50-53
  num_other_agents = x[:, 0] (raw)                                          :58
  host = x_norm[:, 1:5]; others = x_norm[:, 5:].reshape(B, M, 7)            :59-61
  LSTMCell(64) over the M others with sequence_length = num_other_agents,
      final hidden state h (state frozen past each row's length)            :64-66
  concat[host(4), h(64)] -> dense256-relu 'layer1' -> dense256-relu 'layer2' :67,103-105
  -> dense256-relu 'fullyconnected1' -> logits_v (1), logits_p (A)          NetworkVPCore.py:66-75
  softmax_p = (softmax(logits_p) + MIN_POLICY) / (1 + MIN_POLICY * A)        :75
Loss (A3C with GA3C's epsilon; NetworkVPCore.py:71-100), optimiser Adam(lr = LEARNING_RATE_RL_START).

Parameters are stored in TensorFlow's layout (dense kernels [in, out]; the LSTM kernel [7+64, 4*64] with
gate order i, j, f, o and forget_bias = 1 added at run time, as tf.contrib.rnn.LSTMCell does) and
``TF_VARIABLE_NAMES`` maps them to the reference checkpoint's variable names (the Saver is keyed by
``var.name``, NetworkVPCore.py:56-57), so a TF1 checkpoint converts by plain assignment.
The dense layers are library GEMMs (rocBLAS/hipBLASLt through torch): tiny and dense, not a kernel target.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

TF_VARIABLE_NAMES = {
    "lstm_kernel": "rnn/lstm_cell/kernel:0", "lstm_bias": "rnn/lstm_cell/bias:0",
    "other_kernel": "other_agent_layer1/kernel:0", "other_bias": "other_agent_layer1/bias:0",
    "layer1_kernel": "layer1/kernel:0", "layer1_bias": "layer1/bias:0",
    "layer2_kernel": "layer2/kernel:0", "layer2_bias": "layer2/bias:0",
    "fc1_kernel": "fullyconnected1/kernel:0", "fc1_bias": "fullyconnected1/bias:0",
    "v_kernel": "logits_v/kernel:0", "v_bias": "logits_v/bias:0",
    "p_kernel": "logits_p/kernel:0", "p_bias": "logits_p/bias:0",
}


class _SplitKLinear(torch.autograd.Function):
    """``x @ w (+ b)`` whose weight gradient ``x^T @ g`` is computed split-K: the trainer's batches are tens of
    thousands of rows against 256-wide layers, so the plain library GEMM for [in, B] x [B, out] launches
    (in/32) x (out/64) = 32 workgroups on a 256-CU part and runs at ~12 % of it (140 us per layer at B = 32768).
    Slicing B into S batched GEMMs of ~2048 rows each lets the library pick a full-size tile and fill the chip
    (measured at [256, 32768] x [32768, 256]: plain 141 us, S = 64: 129, S = 256: 62, S = 16: 45); the S partial
    [in, out] products are summed afterwards."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, w) if b is not None else x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w.t() if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            B = x.shape[0]
            S = split_k_factor(B)
            if S > 1 and x.is_cuda:
                gw = torch.bmm(x.reshape(S, B // S, -1).transpose(1, 2), g.view(S, B // S, -1)).sum(dim=0)
            else:
                gw = x.t() @ g
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=0)
        return gx, gw, gb


def split_k_factor(rows: int, target: int = 2048) -> int:
    """Number of row slices for a split-K weight-gradient GEMM: the largest divisor of ``rows`` that keeps >= ``target``
    rows per slice (1 if ``rows`` is small or awkward)."""
    for s in range(rows // target, 1, -1):
        if rows % s == 0:
            return s
    return 1


def _linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _SplitKLinear.apply(x, w, b)


def _glorot(shape, gen) -> torch.Tensor:
    limit = float(np.sqrt(6.0 / (shape[0] + shape[1])))
    return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * limit


class NetworkVP_rnn(nn.Module):
    HOST, OTHER, HIDDEN, WIDTH = 4, 7, 64, 256

    def __init__(self, config, num_actions: Optional[int] = None, seed: int = 0, arch: Optional[str] = None):
        super().__init__()
        # MULTI_AGENT_ARCH (Config.py:43-49): 'rnn' (LSTM over the others) or 'weight_sharing' (one shared dense filter
        # per observed agent, NetworkVP_rnn.py:69-92)
        if arch is None:
            ws = getattr(config, "MULTI_AGENT_ARCH_WEIGHT_SHARING", None)
            arch = "weight_sharing" if ws is not None and getattr(config, "MULTI_AGENT_ARCH", None) == ws else "rnn"
        if arch not in ("rnn", "weight_sharing"):
            raise ValueError("arch must be 'rnn' or 'weight_sharing'")
        self.arch = arch
        self.num_actions = int(num_actions if num_actions is not None else getattr(config, "NUM_ACTIONS", 11))
        self.max_others = int(config.MAX_NUM_OTHER_AGENTS_OBSERVED)
        self.input_size = 1 + self.HOST + self.OTHER * self.max_others
        self.min_policy = float(getattr(config, "MIN_POLICY", 0.0))
        self.log_epsilon = float(getattr(config, "LOG_EPSILON", 1e-6))
        self.beta = float(getattr(config, "BETA_START", 1e-4))
        self.normalize = bool(getattr(config, "NORMALIZE_INPUT", True))
        avg = getattr(config, "NN_INPUT_AVG_VECTOR", None)
        std = getattr(config, "NN_INPUT_STD_VECTOR", None)
        if avg is None or len(avg) != self.input_size:      # plain EnvConfig: derive them like Config.py:64-71
            avg, std = input_normalisation(config)
        self.register_buffer("avg", torch.as_tensor(np.asarray(avg), dtype=torch.float32))
        self.register_buffer("std", torch.as_tensor(np.asarray(std), dtype=torch.float32))
        g = torch.Generator().manual_seed(seed)
        H, Wd = self.HIDDEN, self.WIDTH
        # TF gate order (i, j, f, o) -> ATen fused-cell order (i, f, g, o)
        self.register_buffer("gate_perm", torch.cat([torch.arange(0, H), torch.arange(2 * H, 3 * H), torch.arange(H, 2 * H),
                                                     torch.arange(3 * H, 4 * H)]), persistent=False)
        if self.arch == "rnn":
            self.lstm_kernel = nn.Parameter(_glorot((self.OTHER + H, 4 * H), g))
            self.lstm_bias = nn.Parameter(torch.zeros(4 * H))
            summary = H
        else:
            self.other_kernel = nn.Parameter(_glorot((self.OTHER + 1, H), g))       # 'other_agent_layer1', shared by all slots
            self.other_bias = nn.Parameter(torch.zeros(H))
            summary = H * self.max_others
        self.layer1_kernel = nn.Parameter(_glorot((self.HOST + summary, Wd), g)); self.layer1_bias = nn.Parameter(torch.zeros(Wd))
        self.layer2_kernel = nn.Parameter(_glorot((Wd, Wd), g)); self.layer2_bias = nn.Parameter(torch.zeros(Wd))
        self.fc1_kernel = nn.Parameter(_glorot((Wd, Wd), g)); self.fc1_bias = nn.Parameter(torch.zeros(Wd))
        self.v_kernel = nn.Parameter(_glorot((Wd, 1), g)); self.v_bias = nn.Parameter(torch.zeros(1))
        self.p_kernel = nn.Parameter(_glorot((Wd, self.num_actions), g)); self.p_bias = nn.Parameter(torch.zeros(self.num_actions))

    # ---------------------------------------------------------------------------------------------
    def _lstm_final_h(self, seq: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """tf.nn.dynamic_rnn(LSTMCell(64), seq, sequence_length=lengths)[1].h : rows stop updating once
        their own length is reached (zero state for length 0)."""
        B = seq.shape[0]
        h = seq.new_zeros((B, self.HIDDEN))
        c = seq.new_zeros((B, self.HIDDEN))
        for t in range(self.max_others):
            gates = torch.addmm(self.lstm_bias, torch.cat([seq[:, t, :], h], dim=1), self.lstm_kernel)
            i, j, f, o = gates.chunk(4, dim=1)
            c_new = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
            h_new = torch.sigmoid(o) * torch.tanh(c_new)
            live = (lengths > t).unsqueeze(1)
            c = torch.where(live, c_new, c)
            h = torch.where(live, h_new, h)
        return h

    def _weight_sharing_summary(self, others: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """NetworkVP_rnn.py:69-92: every observed-agent slot goes through the SAME dense(8 -> 64, relu) filter
        (its 7 features + an 'is this slot filled' flag); the M outputs are concatenated."""
        B, M = others.shape[0], self.max_others
        slot = torch.arange(1, M + 1, device=others.device, dtype=lengths.dtype)
        is_on = (lengths.unsqueeze(1) >= slot).to(others.dtype).unsqueeze(2)              # [B, M, 1]
        inp = torch.cat([others, is_on], dim=2).reshape(B * M, self.OTHER + 1)
        return torch.relu(torch.addmm(self.other_bias, inp, self.other_kernel)).reshape(B, M * self.HIDDEN)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """x [B, NN_INPUT_SIZE] -> (logits_p [B, A], softmax_p [B, A], v [B])"""
        x = x.to(torch.float32)
        xn = (x - self.avg) / self.std if self.normalize else x
        lengths = x[:, 0]
        host = xn[:, 1:1 + self.HOST]
        others = xn[:, 1 + self.HOST:].reshape(-1, self.max_others, self.OTHER)
        if self.arch != "rnn":
            h = self._weight_sharing_summary(others, lengths)
        elif x.is_cuda:          # same recurrence through ATen's fused (and differentiable) LSTM-cell kernel: the trainer's
            h = self._lstm_final_h_fused(others, lengths)      # forward + backward drop ~25 pointwise launches per step
        else:
            h = self._lstm_final_h(others, lengths)
        z = torch.relu(_linear(torch.cat([host, h], dim=1), self.layer1_kernel, self.layer1_bias))
        z = torch.relu(_linear(z, self.layer2_kernel, self.layer2_bias))
        z = torch.relu(_linear(z, self.fc1_kernel, self.fc1_bias))
        v = _linear(z, self.v_kernel, self.v_bias).squeeze(1)
        logits = _linear(z, self.p_kernel, self.p_bias)
        p = (torch.softmax(logits, dim=1) + self.min_policy) / (1.0 + self.min_policy * self.num_actions)
        return logits, p, v

    def _lstm_final_h_fused(self, seq: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """Same recurrence on the GPU inference path: ONE input-projection GEMM for all M steps, then per step one
        [B,64]x[64,256] GEMM and ATen's fused LSTM-cell kernel (gate order i,f,g,o: the TF-layout columns are
        permuted on the fly and the forget bias folded into the bias) -- ~3 kernels per step instead of ~12."""
        H, B, M = self.HIDDEN, seq.shape[0], self.max_others
        perm = self.gate_perm
        w = self.lstm_kernel.index_select(1, perm)
        bias = self.lstm_bias.index_select(0, perm).clone()
        bias[H:2 * H] += 1.0                                   # tf.contrib.rnn.LSTMCell forget_bias
        xproj = _linear(seq.reshape(B * M, self.OTHER), w[:self.OTHER], bias).view(B, M, 4 * H)
        wh = w[self.OTHER:]
        h = seq.new_zeros((B, H))
        c = seq.new_zeros((B, H))
        for t in range(M):
            h_new, c_new, _ = torch.ops.aten._thnn_fused_lstm_cell(xproj[:, t].contiguous(), _linear(h, wh), c)
            live = (lengths > t).unsqueeze(1)
            c = torch.where(live, c_new, c)
            h = torch.where(live, h_new, h)
        return h

    @torch.no_grad()
    def predict_p_and_v(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """NetworkVPCore.predict_p_and_v (:175-176)."""
        if not x.is_cuda:
            _, p, v = self.forward(x)
            return p, v
        x = x.to(torch.float32)
        xn = (x - self.avg) / self.std if self.normalize else x
        host = xn[:, 1:1 + self.HOST]
        others = xn[:, 1 + self.HOST:].reshape(-1, self.max_others, self.OTHER)
        h = self._lstm_final_h_fused(others, x[:, 0]) if self.arch == "rnn" else self._weight_sharing_summary(others, x[:, 0])
        z = torch.relu_(torch.addmm(self.layer1_bias, torch.cat([host, h], dim=1), self.layer1_kernel))
        z = torch.relu_(torch.addmm(self.layer2_bias, z, self.layer2_kernel))
        z = torch.relu_(torch.addmm(self.fc1_bias, z, self.fc1_kernel))
        v = torch.addmm(self.v_bias, z, self.v_kernel).squeeze(1)
        p = torch.softmax(torch.addmm(self.p_bias, z, self.p_kernel), dim=1)
        if self.min_policy != 0.0:
            p = (p + self.min_policy) / (1.0 + self.min_policy * self.num_actions)
        return p, v

    def loss(self, x: torch.Tensor, y_r: torch.Tensor, a_onehot: torch.Tensor):
        """cost_all = cost_p + cost_v of NetworkVPCore.py:71-100 (sums over the batch, not means)."""
        _, p, v = self.forward(x)
        y_r = y_r.to(torch.float32)
        cost_v = 0.5 * torch.sum((y_r - v) ** 2)
        selected = torch.sum(p * a_onehot, dim=1)
        advant = torch.log(torch.clamp_min(selected, self.log_epsilon)) * (y_r - v.detach())
        entropy = -self.beta * torch.sum(torch.log(torch.clamp_min(p, self.log_epsilon)) * p, dim=1)
        cost_p = -(advant.sum() + entropy.sum())
        return cost_p + cost_v, cost_p, cost_v


def export_tf_variables(net: "NetworkVP_rnn") -> dict:
    """{TensorFlow variable name: float32 ndarray} in the reference checkpoint's names and layouts (what
    ``tf.train.load_checkpoint(path).get_tensor(name)`` returns there): dense kernels [in, out], LSTM kernel [7+64, 256]."""
    return {TF_VARIABLE_NAMES[n]: p.detach().cpu().numpy().astype(np.float32).copy() for n, p in net.named_parameters()}


def load_tf_variables(net: "NetworkVP_rnn", variables: dict, strict: bool = True) -> list:
    """Assign the arrays of a reference checkpoint (dumped as {variable name: array}, e.g. an .npz written from
    ``tf.train.load_checkpoint``) to the module.  Returns the names it did not find (raises if ``strict``)."""
    missing = []
    with torch.no_grad():
        for n, p in net.named_parameters():
            tf_name = TF_VARIABLE_NAMES[n]
            key = tf_name if tf_name in variables else tf_name.split(":")[0]
            if key not in variables:
                missing.append(tf_name)
                continue
            arr = torch.as_tensor(np.asarray(variables[key]), dtype=p.dtype)
            if tuple(arr.shape) != tuple(p.shape):
                raise ValueError("%s: checkpoint shape %s, module shape %s" % (tf_name, tuple(arr.shape), tuple(p.shape)))
            p.copy_(arr.to(p.device))
    if missing and strict:
        raise KeyError("variables missing from the checkpoint: %s" % ", ".join(missing))
    return missing


def input_normalisation(config):
    """NN_INPUT_AVG_VECTOR / NN_INPUT_STD_VECTOR exactly as Config.py:64-71 builds them."""
    avg, std = [], []
    for state in config.STATES_IN_OBS:
        if state in config.STATES_NOT_USED_IN_POLICY:
            continue
        avg.append(np.asarray(config.STATE_INFO_DICT[state]["mean"]).flatten())
        std.append(np.asarray(config.STATE_INFO_DICT[state]["std"]).flatten())
    return np.hstack(avg), np.hstack(std)


class A3CTrainer(object):
    """``Server.train_model`` (Server.py:114-124) without the queue: one Adam step per batch.

    Multi-GPU (one process per GPU, a policy replica each): pass ``group`` (or rely on the default
    process group) and every step sums the replicas' gradients with ONE all-reduce of a flat buffer
    -- the model is 0.6 MB, one bucket is the right bucketing for the per-link-bound xGMI rings -- so
    the replicas take the identical Adam step a single trainer would take on the concatenated batch
    (the A3C loss is a SUM over rows, NetworkVPCore.py:71-100)."""

    def __init__(self, model: NetworkVP_rnn, learning_rate: float = 2e-5, group=None, distributed: Optional[bool] = None):
        import torch.distributed as dist
        self.model = model
        on_gpu = next(model.parameters()).is_cuda
        self.opt = torch.optim.Adam(model.parameters(), lr=learning_rate, eps=1e-8,     # tf.train.AdamOptimizer defaults
                                    fused=True if on_gpu else None)                     # one kernel for all 12 variables
        self.training_step = 0
        self.frame_counter = 0
        self.group = group
        self.distributed = (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1) \
            if distributed is None else distributed
        self._params = [p for p in model.parameters()]
        self._flat = None

    def _allreduce_grads(self) -> None:
        import torch.distributed as dist
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self._params]
        if self._flat is None:
            self._flat = torch.empty(sum(g.numel() for g in grads), dtype=grads[0].dtype, device=grads[0].device)
        torch.cat([g.reshape(-1) for g in grads], out=self._flat)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p, g in zip(self._params, grads):
            n = g.numel()
            p.grad = self._flat[off:off + n].view_as(p).clone()
            off += n

    def train(self, x: torch.Tensor, y_r: torch.Tensor, a_onehot: torch.Tensor) -> float:
        self.opt.zero_grad(set_to_none=True)
        total, _, _ = self.model.loss(x, y_r, a_onehot)
        total.backward()
        if self.distributed:
            self._allreduce_grads()
        self.opt.step()
        self.training_step += 1
        self.frame_counter += int(x.shape[0])
        return float(total.detach())

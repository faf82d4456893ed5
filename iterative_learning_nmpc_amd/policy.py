"""Host mirror of the reference's policy network and behaviour-cloning step on the device.

    GoalConditionedPolicyNet(input_size, output_size, num_hidden_layer, hidden_dim, batch_norm)
        DAgger/utils/network.py:7-81          -> DevicePolicy (forward = eval mode)
    BehavioralCloning.train_network, inner loop   DAgger/utils/train_locosafedagger.py:93-102
        -> DevicePolicy.train_step(x, y, lr)   (L1 loss, Adam)

The kernels are in csrc/nmpc_policy.hip behind include/nmpc_policy.h; tensors stay on the device
(torch is the container only).  There is no CPU path."""
import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def parameter_layout(n_in, n_out, n_hidden, hidden, batch_norm):
    """[(name, shape, offset)] of the flat parameter vector (the order of net.parameters())."""
    items, off = [], 0
    for l in range(n_hidden):
        fan_in = n_in if l == 0 else hidden
        names = [(f"net.{l}.W", (hidden, fan_in)), (f"net.{l}.b", (hidden,))]
        if batch_norm:
            names += [(f"net.{l}.gamma", (hidden,)), (f"net.{l}.beta", (hidden,))]
        for name, shape in names:
            items.append((name, shape, off)); off += int(np.prod(shape))
    for name, shape in ((f"net.{n_hidden}.W", (n_out, hidden)), (f"net.{n_hidden}.b", (n_out,))):
        items.append((name, shape, off)); off += int(np.prod(shape))
    return items, off


class DevicePolicy:
    """The reference's MLP policy with its parameters, BatchNorm buffers and Adam state on one GPU."""

    def __init__(self, input_size: int, output_size: int, num_hidden_layer: int = 3, hidden_dim: int = 512,
                 batch_norm: bool = True, batch_max: int = 1024, device="cuda:0", seed: Optional[int] = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePolicy needs a HIP device; there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.dims = (int(input_size), int(output_size), int(num_hidden_layer), int(hidden_dim), bool(batch_norm))
        self.batch_max = int(batch_max)
        d = _lib.NmpcPolicyDims(*self.dims[:4], int(batch_norm), self.batch_max)
        self._h = ctypes.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.nmpc_policy_create(ctypes.byref(d), idx, ctypes.byref(self._h))
        if rc:
            raise _lib.NmpcError(f"nmpc_policy_create: {self.lib.nmpc_policy_last_error(None).decode()}")
        self.items, self.n_theta = parameter_layout(*self.dims)
        assert self.n_theta == self.lib.nmpc_policy_param_count(self._h)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        if seed is not None:
            self.init_weight(seed)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.nmpc_policy_destroy(h)
            self._h = None

    def _check(self, rc, what):
        if rc:
            raise _lib.NmpcError(f"{what}: {self.lib.nmpc_policy_last_error(self._h).decode()}")

    # -- parameters ---------------------------------------------------------------------------------
    def init_weight(self, seed: int = 0):
        """Kaiming-normal weights (fan_in, relu), zero biases, gamma = 1, beta = 0 (network.py:59-70)."""
        n_in, n_out, L, hidden, bn = self.dims
        g = torch.Generator().manual_seed(seed)
        theta = torch.zeros(self.n_theta)
        for name, shape, off in self.items:
            n = int(np.prod(shape))
            if name.endswith(".W"):
                theta[off:off + n] = (torch.randn(shape, generator=g) * (2.0 / shape[1]) ** 0.5).reshape(-1)
            elif name.endswith(".gamma"):
                theta[off:off + n] = 1.0
        self.set_parameters(theta, torch.zeros(L, hidden), torch.ones(L, hidden))

    def set_parameters(self, theta, running_mean=None, running_var=None):
        n_in, n_out, L, hidden, bn = self.dims
        dev = lambda t: torch.as_tensor(t, dtype=torch.float32).contiguous().to(self.device)
        theta = dev(theta)
        assert theta.numel() == self.n_theta
        rm = dev(running_mean) if bn else None
        rv = dev(running_var) if bn else None
        self._check(self.lib.nmpc_policy_set_params(self._h, _ptr(theta), _ptr(rm), _ptr(rv), _stream(self.device)),
                    "nmpc_policy_set_params")
        torch.cuda.current_stream(self.device).synchronize()      # the staging tensors go out of scope

    def get_parameters(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        n_in, n_out, L, hidden, bn = self.dims
        theta = torch.empty(self.n_theta, dtype=torch.float32, device=self.device)
        # (without BatchNorm there are no running statistics: the library leaves these two as they are -- zeros / ones,
        #  the state of a fresh BatchNorm layer, rather than uninitialised memory)
        rm = torch.zeros(L, hidden, dtype=torch.float32, device=self.device)
        rv = torch.ones(L, hidden, dtype=torch.float32, device=self.device)
        self._check(self.lib.nmpc_policy_get_params(self._h, _ptr(theta), _ptr(rm), _ptr(rv), _stream(self.device)),
                    "nmpc_policy_get_params")
        return theta, rm, rv

    def load_state_dict(self, state: Dict[str, "np.ndarray"]):
        """Parameters from the reference's state_dict ('net.<i>.weight', BatchNorm buffers ...)."""
        n_in, n_out, L, hidden, bn = self.dims
        theta = np.zeros(self.n_theta, np.float32)
        view = {n: theta[o:o + int(np.prod(s))].reshape(s) for n, s, o in self.items}
        rm, rv = np.zeros((L, hidden), np.float32), np.ones((L, hidden), np.float32)
        idx = 0
        for l in range(L + 1):
            view[f"net.{l}.W"][:] = np.asarray(state[f"net.{idx}.weight"]); view[f"net.{l}.b"][:] = np.asarray(state[f"net.{idx}.bias"])
            idx += 1
            if l < L:
                if bn:
                    view[f"net.{l}.gamma"][:] = np.asarray(state[f"net.{idx}.weight"]); view[f"net.{l}.beta"][:] = np.asarray(state[f"net.{idx}.bias"])
                    rm[l] = np.asarray(state[f"net.{idx}.running_mean"]); rv[l] = np.asarray(state[f"net.{idx}.running_var"])
                    idx += 1
                idx += 1                                  # ReLU
        self.set_parameters(theta, rm, rv)

    # -- forward / training ---------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """network.eval(); network(x)"""
        B = x.shape[0]
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape == (B, self.dims[0])
        y = torch.empty(B, self.dims[1], dtype=torch.float32, device=self.device)
        self._check(self.lib.nmpc_policy_forward(self._h, B, _ptr(x), _ptr(y), _stream(self.device)), "nmpc_policy_forward")
        return y

    __call__ = forward

    def train_step(self, x: torch.Tensor, y: torch.Tensor, lr: float = 1e-3, return_pred: bool = False):
        """optimizer.zero_grad(); loss = L1Loss(network(x), y); loss.backward(); optimizer.step().
        Returns the loss as a device scalar (and the train-mode prediction)."""
        B = x.shape[0]
        assert x.is_cuda and y.is_cuda and x.dtype == y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
        assert x.shape == (B, self.dims[0]) and y.shape == (B, self.dims[1])
        pred = torch.empty(B, self.dims[1], dtype=torch.float32, device=self.device) if return_pred else None
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        self._check(self.lib.nmpc_policy_train_step(self._h, B, _ptr(x), _ptr(y), float(lr), _ptr(loss), _ptr(pred),
                                                    _stream(self.device)), "nmpc_policy_train_step")
        return (loss, pred) if return_pred else loss


def weighted_sample(weights: torch.Tensor, num_samples: int, seed: int) -> torch.Tensor:
    """WeightedRandomSampler(weights, num_samples, replacement=True) on device weights
    (test_train_policy.py:128-134): int32 indices [num_samples], reproducible for a seed."""
    lib = _lib.load()
    w = weights.reshape(-1)
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    n = w.numel()
    scratch = torch.empty(n + n // 2048 + 2, dtype=torch.float64, device=w.device)
    idx = torch.empty(num_samples, dtype=torch.int32, device=w.device)
    rc = lib.nmpc_weighted_sample(_ptr(w), n, int(num_samples), int(seed) & (2 ** 64 - 1), _ptr(scratch), _ptr(idx), _stream(w.device))
    if rc:
        raise _lib.NmpcError(f"nmpc_weighted_sample: {lib.nmpc_policy_last_error(None).decode()}")
    return idx


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Batch assembly behind the sampler: src[idx] for a [rows, features] fp32 table."""
    lib = _lib.load()
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 2
    assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()
    dst = torch.empty(idx.numel(), src.shape[1], dtype=torch.float32, device=src.device)
    rc = lib.nmpc_gather_rows(_ptr(src), src.shape[0], src.shape[1], _ptr(idx), idx.numel(), _ptr(dst), _stream(src.device))
    if rc:
        raise _lib.NmpcError(f"nmpc_gather_rows: {lib.nmpc_policy_last_error(None).decode()}")
    return dst

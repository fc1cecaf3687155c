"""Model base (reference core/model.py:5-31) plus flat parameter/gradient
buffers: `flatten_()` re-points every Parameter (and its .grad) at a view of one
contiguous fp32 tensor, so the fused optimizer / Polyak kernels, the RCCL
gradient all-reduce and the weight publish each touch exactly one buffer.
state_dict keys and shapes are unchanged by flattening.
"""
import torch
import torch.nn as nn


class Model(nn.Module):
    def __init__(self, args, norm_val, input_dims, output_dims, action_dims):
        super().__init__()
        self.norm_val = norm_val
        self.input_dims = input_dims
        self.output_dims = output_dims
        self.action_dims = action_dims
        self._flat = None
        self._flat_grad = None

    def _init_weights(self):
        raise NotImplementedError("not implemented in base class")

    def _reset(self):   # children call this at the end of __init__
        self._init_weights()

    def forward(self, input):
        raise NotImplementedError("not implemented in base class")

    def get_action(self, input):
        raise NotImplementedError("not implemented in base class")

    # -- flat buffers ------------------------------------------------------
    def param_numel(self):
        return sum(p.numel() for p in self.parameters())

    def flatten_(self, with_grad=True):
        """Move all parameters into one flat buffer (idempotent per device)."""
        params = list(self.parameters())
        dev = params[0].device
        if self._flat is not None and self._flat.device == dev and \
                all(p.data_ptr() >= self._flat.data_ptr() and
                    p.data_ptr() < self._flat.data_ptr() + self._flat.numel() * 4 for p in params):
            return self._flat
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev) if with_grad else None
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view_as(p.data)
            if with_grad and p.requires_grad:
                p.grad = grad[off:off + n].view_as(p.data)
            off += n
        self._flat, self._flat_grad = flat, grad
        return flat

    def _apply(self, fn, *a, **k):      # .to()/.cuda() invalidates the views
        self._flat = None
        self._flat_grad = None
        return super()._apply(fn, *a, **k)

    @property
    def flat_params(self):
        if self._flat is None:
            self.flatten_()
        return self._flat

    @property
    def flat_grads(self):
        if self._flat is None:
            self.flatten_()
        return self._flat_grad

    def param_segments(self):
        """[(name, offset, shape)] in flat-buffer order."""
        out, off = [], 0
        for name, p in self.named_parameters():
            out.append((name, off, tuple(p.shape)))
            off += p.numel()
        return out

    def zero_flat_grad(self):
        if self._flat_grad is not None:
            self._flat_grad.zero_()
        else:
            for p in self.parameters():
                p.grad = None

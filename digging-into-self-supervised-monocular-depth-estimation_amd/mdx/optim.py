"""The optimiser of the training step: torch.optim.Adam(fused=True) with its step as ONE launch of csrc/adam.hip.

Same class from the outside (reference model_tool/loader.py:93-97 builds torch.optim.Adam): parameter groups, `state_dict()` /
`load_state_dict()` (per-parameter "step", "exp_avg", "exp_avg_sq"), capturable learning-rate tensors (model_train.graphed_step).
What the kernel does not cover -- weight decay, amsgrad, maximize, a closure, a GradScaler's grad_scale / found_inf, parameters that
are not dense float32 GPU tensors -- goes through torch's own step."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, stream


class _Table(C.Structure):
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p), ("n", C.c_int64)]


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, **kw):
        kw.setdefault("fused", True)
        super().__init__(params, lr, **kw)
        self._plans = {}

    # -- what one launch needs, built once per parameter group -------------------------------------------------------------
    def _plan(self, gi, params, exp_avgs, exp_avg_sqs, steps):
        key = tuple(t.data_ptr() for ts in (params, exp_avgs, exp_avg_sqs, steps) for t in ts)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        assert C.sizeof(_Table) == lib().mdx_adam_table_entry_bytes()
        dev = params[0].device
        n = len(params)
        host = (_Table * n)()
        for i, (p, m, v, s) in enumerate(zip(params, exp_avgs, exp_avg_sqs, steps)):
            host[i] = _Table(p.data_ptr(), m.data_ptr(), v.data_ptr(), s.data_ptr(), p.numel())
        raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).clone()
        table = raw.to(dev)
        chunk, per = lib().mdx_adam_chunk(), lib().mdx_adam_max_tensors()
        launches = []
        for first in range(0, n, per):
            count = min(per, n - first)
            bm = []
            for t in range(count):
                bm += [(t, c) for c in range((params[first + t].numel() + chunk - 1) // chunk)]
            blockmap = torch.tensor(bm, dtype=torch.int32).to(dev)
            launches.append((first, count, blockmap, len(bm)))
        plan = dict(key=key, table=table, launches=launches)
        self._plans[gi] = plan
        return plan

    @staticmethod
    def _dense(a, b):
        return a.stride() == b.stride() and a.shape == b.shape

    def _native_ok(self, group, params, grads, exp_avgs, exp_avg_sqs, steps):
        if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"] or group.get("differentiable"):
            return False
        if getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            return False
        dev = params[0].device
        for p, g, m, v, s in zip(params, grads, exp_avgs, exp_avg_sqs, steps):
            if not (p.is_cuda and p.device == dev and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse
                    and (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last))
                    and self._dense(g, p) and self._dense(m, p) and self._dense(v, p)
                    and torch.is_tensor(s) and s.is_cuda and s.dtype == torch.float32 and s.numel() == 1):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            return super().step(closure)
        todo = []
        for gi, group in enumerate(self.param_groups):
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            if not self._native_ok(group, params, grads, exp_avgs, exp_avg_sqs, steps):
                return super().step()
            todo.append((gi, group, params, grads, exp_avgs, exp_avg_sqs, steps))
        for gi, group, params, grads, exp_avgs, exp_avg_sqs, steps in todo:
            if params[0].device.index != torch.cuda.current_device():
                raise _lib.MdxError("mdx.optim.Adam: parameters live on %s but the current device is cuda:%d"
                                    % (params[0].device, torch.cuda.current_device()))
            plan = self._plan(gi, params, exp_avgs, exp_avg_sqs, steps)
            torch._foreach_add_(steps, 1)
            lr = group["lr"]
            lr_ptr = C.c_void_p(lr.data_ptr()) if torch.is_tensor(lr) else None
            if torch.is_tensor(lr) and not (lr.is_cuda and lr.dtype == torch.float32):
                lr, lr_ptr = float(lr), None
            beta1, beta2 = group["betas"]
            for first, count, blockmap, nblocks in plan["launches"]:
                garr = (C.c_void_p * count)(*[g.data_ptr() for g in grads[first:first + count]])
                check(lib().mdx_adam_step(C.c_void_p(plan["table"].data_ptr()), first, count, garr,
                                          C.c_void_p(blockmap.data_ptr()), nblocks, lr_ptr,
                                          C.c_double(0.0 if lr_ptr is not None else float(lr)), C.c_double(beta1), C.c_double(beta2),
                                          C.c_double(group["eps"]), stream()), "mdx_adam_step")
        return None

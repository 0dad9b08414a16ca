"""Whole-model training step replayed from ONE HIP graph (multi-layer step).

A step of the reference's trainer (engineer/trainer/trainer.py:204-227) is
    optimizer.zero_grad(); loss, _ = model(batch); loss.backward(); optimizer.step()
i.e. for the hulls model ~3 x 4 fused CEMLP launches between ~150 small PyTorch launches (embedding
gathers, attribute concatenation, pooling, Adam). On complexes of the reference's sizes (a few thousand
simplices per batch) the GPU work of a step is shorter than the host time to launch it, so the step is
launch-bound. For batches of FIXED topology (same complexes, new vertex features / targets: MD17
trajectories of one molecule, repeated epochs over cached batches) this module captures the whole step -
embedding, every EGCL layer forward and backward, readout, loss and the optimizer update - once and replays
it; per step the host copies the new features into the captured buffers and launches one graph.

The kernels launched inside are the same C-ABI calls as in eager mode (they take the current stream); the
CSR and the embedding index tables are built before capture (their only host round trips).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch


def flatten_parameters(module: torch.nn.Module) -> torch.nn.Parameter:
    """Re-homes the module's trainable float32 parameters - and their .grad - as views of ONE flat buffer each and returns the
    flat parameter (its .grad is the flat gradient buffer). An optimizer built over `[flat]` then updates every parameter of
    the module with ONE fused launch and zeroes all gradients with one fill: the task models carry 700-1 100 small parameter
    tensors (226 per EGCL layer), which torch's multi-tensor Adam walks in 5-7 launches of ~9 us each, its zero_grad in 3.
    Names, shapes and state_dict() of the module are unchanged (the parameters stay the module's own objects; only their
    storage moves). Elementwise optimizers only (Adam, SGD, ...): per-tensor statistics would see one tensor.

    ZEROING THE GRADIENTS: the aliasing lives in the .grad attributes. `optimizer.zero_grad()` / `module.zero_grad()` with
    PyTorch's default `set_to_none=True` REPLACE them - the optimizer then sees `flat.grad is None` and skips the only
    parameter it has while the module's gradients accumulate elsewhere: training stops silently. Use
    `zero_flat_grad(flat)` (one fill) or `zero_grad(set_to_none=False)`; `flat_gradients_intact(flat)` tells whether the
    views still alias (GraphedTrainStep checks it before it captures)."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        raise ValueError("module has no trainable parameters")
    dev, dt = params[0].device, params[0].dtype
    if any(p.device != dev or p.dtype != dt for p in params):
        raise ValueError("flatten_parameters: all trainable parameters must share device and dtype")
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4      # 16-byte aligned starts
    flat = torch.zeros(total, dtype=dt, device=dev)
    gflat = torch.zeros(total, dtype=dt, device=dev)
    with torch.no_grad():
        for p, o in zip(params, offs):
            n = p.numel()
            flat[o:o + n].copy_(p.detach().reshape(-1))
            if p.grad is not None:
                gflat[o:o + n].copy_(p.grad.reshape(-1))
            p.data = flat[o:o + n].view(p.shape)
            p.grad = gflat[o:o + n].view(p.shape)
    flat_param = torch.nn.Parameter(flat)
    flat_param.grad = gflat
    flat_param._csmpn_flat = (gflat, params, offs)
    return flat_param


def zero_flat_grad(flat_param: torch.nn.Parameter) -> None:
    """Zero every gradient of a flatten_parameters() module with one fill, keeping the views (the safe zero_grad)."""
    flat_param._csmpn_flat[0].zero_()


def flat_gradients_intact(flat_param: torch.nn.Parameter) -> bool:
    """True while flat.grad is the flat gradient buffer and every parameter's .grad is still its view of it (False after a
    zero_grad(set_to_none=True) on the optimizer or the module)."""
    gflat, params, offs = flat_param._csmpn_flat
    if flat_param.grad is not gflat:
        return False
    esz = gflat.element_size()
    return all(p.grad is not None and p.grad.data_ptr() == gflat.data_ptr() + o * esz for p, o in zip(params, offs))


class GraphedTrainStep:
    """step(features) -> loss (a device scalar that the next replay overwrites).

    model(batch) must return (loss, aux) like the reference's task models; `optimizer` must be
    graph-capturable (torch.optim.Adam(..., capturable=True)); `batch` is a device SimplicialBatch
    whose tensors become the captured input buffers. `feature_names`: the batch attributes that change
    from step to step (everything else - topology, types - is part of the captured graph)."""

    def __init__(self, model, optimizer, batch, feature_names: Iterable[str], warmup: int = 2, max_dim: Optional[int] = None):
        dev = batch.edge_index.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs a device batch (no CPU fallback)")
        self.model, self.optimizer, self.batch = model, optimizer, batch
        self.feature_names = list(feature_names)
        for name in self.feature_names:
            if not hasattr(batch, name):
                raise AttributeError(f"batch has no feature {name!r}")
        batch.plan(getattr(model, "max_dim", 2) if max_dim is None else max_dim)
        batch.csr()
        # warm-up steps run eagerly on a side stream (allocator pools, lazily built workspaces, Adam state
        # tensors); parameters and optimizer state are restored afterwards, so the first replayed step is
        # step 1 of the trajectory
        params = [p for g in optimizer.param_groups for p in g["params"]]
        for p in params:
            if hasattr(p, "_csmpn_flat") and not flat_gradients_intact(p):
                raise RuntimeError("flatten_parameters(): the gradient views no longer alias the flat buffer (a zero_grad with "
                                   "set_to_none=True ran?) - the optimizer would skip every update; use zero_flat_grad()")
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        saved_params = [p.detach().clone() for p in params]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for p, s in zip(params, saved_params):
                p.copy_(s)
            # optimizer state (exp_avg, exp_avg_sq, step) exists now: reset it in place, the captured
            # graph updates these very tensors
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.aux = self._eager_step()
        # the capture itself executes nothing: parameters and optimizer state are still those of step 0

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=False)
        loss, aux = self.model(self.batch)
        loss.backward()
        self.optimizer.step()
        return loss, aux

    def load(self, features: Dict[str, torch.Tensor]):
        """Copy new feature tensors (same shapes) into the captured buffers."""
        for name, value in features.items():
            if name not in self.feature_names:
                raise KeyError(f"{name!r} was not declared as a changing feature")
            getattr(self.batch, name).copy_(value, non_blocking=True)

    def step(self, features: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        if features:
            self.load(features)
        self.graph.replay()
        return self.loss

"""Training-step plumbing for the pre-training hot path: flat parameter / gradient buffers, the
data-parallel gradient exchange (one RCCL all-reduce over xGMI per step), gradient clipping and
AdamW with the config's `norm` no-decay rule.

Reference behaviour being matched (external to its tree, mmcv/mmdet): MMDistributedDataParallel
(bucketed all-reduce, gradients AVERAGED over ranks), OptimizerHook(grad_clip=dict(max_norm=10,
norm_type=2)), AdamW(lr=1e-5, betas=(0.9,0.999), weight_decay=0.05, paramwise_cfg custom_keys
{'norm': decay_mult 0}) -- configs/_base_/schedules/cosine_2x.py:1-17.

Design for one node of 8 MI355X: the model has 2.76 M parameters (11 MB fp32), i.e. ONE bucket.
Parameters and gradients live in two flat fp32 buffers (every nn.Parameter is a view), so the
exchange is a single in-place all_reduce of the gradient buffer with no flatten/unflatten copies,
clipping is one norm over one buffer, and the optimizer is one fused update over two segments
(decayed / undecayed).  Host logic is device agnostic (tested with gloo on CPU, world_size 2).
"""
import torch
from torch import distributed as dist


class FlatParams:
    """Re-homes every parameter of `model` (and its .grad) into two contiguous fp32 buffers.
    Order: undecayed ('norm' in the name, or listed in no_decay_keys) first, then decayed."""

    def __init__(self, model, no_decay_keys=("norm",)):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        nd = [(n, p) for n, p in named if any(k in n for k in no_decay_keys)]
        dc = [(n, p) for n, p in named if not any(k in n for k in no_decay_keys)]
        self.names = [n for n, _ in nd + dc]
        self.params = [p for _, p in nd + dc]
        self.n_no_decay = sum(p.numel() for _, p in nd)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.grad[off:off + n].view_as(p.data)
            off += n
        self.total = total

    def zero_grad(self):
        self.grad.zero_()

    def check_views(self):
        """Autograd must have accumulated in place; re-point any .grad that was replaced."""
        if getattr(self, "_ptrs", None) is None:
            base, item, off, self._ptrs = self.grad.data_ptr(), self.grad.element_size(), 0, []
            for p in self.params:
                self._ptrs.append((base + off * item, off, p.numel()))
                off += p.numel()
        for p, (ptr, off, n) in zip(self.params, self._ptrs):
            g = p.grad
            if g is None:
                p.grad = self.grad[off:off + n].view_as(p.data)
            elif g.data_ptr() != ptr:
                view = self.grad[off:off + n]
                view.copy_(g.reshape(-1))
                p.grad = view.view_as(p.data)


def allreduce_gradients(flat, group=None):
    """DDP semantics: gradients averaged over ranks; one collective on the single bucket."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=group)
    flat.grad.mul_(1.0 / dist.get_world_size(group))


def clip_grad_norm(flat, max_norm, norm_type=2):
    """torch.nn.utils.clip_grad_norm_ on the flat buffer (mmcv OptimizerHook.clip_grads)."""
    assert norm_type == 2
    total = torch.linalg.vector_norm(flat.grad)
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
    flat.grad.mul_(coef)
    return total


class FlatAdamW:
    """AdamW over the flat buffers; weight decay only on the decayed segment.  Update rule of
    torch.optim.AdamW: p *= 1 - lr*wd ; m,v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)."""

    def __init__(self, flat, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0

    @torch.no_grad()
    def step(self):
        f = self.flat
        self.step_count += 1
        b1, b2 = self.betas
        g = f.grad
        if self.weight_decay != 0:
            f.flat[f.n_no_decay:].mul_(1 - self.lr * self.weight_decay)
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        denom = (self.exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        f.flat.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr)


class Trainer:
    """One process per GPU.  train_step = forward_train + backward + gradient exchange + clip + AdamW."""

    def __init__(self, model, optimizer_cfg=None, grad_clip=None):
        from .configs import GRAD_CLIP, OPTIMIZER
        ocfg = dict(optimizer_cfg or OPTIMIZER)
        assert ocfg.pop("type") == "AdamW"
        pw = ocfg.pop("paramwise_cfg", None) or {}
        keys = tuple(k for k, v in pw.get("custom_keys", {}).items() if v.get("decay_mult", 1.0) == 0.0)
        self.model = model
        self.flat = FlatParams(model, no_decay_keys=keys or ("\0",))
        self.opt = FlatAdamW(self.flat, **ocfg)
        self.grad_clip = dict(GRAD_CLIP if grad_clip is None else grad_clip)

    def train_step(self, points, next_points=None, **kw):
        """next_points: the batch of the FOLLOWING step (the same list object must be passed as `points`
        then); its voxelization / pillar sort is enqueued ahead of this step so that its count readback is
        off the critical path (detector.prefetch)."""
        self.flat.zero_grad()
        pre = getattr(self.model, "_prefetched", None)
        if pre is not None and pre[0] is points:
            pre[1][4].sync_counts()             # already landed: claim it before the next readback is queued
        keep = pre if (pre is not None and pre[0] is points) else None
        if next_points is not None and hasattr(self.model, "prefetch"):
            self.model.prefetch(next_points)
            nxt = self.model._prefetched
            self.model._prefetched = keep
            losses = self.model.forward_train(points, None, **kw)
            self.model._prefetched = nxt
        else:
            losses = self.model.forward_train(points, None, **kw)
        total = sum(losses.values())
        total.backward()
        self.flat.check_views()
        allreduce_gradients(self.flat)
        gnorm = clip_grad_norm(self.flat, **self.grad_clip)
        self.opt.step()
        return losses, gnorm

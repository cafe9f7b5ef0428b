"""torch-CPU (oneDNN) restatement of the ResNet-v1.5 training step — TEST / BASELINE INFRASTRUCTURE ONLY.

This is the "well-optimised CPU library" stand-in of BASELINE.md §4 / SURVEY.md §8d for the reference's `num_gpus=0`
TensorFlow-CPU path, which cannot run in this image (TensorFlow absent).  It is NOT TensorFlow and is labelled so
wherever it is reported.  Only bench.py's `cpu_baseline` leg and tests/ import it; the product never does.

Same step as oracle.net.train_step (the NumPy oracle of record) and as the HIP path: input prep (convnet.py:452-466),
conv with TF SAME pads (:1659), training-mode batch norm with biased normalisation and Bessel-corrected running update
(:1883-1914), ReLU, 3x3/2 max-pool, residual adds, global mean, fc, softmax-CE + L2 (:528-601), autograd of all of it,
Nesterov momentum (optimizers.py:676) and the EMA shadows (convnet.py:183, decay min(d, (1+t)/(10+t))).
tests/test_oracle_vs_torch.py::test_torch_cpu_step_matches_oracle pins it to the NumPy oracle.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _same(n, k, s, d=1):
    out = -(-n // s)
    tot = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return tot // 2, tot - tot // 2


class ResNetTorchCPU(object):
    """Parameters keyed by the reference's variable names; conv weights kept OIHW (oneDNN's layout), converted from / to
    the reference's HWIO at the edges."""

    def __init__(self, spec, params, stats, hp=None, channels_last=True):
        self.spec = spec
        self.hp = dict(image_mean=0.5, scale_factor=2.0, l2_reg=1e-4, momentum=0.9, base_learning_rate=0.1, moving_average_decay=0.99,
                       batch_norm_decay=0.99, eps=1e-3)
        self.hp.update(hp or {})
        self.cl = channels_last
        self.P, self.S = {}, {}
        for k, v in params.items():
            t = torch.as_tensor(np.asarray(v))
            if v.ndim == 4:
                t = t.permute(3, 2, 0, 1).contiguous()                 # HWIO -> OIHW
                if channels_last:
                    t = t.contiguous(memory_format=torch.channels_last)
            self.P[k] = t.clone().requires_grad_(True)
        for k, v in stats.items():
            self.S[k] = torch.as_tensor(np.asarray(v)).clone()
        self.accum = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.ema = {k: v.detach().clone() for k, v in self.P.items()}
        self.ema_stats = {k: v.clone() for k, v in self.S.items()}
        self.step = 0

    # ---- forward (NCHW tensors) -----------------------------------------------------------------------------------
    def _conv(self, h, name, s):
        w = self.P[name + '/weights']
        k = w.shape[-1]
        pt, pb = _same(h.shape[2], k, s)
        pl, pr = _same(h.shape[3], k, s)
        if pt == pb and pl == pr:
            return F.conv2d(h, w, None, s, (pt, pl))
        return F.conv2d(F.pad(h, (pl, pr, pt, pb)), w, None, s, 0)

    def _bn(self, h, name, batch_stats):
        # F.batch_norm(training=True): biased variance for normalisation; the running update is done by hand below with
        # TF's convention (Bessel-corrected variance, decay 0.99) so that it matches convnet.py:1898-1901
        g, b = self.P[name + '/gamma'], self.P[name + '/beta']
        y = F.batch_norm(h, None, None, g, b, True, 0.0, self.hp['eps'])
        with torch.no_grad():
            n = h.numel() // h.shape[1]
            var, mean = torch.var_mean(h, dim=(0, 2, 3), unbiased=False)
            batch_stats[name] = (mean, var * (n / max(n - 1, 1)))
        return y

    def forward(self, x_nhwc, batch_stats):
        sp = self.spec
        h = torch.as_tensor(x_nhwc).permute(0, 3, 1, 2)
        h = (h - self.hp['image_mean']) * self.hp['scale_factor']
        if self.cl:
            h = h.contiguous(memory_format=torch.channels_last)
        h = F.relu(self._bn(self._conv(h, 'block_0/conv_0', sp.strides[0]), 'block_0/conv_0/bn', batch_stats))
        pt, pb = _same(h.shape[2], 3, 2)
        pl, pr = _same(h.shape[3], 3, 2)
        h = F.max_pool2d(F.pad(h, (pl, pr, pt, pb), value=float('-inf')), 3, 2)
        ch = sp.channels
        cin = ch[0]
        for i in range(1, len(ch)):
            for j in range(sp.res_units[i]):
                s = sp.strides[i] if j == 0 else 1
                nm = 'block_{}/res_{}'.format(i, j)
                cout = ch[i]
                if cin == cout:
                    skip = F.max_pool2d(h, s, s) if s > 1 else h
                else:
                    skip = self._bn(self._conv(h, nm + '/conv_skip', s), nm + '/conv_skip/bn', batch_stats)
                if sp.bottleneck:
                    y = F.relu(self._bn(self._conv(h, nm + '/conv_0', 1), nm + '/conv_0/bn', batch_stats))
                    y = F.relu(self._bn(self._conv(y, nm + '/conv_1', s), nm + '/conv_1/bn', batch_stats))
                    y = self._bn(self._conv(y, nm + '/conv_2', 1), nm + '/conv_2/bn', batch_stats)
                else:
                    y = F.relu(self._bn(self._conv(h, nm + '/conv_0', s), nm + '/conv_0/bn', batch_stats))
                    y = self._bn(self._conv(y, nm + '/conv_1', 1), nm + '/conv_1/bn', batch_stats)
                h = F.relu(y + skip)
                cin = cout
        h = h.mean(dim=(2, 3))
        return h @ self.P['block_None/logits/weights'] + self.P['block_None/logits/biases']

    # ---- one optimisation step ----------------------------------------------------------------------------------------
    def train_step(self, x_nhwc, y_ids, batch_total=None, lr_mult=1.0):
        hp = self.hp
        bstats = {}
        logits = self.forward(x_nhwc, bstats)
        ids = torch.as_tensor(np.asarray(y_ids)).long()
        ce = F.cross_entropy(logits.float() if logits.dtype != torch.float64 else logits, ids, reduction='mean')
        l2 = sum((p * p).sum() for k, p in self.P.items() if k.endswith('/weights')) * (0.5 * hp['l2_reg'])
        loss = ce + l2
        for p in self.P.values():
            p.grad = None
        loss.backward()
        btot = batch_total if batch_total is not None else len(ids)
        lr = hp['base_learning_rate'] * btot / 256.0 * lr_mult
        d = min(hp['moving_average_decay'], (1.0 + self.step) / (10.0 + self.step))
        m, mom = hp['batch_norm_decay'], hp['momentum']
        with torch.no_grad():
            for k in self.S:
                self.ema_stats[k].mul_(d).add_(self.S[k], alpha=1.0 - d)
            for name, (mean, var) in bstats.items():
                self.S[name + '/mu'].mul_(m).add_(mean, alpha=1.0 - m)
                self.S[name + '/sigma'].mul_(m).add_(var, alpha=1.0 - m)
            for k, p in self.P.items():
                self.ema[k].mul_(d).add_(p, alpha=1.0 - d)                 # EMA of the PRE-update value (optimizers.py:159,175)
                g = p.grad
                a = self.accum[k]
                a.mul_(mom).add_(g)                                         # a <- m a + g ; w <- w - lr g - lr m a
                p.add_(g, alpha=-lr).add_(a, alpha=-lr * mom)
        self.step += 1
        return float(loss.item()), logits.detach()

    def params_hwio(self):
        out = {}
        for k, p in self.P.items():
            t = p.detach()
            out[k] = (t.permute(2, 3, 1, 0) if t.ndim == 4 else t).contiguous().numpy().copy()
        return out

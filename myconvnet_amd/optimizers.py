"""Optimizer — host-side mirror of reference optimizers.py (Optimizer.__init__ :19-63, _optimize_and_update
:89-177, train :179-563, _step :565-606, _update_learning_rate :608-632, MomentumOptimizer :668-677).

What is MI355X-native here: the parameter-server tower averaging (optimizers.py:121-147) is replaced by an RCCL
all-reduce of the flat gradient buffer, bucketed in reverse layer order and launched on a side stream while the
remaining backward kernels run (dist.py); Nesterov momentum, the L2 term, the EMA shadows and the optional
decoupled decay are one fused kernel over the flat parameter buffer; the per-step device->host copy of
Y_all / pred (optimizers.py:590-594) is optional (`fetch=False`) so the training loop never synchronises.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import _ffi
from ._ffi import lib
from .graph import Program


class Optimizer(object):
    def __init__(self, model, train_set, evaluator, val_set=None, **kwargs):
        self.model = model
        self.train_set = train_set
        self.evaluator = evaluator
        self.val_set = val_set
        if train_set is not None:
            assert model.compute_device == train_set.compute_device, 'Device mismatch between the model and dataset'
            assert model.num_devices == train_set.num_shards, 'Number of devices mismatch between the model and dataset'
            assert model.device_offset == train_set.device_offset, 'Device offset mismatch between the model and dataset'
            self.batch_size = train_set.batch_size
        else:
            self.batch_size = model.batch_size
        self.num_epochs = kwargs.get('num_epochs', 100)
        self.monte_carlo = kwargs.get('monte_carlo', False)
        self.augment_train = kwargs.get('augment_train', False)
        self.init_learning_rate = kwargs.get('base_learning_rate', 0.1) * self.batch_size / 256
        self.gradient_threshold = kwargs.get('gradient_threshold', None)
        self.warmup_epoch = kwargs.get('learning_warmup_epochs', kwargs.get('learning_warmup_epoch', 1.0))
        self.decay_method = kwargs.get('learning_rate_decay_method', None)
        self.decay_params = kwargs.get('learning_rate_decay_params', (0.94, 2))
        self.update_vars = [v for v in model.store.variables if v.trainable]
        self.optimization_operation = self._optimize_and_update(self._optimizer(**kwargs), **kwargs)
        # use_graph=True / MCN_GRAPH=1: one hipGraph launch per step (single process).  Off by default: measured on MI355X the
        # replay is bit-identical but SLOWER than the eager launch lists (bf16 25.04 vs 24.10 ms, fp32 73.4 vs 72.0 ms per step):
        # the host is ~2x ahead of the GPU anyway, and the replay loses the wgrad / BN-backward overlap of the two eager streams
        self.use_graph = bool(kwargs.get('use_graph', os.environ.get('MCN_GRAPH', '0') == '1')) and model.device.type == 'cuda'
        self._graph, self._graph_low, self._eager_steps = None, None, 0
        self._reset()
        self.steps_per_epoch = kwargs.get('steps_per_epoch', None)
        if self.steps_per_epoch is None and train_set is not None:
            self.steps_per_epoch = int(np.ceil(train_set.num_examples / self.batch_size))
        self.total_steps = (self.steps_per_epoch or 1) * self.num_epochs

    def _reset(self):
        self.curr_step = 0
        self.curr_epoch = 1
        self.best_score = self.evaluator.worst_score if self.evaluator is not None else 0.0
        self.learning_rate_update = 0
        self.curr_multiplier = 1.0

    name = 'Optimizer'

    def _optimizer(self, **kwargs):
        raise NotImplementedError

    # ---- update program ------------------------------------------------------------------------------------------------
    def _optimize_and_update(self, optimizer, **kwargs):
        """reference optimizers.py:89-177.  Returns the update Program; gradients come from the model's backward
        launch list (compute_gradients), averaged over ranks by the all-reduce (mean over towers, :138)."""
        m = self.model
        # the reference scales only for a factor > 1 (optimizers.py:102-103, 109-111)
        self.loss_scaling_factor = max(float(kwargs.get('loss_scaling_factor', 1.0)), 1.0)
        if self.loss_scaling_factor != m.loss_scale:
            m.compile(loss_scale=self.loss_scaling_factor)        # re-lowers only: variables and optimizer state are kept
        self.weight_decay = kwargs.get('base_weight_decay', 0.0) * self.batch_size / 256
        self.weight_decay_scheduling = kwargs.get('weight_decay_scheduling', True)
        # decay variant (optimizers.py:163-170): pseudo-Huber wins over L1 when both are given
        self.huber_decay_delta = kwargs.get('huber_decay_delta', None)
        if self.huber_decay_delta is not None:
            self.decay_mode = _ffi.DECAY_HUBER
        else:
            self.decay_mode = _ffi.DECAY_L1 if kwargs.get('l1_weight_decay', False) else _ffi.DECAY_L2
        self.momentum = optimizer['momentum']
        self.l2_reg = float(m._parameters.get('l2_reg', 1e-4))
        self.l1_reg = float(m._parameters.get('l1_reg', 0.0))              # convnet.py:529,553-557: l1_factor * sum |w| over the regularised variables
        self.use_ema = bool(kwargs.get('update_ema', True))
        st = m.store
        nw, n = m.n_l2_elems, st.size
        frozen = any(not v.trainable for v in st.variables)
        # update_vars = tf.trainable_variables() (optimizers.py:53): maximal contiguous runs of trainable / frozen variables of
        # the flat store, cut at the end of the regularised range.  A frozen run only moves its EMA shadow (ema.apply runs
        # for every variable, convnet.py:1400-1404).
        runs = []                                                            # (start, end, trainable)
        for v in st.variables:
            s0, e0 = v.offset, v.offset + (v.size + 3) // 4 * 4
            for s1, e1 in ((s0, min(e0, nw)), (max(s0, nw), e0)):
                if s1 >= e1:
                    continue
                if runs and runs[-1][1] == s1 and runs[-1][2] == v.trainable and not (s1 == nw and nw > 0):
                    runs[-1][1] = e1
                else:
                    runs.append([s1, e1, v.trainable])
        P = Program()
        ema_w = st.ema.data_ptr() if self.use_ema else 0
        # per-step scalars {lr, wd, ema_decay, grad_scale} live in a 4-float device buffer that the update kernels read, so every
        # launch of a step has the same arguments every step: the step is replayable as ONE hipGraph (_capture_step)
        self._hyper_host = torch.zeros(4, dtype=torch.float32).pin_memory() if m.device.type == 'cuda' else torch.zeros(4, dtype=torch.float32)
        self._hyper = torch.zeros(4, dtype=torch.float32, device=m.device)
        hp = self._hyper.data_ptr()
        separate_decay = self.decay_mode != _ffi.DECAY_L2 and self.weight_decay > 0.0
        clipping = self.gradient_threshold is not None
        for s1, e1, trainable in runs:
            off = s1 * 4
            if not trainable:
                if ema_w:
                    P.add(lib.mcn_ema_update_h, ema_w + off, st.data.data_ptr() + off, e1 - s1, hp)
                continue
            reg = s1 < nw
            if reg and self.l1_reg > 0.0 and not clipping:   # d/dw l1 * |w| = l1 * sign(w), added unscaled by the tower mean (the update kernel scales g by 1 / towers)
                P.add(lib.mcn_l1_grad_h, st.grad.data_ptr() + off, st.data.data_ptr() + off, e1 - s1, self.l1_reg, hp)
            # (with clipping the L2 gradient was folded into g by mcn_clip_by_global_norm: l2 = 0 here)
            P.add(lib.mcn_sgd_nesterov_fused_h, st.data.data_ptr() + off, st.grad.data_ptr() + off, st.accum.data_ptr() + off,
                  (ema_w + off) if ema_w else 0, e1 - s1, hp, self.momentum, self.l2_reg if (reg and not clipping) else 0.0,
                  1 if (reg and not separate_decay) else 0)
            # L1 / pseudo-Huber decay of the decayed range (weights, or everything with bias_norm_decay): a pass of its own
            # after the update, as in the reference; the plain w -= wd*w stays fused in the update kernel
            if reg and separate_decay:
                P.add(lib.mcn_decoupled_decay_h, st.data.data_ptr() + off, e1 - s1, hp, self.decay_mode, float(self.huber_decay_delta or 0.0))
        # per-tower clipping by global norm (optimizers.py:112-113): folds the L2 gradient into g, then scales
        self._clip = Program()
        if self.gradient_threshold is not None:
            self.grad_norm = torch.zeros(1, dtype=torch.float32, device=m.device)
            low = m._train_low
            if self.l1_reg > 0.0:
                # l1_reg with clipping: the reference clips the gradient of the FULL loss per tower (optimizers.py:106-113; the L1 term is part of it,
                # convnet.py:553-557), so l1 * sign(w) enters this tower's gradient — as is, no tower-mean factor — in front of the clip
                self._hyper_one = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float32, device=m.device)
                for s1, e1, trainable in runs:
                    if trainable and s1 < nw:
                        self._clip.add(lib.mcn_l1_grad_h, st.grad.data_ptr() + 4 * s1, st.data.data_ptr() + 4 * s1, e1 - s1, self.l1_reg, self._hyper_one.data_ptr())
            if not frozen:
                self._clip.add(lib.mcn_clip_by_global_norm, st.grad.data_ptr(), st.data.data_ptr(), n, nw, self.l2_reg, float(self.gradient_threshold),
                               self.grad_norm.data_ptr(), low.ws_ptr, low.ws_bytes)
            else:
                # clipping with blocks_to_train: norm and L2 fold over update_vars only (optimizers.py:53,106,112-113) = the trainable runs
                tr = [(s1, e1, min(max(nw, s1), e1)) for s1, e1, trainable in runs if trainable]
                merged = []
                for s1, e1, le in tr:                      # re-join runs that were only cut at the end of the regularised range
                    if merged and merged[-1][1] == s1 and merged[-1][2] == merged[-1][1]:
                        merged[-1] = (merged[-1][0], e1, le)
                    else:
                        merged.append((s1, e1, le))
                self._clip_runs = (ctypes.c_int64 * (3 * len(merged)))(*[x for r in merged for x in r])       # host array, read at launch time
                self._clip_ws = torch.zeros(len(merged) * 1024 + 8, dtype=torch.float32, device=m.device)   # one partial row per run
                self._clip.add(lib.mcn_clip_by_global_norm_runs, st.grad.data_ptr(), st.data.data_ptr(), self._clip_runs, len(merged), self.l2_reg,
                               float(self.gradient_threshold), self.grad_norm.data_ptr(), self._clip_ws.data_ptr(), self._clip_ws.numel() * 4)
        # EMA of the BN running statistics (pre-assign value), launched before the forward pass
        self._pre = Program()
        if self.use_ema and m.stats.size > 0:
            self._pre.add(lib.mcn_ema_update_h, m.stats.ema.data_ptr(), m.stats.data.data_ptr(), m.stats.size, hp)
        # cross-rank running-statistics chain (convnet.py:1899-1909)
        self._post_fwd = Program()
        self.dp = None
        if m.world_size > 1 or kwargs.get('force_data_parallel', False):      # (forced with one rank: exercises the RCCL calls in tests)
            from .dist import DataParallel
            self.dp = DataParallel(m, bucket_mb=float(kwargs.get('allreduce_bucket_mb', 25.0)))
            if m.world_size > 1:                         # with one rank the BN kernel itself updates the running statistics
                # Only BNs that update their statistics take part (a frozen BN — update_batch_norm=False / outside blocks_to_train — has
                # no update op in the reference, convnet.py:1915-1923; its batch-statistics slots are never written and its backward
                # pass READS the running statistics while this chain runs on its own stream): maximal runs of updating ranges
                runs = []
                for n_ in m.graph.nodes:
                    if n_.op == 'bn' and n_.attrs['update']:
                        for v in (n_.attrs['mu'], n_.attrs['sigma']):
                            s0, e0 = v.offset, v.offset + (v.size + 3) // 4 * 4
                            runs.append([s0, e0])
                runs.sort()
                merged_runs = []
                for s0, e0 in runs:
                    if merged_runs and merged_runs[-1][1] == s0:
                        merged_runs[-1][1] = e0
                    else:
                        merged_runs.append([s0, e0])
                stride = self.dp.gathered_stats.shape[1]
                for s0, e0 in merged_runs:
                    self._post_fwd.add(lib.mcn_bn_running_chain_strided, m.stats.data.data_ptr() + 4 * s0, self.dp.gathered_stats.data_ptr() + 4 * s0,
                                       m.world_size, min(e0, m.stats.size) - s0, stride, float(m.batch_norm_decay))
        return P

    def _set_hyper(self):
        """Per-step scalars -> the device buffer the update kernels read (one 16-byte stream-ordered copy per step)."""
        m = self.model
        lr = self.init_learning_rate * self.curr_multiplier
        d = min(m.moving_average_decay, (1.0 + m.global_step) / (10.0 + m.global_step))     # tf EMA num_updates rule
        wd = self.weight_decay * (self.curr_multiplier if self.weight_decay_scheduling else 1.0)
        gscale = 1.0 / m.world_size                                                          # tower mean, optimizers.py:138
        h = self._hyper_host
        h[0], h[1], h[2], h[3] = lr, wd, d, gscale
        self._hyper.copy_(h, non_blocking=True)
        return lr

    # ---- one step ---------------------------------------------------------------------------------------------------------------
    def _step_body(self):
        """Every launch of one optimisation step on the current stream (single process): replayable as a hipGraph."""
        m = self.model
        sp = m.stream_ptr()
        self._pre.run(sp)
        m.forward(train=True)
        m.backward()
        if len(self._clip) > 0:
            self._clip.run(sp)
        self.optimization_operation.run(sp)

    def _capture_step(self):
        """Capture _step_body once (forward, backward incl. the wgrad side stream's fork / join, clipping, update: ~600 launches in
        bf16) into a hipGraph through torch's capture API; addresses are static (compile() allocates everything once) and the
        per-step scalars come from the device buffer, so a step is then ONE graph launch + the 16-byte hyper-parameter copy.
        Results are bit-identical to the eager launch lists (same kernels, same order, same streams)."""
        m = self.model
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=m.device)
        cap.wait_stream(torch.cuda.current_stream(m.device))
        with torch.cuda.graph(g, stream=cap):
            self._step_body()
        torch.cuda.current_stream(m.device).wait_stream(cap)
        self._graph = g
        self._graph_low = m._train_low                   # a re-lowering (compile, autotune) invalidates the capture

    def _step(self, handles, merged=None, writer=None, summary=False, log_trace=False, fetch=True):
        """reference optimizers.py:565-606: one optimisation step on the batch currently in the model's input buffers.
        Returns (loss, Y_true, Y_pred) as numpy when fetch=True (the reference's behaviour), else device tensors
        without synchronising."""
        m = self.model
        self._set_hyper()
        if m._random_nodes:
            m.sample_random_masks()
        if self.dp is None and self.use_graph:
            if self._graph is not None and self._graph_low is not m._train_low:
                self._graph = None
            if self._graph is None and self._eager_steps >= 2:          # two eager steps first (lazy event / scratch creation)
                self._capture_step()
            if self._graph is not None:
                self._graph.replay()
            else:
                self._step_body()
                self._eager_steps += 1
        elif self.dp is not None:
            sp = m.stream_ptr()
            self._pre.run(sp)
            m.forward(train=True)
            clip = len(self._clip) > 0
            # batch statistics of all ranks -> chained running update, on a stream of its own (joined before the update below)
            self.dp.gather_bn_stats_async(self._post_fwd)
            if clip:                                     # towers clip their own gradient before the mean
                m.backward()
                self._clip.run(sp)
                self.dp.reduce_all()
            else:
                m.backward(self.dp.hooks())
                self.dp.finish()
            self.dp.join_bn_stats()
            self.optimization_operation.run(sp)
        else:
            self._step_body()
        m.global_step += 1
        if fetch:
            loss = self._mean_loss()
            y, pred = m.Y.buf, m.pred.buf
            if self.dp is not None and m.world_size > 1:                 # Y_all / pred over all towers (convnet.py:504,508)
                from .dist import all_gather_rows
                y, pred = all_gather_rows(y), all_gather_rows(pred)
            return loss, y.cpu().numpy(), pred.cpu().numpy()
        return m.loss_buf, m.Y.buf, m.pred.buf

    def _mean_loss(self):
        m = self.model
        if self.dp is not None:
            return self.dp.mean_scalar(m.loss_buf[0])
        return float(m.loss_buf[0].item())

    def _update_learning_rate(self):
        """reference optimizers.py:608-632."""
        warmup_steps = np.around(self.warmup_epoch * self.steps_per_epoch)
        if self.curr_step < warmup_steps:
            self.curr_multiplier = (self.curr_step + 1) / warmup_steps
        elif self.decay_method is not None:
            method = self.decay_method.lower()
            if method == 'step':
                self.curr_multiplier = 1.0
                for n in range(len(self.decay_params) - 1):
                    self.curr_multiplier *= np.power(self.decay_params[0], np.maximum(np.sign(self.curr_epoch - self.decay_params[n + 1]), 0.0))
            elif method == 'exponential':
                self.curr_multiplier = self.decay_params[0] ** ((self.curr_step - warmup_steps) / self.steps_per_epoch / self.decay_params[1])
            elif method in ('poly', 'polynomial'):
                power = self.decay_params[0] if isinstance(self.decay_params, (list, tuple)) else self.decay_params
                total_steps = self.steps_per_epoch * self.num_epochs - warmup_steps
                self.curr_multiplier = (1 - (self.curr_step - warmup_steps) / total_steps) ** power
            else:  # cosine
                anneal = self.decay_params[0] if isinstance(self.decay_params, (list, tuple)) else self.decay_params
                anneal = 0 if anneal is None else int(anneal)
                total_steps = self.steps_per_epoch * self.num_epochs - warmup_steps
                curr_prog = ((anneal + 1) * (self.curr_step - warmup_steps) / total_steps) % 1.0
                self.curr_multiplier = 0.5 * (1 + np.cos(curr_prog * np.pi))

    # ---- training loop ----------------------------------------------------------------------------------------------------------------
    def train(self, save_dir='./tmp', transfer_dir=None, details=False, verbose=True, show_each_step=False, show_percentage=True,
              **kwargs):
        """reference optimizers.py:179-563 without checkpoint files, TensorBoard and plots (out of scope)."""
        if transfer_dir is not None:
            raise NotImplementedError('transfer learning / checkpoint restore is out of scope')
        m = self.model
        train_size = self.train_set.num_examples
        self.steps_per_epoch = int(np.ceil(train_size / self.batch_size))
        num_steps = self.steps_per_epoch * self.num_epochs
        self.total_steps = num_steps
        validation_frequency = kwargs.get('validation_frequency', None) or self.steps_per_epoch
        train_losses, train_scores, eval_losses, eval_scores = [], [], [], []
        step_losses, step_scores = 0.0, 0.0
        self.train_set.initialize()
        start_time = time.time()
        for i in range(num_steps):
            self._update_learning_rate()
            X, Y = self.train_set.next_batch(m.device_batch, shard=m.rank)
            m.feed(X, Y)
            step_loss, step_Y_true, step_Y_pred = self._step(None)
            step_score = self.evaluator.score(step_Y_true, step_Y_pred)
            step_losses += step_loss
            step_scores += step_score
            self.curr_step += 1
            if (i + 1) % validation_frequency == 0:
                if self.val_set is not None:
                    _, eval_Y_true, eval_Y_pred, eval_loss = m.predict(self.val_set, verbose=False, return_images=False, **kwargs)
                    eval_score = self.evaluator.score(eval_Y_true, eval_Y_pred)
                    eval_scores.append(eval_score)
                    eval_losses.append(eval_loss)
                    curr_score = eval_score
                else:
                    curr_score = step_scores / validation_frequency
                if self.evaluator.is_better(curr_score, self.best_score, **kwargs):
                    self.best_score = curr_score
                train_losses.append(step_losses / validation_frequency)
                train_scores.append(step_scores / validation_frequency)
                step_losses, step_scores = 0.0, 0.0
            if (i + 1) % self.steps_per_epoch == 0:
                self.train_set.initialize()
                if verbose and m.rank == 0:
                    msg = '[epoch {}/{}]\tTrain loss: {:.5f}  |Train score: {:.5f}'.format(self.curr_epoch, self.num_epochs,
                                                                                          train_losses[-1] if train_losses else step_loss,
                                                                                          train_scores[-1] if train_scores else step_score)
                    if eval_losses:
                        msg += '  |Eval loss: {:.5f}  |Eval score: {:.5f}'.format(eval_losses[-1], eval_scores[-1])
                    msg += '  |LR: {:.7f}  |Elapsed time: {:5.0f} sec'.format(self.init_learning_rate * self.curr_multiplier, time.time() - start_time)
                    print(msg)
                self.curr_epoch += 1
        if verbose and m.rank == 0:
            print('Total training time: {:.2f} sec'.format(time.time() - start_time))
        if details:
            return dict(train_losses=train_losses, train_scores=train_scores, eval_losses=eval_losses, eval_scores=eval_scores)


class MomentumOptimizer(Optimizer):
    """reference optimizers.py:668-677: tf.train.MomentumOptimizer(lr, momentum, use_nesterov=True)."""
    name = 'SGD with Momentum'

    def _optimizer(self, **kwargs):
        return dict(kind='nesterov', momentum=float(kwargs.get('momentum', 0.9)))

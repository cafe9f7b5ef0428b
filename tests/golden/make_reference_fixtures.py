"""Generates tests/golden/evaluator.npz by RUNNING THE REFERENCE: imports /root/reference/evaluators.py (numpy-only, the one
module of the path that is importable here — SURVEY.md §8c) and records its scores on seeded inputs.

    python tests/golden/make_reference_fixtures.py          (build container only: /root/reference does not travel)

The .npz holds inputs and the reference's outputs — data, no reference source.  It pins (bit-exactly, float64) the rule
that turns the path's integer arg-max results into a score: myconvnet_amd/evaluators.py and oracle.ops.accuracy_score
must reproduce every stored score (tests/test_evaluators_reference.py).
"""
import contextlib
import importlib.util
import io
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/evaluators.py'


def load_reference():
    spec = importlib.util.spec_from_file_location('reference_evaluators', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def one_hot(ids, c):
    """ids < 0 -> all-zero row (the reference's encoding of an ignored sample: convnet.py:441-449)."""
    ids = np.asarray(ids)
    out = np.zeros(ids.shape + (c,), dtype=np.float32)
    ok = ids >= 0
    out[ok, ids[ok]] = 1.0
    return out


def softmax(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def cases():
    rng = np.random.default_rng(20261004)
    out = {}

    def add(name, y_true, y_pred):
        out[name] = (np.ascontiguousarray(y_true), np.ascontiguousarray(y_pred))

    # classification [B, C]: one-hot labels with ignored rows, softmax predictions biased towards the label
    for b, c, p_ign in ((8, 10, 0.0), (32, 1000, 0.1), (5, 3, 0.4), (4, 7, 1.0)):
        ids = rng.integers(0, c, b)
        ids = np.where(rng.random(b) < p_ign, -1, ids)
        z = rng.standard_normal((b, c)) + 3.0 * one_hot(np.where(ids < 0, 0, ids), c) * (rng.random((b, 1)) < 0.6)
        add('cls_b{}_c{}'.format(b, c), one_hot(ids, c), softmax(z))
    # ties in the prediction (arg-max must take the first maximum, argsort order decides top-N)
    yt = one_hot(np.array([2, 0, 1, 1]), 4)
    yp = np.array([[0.25, 0.25, 0.25, 0.25], [0.4, 0.4, 0.1, 0.1], [0.1, 0.45, 0.45, 0.0], [0.0, 0.0, 0.5, 0.5]], dtype=np.float32)
    add('cls_ties', yt, yp)
    # segmentation [N, H, W, C] one-hot with ignored pixels; one image entirely ignored
    for n, h, w, c in ((3, 9, 11, 5), (2, 17, 17, 19)):
        ids = rng.integers(0, c, (n, h, w))
        ids = np.where(rng.random((n, h, w)) < 0.3, -1, ids)
        ids[-1] = -1 if n == 3 else ids[-1]
        z = rng.standard_normal((n, h, w, c)) + 2.5 * one_hot(np.where(ids < 0, 0, ids), c) * (rng.random((n, h, w, 1)) < 0.7)
        add('seg_n{}_c{}'.format(n, c), one_hot(ids, c), softmax(z))
        # argmax_output forms (convnet.py:195-203): class ids in a trailing axis of 1, -1 = ignored
        pred_ids = softmax(z).argmax(axis=-1)
        add('seg_ids_n{}_c{}'.format(n, c), ids[..., None].astype(np.float32), pred_ids[..., None].astype(np.float32))
        add('seg_mixed_n{}_c{}'.format(n, c), one_hot(ids, c), pred_ids[..., None].astype(np.float32))
    # classification with class ids on both sides
    ids = np.array([3, -1, 0, 2, 2, 1], dtype=np.float32)
    add('cls_ids', ids[:, None], np.array([3, 0, 1, 2, 0, 1], dtype=np.float32)[:, None])
    return out


def main():
    with contextlib.redirect_stdout(io.StringIO()):           # the reference prints its name on construction
        ref = load_reference()
        evs = {'accuracy': ref.AccuracyEvaluator(), 'error': ref.ErrorEvaluator(), 'top1': ref.AccuracyTop1Evaluator(),
               'top5': ref.AccuracyTop5Evaluator(), 'miou': ref.MeanIoUEvaluator(), 'miou_bkgd0': ref.MeanIoUEvaluator(bkgd_idx=0)}
    blob = {}
    for name, (yt, yp) in cases().items():
        blob[name + '/y_true'] = yt
        blob[name + '/y_pred'] = yp
        for key, ev in evs.items():
            if key.startswith('top') and (yp.shape[-1] == 1 or yp.shape[-1] < 5 and key == 'top5'):
                continue
            if key.startswith('miou') and yt.shape[-1] == 1 and yt.max() < 0:
                continue
            blob[name + '/' + key] = np.float64(ev.score(yt.copy(), yp.copy()))
    # is_better: (curr, best, reference answer) for the max- and the min-mode rule (evaluators.py:59-70, 228-232)
    pairs = np.array([[0.5, 0.5], [0.5005, 0.5], [0.5006, 0.5], [0.4, 0.5], [0.0, 0.0], [1.0, 0.999]], dtype=np.float64)
    blob['is_better/pairs'] = pairs
    blob['is_better/accuracy'] = np.array([evs['accuracy'].is_better(a, b) for a, b in pairs])
    blob['is_better/error'] = np.array([evs['error'].is_better(a, b) for a, b in pairs])
    np.savez_compressed(os.path.join(HERE, 'evaluator.npz'), **blob)
    print('wrote evaluator.npz: {} arrays, {} scores'.format(len(blob), sum(1 for k in blob if k.split('/')[-1] in evs)))


if __name__ == '__main__':
    main()

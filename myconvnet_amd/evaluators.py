"""Accuracy evaluator used by the training loop (reference evaluators.py:73-109 semantics: arg-max of the one-hot
label vs arg-max of the prediction over valid rows; invalid rows count as correct)."""
import numpy as np


class AccuracyEvaluator(object):
    name = 'Accuracy'
    worst_score = 0.0
    mode = 'max'

    def score(self, y_true, y_pred):
        y_t = y_true.argmax(axis=-1)
        valid = np.isclose(y_true.sum(axis=-1), 1)
        y_p = y_pred.argmax(axis=-1)
        right = np.equal(y_t, y_p) & valid
        return float(np.where(valid, right, True).astype(float).mean())

    def is_better(self, curr, best, **kwargs):
        return curr >= best * (1.0 + kwargs.get('score_threshold', 1e-3))

"""In-memory data feeder: the `from_memory=True` mode of reference dataset.py (:46, :134) and its shard-per-device
contract (:113-129) — everything else in dataset.py (file lists, cv2 decoding, tf.data) is out of scope."""
import numpy as np


class DataSet(object):
    def __init__(self, images, labels, batch_size=16, num_shards=1, shuffle=False, seed=0, **kwargs):
        self.images = images
        self.labels = labels
        self.batch_size = int(batch_size)            # TOTAL batch; per device = batch_size // num_shards (dataset.py:113)
        self.num_shards = int(num_shards)
        self.compute_device = 'gpu'
        self.device_offset = 0
        self.num_examples = len(images)
        self.shuffle = shuffle
        self._rng = np.random.default_rng(seed)
        self.initialize()

    def initialize(self, session=None):
        self._cursor = 0
        self._order = self._rng.permutation(self.num_examples) if self.shuffle else np.arange(self.num_examples)

    def next_batch(self, device_batch, shard=0):
        """Rank `shard` takes images [cursor + shard*B, cursor + (shard+1)*B) of the global batch; wraps at the end."""
        start = self._cursor + shard * device_batch
        idx = self._order[np.arange(start, start + device_batch) % self.num_examples]
        self._cursor += device_batch * self.num_shards
        return self.images[idx], self.labels[idx]


def synthetic(num_examples, input_shape, num_classes, seed=1234):
    """SURVEY §8d synthetic inputs: uniform [0,1) images, uniform integer labels as float32 class ids."""
    rng = np.random.default_rng(seed)
    x = rng.random((num_examples,) + tuple(input_shape), dtype=np.float32)
    y = rng.integers(0, num_classes, num_examples).astype(np.float32)
    return x, y

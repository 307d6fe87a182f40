"""Preprocess transforms of the replay scheme (reference: src/components/transforms.py:11-22)."""
import torch as th


class Transform:
    def transform(self, tensor):
        raise NotImplementedError

    def infer_output_info(self, vshape_in, dtype_in):
        raise NotImplementedError


class OneHot(Transform):
    """actions -> actions_onehot (float32), the `preprocess` entry of src/run.py:194-196."""

    def __init__(self, out_dim):
        self.out_dim = out_dim

    def transform(self, tensor):
        return th.nn.functional.one_hot(tensor.long().squeeze(-1), self.out_dim).to(th.float32)

    def infer_output_info(self, vshape_in, dtype_in):
        return (self.out_dim,), th.float32

"""prodsearch_amd — MI355X-native (gfx950) implementation of ProdSearch's negative-sampled
ranking-loss training step behind the reference's own nn.Module / Optimizer API.

    from prodsearch_amd import ItemTransformerRanker, build_optim, ItemPVBatch, default_args

Everything numerical runs in libprodsearch_hip.so (hand-written HIP kernels, C ABI in
include/prodsearch_hip.h); this package is the thin host side.  No CPU fallback.
"""
from .batch import ItemPVBatch
from .config import default_args, readme_tem_args
from .item_transformer import ItemTransformerRanker
from .optimizers import Optimizer, build_optim

__all__ = ['ItemTransformerRanker', 'Optimizer', 'build_optim', 'ItemPVBatch', 'default_args',
           'readme_tem_args']

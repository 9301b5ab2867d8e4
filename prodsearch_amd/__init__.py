"""prodsearch_amd — MI355X-native (gfx950) implementation of ProdSearch's negative-sampled
ranking-loss training step behind the reference's own nn.Module / Optimizer API.

    from prodsearch_amd import ItemTransformerRanker, ProductRanker, build_optim, ItemPVBatch, default_args

Everything numerical runs in libprodsearch_hip.so (hand-written HIP kernels, C ABI in
include/prodsearch_hip.h); this package is the thin host side.  No CPU fallback.
"""
from . import corpus, evaluate, pyrandom, trainer
from .batch import ItemPVBatch
from .dataloader import ItemPVDataloader
from .config import default_args, readme_tem_args
from .item_transformer import ItemTransformerRanker
from .optimizers import Optimizer, build_optim
from .ps_model import ProductRanker
from .rtm_data import ProdSearchTestBatch, ProdSearchTrainBatch

__all__ = ['ProductRanker', 'ProdSearchTrainBatch', 'ProdSearchTestBatch', 'ItemTransformerRanker', 'Optimizer', 'build_optim', 'ItemPVBatch', 'default_args',
           'readme_tem_args']

"""prodsearch_amd — MI355X-native (gfx950) implementation of ProdSearch's negative-sampled
ranking-loss training step behind the reference's own nn.Module / Optimizer API.

    from prodsearch_amd import ItemTransformerRanker, ProductRanker, build_optim, ItemPVBatch, default_args

Everything numerical runs in libprodsearch_hip.so (hand-written HIP kernels, C ABI in
include/prodsearch_hip.h); this package is the thin host side.  No CPU fallback.
"""
import os as _os

# Kernel arguments in DEVICE memory (the HIP runtime reads this when it initialises, i.e. at the first GPU call): every workgroup of
# every launch starts by reading its argument block, and with the block in host memory the C2 step measured 0.275 ms against 0.235,
# C4 0.501 against 0.445, C5 1.52 against 1.42 (profiles/r03_kernarg.txt).  ROCm 7.2 defaults to device memory already; this only
# keeps an unset environment from depending on that default.  An explicit setting is respected.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

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

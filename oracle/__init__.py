"""CPU oracle for the ProdSearch ranking-loss training step.  TEST INFRASTRUCTURE.

This package restates, in plain PyTorch fp32 CPU ops, the algorithm of the
reference's hot path (kepingbi/ProdSearch ``models/item_transformer.py``,
``models/transformer.py``, ``models/neural.py``, ``models/text_encoder.py``,
``models/optimizers.py``); every function cites the reference file:line it
follows.  It is NOT part of the product:

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import it, and only as the checker / the timed CPU baseline;
* ``prodsearch_amd`` never imports it and has no CPU fallback: without the HIP
  library the product raises.

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``tests/golden/make_golden.py`` (with the
one uint8->bool ``masked_fill`` shim SURVEY.md §8c documents) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks the oracle against
every one of them.
"""

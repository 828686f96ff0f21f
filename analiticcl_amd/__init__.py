"""anx -- MI355X-native variant-scoring engine behind analiticcl's find_variants() API.

Importing this package loads the in-tree HIP library (analiticcl_amd/libanx.so); there is no CPU fallback.
"""
from ._lib import AnxError, lib, set_switch  # noqa: F401
from .model import Batch, Pipeline, SearchParameters, VariantModel, VocabParams, Weights, edit_script  # noqa: F401

__all__ = ["VariantModel", "SearchParameters", "Weights", "VocabParams", "Batch", "Pipeline", "AnxError"]

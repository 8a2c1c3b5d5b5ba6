"""MI355X-native edge-attention hot path of HiVT-GAN behind the reference's module interface.

Public surface (same names, constructor arguments and forward signatures as the reference):
``AAEncoder``, ``ALEncoder``, ``LocalEncoder``, ``TemporalEncoder`` (``models/local_encoder.py``),
``GlobalInteractorLayer``, ``GlobalInteractor`` (``models/global_interactor.py``),
``SingleInputEmbedding``, ``MultipleInputEmbedding`` (``models/embedding.py``), ``MLPDecoder``
(``models/decoder.py``), the three critics (``models/critics``), the loss classes (``losses``),
``HiVT`` (``models/hivt.py``) and ``TemporalData`` / ``DistanceDropEdge`` / ``init_weights`` (``utils.py``).
The sparse arithmetic runs in hand-written gfx950 kernels reached through ``libhivt_mi355.so``.
"""
from .critics import BaseTrajectoryCritic, LongScaleCritic, MidScaleCritic, ShortScaleCritic
from .data import BatchSpec, TemporalData, collate, make_batch, make_scene, pad_batch
from .dataset import SceneDataset, batches, load_scene, sanitize, save_scene
from .decoder import MLPDecoder
from .embedding import MultipleInputEmbedding, SingleInputEmbedding, init_weights
from .global_interactor import GlobalInteractor, GlobalInteractorLayer
from .hivt import HiVT
from .local_encoder import (AAEncoder, ALEncoder, DistanceDropEdge, EdgePlan, LocalEncoder, TemporalEncoder,
                            TemporalEncoderLayer, build_edge_plan, subgraph)
from .losses import (AdversarialDiscriminatorLoss, AdversarialGeneratorLoss, LaplaceNLLLoss, R1Regularization,
                     SoftTargetCrossEntropyLoss, fused_supervised_losses)
from .metrics import ADE, FDE, MR, EndpointDiversity, Jerk, SpeedViolation, ValidationMetrics
from .trainer import GanTrainer

__all__ = [
    "AAEncoder", "ALEncoder", "LocalEncoder", "TemporalEncoder", "TemporalEncoderLayer", "GlobalInteractor",
    "GlobalInteractorLayer", "SingleInputEmbedding", "MultipleInputEmbedding", "MLPDecoder", "BaseTrajectoryCritic",
    "ShortScaleCritic", "MidScaleCritic", "LongScaleCritic", "LaplaceNLLLoss", "SoftTargetCrossEntropyLoss",
    "R1Regularization", "AdversarialDiscriminatorLoss", "AdversarialGeneratorLoss", "fused_supervised_losses", "HiVT",
    "TemporalData", "collate", "make_scene", "make_batch", "DistanceDropEdge", "EdgePlan", "build_edge_plan",
    "subgraph", "init_weights", "GanTrainer", "BatchSpec", "pad_batch", "ADE", "FDE", "MR", "Jerk", "SpeedViolation",
    "EndpointDiversity", "ValidationMetrics", "SceneDataset", "batches", "load_scene", "save_scene", "sanitize",
]

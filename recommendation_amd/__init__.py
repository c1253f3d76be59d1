"""recommendation_amd — MI355X-native (gfx950) hot path for LightGCN-family graph-contrastive
recommenders: CSR SpMM message pass, InfoNCE / prototype contrast, BPR + negative sampler,
as hand-written HIP behind the C ABI of include/gcr.h, with a host-side mirror of the reference's
model-class / loss-function interface (Cmint22/Recommendation: lightgcn.py, ncl.py, ssl4rec.py,
gcl.py and univariate/)."""
from . import _lib
from .graph import CsrGraph, SpmmPlan
from . import functional

__all__ = ["CsrGraph", "SpmmPlan", "functional", "_lib"]
__version__ = "0.1.0"

"""MI355X-native E(3)-equivariant diffusion denoiser (drop-in for the eps_theta path of
Ren-Okubo/diffusion_model).  Host-side mirror of the reference's operator interface; all
arithmetic on the path runs in libegnn_amd.so (hand-written HIP for gfx950) through the C ABI
declared in include/egnn_amd.h."""
from .egnn import EGCL, EquivariantGNN  # noqa: F401
from .diffusion import E3DiffusionProcess, E3DiffusionProcessLegacy, E3DiffusionProcessXOnly, remove_mean  # noqa: F401
from .graph import GraphPlan, fully_connected_edge_index, fully_connected_plan, plan_edge_index, radius_plan  # noqa: F401
from . import partition, stats  # noqa: F401
from .partition import PartitionedSampler  # noqa: F401
from .optim import RAdamScheduleFree, define_optimizer  # noqa: F401
from .sampler import DeviceSampler, generate  # noqa: F401
from .preprocessor import SpectrumCompressor  # noqa: F401
from .snr import GammaNetwork, PositiveLinear  # noqa: F401
from .training import EarlyStopping, GradAllReducer, diffuse_as_batch, eval_epoch, train_epoch, train_step, training_loss  # noqa: F401
from .checkpoint import load_model_state, save_model_state  # noqa: F401
from .data import Batch, GraphData, GraphLoader, collate, load_dataset, make_graph, save_dataset  # noqa: F401

__all__ = ["EGCL", "EquivariantGNN", "E3DiffusionProcess", "remove_mean", "GraphPlan",
           "fully_connected_edge_index", "DeviceSampler", "generate", "SpectrumCompressor",
           "GammaNetwork", "PositiveLinear"]

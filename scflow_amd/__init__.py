"""scflow_amd -- MI355X-native implementation of SCFlow's recurrent flow/pose
refinement hot path.

Public surface = the reference's registry API for this path:
``build_refiner(cfg)`` / ``REFINERS`` / ``ENCODERS`` / ``DECODERS`` / ``HEAD`` and
the registered classes (``SCFlowRefiner``, ``RAFTEncoder``, ``SCFlowDecoder``,
``MultiClassPoseHead``) plus the operator-level seam (``CorrelationPyramid``,
``CorrLookup``).  All arithmetic runs in libscflow_hip.so (include/scflow_hip.h).
"""
__version__ = '0.1.0'

from .registry import (DECODERS, ENCODERS, HEAD, REFINERS, Registry, build_decoder,  # noqa: F401
                       build_encoder, build_from_cfg, build_head, build_refiner)
from .modules import (ConvGRU, CorrelationPyramid, CorrLookup, MotionEncoder,  # noqa: F401
                      MultiClassPoseHead, RAFTDecoder, RAFTDecoderMask, RAFTEncoder,
                      SCFlowDecoder, XHead)
from .refiner import RAFTRefinerFlow, RAFTRefinerFlowMask, SCFlowRefiner  # noqa: F401
from .metrics import (cal_epe, eval_pose_error, eval_rot_error,  # noqa: F401
                      eval_tran_error, filter_flow_by_mask,
                      get_flow_from_delta_pose_and_depth)
from .config import raft_model_cfg, scflow_model_cfg  # noqa: F401
from .weights import fill_state_dict  # noqa: F401
from .synthetic import make_inputs  # noqa: F401

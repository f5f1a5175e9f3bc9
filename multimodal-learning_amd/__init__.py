"""MI355X-native drop-in for the pathomic teacher/student distillation hot path of
CityU-AIM-Group/MultiModal-learning (MICCAI-2022/train_test_path_multi_distill.py).

Same Python module API as the reference (networks_new.py / resnets.py / fusion.py / KD_loss.py /
CL_utils / the five hot-loop functions), implemented over hand-written HIP kernels behind the C-ABI of
include/pathomic_hip.h.  There is no CPU or eager-PyTorch fallback: without libpathomic_hip.so the
package raises at first use.
"""
from . import _lib
from ._lib import build, lib, LIB_PATH
from .ops import set_precision, get_precision
from .utils import init_net, init_max_weights, count_parameters
from .resnets import ResNet, ResNet18, BasicBlock
from .fusion import BilinearFusion
from .networks_new import (define_net, define_optimizer, define_reg, define_scheduler, define_act_layer,
                           define_bifusion, MaxNet, PathomicNet, get_resnet)
from .kd_loss import DistillKL
from .CL_utils import CRDLoss, ContrastLoss_v2, Embed, Normalize, ContrastMemory_v3
from .train_step import (AEKD_loss, momentum_AEKD_loss, update_ema_variables, DistillStep, FusedAdam, FusedAdagrad, FlatParams,
                         TeacherStage1Step)
from . import dist
from . import mia2023
from . import sampler
from . import distiller_zoo
from .sampler import ContrastIndexSampler
from . import tsvd
from . import evaluate
from . import superpixel
from . import augment
from .options import stage2_opt

__all__ = ["build", "lib", "set_precision", "get_precision", "init_net", "init_max_weights", "count_parameters",
           "ResNet", "ResNet18", "BasicBlock", "BilinearFusion", "define_net", "define_optimizer", "define_reg",
           "define_scheduler", "define_act_layer", "define_bifusion", "MaxNet", "PathomicNet", "get_resnet",
           "DistillKL", "CRDLoss", "ContrastLoss_v2", "Embed", "Normalize", "ContrastMemory_v3", "AEKD_loss", "momentum_AEKD_loss",
           "update_ema_variables", "DistillStep", "FusedAdam", "FusedAdagrad", "FlatParams", "TeacherStage1Step", "dist", "stage2_opt"]

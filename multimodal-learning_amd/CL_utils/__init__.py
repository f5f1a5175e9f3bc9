from .CRD_loss import CRDLoss, ContrastLoss_v2, Embed, Normalize   # noqa: F401
from .memory_new import ContrastMemory_v3                           # noqa: F401
from . import CRD_criterion_v3                                     # noqa: F401  (MIA-2022 variant)
from . import CRD_criterion_v10                                    # noqa: F401  (MIA-2023 variant)

"""CPU oracle for the pathomic teacher/student distillation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU (fp32) restatement of the
reference algorithm (CityU-AIM-Group/MultiModal-learning, MICCAI-2022/*), each function citing
the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``multimodal-learning_amd``) never does and fails loudly when its HIP library is missing.

Parity pin: the restatement is asserted against golden vectors produced by importing and
running the reference itself on CPU (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
checked by ``tests/test_oracle_golden.py``).
"""
from .weights import make_state_dict, student_shapes, teacher_shapes, embed_shapes  # noqa: F401
from .nets import (resnet_forward, maxnet_forward, bilinear_fusion_forward,  # noqa: F401
                   pathomic_forward, Rounding)
from .losses import (distill_kl, embed_forward, contrast_memory_v3, contrast_loss_v2,  # noqa: F401
                     crd_loss, aekd_loss, nll_loss, CRDState)
from .step import DistillOracle, synthetic_batch, default_opt  # noqa: F401

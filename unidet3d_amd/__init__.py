"""MI355X-native implementation of UniDet3D's detection hot path (sparse-voxel U-Net backbone,
superpoint pooling, transformer query decoder) behind the reference's registry API.
Importing the package registers ``UniDet3D``, ``SpConvUNet``, ``UniDet3DEncoder``,
``UniDet3DCriterion`` and the loss / matcher classes (what ``custom_imports`` of the reference
configs triggers, configs/unidet3d_1xb8_scannet.py:2)."""
from .registry import MODELS, TASK_UTILS  # noqa: F401
from .spconv_unet import SpConvUNet  # noqa: F401
from .encoder import UniDet3DEncoder  # noqa: F401
from .criterion import (UniDet3DCriterion, UniDet3DAxisAlignedIoULoss, UniDet3DRotatedIoU3DLoss,  # noqa: F401
                        UniMatcher, QueryClassificationCost, BboxCostJointTraining)
from .data_preprocessor import Det3DDataPreprocessor_  # noqa: F401
from .unidet3d import UniDet3D  # noqa: F401
from .structures import InstanceData_  # noqa: F401
from . import transforms, evaluation  # noqa: F401  (registers the pipeline transforms)

__version__ = '0.1.0'

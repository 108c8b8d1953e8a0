"""The library's ResNet family under the module name scripts of the reference
import (examples/cnn/model/resnet.py there)."""
from singa_amd.models.resnet import *      # noqa: F401,F403
from singa_amd.models.resnet import (ResNet, resnet18, resnet34,  # noqa: F401
                                     resnet50, resnet101, resnet152)

"""The library's model zoo network under the module name scripts of the
reference import (examples/cnn/model/alexnet.py there)."""
from singa_amd.models.alexnet import *          # noqa: F401,F403
from singa_amd.models.alexnet import create_model  # noqa: F401

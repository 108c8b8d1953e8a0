"""The library's model zoo network under the module name scripts of the
reference import (examples/cnn/model/cnn.py there)."""
from singa_amd.models.cnn import *          # noqa: F401,F403
from singa_amd.models.cnn import create_model  # noqa: F401

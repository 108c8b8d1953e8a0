"""Packaging: ``pip install .`` builds the native pieces in-tree with
``singa_amd/build.py`` (g++ for the runtime, hipcc --offload-arch=gfx950 for
the kernels) and installs ``singa_amd`` plus the ``singa`` import alias."""
import os
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildNative(build_py):

    def run(self):
        sys.path.insert(0, HERE)
        from singa_amd import build as native
        native.build_all(verbose=True)
        super().run()


setup(
    name="singa-amd",
    version="0.1.0",
    description="MI355X-native deep-learning framework with the Apache SINGA API",
    packages=find_packages(include=["singa_amd", "singa_amd.*", "singa"]),
    package_data={"singa_amd": ["*.so", "lib/*.so", "bin/*",
                                "csrc/**/*.h", "csrc/**/*.cuh"]},
    python_requires=">=3.8",
    install_requires=["numpy", "torch", "protobuf", "pillow"],
    cmdclass={"build_py": BuildNative},
)

"""Packaging entry for setuptools that predate PEP 621 (this image ships 59.6, which ignores the [project] table of pyproject.toml):
the same metadata as pyproject.toml, and the same build hook (`_build_hook.BuildWithHip`: hipcc builds
brainevent_amd/lib/libbrainevent_amd.so before the Python files are collected)."""
import os
import sys

from setuptools import setup

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _build_hook import BuildWithHip  # noqa: E402

version = {}
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'brainevent_amd', '_version.py')).read(), version)

setup(
    name='brainevent_amd',
    version=version['__version__'],
    description='MI355X-native event-driven sparse matmul: the BinaryArray @ CSR / dense / JITC / FixedNumConn hot path of brainevent '
                'as hand-written HIP kernels behind a C ABI',
    packages=['brainevent_amd'],
    package_data={'brainevent_amd': ['csrc/*.hip', 'csrc/*.h', 'lib/*.so']},
    python_requires='>=3.10',
    install_requires=['numpy', 'torch'],
    cmdclass={'build_py': BuildWithHip},
)

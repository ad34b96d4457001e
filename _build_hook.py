"""setuptools hook of pyproject.toml: `build_py` first compiles the HIP sources into brainevent_amd/lib/libbrainevent_amd.so
(``brainevent_amd._lib.build``: hipcc --offload-arch=gfx950, no GPU needed) so that the wheel carries the library next to the
Python files.  BE_SKIP_NATIVE_BUILD=1 packages the tree as it is (a library built earlier, or none: the package then raises
KernelLoadError at first use — there is no CPU fallback)."""
import importlib.util
import os
import sys
from pathlib import Path

from setuptools.command.build_py import build_py


class BuildWithHip(build_py):
    def run(self):
        if os.environ.get('BE_SKIP_NATIVE_BUILD') != '1':
            root = Path(__file__).resolve().parent
            # load brainevent_amd/_lib.py by path: importing the package would import torch, which a build does not need
            for name, rel in (('brainevent_amd._error', 'brainevent_amd/_error.py'), ('brainevent_amd._lib', 'brainevent_amd/_lib.py')):
                spec = importlib.util.spec_from_file_location(name, root / rel)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[name] = mod
                spec.loader.exec_module(mod)
            # the header lives beside the package in a checkout (include/brainevent_amd.h) and the sources include it relatively
            print('[brainevent_amd] building', sys.modules['brainevent_amd._lib'].build(verbose=False))
        super().run()

"""Packaging of the `_richdem` extension module on the MI355X engine (the counterpart of the reference's
wrappers/pyrichdem/setup.py, which compiles pywrapper.cpp against the header-only library).

    python setup.py build_ext --inplace        # same result as `make`: ./_richdem<ext-suffix>.so

librdgpu.so must exist (richdem_amd.build()); it is found at run time through an rpath relative to this directory.
A distribution that ships the reference's `richdem/` Python package next to this module is a drop-in `richdem`."""
import os

import pybind11
from setuptools import Extension, setup

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

setup(
    name="richdem-rdgpu-engine",
    version="0.1",
    description="_richdem, the calculation module of RichDEM's Python package, on the MI355X engine (librdgpu.so)",
    ext_modules=[
        Extension(
            "_richdem",
            sources=[os.path.relpath(os.path.join(HERE, "pywrapper_gpu.cpp"))],
            include_dirs=[os.path.join(ROOT, "include"), pybind11.get_include()],
            library_dirs=[os.path.join(ROOT, "richdem_amd")],
            libraries=["rdgpu"],
            runtime_library_dirs=["$ORIGIN/../../richdem_amd"],
            extra_compile_args=["-O2", "-std=c++17", "-fvisibility=hidden"],
            language="c++",
        )
    ],
)

#!/usr/bin/env python3
"""Build of the `genomeworks` Python package (Cython extensions over the MI355X-native libraries).

  python setup.py build_ext --inplace      # what __graft_entry__.build() runs (in-tree .so files, no install)
  pip install .                            # a regular install / wheel (python -m build)

The extensions compile against <repo>/include (the reference's public C++ headers, re-implemented) and the HIP
runtime headers, and link with <repo>/genomeworks_amd/lib/libgenomeworks_amd.so (built by genomeworks_amd/build.py).
Same three extension modules as the reference's pygenomeworks/setup.py:127-166."""
import os

from Cython.Build import cythonize
from setuptools import Extension, find_packages, setup

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
LIB_DIR = os.path.join(ROOT, "genomeworks_amd", "lib")

common = dict(
    include_dirs=[os.path.join(ROOT, "include"), os.path.join(ROCM, "include")],
    define_macros=[("__HIP_PLATFORM_AMD__", "1")],
    library_dirs=[LIB_DIR, os.path.join(ROCM, "lib")],
    runtime_library_dirs=[os.path.join(ROCM, "lib")],
    # in-tree build: the host library sits three directories above an extension module
    extra_link_args=["-Wl,-rpath,$ORIGIN/../../../genomeworks_amd/lib"],
    language="c++",
    extra_compile_args=["-std=c++17", "-O2"],
)

extensions = [
    Extension("genomeworks.cuda.cuda", ["genomeworks/cuda/cuda.pyx"], libraries=["amdhip64"], **common),
    Extension("genomeworks.cudapoa.cudapoa", ["genomeworks/cudapoa/cudapoa.pyx"], libraries=["genomeworks_amd", "amdhip64"], **common),
    Extension("genomeworks.cudaaligner.cudaaligner", ["genomeworks/cudaaligner/cudaaligner.pyx"],
              libraries=["genomeworks_amd", "amdhip64"], **common),
]

setup(
    name="genomeworks",
    version="0.6.0+mi355x",
    description="Python bindings of the MI355X-native cudapoa / cudaaligner libraries (pygenomeworks API)",
    packages=find_packages(where=HERE, include=["genomeworks", "genomeworks.*"]),
    package_data={"genomeworks": ["cuda/*.pxd", "cudapoa/*.pxd", "cudaaligner/*.pxd"]},
    ext_modules=cythonize(extensions, compiler_directives={"embedsignature": True, "language_level": 3}, build_dir="build"),
    install_requires=["networkx"],
    python_requires=">=3.8",
    zip_safe=False,
)

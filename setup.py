"""pip install --no-build-isolation .

The build step is pypwt_amd.build: hipcc cross-compiles the HIP libraries for gfx950 (libpypwt_amd.so, libpypwt_amd_f64.so)
and cython + gcc build the compiled binding pypwt_amd/_cy/_wavelets; the wheel carries them as package data.  Three import
names: pypwt_amd (everything), pycudwt and pypwt (the reference's names: `from pycudwt import Wavelets`)."""
import os
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class build_with_hip(build_py):
    def run(self):
        sys.path.insert(0, HERE)
        from pypwt_amd import build as b
        for variant in ("f32", "f64"):
            b.build_library(verbose=True, variant=variant)
        if b.build_cython(verbose=True) is None:
            print("pypwt_amd: cython is not installed -- the package will bind through ctypes")
        super().run()


setup(
    name="pypwt_amd",
    version="0.1.0",
    description="MI355X-native (gfx950, hand-written HIP) discrete wavelet transform: drop-in for pycudwt's Wavelets class",
    install_requires=["numpy"],
    packages=["pypwt_amd", "pypwt_amd._cy", "pycudwt", "pypwt"],
    package_data={"pypwt_amd": ["libpypwt_amd.so", "libpypwt_amd_f64.so", "csrc/*", "include/*.h"],
                  "pypwt_amd._cy": ["*.so", "*.pyx.in", "*.pxi"]},
    cmdclass={"build_py": build_with_hip},
    zip_safe=False,
)

"""Builds ucd_amd/_abn_node*.so (the C++ autograd node, host code only) with g++ against the installed PyTorch.
Usage: python ucd_amd/csrc/build_node.py   (called by __graft_entry__.build(); needs libucd_hip.so built first)"""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension as ce

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
NAME = "_abn_node"


def output_path():
    return os.path.join(PKG, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False):
    out, src = output_path(), os.path.join(HERE, "abn_node.cpp")
    deps = [src, os.path.join(PKG, "..", "include", "ucd_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    import pybind11
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out,
           f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    for inc in ce.include_paths() + [sysconfig.get_paths()["include"], pybind11.get_include()]:
        cmd += ["-isystem", inc]
    cmd += [f"-L{torch_lib}", "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", f"-Wl,-rpath,{torch_lib}",
            f"-L{PKG}", "-lucd_hip", "-Wl,-rpath,$ORIGIN"]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

"""Build oracle/_ref/_gsdf_reference*.so: the REFERENCE'S OWN in-tree host sources of the hot path, compiled where they lie under
/root/reference/include and linked with this repository's drop-in operator libraries (TEST INFRASTRUCTURE; nothing is copied, the
output directory is git-ignored and travels to the GPU box with the snapshot).

    python oracle/ref_link/build.py            # no-op with a message when /root/reference is absent

Sources (all of the reference's host code on the path that needs nothing but libtorch + the replaced submodules):
    neural_net/{encoding_map,sub_map,local_map}.cpp   neural_gaussian/neural_gaussian.cpp
    optimizer/loss.cpp  optimizer/loss_utils/loss_utils.cpp  optimizer/optimizer_utils/optimizer_utils.cpp
    utils/utils.cpp  utils/coordinates.cpp  utils/ray_utils/ray_utils.cpp  mesher/mesher.cpp  mesher/cumcubes/src/cumcubes.cpp
Include path: tests/ref_compile_stubs (inert stand-ins for OpenCV / PCL / the CUDA runtime header / llog / tinyply, none on the path),
the reference's include/, this repository's drop-in headers (gs-sdf_amd/host: gsplat_cpp, tcnn_binding, kaolin_wisp_cpp, spatial.h).
NOT buildable here and why: neural_mapping/neural_mapping.cpp (ROS, tf, Eigen, OpenCV I/O, the data loader), params/params.cpp
(cv::FileStorage) — the latter's globals are defined in shim.cpp and set from Python."""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/include"
OUT = os.path.join(ROOT, "oracle", "_ref")
NAME = "_gsdf_reference"
SOURCES = ["neural_net/encoding_map.cpp", "neural_net/sub_map.cpp", "neural_net/local_map.cpp", "neural_gaussian/neural_gaussian.cpp",
           "optimizer/loss.cpp", "optimizer/loss_utils/loss_utils.cpp", "optimizer/optimizer_utils/optimizer_utils.cpp", "utils/utils.cpp",
           "utils/coordinates.cpp", "utils/ray_utils/ray_utils.cpp", "mesher/mesher.cpp", "mesher/cumcubes/src/cumcubes.cpp"]
OWN = ["shim.cpp", "binding.cpp"]


def module_path():
    return os.path.join(OUT, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build(verbose=True):
    if not os.path.isdir(REF):
        if verbose:
            print("oracle/ref_link: /root/reference is not present, nothing built (the prebuilt module, if any, is used as is)")
        return None
    import torch
    import torch.utils.cpp_extension as ce
    lib = os.path.join(ROOT, "gs-sdf_amd", "lib")
    for so in ("libgsdf_torch.so", "libgsdf_hip.so"):
        if not os.path.exists(os.path.join(lib, so)):
            raise RuntimeError(f"oracle/ref_link: {so} is not built (make -C gs-sdf_amd/csrc && make -C gs-sdf_amd/host)")
    obj = os.path.join(OUT, "obj")
    os.makedirs(obj, exist_ok=True)
    inc = [os.path.join(ROOT, "tests", "ref_compile_stubs"), REF, os.path.join(REF, "mesher", "cumcubes", "include"),
           os.path.join(ROOT, "gs-sdf_amd", "host"), os.path.join(ROOT, "gs-sdf_amd", "host", "compat"), os.path.join(ROOT, "include"),
           sysconfig.get_paths()["include"]] + ce.include_paths()
    flags = ["-std=c++17", "-O2", "-fPIC", "-w", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-DTORCH_EXTENSION_NAME={NAME}"]
    flags += [f"-I{p}" for p in inc]
    own_dir = os.path.dirname(os.path.abspath(__file__))
    jobs = [(os.path.join(REF, f), os.path.join(obj, f.replace("/", "_")[:-4] + ".o")) for f in SOURCES]
    jobs += [(os.path.join(own_dir, f), os.path.join(obj, "own_" + f[:-4] + ".o")) for f in OWN]
    stubs = os.path.join(ROOT, "tests", "ref_compile_stubs")
    newest_header = max(os.path.getmtime(os.path.join(d, f)) for base in (stubs, os.path.join(ROOT, "gs-sdf_amd", "host"))
                        for d, _, fs in os.walk(base) for f in fs if f.endswith((".h", ".hpp")))

    def compile_one(job):
        src, o = job
        if os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(src), newest_header):
            return src, 0, ""
        r = subprocess.run(["g++", *flags, "-c", src, "-o", o], capture_output=True, text=True)
        return src, r.returncode, r.stderr[-4000:]
    with ThreadPoolExecutor(min(8, os.cpu_count() or 2)) as ex:
        res = list(ex.map(compile_one, jobs))
    bad = [f"{s}:\n{e}" for s, rc, e in res if rc != 0]
    if bad:
        raise RuntimeError("oracle/ref_link: the reference's sources do not compile against the drop-in headers:\n" + "\n".join(bad))
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    out = module_path()
    cmd = ["g++", "-shared", "-o", out, *[o for _, o in jobs], f"-L{lib}", "-lgsdf_torch", "-lgsdf_hip", f"-L{tlib}", "-ltorch", "-ltorch_cpu",
           "-ltorch_python", "-lc10", "-Wl,--no-undefined", "-Wl,-rpath,$ORIGIN/../../gs-sdf_amd/lib", f"-Wl,-rpath,{tlib}",
           f"-L{sysconfig.get_config_var('LIBDIR')}", f"-lpython{sysconfig.get_config_var('LDVERSION')}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle/ref_link: link failed:\n" + r.stderr[-6000:])
    if verbose:
        print(f"oracle/ref_link: built {os.path.relpath(out, ROOT)} from {len(SOURCES)} reference sources")
    return out


def load():
    """import the module (prebuilt or just built); None when it does not exist"""
    p = module_path()
    if not os.path.exists(p):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, p)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except (ImportError, OSError) as e:   # a module left over from before libgsdf_torch.so changed: rebuild (python oracle/ref_link/build.py)
        import warnings
        warnings.warn(f"oracle/ref_link: {os.path.basename(p)} does not load ({e}); rebuild it with python oracle/ref_link/build.py")
        return None
    return mod


if __name__ == "__main__":
    build()
    sys.exit(0)

"""The C ABI: struct layouts, and that each library exports every symbol its header declares (no compute calls)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_device_library_exports_every_declared_symbol(pkg):
    L = pkg.render.lib()
    names = declared("ptx_core.h", "ptx_")
    assert len(names) >= 27
    for n in names:
        assert hasattr(L, n), f"libmi355_pt.so does not export {n}"
    assert sorted(pkg.render.EXPORTS) == names


def test_host_library_exports_every_declared_symbol(pkg):
    L = pkg.host.lib()
    for n in declared("pt_host.h", "pth_"):
        assert hasattr(L, n), f"libpt_host.so does not export {n}"


def test_struct_sizes_match_the_c_compiler(pkg):
    abi = pkg._abi
    names = ["PtxBvh8Node", "PtxBvh8", "PtxMesh", "PtxInstance", "PtxMaterial", "PtxTexture", "PtxSky", "PtxSun",
             "PtxBlueNoise", "PtxSceneDesc", "PtxCamera", "PtxRay", "PtxHitRecord", "PtxCounters", "PtxAdaptiveConfig", "PtxTimings", "PtxFrameStats", "PtxDebugStats"]
    src = '#include "ptx_core.h"\n#include <stdio.h>\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")], text=True)
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n in names:
        assert C.sizeof(getattr(abi, n)) == int(sizes[n]), n
    # the reference's own layout facts (accel/bvh.rs:86-98): 256 bytes, 64-byte aligned
    assert int(sizes["PtxBvh8Node"]) == 256
    assert abi.ray_dtype().itemsize == int(sizes["PtxRay"]) and abi.hit_dtype().itemsize == int(sizes["PtxHitRecord"])


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/ (task rule 3)."""
    pkg_dir = os.path.join(ROOT, "rust-pathtracer_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle_py" not in text and "oracle/" not in text, f

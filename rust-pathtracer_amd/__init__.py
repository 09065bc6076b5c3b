"""rust-pathtracer_amd — MI355X-native render core for GMunozRodr/Rust-Pathtracer.

The directory name carries a hyphen (it is the project's name), so import it
through `__graft_entry__.load_package()`, which registers it as the module
`rust_pathtracer_amd`.

Layout
  csrc/device/   hand-written gfx950 HIP kernels + the C ABI (include/ptx_core.h)
  csrc/host/     C++ stand-in for the reference's Rust host (include/pt_host.h)
  host.py        ctypes over libpt_host.so   (Scene, SceneBuilder, Camera)
  render.py      ctypes over libmi355_pt.so  (HipRenderLoop: the RenderLoop mirror)
  gltf.py        loader/gltf.rs in Python    (GLB/glTF -> flat scene)
  exporter.py    PngExporter                 (device tone map + sRGB8, PIL writes the PNG)
  textures.py    Texture::from_file          (sky / texture files: Radiance .hdr, 8/16-bit LDR)
"""
from . import _abi  # noqa: F401
from . import build  # noqa: F401
from . import host  # noqa: F401
from . import gltf  # noqa: F401
from . import render  # noqa: F401
from . import textures  # noqa: F401
from . import exporter  # noqa: F401
from .exporter import PngExporter  # noqa: F401
from .host import Camera, Scene, SceneBuilder, default_material  # noqa: F401
from .render import (AdaptiveSamplingConfig, DeviceFramebuffer, HipRenderLoop,  # noqa: F401
                     PathTracer, RenderCoreError, UnshadedRenderer)

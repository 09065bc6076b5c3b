"""`PngExporter` (exporter/png.rs:7-149) over the device framebuffer: tone mapping and `linear_to_srgb_u8` run on the
GPU (`ptx_fb_tonemap_srgb8`, exporter/tonemapping.rs:12-115), three bytes per pixel come back and PIL writes the file
(the `image` crate's job in the reference).  `with_denoiser` has no counterpart here: OIDN is not installed, so asking
for it raises instead of silently exporting the noisy image."""
import numpy as np

TONEMAPS = ("none", "aces", "reinhard", "agx")


class PngExporter:
    def __init__(self, tonemap="none", exposure=1.0):
        if tonemap not in TONEMAPS:
            raise ValueError(f"tonemap must be one of {TONEMAPS}, not {tonemap!r}")
        self.tonemap, self.exposure = tonemap, float(exposure)

    @classmethod
    def srgb(cls):                                   # png.rs:14-20
        return cls("none", 1.0)

    @classmethod
    def with_tonemap(cls, tonemap):                  # png.rs:22-28
        return cls(tonemap, 1.0)

    def with_exposure(self, exposure):               # png.rs:32-35
        return PngExporter(self.tonemap, exposure)

    def with_denoiser(self, denoiser):               # png.rs:37-43
        raise NotImplementedError("Open Image Denoise is not available in this environment; export without it or "
                                  "denoise on the host from render_first_pass_with_aov's buffers")

    def pixels(self, framebuffer):
        """(H, W, 3) uint8: `vec3_to_rgb` of every averaged pixel (png.rs:45-48)."""
        return framebuffer.to_srgb8(self.tonemap, self.exposure)

    def export(self, framebuffer, path):             # png.rs:50-56, 134-148; export_adaptive: 93-99
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(self.pixels(framebuffer)), "RGB").save(path, format="PNG")

    export_adaptive = export

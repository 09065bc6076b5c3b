"""In-tree builds: libpt_host.so (g++) and libmi355_pt.so (hipcc, gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
INC = os.path.join(ROOT, "include")

HOST_SRCS = ["csrc/host/pt_host.cpp", "csrc/host/pt_scenes.cpp"]
DEVICE_SRCS = ["csrc/device/pt_core.hip"]
DEVICE_DEPS = ["csrc/device/pt_core.hip", "csrc/device/pt_detmath.h", "csrc/device/pt_traverse.h",
               "csrc/device/pt_shade.h", "csrc/device/pt_types.h", "csrc/device/pt_vec.h"]

# -ffp-contract=off everywhere: the reference is rustc output, which never
# fuses a*b+c, and bit-exact hit records depend on it.
HOST_FLAGS = ["-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall", "-Wextra"]
# -fno-slp-vectorize: left on, the SLP vectorizer packs neighbouring scalar f32 operations into v_pk_mul_f32 / v_pk_add_f32,
# which issue no faster than the two scalar instructions on gfx950 and need v_mov's to pair their operands: k_extend
# 80 -> 74 VGPRs, 284 -> 197 moves, frame 550 -> 540 ms ("What the compiler was doing").
DEVICE_FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"build failed: {' '.join(cmd)}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libpt_host.so")
    deps = [os.path.join(HERE, s) for s in HOST_SRCS] + [
        os.path.join(HERE, "csrc/host/pt_host_internal.h"), os.path.join(HERE, "csrc/host/pt_hostmath.h"),
        os.path.join(INC, "ptx_core.h"), os.path.join(INC, "pt_host.h")]
    if force or _stale(out, deps):
        _run(["g++", *HOST_FLAGS, f"-I{INC}", *HOST_SRCS, "-o", out])
    return out


def build_device(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libmi355_pt.so")
    deps = [os.path.join(HERE, s) for s in DEVICE_DEPS] + [os.path.join(INC, "ptx_core.h")]
    if force or _stale(out, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, *DEVICE_FLAGS, f"-I{INC}", *DEVICE_SRCS, "-o", out])
    return out


def build_bench(force=False):
    """pt_bench: the C++ host driver with the reference's `--benchmark` entry point (src/main.rs:59-153)."""
    out = os.path.join(LIB, "pt_bench")
    src = "csrc/host/pt_bench.cpp"
    deps = [os.path.join(HERE, src), os.path.join(INC, "ptx_core.h"), os.path.join(INC, "pt_host.h"),
            os.path.join(LIB, "libpt_host.so"), os.path.join(LIB, "libmi355_pt.so")]
    if force or _stale(out, deps):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", f"-I{INC}", src, f"-L{LIB}", "-lpt_host", "-lmi355_pt",
              "-Wl,-rpath,$ORIGIN", "-o", out])
    return out


def build_all(force=False):
    return build_host(force), build_device(force), build_bench(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)

"""TEST INFRASTRUCTURE ONLY -- compile the reference's own CPU filter, from the
sources where they lie under /root/reference, into oracle/_ref/.

Nothing is copied: g++ (driven by torch.utils.cpp_extension, the same
mechanism the reference uses at bilateral_kernel.py:71-74) reads
/root/reference/gpytorch_lattice_kernel/cpp/lattice.cpp in place and writes
only into oracle/_ref/ (git-ignored; it does travel to the GPU box).

Two builds of the same one-file extension:
  cpu_lattice_ref      -O3            the parity reference and cpu_baseline
  cpu_lattice_ref_dbg  -O2 -DDEBUG    prints "Hash table size" + stage ns (h:298-336)

-march=native is deliberately absent: the .so must run on the GPU box's host
CPU, and FMA contraction would perturb the discrete front end.

On the GPU box /root/reference does not exist; build() there is a no-op and
load() just imports the prebuilt module.
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SRC = "/root/reference/gpytorch_lattice_kernel/cpp/lattice.cpp"

_VARIANTS = {
    "cpu_lattice_ref": ["-O3"],
    "cpu_lattice_ref_dbg": ["-O2", "-DDEBUG"],
}
# our own stage-dump driver (oracle/ref_stage_driver.cpp) compiled against the reference header in place
_DRIVER = ("ref_stages", os.path.join(HERE, "ref_stage_driver.cpp"), ["-O2", "-I" + os.path.dirname(SRC)])


def available(name="cpu_lattice_ref"):
    return os.path.exists(os.path.join(OUT, name + ".so"))


def build(verbose=False):
    """Build every variant if the reference sources are present. Returns the
    list of module names that exist afterwards."""
    if os.path.exists(SRC):
        os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
        from torch.utils.cpp_extension import load
        for name, flags in _VARIANTS.items():
            if available(name):
                continue
            bdir = os.path.join(OUT, "build_" + name)
            os.makedirs(bdir, exist_ok=True)
            load(name=name, sources=[SRC], extra_cflags=flags,
                 build_directory=bdir, verbose=verbose)
            os.replace(os.path.join(bdir, name + ".so"), os.path.join(OUT, name + ".so"))
        name, src, flags = _DRIVER
        if not available(name):
            bdir = os.path.join(OUT, "build_" + name)
            os.makedirs(bdir, exist_ok=True)
            load(name=name, sources=[src], extra_cflags=flags, build_directory=bdir, verbose=verbose)
            os.replace(os.path.join(bdir, name + ".so"), os.path.join(OUT, name + ".so"))
    return [n for n in list(_VARIANTS) + [_DRIVER[0]] if available(n)]


def load(name="cpu_lattice_ref"):
    """Import a prebuilt reference module (torch must be imported first so
    that libtorch is resolvable)."""
    import torch  # noqa: F401
    path = os.path.join(OUT, name + ".so")
    if not os.path.exists(path):
        raise FileNotFoundError(path + " (run oracle/build_ref.py in the dev container)")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))

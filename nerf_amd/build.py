"""Build libnerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(CSRC, "libnerf_hip.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return hs + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    newest = max(os.path.getmtime(p) for p in sources() + headers())
    return os.path.getmtime(OUT) >= newest


def build(force=False, verbose=False, out=None, defines=()):
    """Compile every .hip under csrc/ into one shared library.  ``out``/``defines`` build an
    experimental variant next to the product library (scripts/ablate.py)."""
    if out is None and not force and up_to_date():
        return OUT
    out = out or OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", INCLUDE, "-I", CSRC, "-Wno-unused-value", "-o", out]
    cmd += [f"-D{d}" for d in defines] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))

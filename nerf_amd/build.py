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


# -fno-slp-vectorize: the SLP vectoriser turns pairs of scalar fp32 operations into packed ones
# (v_pk_mul_f32 ...) and folds operand swaps into their op_sel modifiers.  On MI355X a packed fp32
# instruction whose LOW result selects the HIGH register of a source (op_sel bit set) returns that
# result as if the operand were 0 in lanes 48-63 while the SIMD's other wave executes
# v_mfma_f32_16x16x32_{f16,bf16} (scripts/probes/pk_vs_mfma_coexec.hip): the
# kernels must not contain such instructions, and scripts/isa_hazards.py (rule R5) checks that they
# do not.  Hand-written f32x2 code never swaps halves, so it is unaffected.
CODEGEN_FLAGS = ["-O3", "-std=c++17"]
# per source file: the render kernels' front ends overlap a partner wave's 16x16x32 MFMAs, and SLP is
# what produced the op_sel form there.  The backward kernels keep SLP (their LayerNorm-backward VALU
# phases gain 20 % from packed fp32; their MFMAs are 16x16x4 fp32 / 32x32x16 bf16, which the probe
# shows unaffected) — rule R5 still checks their ISA, so the form cannot slip in unnoticed.
FILE_FLAGS = {"nerf_render.hip": ["-fno-slp-vectorize"], "nerf_legacy.hip": ["-fno-slp-vectorize"]}


def flags_for(path):
    return CODEGEN_FLAGS + FILE_FLAGS.get(os.path.basename(path), [])


def build(force=False, verbose=False, out=None, defines=()):
    """Compile every .hip under csrc/ into one shared library.  ``out``/``defines`` build an
    experimental variant next to the product library (scripts/ablate.py)."""
    if out is None and not force and up_to_date():
        return OUT
    out = out or OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    common = ["--offload-arch=gfx950", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-Wno-unused-value"]
    common += [f"-D{d}" for d in defines]
    objects, procs = [], []
    for src in sources():                      # one compile per file (own flags), in parallel
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, *common, *flags_for(src), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objects.append(obj)
    for cmd, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objects]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))

"""Build libnerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every compile also keeps the device assembly (-save-temps) and runs the static ISA scan of
nerf_amd/isa_scan.py over it: a hazard-rule hit (inline asm inside an MFMA wait-state window, a
packed-fp32 instruction with an op_sel bit, an over-counted vmcnt hand-over) FAILS the build, so the
rules hold for whatever hipcc the library is built with, not only for the one the tests ran under.
"""
import hashlib
import os
import shutil
import subprocess

from . import isa_scan

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(CSRC, "libnerf_hip.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return sorted(hs + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")])


# -fno-slp-vectorize: the SLP vectoriser turns pairs of scalar fp32 operations into packed ones
# (v_pk_mul_f32 ...) and folds operand swaps into their op_sel modifiers.  On MI355X a packed fp32
# instruction whose LOW result selects the HIGH register of a source (op_sel bit set) returns that
# result as if the operand were 0 in lanes 48-63 while the SIMD's other wave executes
# v_mfma_f32_16x16x32_{f16,bf16} (scripts/probes/pk_vs_mfma_coexec.hip, NOTES.md section 7b): the
# kernels must not contain such instructions, and rule R5 of the ISA scan — run on every build, below —
# fails the build if one appears.  Hand-written f32x2 code never swaps halves, so it is unaffected.
CODEGEN_FLAGS = ["-O3", "-std=c++17"]
# per source file.  The render kernels' front ends overlap a partner wave's 16x16x32 MFMAs and SLP is
# what produced the op_sel form there: they are built without it.  nerf_backward.hip keeps SLP although
# it ALSO runs beside 16x16x32 MFMAs (nerf_bwd_data_h_kernel: layer_wide_h loops of one wave against
# the LayerNorm-backward VALU phase of its SIMD partner): the packed fp32 SLP produces in those phases
# is worth 20 % of the data-gradient kernels' time, and the forms it emits there carry op_sel_hi only
# (the harmless direction: 0 wrong of 1.6e8 in the probe).  That is a property of this compiler's
# output, not a guarantee — hence R5 at build time, not only in the test suite.
# nerf_sampler.hip (gather / resampler / Adam: no MFMAs of their own, but they may share a SIMD with another
# stream's): SLP gains nothing there and did produce the op_sel form in the Adam kernel's powf.
FILE_FLAGS = {"nerf_render.hip": ["-fno-slp-vectorize"], "nerf_legacy.hip": ["-fno-slp-vectorize"],
              "nerf_sampler.hip": ["-fno-slp-vectorize"]}


def flags_for(path):
    return CODEGEN_FLAGS + FILE_FLAGS.get(os.path.basename(path), [])


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def stamp(defines=()):
    """"<tree hash>-<compiler hash>" of everything the library's bytes depend on: source and header
    CONTENTS (mtimes do not survive a snapshot copy to the GPU box), this file (the flags live here), the
    scanner's rules and the -D list; then the compiler's version string."""
    return source_stamp(defines) + "-" + _compiler_id()


def _compiler_id():
    try:
        out = subprocess.run([_hipcc(), "--version"], capture_output=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        out = b"no-hipcc"           # a box without the compiler can only use what it was given
    return hashlib.sha256(out).hexdigest()[:16]


def source_stamp(defines=()):
    """The part of ``stamp`` that depends on the tree only (what the loader holds a library to:
    nerf_amd/_lib.py refuses a library whose sources have changed since it was built)."""
    h = hashlib.sha256()
    for p in sources() + headers() + [os.path.abspath(__file__), isa_scan.__file__]:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(repr((CODEGEN_FLAGS, sorted(FILE_FLAGS.items()), tuple(defines))).encode())
    return h.hexdigest()[:40]


def up_to_date(out=None, defines=()):
    out = out or OUT
    try:
        with open(out + ".stamp") as f:
            return os.path.exists(out) and f.read().strip() == stamp(defines)
    except OSError:
        return False


class IsaHazard(RuntimeError):
    pass


def build(force=False, verbose=False, out=None, defines=()):
    """Compile every .hip under csrc/ into one shared library.  ``out``/``defines`` build an
    experimental variant next to the product library (pass "NERF_HIP_EXPERIMENT=name" among the
    defines: the loader then refuses the result unless it is selected through NERF_HIP_LIB)."""
    out = out or OUT
    if not force and up_to_date(out, defines):
        return out
    hipcc = _hipcc()
    objdir = out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    common = ["--offload-arch=gfx950", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-Wno-unused-value"]
    common += [f"-D{d}" for d in defines]
    objects, procs = [], []
    for src in sources():                      # one compile per file (own flags), in parallel
        name = os.path.basename(src)
        tmp = os.path.join(objdir, name + ".tmp")          # -save-temps=obj writes next to the object
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        obj = os.path.join(tmp, name + ".o")
        cmd = [hipcc, *common, *flags_for(src), "-save-temps=obj", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, tmp, subprocess.Popen(cmd)))
        objects.append(os.path.join(objdir, name + ".o"))
    for (cmd, tmp, proc), final in zip(procs, objects):
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        if len(asm) != 1:
            raise IsaHazard(f"no device assembly found under {tmp}: the ISA scan cannot run")
        hits = isa_scan.scan(os.path.join(tmp, asm[0]))
        if hits:
            lines = "\n".join(f"  [{rule}] {kern} line {ln}: {text}\n      {detail}"
                              for kern, ln, text, rule, detail in hits[:8])
            raise IsaHazard(f"{os.path.basename(cmd[-3])}: {len(hits)} ISA hazard-rule violation(s) "
                            f"(nerf_amd/isa_scan.py); library NOT built\n{lines}")
        base = os.path.basename(final)
        os.replace(os.path.join(tmp, base), final)
        os.replace(os.path.join(tmp, asm[0]), final[:-2] + ".s")      # kept for scripts/isa_stats.py
        shutil.rmtree(tmp, ignore_errors=True)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objects]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    with open(out + ".stamp", "w") as f:
        f.write(stamp(defines) + "\n")
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))

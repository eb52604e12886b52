"""CPU checks of the build itself (hipcc cross-compiles gfx950 without a GPU):
* the emitted ISA obeys the inline-asm hazard rules of scripts/isa_hazards.py (hipcc inserts no wait
  states inside asm statements, so the kernels may only touch registers there that a compiler-visible
  instruction wrote last) — and the scanner does catch a violation when shown one;
* a library compiled as an experiment (-DNERF_HIP_EXPERIMENT=...) is refused by the default loader;
* the Python mirror of the training workspace layout matches the library's."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scanner():
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "scripts", "isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scanner_flags_unshielded_asm_accesses(tmp_path):
    haz = _scanner()
    bad = tmp_path / "bad.s"
    bad.write_text("""
_Zkernel:
	v_mfma_f32_16x16x32_f16 v[12:15], v[84:87], v[68:71], v[12:15]
	v_mfma_f32_16x16x32_f16 v[48:51], v[80:83], v[56:59], v[12:15]
	;;#ASMSTART
	v_fma_mix_f32 v12, v64, -1.0, v66 op_sel_hi:[1,0,0]
	;;#ASMEND
	v_readlane_b32 s4, v200, 1
	;;#ASMSTART
	s_mov_b32 m0, s9
	global_load_lds_dwordx4 v138, s[4:5]
	;;#ASMEND
	v_pk_mul_f32 v[54:55], v[142:143], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]
	v_pk_mul_f32 v[56:57], v[142:143], v[8:9] op_sel_hi:[1,0]
""")
    hits = haz.scan(str(bad))
    assert sorted({h[3] for h in hits}) == ["R1-WAR", "R1-WAW", "R2", "R3", "R5"]
    assert sum(h[3] == "R5" for h in hits) == 1            # op_sel_hi alone is the harmless form
    good = tmp_path / "good.s"
    good.write_text("""
_Zkernel:
	v_mfma_f32_16x16x32_f16 v[48:51], v[80:83], v[56:59], v[12:15]
	s_nop 6
	v_max_f32_e32 v12, 0, v20
	;;#ASMSTART
	v_fma_mix_f32 v12, v64, -1.0, v12 op_sel_hi:[1,0,0]
	;;#ASMEND
	v_readlane_b32 s4, v200, 1
	;;#ASMSTART
	s_mov_b32 s20, m0
	s_mov_b32 m0, s9
	s_nop 2
	global_load_lds_dwordx4 v138, s[4:5]
	;;#ASMEND
""")
    assert haz.scan(str(good)) == []


def test_scanner_flags_an_overcounted_handover_wait(tmp_path):
    """R6: `s_waitcnt vmcnt(4 + k)` in asm claims k compiler-emitted vector-memory ops sit between the
    DMA of the stage being opened and the wait; with fewer, the youngest 4 + k reach into that DMA."""
    haz = _scanner()
    dma = "\t;;#ASMSTART\n" + "\tglobal_load_lds_dwordx4 v1, s[4:5]\n" * 4 + "\t;;#ASMEND\n"
    stores = "\tglobal_store_dwordx4 v[2:3], v[4:7], off\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(%d) lgkmcnt(0)\n\t;;#ASMEND\n"
    body = "_Zkernel:\n\ts_mov_b32 m0, s9\n\ts_nop 2\n" + dma + stores * 2 + dma
    ok, bad = tmp_path / "ok.s", tmp_path / "bad.s"
    ok.write_text(body + wait % 6)                 # 2 stores + the next stage's 4 pieces may fly
    bad.write_text(body + wait % 7)                # one more: a piece of the opened stage may fly
    assert haz.scan(str(ok)) == []
    assert [h[3] for h in haz.scan(str(bad))] == ["R6"]
    # a block entered only by a jump (a rotated loop body: its real predecessor is its own latch): the history above
    # the label is not its history — R6 assumes LDS-DMA pieces for everything it cannot see
    jumped = "_Zkernel:\n\ts_mov_b32 m0, s9\n\ts_nop 2\n\ts_branch .LBB0_2\n.LBB0_1:\n" + stores
    rot_ok, rot_bad = tmp_path / "rot_ok.s", tmp_path / "rot_bad.s"
    rot_ok.write_text(jumped + wait % 5)           # its own store + at most 4 unseen pieces
    rot_bad.write_text(jumped + wait % 7)          # 6 unseen operations may all be pieces of the opened stage
    assert haz.scan(str(rot_ok)) == []
    assert [h[3] for h in haz.scan(str(rot_bad))] == ["R6"]
    # a TWO-slot ring (`; nerf_ring_depth=2` on the wait, nerf_device.h): the DMA of the opened stage was issued at
    # the previous hand-over, so only what follows it may fly and NO piece of any DMA may (the narrow
    # split-precision training forward counted the previous stage's stores too and left two pieces in flight)
    wait2 = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(%d) lgkmcnt(0) ; nerf_ring_depth=2\n\t;;#ASMEND\n"
    body2 = "_Zkernel:\n\ts_mov_b32 m0, s9\n\ts_nop 2\n" + stores * 2 + dma + stores * 2
    two_ok, two_bad = tmp_path / "two_ok.s", tmp_path / "two_bad.s"
    two_ok.write_text(body2 + wait2 % 2)
    two_bad.write_text(body2 + wait2 % 4)          # the round-5 count: [piece, piece, store, store]
    assert haz.scan(str(two_ok)) == []
    assert [h[3] for h in haz.scan(str(two_bad))] == ["R6"]
    pure = tmp_path / "pure.s"                     # DMA-only waits (the weight gradient's ring) are by construction
    pure.write_text("_Zkernel:\n\ts_mov_b32 m0, s9\n\ts_nop 2\n" + dma + dma + wait % 8)
    assert haz.scan(str(pure)) == []


def test_build_refuses_a_library_with_a_hazard(tmp_path, monkeypatch):
    """build() scans the assembly it just produced and raises instead of linking (the rules
    must hold for whatever hipcc builds the library, not only for the one the tests ran under)."""
    from nerf_amd import build as nerf_build, isa_scan
    monkeypatch.setattr(isa_scan, "scan", lambda path: [("k", 1, "v_pk_mul_f32 ...", "R5", "planted")])
    with pytest.raises(nerf_build.IsaHazard, match="R5"):
        nerf_build.build(out=str(tmp_path / "lib_hazard.so"), defines=("NERF_HIP_EXPERIMENT=hazard_probe",))
    assert not os.path.exists(tmp_path / "lib_hazard.so")


def test_stamp_tracks_flags_and_sources(tmp_path, monkeypatch):
    from nerf_amd import build as nerf_build
    a = nerf_build.stamp()
    assert a == nerf_build.stamp() and a != nerf_build.stamp(defines=("X=1",))
    monkeypatch.setattr(nerf_build, "CODEGEN_FLAGS", ["-O2", "-std=c++17"])
    assert nerf_build.stamp() != a                 # a library built with other flags is not "up to date"
    assert not nerf_build.up_to_date()


def test_kernels_obey_the_inline_asm_hazard_rules(tmp_path):
    haz = _scanner()
    for name in sorted(f for f in os.listdir(haz.CSRC) if f.endswith(".hip")):
        out = str(tmp_path / (name + ".s"))
        haz.compile_to_asm(os.path.join(haz.CSRC, name), out)
        hits = haz.scan(out)
        assert hits == [], hits[:3]


def test_default_loader_refuses_an_experiment_build(tmp_path):
    from nerf_amd import build as nerf_build
    out = str(tmp_path / "libnerf_hip_exp.so")
    nerf_build.build(out=out, defines=("NERF_HIP_EXPERIMENT=probe_build",))
    code = ("import nerf_amd._lib as L, sys\n"
            f"L.LIB_PATH = {out!r}\n"
            "try:\n    L.lib()\nexcept RuntimeError as e:\n    print('REFUSED', e); sys.exit(0)\n"
            "print('LOADED', L.build_flags())\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("NERF_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "REFUSED" in r.stdout and "probe_build" in r.stdout, r.stdout + r.stderr
    env["NERF_HIP_LIB"] = out                       # asked for by path: allowed (scripts/ablate.py)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert "LOADED ['probe_build']" in r.stdout, r.stdout + r.stderr


def test_loader_refuses_a_library_older_than_its_sources(tmp_path):
    """The ABI number only changes with the header; a kernel edit does not bump it.  The loader therefore holds
    the in-tree library to the content hash ``build()`` wrote beside it (nerf_amd/_lib.py: _check_stamp): with a
    source file changed after the build (here: a copy of csrc/ with one byte appended) it raises instead of
    running yesterday's kernels."""
    import shutil
    from nerf_amd import build as nerf_build
    nerf_build.build()
    pkg = tmp_path / "nerf_amd"
    shutil.copytree(os.path.join(ROOT, "nerf_amd"), pkg, ignore=shutil.ignore_patterns("*.obj", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    code = ("import nerf_amd._lib as L, sys\n"
            "try:\n    L.lib()\nexcept RuntimeError as e:\n    print('REFUSED', e); sys.exit(0)\n"
            "print('LOADED')\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    env.pop("NERF_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert "LOADED" in r.stdout, r.stdout + r.stderr            # an exact copy of the tree loads
    with open(pkg / "csrc" / "nerf_device.h", "a") as f:
        f.write("\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert "REFUSED" in r.stdout and "stale" in r.stdout, r.stdout + r.stderr


# Scratch (register spills) per kernel of the product build: bytes per lane from the kernel descriptor and the number
# of scratch accesses that sit INSIDE an MFMA loop (an MFMA within 40 instructions on both sides; a spill there is a
# vector-memory operation in the in-order vmcnt queue of the stage hand-overs).  The render kernels of both networks
# and every weight-gradient loop are spill-free inside their loops; the 256-register training kernels spill in their
# VALU phases (measured, round 4: halving those accesses in nerf_legacy_bwd_data_kernel moved its training step by
# 0.5 %).  A kernel not listed must not spill at all; budgets only ever go DOWN.
SCRATCH_BUDGET = {            # kernel name fragment: (bytes per lane, accesses between two MFMAs)
    # the split-precision render kernel at 8 register tiles runs THREE workgroups per CU (<= 168 registers, no LDS
    # stash): 7 values of the front end / compositing state spill around the MLP, none inside its loops
    "nerf_render_fwd_kernelILb0ELb1ELb0ELi8E": (24, 0),
    # the split-precision kernel WITH per-sample outputs (NeRF.forward under no_grad, the hierarchical sampler's coarse
    # pass): since color_outputs is a run-time count its out_raw store computes the column of each slot of output tile 0
    # per lane (nerf_layout.h: row_of_tile0) — three 8-byte addresses hoisted in the front end are parked around the
    # MLP, 8 scratch instructions per ray, none inside a loop.  Not a bench path.
    "nerf_render_fwd_kernelILb0ELb1ELb1ELi16E": (20, 0),
    "nerf_bwd_data_h_kernel": (8, 0),
    "nerf_wgrad_h_kernel": (12, 0),
    "nerf_legacy_fwd_kernelILb1E": (68, 0),
    "nerf_legacy_fwd_h_kernelILb0E": (60, 0),
    "nerf_legacy_fwd_h_kernelILb1E": (124, 1),
    "nerf_legacy_bwd_data_kernel": (272, 0),
    "nerf_legacy_bwd_data_h_kernel": (68, 0),
    "nerf_legacy_wgrad_h_kernel": (12, 0),
}


# A frame that NO instruction touches is not a spill: LLVM keeps the stack slot of a scalar-register tuple it then parked
# in vector-register lanes (resource remarks: `SGPRs Spill: 48`, `VGPRs Spill: 0`; MIR after prologepilog: one dead
# 32-byte spill slot + the 4 bytes the register scavenger reserves once a frame exists).  Listed so that it cannot
# grow or start being used unnoticed: (bytes, 0 scratch instructions).
DEAD_FRAMES = {
    "nerf_render_fwd_kernelILb1ELb1ELb0ELi8E": 36,      # narrow split-precision training forward, three workgroups per CU
    "nerf_render_fwd_kernelILb1ELb1ELb0ELi16E": 36,     # full-width split-precision training forward (230 registers)
}


def test_scratch_stays_within_the_per_kernel_budget():
    import glob
    import re
    from nerf_amd import build as nerf_build, isa_scan
    nerf_build.build()
    seen = set()
    files = sorted(glob.glob(os.path.join(nerf_build.OUT + ".obj", "*.s")))
    assert len(files) == len(nerf_build.sources())
    for path in files:
        for kernel, (size, inside, instructions) in isa_scan.scratch_report(path).items():
            short = re.sub(r"^_ZN?\d*_GLOBAL__N_1\d+", "", kernel)
            dead = [k for k in DEAD_FRAMES if short.startswith(k)]
            if dead:
                seen.add(dead[0])
                assert instructions == 0 and size <= DEAD_FRAMES[dead[0]], (kernel, size, instructions)
                continue
            keys = [k for k in SCRATCH_BUDGET if short.startswith(k)]
            if not keys:
                assert (size, inside, instructions) == (0, 0, 0), \
                    f"{kernel} spills ({size} B per lane, {instructions} scratch instructions, {inside} inside its MFMA loops)"
                continue
            key = max(keys, key=len)
            seen.add(key)
            budget = SCRATCH_BUDGET[key]
            assert size <= budget[0] and inside <= budget[1], (kernel, (size, inside), budget)
    # a kernel that no longer spills: delete its entry
    assert seen == set(SCRATCH_BUDGET) | set(DEAD_FRAMES), (set(SCRATCH_BUDGET) | set(DEAD_FRAMES)) - seen


def test_workspace_mirror_matches_the_library():
    from nerf_amd import _lib, build as nerf_build
    import workspace_mirror as W
    nerf_build.build()
    lib = _lib.lib()
    assert _lib.build_flags() == []
    for n, S in ((1, 2), (3, 17), (5, 9), (130, 64), (256, 100), (4096, 64)):
        assert W.train_layout(n, S)["total"] * 4 == lib.nerf_hip_train_workspace_bytes(n, S)
        assert W.legacy_train_layout(n, S)["total"] * 4 == lib.nerf_hip_legacy_train_workspace_bytes(n, S)
    assert sorted(W.layer0_feature_order().tolist()) == list(range(96))

"""Static hazard scan of the gfx950 ISA the kernels compile to, around INLINE ASM (no GPU needed).
Run by nerf_amd.build.build() on every compile (a hit fails the build) and by tests/test_build_hygiene.py.

hipcc's hazard recognizer inserts the wait states the hardware needs between dependent instructions
(MFMA result -> VALU read, MFMA srcC read -> VALU overwrite, VALU-written SGPR -> VMEM address, ...)
but it never looks INSIDE an inline-asm statement: an instruction written in asm gets none of them.
The kernels therefore follow one rule, and this scan enforces it on the emitted code:

  R1  a vector instruction inside asm may only read or write VGPRs whose most recent MFMA access
      (as destination, or as the srcC operand) is either more than WINDOW wait states old or has
      since been followed by a compiler-visible VALU access of that register (which carries the
      compiler's own wait states: a visible read proves the MFMA result has landed, a visible write
      proves the WAW / WAR windows have passed);
  R2  an SGPR read by a vector-memory instruction inside asm was not written by a VALU instruction
      (v_readlane / v_readfirstlane / v_cmp ...) less than 5 wait states earlier;
  R3  `s_mov_b32 m0` is at least one wait state ahead of the LDS-DMA that uses it;
  R4  a v_permlane*_swap inside asm does not read a VGPR written by a VALU less than 2 wait states
      earlier.

and one rule that is not about asm but about an MI355X erratum measured by
scripts/probes/pk_vs_mfma_coexec.hip (compiler-generated code must obey it too):

  R5  no packed-fp32 instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) carries an op_sel bit,
      i.e. none lets its LOW result select the HIGH register of a source pair: while the SIMD's
      other wave executes v_mfma_f32_16x16x32_{f16,bf16}, such an instruction returns the low result
      as if that operand were 0 in lanes 48-63 (22-25 % of the executions of v_pk_mul_f32).
      op_sel_hi (the HIGH result selecting a LOW register) is unaffected.  The kernels are built
      with -fno-slp-vectorize (nerf_amd/build.py), which is what keeps hipcc from producing the form.

and one rule about the hand-written COUNTED waits (`s_waitcnt vmcnt(N)` inside asm: the stage hand-overs of
WeightPipe::open_stage<kYounger>), whose immediates hard-code how many vector-memory operations the COMPILER
emits between two LDS-DMA issues.  A hand-over wait names the depth D of its weight ring in an assembler
comment (`; nerf_ring_depth=D`, nerf_device.h); a counted wait without the comment is taken as D = 3:

  R6  the N youngest vector-memory instructions in front of such a wait (program text order), when
      they include compiler-emitted loads / stores, contain at most D - 2 stages' LDS-DMA pieces
      (4 global_load_lds per stage): with 3 slots the DMA of the FOLLOWING stage may still fly, with 2
      slots nothing of any DMA may (the stage being opened was issued at the previous hand-over).
      With more, N over-counts and the wait would leave pieces of the stage being opened in flight.
      (An under-count only makes the wait stricter.)

WINDOW = 20 covers the largest requirement of the MFMAs used here (8-pass XDL: 12).
CLI: python scripts/isa_hazards.py [-DNAME ...]      exit code 1 when a rule is violated.
"""
import re

WINDOW = 20


def _vregs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def _sregs(tok):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", tok):
        out.add(int(m.group(1)))
    if re.search(r"\bvcc\b", tok):
        out.add("vcc")
    return out


def _operands(s):
    parts = s.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


def scan(path):
    """-> list of (kernel, line number, instruction, rule, detail)."""
    found = []
    kern = None
    in_asm = False
    ws = 0                      # wait-state clock
    mfma_dst, mfma_srcc = {}, {}     # vgpr -> (clock, text)
    valu_vgpr, valu_sgpr = {}, {}    # register -> clock of the last VALU write
    m0_write = -100
    vmem = []                   # vector-memory instructions in text order: (is LDS-DMA, inside asm)
    no_fallthrough = False      # the previous instruction was an unconditional branch
    unknown_history = False     # the block was entered by a jump: operations older than `vmem` exist but were not seen
    for ln, raw in enumerate(open(path).read().splitlines(), 1):
        s = raw.strip()
        m = re.match(r"^(_Z\S+):", s)
        if m:
            kern = m.group(1)
            unknown_history = False
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        label = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
        if label:                # a label: other paths join here, keep the state (conservative) ...
            if no_fallthrough:   # ... unless the text above cannot fall into it (it ended in s_branch): the vector-memory
                vmem = []        # history of whoever jumps here is not the text above.  It is UNKNOWN, not empty: such a
                unknown_history = True        # block is typically a rotated loop body whose real predecessor is its own
                no_fallthrough = False        # latch, full of LDS-DMA issues — R6 assumes the worst for what it cannot see
            continue
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        ops = _operands(s)
        no_fallthrough = op in ("s_branch", "s_endpgm", "s_setpc_b64")
        if op == "s_nop":
            ws += int(ops[0]) + 1
            continue
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vmem.append((op.startswith(("global_load_lds", "buffer_load")) and "lds" in s, in_asm))
        if in_asm and op == "s_waitcnt":
            cnt = re.search(r"vmcnt\((\d+)\)", s)
            depth = re.search(r"nerf_ring_depth=(\d+)", s)
            allowed = 4 * ((int(depth.group(1)) if depth else 3) - 2)     # LDS-DMA pieces the wait may leave in flight
            if cnt and int(cnt.group(1)) > allowed:
                n = int(cnt.group(1))
                young = vmem[-n:]
                if unknown_history and len(young) < n:       # what the block did not issue itself: assumed LDS-DMA pieces
                    young = [(True, False)] * (n - len(young)) + young
                if any(not dma for dma, _ in young) and sum(dma for dma, _ in young) > allowed:
                    found.append((kern, ln, s, "R6", f"the {cnt.group(1)} youngest vector-memory ops hold "
                                  f"{sum(dma for dma, _ in young)} LDS-DMA pieces (a {allowed // 4 + 2}-slot ring allows "
                                  f"{allowed}): the count is too high"))
        if op.startswith("v_pk_") and op.endswith("_f32"):
            sel = re.search(r"op_sel:\[([01,]+)\]", s)
            if sel and "1" in sel.group(1):
                found.append((kern, ln, s, "R5", "packed fp32 with an op_sel bit: wrong in lanes 48-63 beside 16x16x32 MFMAs"))
        if op.startswith("v_mfma"):
            for r in _vregs(ops[0]):
                mfma_dst[r] = (ws, s)
            for r in _vregs(ops[3]):
                mfma_srcc[r] = (ws, s)
        elif op.startswith("v_"):
            dst = _vregs(ops[0]) if ops else set()
            src = set()
            for o in ops[1:]:
                src |= _vregs(o)
            swap = "permlane" in op and "swap" in op
            if swap:
                src |= dst | _vregs(ops[1])
                dst |= _vregs(ops[1])
            if in_asm:
                for r in src:
                    if r in mfma_dst and ws - mfma_dst[r][0] < WINDOW:
                        found.append((kern, ln, s, "R1-RAW", f"{ws - mfma_dst[r][0]} wait states after {mfma_dst[r][1]}"))
                for r in dst:
                    if r in mfma_dst and ws - mfma_dst[r][0] < WINDOW:
                        found.append((kern, ln, s, "R1-WAW", f"{ws - mfma_dst[r][0]} wait states after {mfma_dst[r][1]}"))
                    if r in mfma_srcc and ws - mfma_srcc[r][0] < WINDOW:
                        found.append((kern, ln, s, "R1-WAR", f"{ws - mfma_srcc[r][0]} wait states after {mfma_srcc[r][1]}"))
                if swap:
                    for r in src:
                        if r in valu_vgpr and ws - valu_vgpr[r] < 3:          # fewer than 2 wait states between
                            found.append((kern, ln, s, "R4", f"source written by a VALU {ws - valu_vgpr[r] - 1} wait states earlier"))
            else:
                # compiler-visible VALU: its own hazards are handled; it shields later asm accesses
                texts = {mfma_dst[r][1] for r in src if r in mfma_dst}
                for r in [r for r, v in mfma_dst.items() if v[1] in texts]:
                    del mfma_dst[r]
                for r in dst:
                    mfma_dst.pop(r, None)
                    mfma_srcc.pop(r, None)
            for r in dst:
                valu_vgpr[r] = ws
            if ops and re.match(r"^(s\d+|s\[\d+:\d+\]|vcc)$", ops[0]):       # VALU writing an SGPR
                for r in _sregs(ops[0]):
                    valu_sgpr[r] = ws
            if op.startswith("v_cmp") and not re.match(r"^(s|vcc)", ops[0] if ops else ""):
                valu_sgpr["vcc"] = ws
        elif op == "s_mov_b32" and ops and ops[0] == "m0":
            m0_write = ws
        elif op.startswith(("global_load_lds", "buffer_load")) and "lds" in s:
            if ws - m0_write < 2:                                          # no wait state between
                found.append((kern, ln, s, "R3", "m0 written in the previous wait state"))
        if in_asm and op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            for o in ops:
                for r in _sregs(o):
                    if r in valu_sgpr and ws - valu_sgpr[r] < 6:            # fewer than 5 between
                        found.append((kern, ln, s, "R2", f"s{r} written by a VALU {ws - valu_sgpr[r] - 1} wait states earlier"))
        ws += 1
    return found


def scratch_report(path):
    """-> {kernel: (scratch bytes per lane, scratch accesses between two MFMAs, scratch instructions in the kernel)}
    of one kept assembly file.
    Bytes = the kernel descriptor's `.amdhsa_private_segment_fixed_size`; "between two MFMAs" = scratch_load /
    scratch_store instructions with an MFMA within 40 instructions on BOTH sides, i.e. inside an MFMA loop proper
    (the VALU phases between two loops — LayerNorm, encoding, compositing — are hundreds of instructions long).
    A spill inside a loop is a vector-memory operation in the in-order vmcnt queue of the stage hand-overs; one in a
    VALU phase costs its issue slot.  A frame WITHOUT any scratch instruction is not a spill: LLVM keeps the stack
    slot of a scalar-register tuple it then parked in vector-register lanes (`SGPRs Spill` in the resource remarks,
    `VGPRs Spill: 0`), plus the 4 bytes its register scavenger reserves once a frame exists — no lane ever touches
    that memory.  tests/test_build_hygiene.py holds every kernel to a budget."""
    import bisect
    text = open(path).read()
    sizes = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\s+(?:.*\n)*?\s*\.amdhsa_private_segment_fixed_size (\d+)", text):
        sizes[m.group(1)] = int(m.group(2))
    out = {}
    for name, size in sizes.items():
        start = text.find("\n" + name + ":")
        if start < 0:
            out[name] = (size, 0, 0)
            continue
        ops = [x.strip() for x in text[start:text.index(".Lfunc_end", start)].splitlines()]
        mf = [i for i, x in enumerate(ops) if x.startswith("v_mfma")]
        near = total = 0
        for i, x in enumerate(ops):
            if x.startswith("scratch_"):
                total += 1
                if mf:
                    k = bisect.bisect_left(mf, i)
                    if 0 < k < len(mf) and i - mf[k - 1] <= 40 and mf[k] - i <= 40:
                        near += 1
        out[name] = (size, near, total)
    return out
